cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
timeout -k 10 300 python bench.py > gpurun_out/b7.json 2> gpurun_out/b7.err || { tail -20 gpurun_out/b7.err; exit 1; }
cat gpurun_out/b7.json
