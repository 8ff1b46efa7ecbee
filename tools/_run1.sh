cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "twice or graph" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
