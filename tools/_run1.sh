cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_edge_cases.py -m gpu -x -q -k "full_size" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
