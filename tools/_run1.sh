cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "texture" > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
for fr in 64 8; do timeout -k 10 100 python tools/kbench.py --what tex --frames $fr | grep tex_bwd; done
timeout -k 10 100 python tools/kbench.py --what tex --mesh horse | grep tex_bwd
