#!/bin/bash
for d in 1 2 4 8 16; do echo "== cover div $d (prefill 1)"; python tools/kbench.py --kout 1 --what sil,tex --iters 20 --div 0,$d,0 2>&1 | grep -E "k_raster_fwd<1"; done > gpurun_out/r3l_div.txt 2>&1
ACFM_LIB=$PWD/acfm_video_3d_reconstruction_amd/libacfm_hip_diag.so python tools/stamps.py 64 sil > gpurun_out/r3l_stamps.txt 2>&1
python tools/kbench.py --kout 1 --what sil --iters 30 2>&1 | grep -E "k_raster_fwd<K|k_sil_bwd" > gpurun_out/r3l_k.txt
python -m pytest tests/test_gpu_render.py tests/test_gpu_fused.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3
cat gpurun_out/r3l_div.txt gpurun_out/r3l_k.txt; head -8 gpurun_out/r3l_stamps.txt; tail -14 gpurun_out/r3l_stamps.txt
