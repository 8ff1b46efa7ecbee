#!/bin/bash
tools/ab.sh "--kout 1 --what sil --iters 30" - bub nopf e88 e112 > gpurun_out/r3j_ab.txt 2>&1
for fr in 16 32 48; do for lib in base -; do echo "== frames $fr lib $lib"; if [ "$lib" = "-" ]; then L=acfm_video_3d_reconstruction_amd/libacfm_hip.so; else L=acfm_video_3d_reconstruction_amd/libacfm_hip_$lib.so; fi; ACFM_LIB=$PWD/$L python tools/kbench.py --kout 1 --what sil --iters 20 --frames $fr 2>&1 | grep -E "k_raster_fwd<K|k_sil_bwd"; done; for sp in -3 -8; do echo "== frames $fr shipping split $sp";  python tools/kbench.py --kout 1 --what sil --iters 20 --frames $fr --split $sp 2>&1 | grep -E "k_raster_fwd<K|k_sil_bwd"; done; done > gpurun_out/r3j_split.txt 2>&1
python bench.py --steps 20 --warmup 5 --cpu-seconds 5 > gpurun_out/r3j_bench2.json 2> gpurun_out/r3j_bench2.err
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_render.py -m gpu -q > gpurun_out/r3j_tests.log 2>&1
tail -5 gpurun_out/r3j_tests.log
