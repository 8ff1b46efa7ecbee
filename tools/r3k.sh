#!/bin/bash
tools/ab.sh "--kout 1 --what sil --iters 30" - ne nofill bwe > gpurun_out/r3k_ab.txt 2>&1
for pf in 1 0; do echo "== prefill $pf"; python tools/kbench.py --kout 1 --what sil,tex --iters 20 --prefill $pf 2>&1 | grep -E "k_raster_fwd|k_tex_bwd|sum of"; done > gpurun_out/r3k_prefill.txt 2>&1
for sp in -5 1 -3; do echo "== frames 32 split $sp"; python tools/kbench.py --kout 1 --what sil --iters 20 --frames 32 --split $sp 2>&1 | grep -E "k_raster_fwd<K|k_sil_bwd"; done > gpurun_out/r3k_split32.txt 2>&1
python -m pytest tests -m gpu -q > gpurun_out/r3k_tests.log 2>&1
tail -5 gpurun_out/r3k_tests.log
