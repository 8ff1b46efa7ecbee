#!/bin/bash
# PMC A/B of kernel variants (through gpurun): tools/ab_pmc.sh variant1 variant2 ...  ("" = shipping)
for v in "$@"; do
  export ACFM_LIB=$PWD/acfm_video_3d_reconstruction_amd/libacfm_hip${v:+_$v}.so
  echo "=== ${v:-shipping}"
  tools/pmc.sh gpurun_out/abp1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 tools/kbench.py --kout 1 --what sil --iters 4 2>&1 | grep -A8 "k_raster_fwd<20\|k_sil_bwd"
  tools/pmc.sh gpurun_out/abp2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/kbench.py --kout 1 --what sil --iters 4 2>&1 | grep -A5 "k_raster_fwd<20\|k_sil_bwd"
  tools/pmc.sh gpurun_out/abp3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_IFETCH -- python3 tools/kbench.py --kout 1 --what sil --iters 4 2>&1 | grep -A7 "k_raster_fwd<20\|k_sil_bwd"
done
