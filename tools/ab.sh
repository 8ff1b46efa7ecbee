#!/bin/bash
# A/B of kernel variants on one box: tools/ab.sh "<kbench args>" variant1 variant2 ...   ("-" = the shipping library)
# (variants are built with `make -C acfm_video_3d_reconstruction_amd/csrc VARIANT=name EXTRA="-D..."`); two interleaved rounds
args="$1"; shift
pat=${AB_PAT:-"k_raster_fwd<K|k_sil_bwd|k_raster_fwd<1|k_tex_bwd|sum of"}
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then lib=acfm_video_3d_reconstruction_amd/libacfm_hip.so; else lib=acfm_video_3d_reconstruction_amd/libacfm_hip_$v.so; fi
    echo "== $v (round $rep)"
    ACFM_LIB=$PWD/$lib python tools/kbench.py $args 2>&1 | grep -E "$pat"
  done
done
