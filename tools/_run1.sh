cd $GRAFT_REPO_ROOT
for d in 4,2,4 4,1,1 4,2,2 4,4,4 2,2,2 8,4,8; do echo "== div $d"; for fr in 64 8; do ACFM_DIV=$d timeout -k 10 120 python tools/kbench.py --what sil,tex --frames $fr 2>&1 | grep -E "raster|sil_bwd" | awk '{printf "%s %s | ", $1, $3} END {print ""}'; done; done
