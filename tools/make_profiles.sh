#!/bin/bash
# Regenerates profiles/rNN_* on the GPU box: tools/make_profiles.sh r01   (run through gpurun)
set -o pipefail
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/profiles_$tag; mkdir -p $o
python bench.py > $o/${tag}_bench.json 2> $o/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python bench.py --steps 20 --warmup 5 --no-cpu --no-lean > $o/stats.log 2>&1 || exit 1
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/${tag}_bench_kernel_stats.csv
tools/pmc.sh $o/pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- --iters 3 > $o/${tag}_pmc_sq.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --iters 3 >> $o/${tag}_pmc_sq.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_f FETCH_SIZE -- --iters 3 > $o/${tag}_pmc_hbm.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_w WRITE_SIZE -- --iters 3 >> $o/${tag}_pmc_hbm.txt 2>&1 || exit 1
(for n in 8 64; do for m in sil tex; do echo "== frames $n kernel $m"; python tools/stamps.py $n $m | grep -E "span|slots|start time|bin_us|^ +[0-9]+ +[0-9]+ +[0-9]+ "; done; done) > $o/${tag}_tile_stamps.txt 2>&1
python tools/pmc_traffic.py $o/${tag}_pmc_hbm.txt $o/${tag}_pmc_sq.txt > $o/${tag}_pmc_traffic.json
# the callers either side of the render path: multiframe training step (eager / one hipGraph) and
# the per-step deformation solve (native fp64 Cholesky vs torch.linalg)
(python tools/step_bench.py; python tools/step_bench.py --graph; python tools/solve_bench.py; python tools/refine_bench.py) > $o/${tag}_step_solve_refine.txt 2>&1
ls -la $o
