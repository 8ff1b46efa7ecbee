#!/bin/bash
# Regenerates profiles/rNN_* on the GPU box: tools/make_profiles.sh r03   (run through gpurun; ~8 minutes)
set -o pipefail
tag=${1:-r03}
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set (run through gpurun)"; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
o=gpurun_out/profiles_$tag; rm -rf $o; mkdir -p $o
B="python3 bench.py --steps 20 --warmup 5 --headline-only --eager"
# 1. PMC passes ON bench.py itself (eager launches so that every kernel is a dispatch of its own), one counter group per pass
tools/pmc.sh $o/pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B > $o/${tag}_pmc_sq.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B >> $o/${tag}_pmc_sq.txt 2>&1 || exit 1
# where the issue stalls go: LDS / scalar / vector-memory shares of the instruction stream
tools/pmc.sh $o/pmc_sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -- $B >> $o/${tag}_pmc_sq.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_f FETCH_SIZE -- $B > $o/${tag}_pmc_hbm.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc_w WRITE_SIZE -- $B >> $o/${tag}_pmc_hbm.txt 2>&1 || exit 1
python3 tools/pmc_traffic.py --frames 64 --img 256 --mesh bird --storage f32 --command "$B" $o/${tag}_pmc_hbm.txt $o/${tag}_pmc_sq.txt > $o/${tag}_pmc_traffic.json
# the same for BASELINE config 5 (one GPU's share: 16 frames @512^2, 5120 faces, half storage): roofline.traffic of --config 5
B5="python3 bench.py --config 5 --steps 10 --warmup 3 --headline-only --eager"
tools/pmc.sh $o/pmc5_f FETCH_SIZE -- $B5 > $o/${tag}_pmc_hbm_config5.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc5_w WRITE_SIZE -- $B5 >> $o/${tag}_pmc_hbm_config5.txt 2>&1 || exit 1
tools/pmc.sh $o/pmc5_sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -- $B5 > $o/${tag}_pmc_sq_config5.txt 2>&1 || exit 1
python3 tools/pmc_traffic.py --frames 16 --img 512 --mesh horse_subdiv1 --storage f16 --command "$B5" $o/${tag}_pmc_hbm_config5.txt $o/${tag}_pmc_sq_config5.txt > $o/cfg5.json
# ... and for configs 3 and 4 (their `roofline.traffic`): FETCH / WRITE + the VALU counters
for c in 3 4; do
  Bc="python3 bench.py --config $c --steps 10 --warmup 3 --headline-only --eager --no-cpu"
  tools/pmc.sh $o/pmc${c}_f FETCH_SIZE -- $Bc > $o/${tag}_pmc_hbm_config$c.txt 2>&1 || exit 1
  tools/pmc.sh $o/pmc${c}_w WRITE_SIZE -- $Bc >> $o/${tag}_pmc_hbm_config$c.txt 2>&1 || exit 1
  tools/pmc.sh $o/pmc${c}_sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -- $Bc > $o/${tag}_pmc_sq_config$c.txt 2>&1 || exit 1
done
python3 tools/pmc_traffic.py --frames 32 --img 256 --mesh horse --storage f32 --command "python3 bench.py --config 3 ..." $o/${tag}_pmc_hbm_config3.txt $o/${tag}_pmc_sq_config3.txt > $o/cfg3.json
python3 tools/pmc_traffic.py --frames 32 --img 256 --mesh mixed --storage f32 --command "python3 bench.py --config 4 ..." $o/${tag}_pmc_hbm_config4.txt $o/${tag}_pmc_sq_config4.txt > $o/cfg4.json
python3 - $o/${tag}_pmc_traffic.json $o/cfg5.json $o/cfg3.json $o/cfg4.json <<'PY'
import json, sys
main = json.load(open(sys.argv[1]))
main["more"] = []
for fn in sys.argv[2:]:
    extra = json.load(open(fn))
    main["more"].append({"workload": extra["workload"], "kernels": extra["kernels"], "source": extra["source"]})
json.dump(main, open(sys.argv[1], "w"), indent=1)
PY
cp $o/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json   # the bench lines below cite this round's counters
# 2. the bench line (default flags) and the rocprof kernel summary of the same program
python3 bench.py > $o/${tag}_bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 5 --headline-only > $o/stats.log 2>&1 || exit 1
cp "$(ls -t $(find $o/stats -name '*kernel_stats.csv') | head -1)" $o/${tag}_bench_kernel_stats.csv
# 3. the other BASELINE workloads: their own bench lines
python3 bench.py --config 5 > $o/${tag}_bench_config5.json 2> $o/bench5.err || { tail -5 $o/bench5.err; exit 1; }
python3 bench.py --config 3 > $o/${tag}_bench_config3.json 2> $o/bench3.err || { tail -5 $o/bench3.err; exit 1; }
python3 bench.py --config 4 > $o/${tag}_bench_config4.json 2> $o/bench4.err || { tail -5 $o/bench4.err; exit 1; }
# 3b. the multi-rank control flow rehearsed on this one GPU (two ranks, collectives through the host with gloo): NOT a
# scaling measurement -- both ranks share the card -- but every line of the --gpus N path runs
for c in 2 4; do
  ACFM_DIST_BACKEND=gloo ACFM_ALL_RANKS_ON_GPU0=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 2951$c bench.py --gpus 2 --config $c --steps 10 --warmup 3 --no-cpu --headline-only 2> $o/world2_$c.err | grep "^{" > $o/${tag}_world2_rehearsal_config$c.json || { tail -5 $o/world2_$c.err; exit 1; }
done
# 4. the callers either side of the render path
(python3 tools/step_bench.py; python3 tools/step_bench.py --graph; python3 tools/solve_bench.py; python3 tools/refine_bench.py) > $o/${tag}_step_solve_refine.txt 2>&1
build_ub/valu_rates > $o/${tag}_valu_rates.txt 2>&1
[ -x tools/ubench/f64_chain ] && tools/ubench/f64_chain > $o/${tag}_f64_chain.txt 2>&1
ls -la $o
