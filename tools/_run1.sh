cd $GRAFT_REPO_ROOT
export ACFM_DIST_BACKEND=gloo ACFM_ALL_RANKS_ON_GPU0=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-lean > gpurun_out/b2r.json 2> gpurun_out/b2r.err || { tail -30 gpurun_out/b2r.err; exit 1; }
python -c "
import json
j=json.loads([l for l in open('gpurun_out/b2r.json') if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], j['launch'], j.get('eager_launch'), j.get('launch_note'))"
unset ACFM_DIST_BACKEND ACFM_ALL_RANKS_ON_GPU0
timeout -k 10 300 python bench.py --no-cpu --no-lean | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('eager_launch'))"
