cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
timeout -k 10 300 python bench.py --no-cpu --no-lean > gpurun_out/b8.json 2> gpurun_out/b8.err || { tail -20 gpurun_out/b8.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/b8.json"))
print(j["value"], j["ms_per_step"], j.get("eager_launch"), j.get("launch_note"))
print({k: round(v["us_per_step"], 1) for k, v in j["kernels"].items()})
PY
