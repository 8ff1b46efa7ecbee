"""profiles/rNN_pmc_hbm.txt (tools/pmc_summary.py output of the FETCH_SIZE and WRITE_SIZE passes)
[+ profiles/rNN_pmc_sq.txt: SQ_INSTS_VALU per launch] -> profiles/rNN_pmc_traffic.json, the file
bench.py reads `roofline.traffic` and the VALU-issue figures from.
usage: python tools/pmc_traffic.py profiles/r01_pmc_hbm.txt [profiles/r01_pmc_sq.txt] > profiles/r01_pmc_traffic.json"""
import json, re, sys

NAMES = [("k_raster_fwd<20, false, false>", "k_raster_fwd<K,soft>"), ("k_sil_bwd", "k_sil_bwd"),
         ("k_raster_fwd<1, true, true>", "k_raster_fwd<1,tex>"), ("k_tex_bwd", "k_tex_bwd"),
         ("k_mask_losses_bwd", "k_mask_losses_bwd"), ("k_mask_losses", "k_mask_losses"),
         ("k_setup", "k_setup"), ("k_tex_mse_bwd", "k_tex_mse_bwd"), ("k_tex_mse", "k_tex_mse")]
kern, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)", line)
    if m and cur:
        kern.setdefault(cur, {})["fetch_bytes" if m.group(1) == "FETCH_SIZE" else "write_bytes"] = int(m.group(2)) * 1024
    elif not line.startswith(" "):
        cur = next((out for key, out in NAMES if key in line), None)
if len(sys.argv) > 2:
    cur = None
    for line in open(sys.argv[2]):
        m = re.match(r"\s+(SQ_INSTS_VALU|SQ_WAVES)\s+(\d+)", line)
        if m and cur:
            kern.setdefault(cur, {})["valu_insts" if m.group(1) == "SQ_INSTS_VALU" else "waves"] = int(m.group(2))
        elif not line.startswith(" "):
            cur = next((out for key, out in NAMES if key in line), None)
print(json.dumps({
    "source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only), "
              "tools/kbench.py --frames 64 --img 256 (same workload as bench.py); counters are KB per dispatch; "
              "FETCH_SIZE not doubled (reads here are 16-byte records, not wide streams: uncalibrated per "
              "MI355X_MICROARCH.md)" % sys.argv[1],
    "workload": {"frames": 64, "img": 256, "K": 20, "mesh": "bird"},
    "kernels": {k: v for k, v in kern.items() if "fetch_bytes" in v and "write_bytes" in v}}, indent=1))
