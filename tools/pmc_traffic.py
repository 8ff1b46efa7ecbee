"""PMC summaries (tools/pmc_summary.py output of the separate --pmc passes) -> profiles/rNN_pmc_traffic.json, the file
bench.py reads `roofline.traffic` and the VALU figures from.
usage: python tools/pmc_traffic.py --frames 64 --img 256 --mesh bird --storage f32 pass1.txt [pass2.txt ...] > out.json
The workload arguments are those of the profiled run (bench.py compares them with its own before quoting a number)."""
import argparse, json, re

NAMES = [("k_raster_fwd<20, false, false>", "k_raster_fwd<K,soft>"), ("k_sil_bwd", "k_sil_bwd"),
         ("k_raster_fwd<1, true, true>", "k_raster_fwd<1,tex>"), ("k_tex_cover", "k_tex_cover"), ("k_tex_bwd", "k_tex_bwd"),
         ("k_mask_losses_bwd", "k_mask_losses_bwd"), ("k_mask_losses", "k_mask_losses"),
         ("k_setup", "k_setup"), ("k_tex_mse_bwd", "k_tex_mse_bwd"), ("k_tex_mse", "k_tex_mse"),
         ("k_sil_loss_finish1", "k_sil_loss_finish1"), ("k_tex_loss_finish1", "k_tex_loss_finish1")]
KEYS = {"FETCH_SIZE": ("fetch_bytes", 1024), "WRITE_SIZE": ("write_bytes", 1024), "SQ_INSTS_VALU": ("valu_insts", 1),
        "SQ_WAVES": ("waves", 1), "SQ_ACTIVE_INST_VALU": ("active_inst_valu", 1), "SQ_THREAD_CYCLES_VALU": ("thread_cycles_valu", 1),
        "GRBM_GUI_ACTIVE": ("grbm_gui_active", 1), "SQ_WAVE_CYCLES": ("wave_cycles", 1), "SQ_WAIT_ANY": ("wait_any", 1),
        "SQ_WAIT_INST_ANY": ("wait_inst_any", 1), "SQ_ACTIVE_INST_ANY": ("active_inst_any", 1),
        "SQ_WAIT_INST_LDS": ("wait_inst_lds", 1), "SQ_INSTS_LDS": ("insts_lds", 1), "SQ_INSTS_SALU": ("insts_salu", 1),
        "SQ_INSTS_VMEM_RD": ("insts_vmem_rd", 1), "SQ_INSTS_VMEM_WR": ("insts_vmem_wr", 1),
        "SQ_ACTIVE_INST_LDS": ("active_inst_lds", 1), "SQ_ACTIVE_INST_SCA": ("active_inst_sca", 1),
        "SQ_INST_CYCLES_VMEM": ("inst_cycles_vmem", 1), "SQ_LDS_BANK_CONFLICT": ("lds_bank_conflict", 1),
        "SQ_BUSY_CYCLES": ("busy_cycles", 1)}
p = argparse.ArgumentParser()
p.add_argument("--frames", type=int, required=True); p.add_argument("--img", type=int, required=True)
p.add_argument("--mesh", default="bird"); p.add_argument("--storage", default="f32"); p.add_argument("--K", type=int, default=20)
p.add_argument("--command", default="")
p.add_argument("files", nargs="+")
a = p.parse_args()
kern = {}
for fn in a.files:
    cur = None
    for line in open(fn):
        m = re.match(r"\s+([A-Z_0-9]+)\s+(\d+)", line)
        if m and cur and m.group(1) in KEYS:
            k, mul = KEYS[m.group(1)]
            kern.setdefault(cur, {})[k] = int(m.group(2)) * mul
        elif not line.startswith(" "):
            cur = next((out for key, out in NAMES if key in line), None)
print(json.dumps({
    "source": "rocprofv3 --pmc <one group per pass> --kernel-trace, per-dispatch means (tools/pmc.sh -> tools/pmc_summary.py); "
              "FETCH_SIZE / WRITE_SIZE are KB per dispatch (x1024 here); FETCH_SIZE not doubled (reads are 16-byte records, not "
              "wide streams: uncalibrated per MI355X_MICROARCH.md); profiled command: " + a.command,
    "workload": {"frames": a.frames, "img": a.img, "K": a.K, "mesh": a.mesh, "storage": a.storage},
    "kernels": {k: v for k, v in kern.items() if "fetch_bytes" in v and "write_bytes" in v}}, indent=1))
