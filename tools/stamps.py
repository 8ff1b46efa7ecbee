"""Diagnostic: per-workgroup start/end stamps of the raster kernels (s_memrealtime, 100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, ops
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, make_cams
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "sil"
H = 256
dev = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz"))
v, f = m["bird_v"], m["bird_f"]
rng = np.random.default_rng(1000)
verts = torch.tensor(batch_verts(v, N, rng, 0.005), device=dev)
cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=dev)
faces = torch.tensor(f, device=dev)[None].repeat(N, 1, 1).contiguous()
atlas = torch.rand(N, f.shape[0], 6, 6, 3, device=dev)
# needs the diagnostic build (make -C acfm_video_3d_reconstruction_amd/csrc DIAG=1; ACFM_LIB=.../libacfm_hip_diag.so)
assert _lib.SO_PATH.endswith("_diag.so"), "run with ACFM_LIB=<path to libacfm_hip_diag.so> (make DIAG=1)"
raw = ctypes.CDLL(_lib.SO_PATH)
nb = N * (H // 8) ** 2 + (8 if N % 8 == 0 else 1) * min(1024, 32 * (N // 8 if N % 8 == 0 else N)) * 4   # blocks + split slots (upper bound)
buf = torch.zeros(nb * 5, dtype=torch.int64, device=dev)
def run():
    if mode == "sil": ops.sil_render(verts, faces, cams, H)
    else: ops.tex_render(verts, faces, cams, atlas, H)
for _ in range(3): run()
torch.cuda.synchronize()
raw.acfm_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
raw.acfm_debug_set_stamp_buffer(None)
b = buf.cpu().numpy().reshape(nb, 5)
b = b[b[:, 0] != 0]          # unused split slots leave no stamp
nb = len(b)
t0, t1, hw = b[:, 0], b[:, 1], b[:, 2]
dur = (t1 - t0) / 100.0  # us
span = (t1.max() - t0.min()) / 100.0
xcc = (hw >> 32) & 7
sub = ((hw >> 36) & 15).astype(np.int64) - 1      # split role: 0..3, else -1
cost = (hw >> 40) & 0xffff                          # the entry's face-box count (k_setup's estimate); 0 = fills only
tw, tf = b[:, 3], b[:, 4]                            # work start, work end (fills before or after)
cu = (hw & 0xffffffff)
print("blocks", nb, "kernel span %.1f us" % span, "block dur: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (
    dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
wpc = raw.acfm_debug_occupancy(0 if mode == "sil" else 1, 0)          # resident one-wave workgroups per CU
slots = wpc * torch.cuda.get_device_properties(0).multi_processor_count
print("sum of block durations %.0f us -> /%d wave slots (%d per CU) = %.1f us" % (dur.sum(), slots, wpc, dur.sum() / slots))
start = (t0 - t0.min()) / 100.0
print("start time percentiles (us): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(start, [10, 50, 90, 100])))
for x in range(8):
    sel = xcc == x
    if sel.any():
        grp = sorted(set((np.nonzero(sel)[0] & 7).tolist()))
        print("xcc", x, "blocks", sel.sum(), "block%8 groups", grp,
              "last end %.1f" % ((t1[sel].max() - t0.min()) / 100.0), "sum dur %.0f" % dur[sel].sum(),
              "idle-weighted util %.2f" % (dur[sel].sum() / (slots / 8 * (t1[sel].max() - t0[sel].min()) / 100.0)))
order = np.argsort(-dur)[:8]
print("heaviest blocks:", [(int(i), round(float(dur[i]), 1), round(float(start[i]), 1)) for i in order])

# occupancy timeline: running workgroups (time-averaged) and starts per 20-us bin
edges = np.arange(0, span + 20, 20.0)
s0, s1 = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0
print("bin_us  running(avg)  starts  mean_dur_of_starts")
for a_, b_ in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(s1, b_) - np.maximum(s0, a_), 0, None).sum() / (b_ - a_)
    st = (s0 >= a_) & (s0 < b_)
    print("%6.0f  %10.0f  %7d  %8.1f" % (a_, ov, st.sum(), dur[st].mean() if st.any() else 0))

# by role and cost: how long the work part of a workgroup lasts (t_end - t_work_start) and when it starts
work = tw != 0
wdur = np.where(work, (tf - tw) / 100.0, 0.0)
fill = np.where(work, ((tw - t0) + (t1 - tf)) / 100.0, dur)
print("workgroups: %d with a work entry (%d split pieces), %d fills only; fill part: mean %.1f us p99 %.1f; fills-only mean %.1f us" % (
    work.sum(), (work & (sub >= 0)).sum(), (~work).sum(), fill[work].mean(), np.percentile(fill[work], 99), dur[~work].mean() if (~work).any() else 0))
print("role    cost      count  work_us: mean   p50    p90    max   | start_us mean  max | end_us p50   p99   max | us per face box")
for role, rs in (("piece", work & (sub >= 0)), ("whole", work & (sub < 0))):
    for lo, hi in ((240, 1 << 16), (200, 240), (160, 200), (112, 160), (80, 112), (56, 80), (36, 56), (20, 36), (1, 20)):
        sel = rs & (cost >= lo) & (cost < hi)
        if not sel.any(): continue
        e = (t1[sel] - t0.min()) / 100.0
        print("%-6s %4d-%-5d %6d  %14.1f %6.1f %6.1f %6.1f | %12.1f %5.1f | %10.1f %5.1f %5.1f | %.3f" % (
            role, lo, min(hi, 9999), sel.sum(), wdur[sel].mean(), np.percentile(wdur[sel], 50), np.percentile(wdur[sel], 90), wdur[sel].max(),
            start[sel].mean(), start[sel].max(), np.percentile(e, 50), np.percentile(e, 99), e.max(), wdur[sel].sum() / cost[sel].sum()))
tot = wdur.sum()
print("work time by role: pieces %.0f us (%.1f %%), whole %.0f us; fills %.0f us" % (
    wdur[work & (sub >= 0)].sum(), 100 * wdur[work & (sub >= 0)].sum() / tot, wdur[work & (sub < 0)].sum(), fill.sum()))
