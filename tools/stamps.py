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
buf = torch.zeros(nb * 3, dtype=torch.int64, device=dev)
def run():
    if mode == "sil": ops.sil_render(verts, faces, cams, H)
    else: ops.tex_render(verts, faces, cams, atlas, H)
for _ in range(3): run()
torch.cuda.synchronize()
raw.acfm_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
raw.acfm_debug_set_stamp_buffer(None)
b = buf.cpu().numpy().reshape(nb, 3)
b = b[b[:, 0] != 0]          # unused split slots leave no stamp
nb = len(b)
t0, t1, hw = b[:, 0], b[:, 1], b[:, 2]
dur = (t1 - t0) / 100.0  # us
span = (t1.max() - t0.min()) / 100.0
xcc = hw >> 32
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
