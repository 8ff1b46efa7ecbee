"""Forward-only timing of the K-nearest silhouette render on the headline workload (no_grad; setup + forward per call):
python tools/fwd_only.py [--frames 64] [--prefill 1] [--iters 30].  Used for A/B runs of library variants (ACFM_LIB)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, ops
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, make_cams
p = argparse.ArgumentParser()
p.add_argument("--frames", type=int, default=64); p.add_argument("--img", type=int, default=256)
p.add_argument("--prefill", type=int, default=1); p.add_argument("--iters", type=int, default=30)
a = p.parse_args()
dev = torch.device("cuda:0")
ops.PREFILL_TEX[0] = bool(a.prefill)
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz"))
v, f = m["bird_v"], m["bird_f"]
rng = np.random.default_rng(1000)
N, H = a.frames, a.img
verts = torch.tensor(batch_verts(v, N, rng, 0.005), device=dev)
cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=dev)
faces = torch.tensor(f, device=dev)[None].repeat(N, 1, 1).contiguous()
lib = _lib.lib()
import ctypes
with torch.no_grad():
    for _ in range(5): ops.sil_render(verts, faces, cams, H)
    torch.cuda.synchronize()
    lib.acfm_prof_enable(1)
    for _ in range(a.iters): ops.sil_render(verts, faces, cams, H)
    torch.cuda.synchronize()
ms = (ctypes.c_float * 24)(); cnt = (ctypes.c_int * 24)()
_lib.check(lib.acfm_prof_collect(ms, cnt, 24), "collect"); lib.acfm_prof_enable(0)
print(os.path.basename(_lib.SO_PATH), "prefill", a.prefill, " ".join("%s %.1f" % (lib.acfm_prof_name(i).decode(), 1e3 * ms[i] / cnt[i]) for i in range(24) if cnt[i]))
