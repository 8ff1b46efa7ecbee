"""Diagnostic: event counters of the K-nearest forward on the headline workload (needs the counting build:
make -C acfm_video_3d_reconstruction_amd/csrc VARIANT=count EXTRA="-DACFM_DIAG -DACFM_DIAG_COUNT";
ACFM_LIB=.../libacfm_hip_count.so python tools/count_events.py)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, ops
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, make_cams
assert "_count" in os.path.basename(_lib.SO_PATH), "run with ACFM_LIB=<libacfm_hip_count.so>"
raw = ctypes.CDLL(_lib.SO_PATH)
d = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")); v, f = m["bird_v"], m["bird_f"]
rng = np.random.default_rng(1000); N, H = 64, 256
verts = torch.tensor(batch_verts(v, N, rng, 0.005), device=d)
cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
faces = torch.tensor(f, device=d)[None].repeat(N, 1, 1).contiguous()
ops.sil_render(verts, faces, cams, H); torch.cuda.synchronize()
c = (ctypes.c_ulonglong * 16)()
raw.acfm_debug_counters(c, 1)
mask, p2f = ops.sil_render(verts, faces, cams, H); torch.cuda.synchronize()
raw.acfm_debug_counters(c, 0)
names = ["blocks with work", "walk calls", "candidates (sum list_n)", "walk iterations (wave)", "lanes in_box", "iterations reaching stage 2",
         "lanes live (stage 2)", "lanes accepted", "insertion blocks executed", "insertion block tests", "lanes having a face", "iterations inserting",
         "(group, face) pairs walked", "(group, face) pairs with an accepting pixel"]
for i, n in enumerate(names): print("%-32s %12d" % (n, c[i]))
it = c[3]
print("per iteration: lanes with a face %.1f, in_box %.1f, live %.1f (of the %.0f%% of iterations that reach stage 2), accepted %.1f; "
      "insertion blocks per inserting iteration %.2f" % (c[10] / it, c[4] / it, c[6] / max(c[5], 1), 100.0 * c[5] / it, c[7] / max(c[11], 1), c[8] / max(c[11], 1)))
cov = (p2f[..., 0] >= 0).sum().item(); kept = (p2f >= 0).sum().item()
print("covered pixels %d, kept (pixel, face) pairs %d (%.1f per covered pixel); accepted pairs %d" % (cov, kept, kept / cov, c[7]))

# ---- the backward's walk (same counters 3, 4, 10, 12: iterations, lanes in a face's box, lanes having a face, pairs)
tv = verts.clone().requires_grad_(True); tc = cams.clone().requires_grad_(True)
gt = (torch.rand(N, H, H, device=d) > 0.5).float()
mk, _ = ops.sil_render(tv, faces, tc, H)
loss = (mk - gt).abs().mean()
torch.cuda.synchronize(); raw.acfm_debug_counters(c, 1)
loss.backward(); torch.cuda.synchronize()
raw.acfm_debug_counters(c, 0)
print("backward: walk iterations %d, lanes having a face %.1f / 64, in a face's box %.1f / 64, (group, face) pairs %d" % (
    c[3], c[10] / max(c[3], 1), c[4] / max(c[3], 1), c[12]))
