"""Config 3 timing (BASELINE.json configs[2]): test-time refinement of a 32-frame horse clip,
eager vs hipGraph-replayed iterations.  usage: python tools/refine_bench.py [--iters 50]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import image_utils as IU
from acfm_video_3d_reconstruction_amd.deform import DeformSolver
from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
from acfm_video_3d_reconstruction_amd.refine import refine_clip
from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams

p = argparse.ArgumentParser(); p.add_argument("--iters", type=int, default=50); p.add_argument("--frames", type=int, default=32)
a = p.parse_args()
d = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")); v, f = m["horse_v"], m["horse_f"]
rng = np.random.default_rng(3); N, H, Kh = a.frames, 256, 16
cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
faces = torch.tensor(f, device=d)[None].repeat(N, 1, 1).contiguous()
solver = DeformSolver(torch.tensor(v, device=d), faces[0], torch.tensor(fps_lbs_logits(v, Kh), device=d))
r = NeuralRenderer(H, pix_to_face_slots=1)
with torch.no_grad():
    gt, _ = r(solver(torch.tensor(rng.normal(0, 0.05, (N, Kh, 3)).astype(np.float32), device=d)), faces, cams)
    gt = (gt > 0.5).float()
edt = IU.compute_dt(gt, norm=False)[:, None].contiguous()
bds = IU.compute_boundaries(gt)[:, :1000].contiguous()
for mode in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, _, _, hist = refine_clip(r, solver, torch.zeros(N, Kh, 3, device=d), cams, faces, gt, edt, bds,
                                num_optim_iter=a.iters, optimize_camera=True, use_graph=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("graph=%-5s %d iterations x %d frames: %.1f ms total, %.3f ms/iter, %.0f frame-iters/s, loss %.5f -> %.5f" % (
        mode, a.iters, N, 1e3 * dt, 1e3 * dt / a.iters, N * a.iters / dt, hist[0], hist[-1]))
# per-kernel breakdown of one eager iteration (hipEvent brackets of the library's own kernels)
from acfm_video_3d_reconstruction_amd import _lib
lib = _lib.lib(); lib.acfm_prof_enable(1)
refine_clip(r, solver, torch.zeros(N, Kh, 3, device=d), cams, faces, gt, edt, bds, num_optim_iter=a.iters,
            optimize_camera=True, use_graph=False)
torch.cuda.synchronize()
prof = _lib.prof_collect(); lib.acfm_prof_enable(0)
tot = 0.0
for k, (ms, c) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("   %-24s %8.1f us/iter  x%.1f" % (k, 1e3 * ms / a.iters, c / a.iters)); tot += 1e3 * ms / a.iters
print("   library kernels total %.1f us/iter" % tot)
