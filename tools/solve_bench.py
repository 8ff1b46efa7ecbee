"""Timing of the native deformation solve (csrc/acfm_solve.hip) vs torch.linalg (rocSOLVER) in fp64.
usage: python tools/solve_bench.py [--mesh horse] [--handles 15]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, ops
from acfm_video_3d_reconstruction_amd.deform import solve_matrix, handle_matrix, DeformSolver
from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits

p = argparse.ArgumentParser(); p.add_argument("--mesh", default="horse"); p.add_argument("--handles", type=int, default=15)
p.add_argument("--iters", type=int, default=50)
a = p.parse_args()
d = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")); v, f = m[a.mesh + "_v"], m[a.mesh + "_f"]
lbs = torch.tensor(fps_lbs_logits(v, a.handles), device=d)
solver = DeformSolver(torch.tensor(v, device=d), torch.tensor(f, device=d), lbs)
L = solver.laplacian()
def t_native():
    lg = lbs.clone().requires_grad_(True)
    P = ops.deform_solve(L, lg); P.sum().backward()
def t_torch():
    lg = lbs.clone().requires_grad_(True)
    P = solve_matrix(L, handle_matrix(lg)).float(); P.sum().backward()
for name, fn in (("native fwd+bwd", t_native), ("torch fp64 fwd+bwd", t_torch)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); print("%-22s %.1f us" % (name, 1e6 * (time.perf_counter() - t0) / a.iters))
P = ops.deform_solve(L, lbs); R = solve_matrix(L, handle_matrix(lbs))
print("max |P - P_torch64| / max|P| = %.2e" % float((P.double() - R).abs().max() / R.abs().max()))
import hashlib
outs = [ops.deform_solve(L, lbs).cpu().numpy().tobytes() for _ in range(20)]
print("P sha1 %s  (20 repeats identical: %s)" % (hashlib.sha1(outs[0]).hexdigest()[:12], all(o == outs[0] for o in outs)))
lib = _lib.lib(); lib.acfm_prof_enable(1)
for _ in range(a.iters): t_native()
torch.cuda.synchronize()
for k, (ms, c) in _lib.prof_collect().items(): print("   %-20s %8.1f us" % (k, 1e3 * ms / c))
lib.acfm_prof_enable(0)
