"""Diagnostic: where the serial chain of the single-launch deformation solve spends its time.
Clock stamps (100 MHz wall clock) of every diagonal job of k_chol_tiles -- job start, own accumulation done, the
previous column's inverse arrived, in-tile factorisation begins / ends, results stored -- and of the four waves of the
in-tile factorisation (potrf32_wg) of the last column.  Needs the diagnostic build:
    make -C acfm_video_3d_reconstruction_amd/csrc DIAG=1      ->  libacfm_hip_diag.so
usage: python tools/solve_stamps.py   (the stamps' own global stores lengthen the in-tile barrier by ~0.5 us)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ACFM_LIB"] = os.path.join(ROOT, "acfm_video_3d_reconstruction_amd", "libacfm_hip_diag.so")
from acfm_video_3d_reconstruction_amd import _lib, ops
from acfm_video_3d_reconstruction_amd.deform import DeformSolver
from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
d = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")); v, f = m["horse_v"], m["horse_f"]
lbs = torch.tensor(fps_lbs_logits(v, 15), device=d)
solver = DeformSolver(torch.tensor(v, device=d), torch.tensor(f, device=d), lbs)
L = solver.laplacian().contiguous()
lib = ctypes.CDLL(os.environ["ACFM_LIB"])
V, Kh = lbs.shape
lib.acfm_deform_solve_workspace_bytes.restype = ctypes.c_size_t
nbytes = lib.acfm_deform_solve_workspace_bytes(V, Kh)
ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
P = torch.empty(V, Kh, device=d)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: ctypes.c_void_p(t.data_ptr())
for it in range(5):
    rc = lib.acfm_deform_solve(vp(L), vp(lbs), V, Kh, vp(P), vp(ws), ctypes.c_size_t(nbytes), st)
    torch.cuda.synchronize()
nb = (V + 31) // 32
out = (ctypes.c_longlong * (8 * nb))()
lib.acfm_debug_solve_stamps(vp(ws), V, out, 8 * nb)
a = np.array(out).reshape(nb, 8)[:, :6]
t0 = a[0, 0]
print("rc", rc, "units: us since the first diagonal job started")
print(" col   start  acc_done  inv_in   pre_potrf  post_potrf  stored |  step")
prev = None
for j in range(nb):
    r = (a[j] - t0) / 100.0
    step = (r[4] - prev) if prev is not None else float("nan")
    print("%4d %7.2f %8.2f %8.2f %9.2f %10.2f %8.2f | %5.2f   wait %.2f trsm+syrk %.2f potrf %.2f store %.2f" % (
        j, r[0], r[1], r[2], r[3], r[4], r[5], step, r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
    prev = r[4]

ps = (ctypes.c_longlong * 16)()
lib.acfm_debug_potrf_stamps(ps)
q = np.array(ps).reshape(4, 4); q0 = q[:, 0].min()
print("potrf of the last diagonal job: per wave [after preload barrier, panel start, panel end, after final barrier] us")
for w in range(4): print("  wave %d  " % w, "  ".join("%6.2f" % ((x - q0) / 100.0) for x in q[w]))
