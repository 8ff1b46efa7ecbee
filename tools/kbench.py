"""Per-kernel micro-benchmark on the headline workload (bird, N frames @HxH): calls the
operator surface with acfm_prof_* enabled and prints avg us per kernel.
usage: python tools/kbench.py [--frames 64] [--img 256] [--iters 20] [--mesh bird] [--subdiv 0]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, ops  # noqa: E402
from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L  # noqa: E402
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, make_cams  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--frames", type=int, default=64)
    p.add_argument("--img", type=int, default=256)
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--mesh", default="bird")
    p.add_argument("--subdiv", type=int, default=0, help="SubdivideMeshes passes (1: 642 v / 1280 f -> 2562 v / 5120 f, BASELINE config 5)")
    p.add_argument("--what", default="sil,tex,loss")
    p.add_argument("--kout", type=int, default=0, help="1: only the nearest-face plane of pix_to_face is written")
    p.add_argument("--det", type=int, default=0, help="1: deterministic (fixed-point) silhouette backward")
    p.add_argument("--fused", type=int, default=0, help="1: the fused render+loss operator (acfm_sil_loss_*)")
    p.add_argument("--split", type=int, default=-5, help="block splitting: < 0 automatic, 0 never, 1 always")
    p.add_argument("--prefill", type=int, default=1, help="0: the silhouette render does not pre-fill the texture render's empty blocks")
    p.add_argument("--div", default="0,0,0", help="workgroups per group = entries / div: fwdK,fwd1,bwd (0 = default)")
    a = p.parse_args()
    dev = torch.device("cuda:0")
    ops.PREFILL_TEX[0] = bool(a.prefill)
    m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz"))
    v, f = m[a.mesh + "_v"], m[a.mesh + "_f"]
    for _ in range(a.subdiv):
        from oracle import oracle as O      # (test infrastructure: only the mesh subdivision, on the host)
        v, f = O.subdivide(v, f)
        f = f.astype(np.int64)
    rng = np.random.default_rng(1000)
    N, H = a.frames, a.img
    verts = torch.tensor(batch_verts(v, N, rng, 0.005), device=dev, requires_grad=True)
    cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=dev, requires_grad=True)
    faces = torch.tensor(f, device=dev)[None].repeat(N, 1, 1).contiguous()
    gt = (torch.rand(N, H, H, device=dev) > 0.5).float()
    edt = torch.rand(N, 1, H, H, device=dev)
    atlas = torch.rand(N, f.shape[0], 6, 6, 3, device=dev, requires_grad=True)
    bds = torch.cat([torch.rand(N, 800, 2, device=dev) * 2 - 1, torch.ones(N, 800, 1, device=dev)], -1)
    lib = _lib.lib()
    # --split / --div: per-call tuning of the raster entry points (AcfmRasterTuning), for experiments
    tune = _lib.raster_tuning(split=a.split, grid_div=tuple(int(x) for x in a.div.split(",")), deterministic=bool(a.det))
    tune.__enter__()

    def run():
        if "sil" in a.what and a.fused:
            los, mask, p2f = ops.sil_render_losses(verts, faces, cams, H, gt, edt, k_out=(1 if a.kout == 1 else None))
            proj = ops.project(verts, cams)[..., :2]
            b = L.bds_loss(proj, bds, faces, p2f, reduce=False)
            (los[:, 0] + los[:, 3] + 0.1 * b).mean().backward()
        elif "sil" in a.what:
            mask, p2f = ops.sil_render(verts, faces, cams, H, k_out=(1 if a.kout == 1 else None))
            if "loss" in a.what:
                l1, iou, e = L.fused_silhouette_losses(mask, gt, edt)
                proj = ops.project(verts, cams)[..., :2]
                b = L.bds_loss(proj, bds, faces, p2f, reduce=False)
                (l1 + e + 0.1 * b).mean().backward()
            else:
                mask.mean().backward()
        if "tex" in a.what:
            img, _, _ = ops.tex_render(verts.detach(), faces, cams.detach(), atlas, H)
            img.mean().backward()

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.acfm_prof_enable(1)
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 24)()
    cnt = (ctypes.c_int * 24)()
    _lib.check(lib.acfm_prof_collect(ms, cnt, 24), "collect")
    lib.acfm_prof_enable(0)
    tot = 0.0
    for i in range(24):
        if cnt[i]:
            per = 1e3 * ms[i] / a.iters
            tot += per
            print("%-24s avg %9.1f us  x%-3d /iter  = %9.1f us/iter" % (
                lib.acfm_prof_name(i).decode(), 1e3 * ms[i] / cnt[i], cnt[i] // a.iters, per))
    print("sum of kernels: %.1f us/iter  (%d frames @%d)" % (tot, N, H))
    with torch.no_grad():
        mask, p2f = ops.sil_render(verts, faces, cams, H)
        cntk = (p2f >= 0).sum(-1)
        print("coverage %.3f  mean faces/covered px %.2f  px with K full %.4f" % (
            (cntk > 0).float().mean().item(), cntk[cntk > 0].float().mean().item(),
            (cntk == p2f.shape[-1]).float().mean().item()))


if __name__ == "__main__":
    main()
