"""Timing of the correlation cost volume (csrc/acfm_correlation.hip) at MaskFlownet-like sizes against
a torch formulation (pad + 81 shifted products).  usage: python tools/corr_bench.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import ops
d = torch.device("cuda:0")
def torch_corr(f1, f2, md):
    N, C, H, W = f1.shape
    p = torch.nn.functional.pad(f2, (md, md, md, md))
    return torch.stack([(f1 * p[:, :, md + tj:md + tj + H, md + ti:md + ti + W]).mean(1)
                        for tj in range(-md, md + 1) for ti in range(-md, md + 1)], 1)
for (N, C, H, W, md) in ((8, 196, 16, 16, 4), (8, 128, 32, 32, 4), (8, 96, 64, 64, 4), (8, 64, 128, 128, 4)):
    f1, f2 = torch.randn(N, C, H, W, device=d), torch.randn(N, C, H, W, device=d)
    res = {}
    for name, fn in (("hip", lambda: ops.correlation(f1, f2, md)), ("torch", lambda: torch_corr(f1, f2, md))):
        with torch.no_grad():
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize(); res[name] = 1e6 * (time.perf_counter() - t0) / 20
    fl = 2.0 * N * H * W * (2 * md + 1) ** 2 * C
    print("N=%d C=%d %dx%d md=%d: hip %.1f us (%.1f TFLOP/s), torch %.1f us" % (N, C, H, W, md, res["hip"], fl / res["hip"] / 1e6, res["torch"]))
