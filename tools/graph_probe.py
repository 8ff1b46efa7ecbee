import sys, os, faulthandler
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from acfm_video_3d_reconstruction_amd import ops
from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, make_cams
which = sys.argv[1]
d = torch.device("cuda:0")
m = np.load("tests/golden/meshes.npz"); v, f = m["bird_v"], m["bird_f"]
rng = np.random.default_rng(0)
tv = torch.tensor(batch_verts(v, 4, rng), device=d, requires_grad=True)
tc = torch.tensor(make_cams(4, rng, extent=0.5), device=d, requires_grad=True)
faces = torch.from_numpy(f)[None].repeat(4, 1, 1).to(d).contiguous()
H = 64
gt = (torch.rand(4, H, H, device=d) > 0.5).float(); edt = torch.rand(4, 1, H, H, device=d)
def step():
    if which == "proj":
        return ops.project(tv, tc)
    if which == "fwd":
        with torch.no_grad():
            return ops.sil_render(tv, faces, tc, H)
    if which == "loss":
        with torch.no_grad():
            return L.fused_silhouette_losses(gt, gt, edt)
    if which == "hard":
        with torch.no_grad():
            return ops.hard_raster(tv, faces, H)
    if which == "lossfwd":
        mask, p2f = ops.sil_render(tv, faces, tc, H)
        return L.fused_silhouette_losses(mask, gt, edt)
    if which == "lossbwd":
        mask, p2f = ops.sil_render(tv, faces, tc, H)
        l1, iou, e = L.fused_silhouette_losses(mask, gt, edt)
        return torch.autograd.grad((l1 + 0.1 * e).sum(), [tv, tc])
    if which == "lossbwd_ret":
        mask, p2f = ops.sil_render(tv, faces, tc, H)
        l1, iou, e = L.fused_silhouette_losses(mask, gt, edt)
        gv, gc = torch.autograd.grad((l1 + 0.1 * e).sum(), [tv, tc])
        return mask, p2f, gv, gc
    mask, p2f = ops.sil_render(tv, faces, tc, H)
    return torch.autograd.grad(mask.sum(), [tv, tc])
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
if len(sys.argv) > 2 and "ref" in sys.argv[2]:
    ref = [t.clone() for t in step()]
if len(sys.argv) > 2 and "sync" in sys.argv[2]:
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print("capturing", which, flush=True)
with torch.cuda.graph(g):
    out = step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed ok", which, flush=True)
