"""Timing of the multiframe training step at the reference's documented sizes
(docs/setup_video.md:25: --batch_size=8 --num_guesses 6, num_frames=2 -> 96 silhouette renders,
192 texture renders, 96 visibility rasters per step @256^2).  usage: python tools/step_bench.py"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acfm_video_3d_reconstruction_amd import _lib, image_utils as IU
from acfm_video_3d_reconstruction_amd.multiframe_step import MultiframeStep
from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams

p = argparse.ArgumentParser()
p.add_argument("--B", type=int, default=8); p.add_argument("--G", type=int, default=6)
p.add_argument("--img", type=int, default=256); p.add_argument("--iters", type=int, default=10)
p.add_argument("--mesh", default="horse"); p.add_argument("--tex", type=int, default=1)
p.add_argument("--torch-profile", action="store_true"); p.add_argument("--graph", action="store_true"); p.add_argument("--ops", action="store_true", help="eager step under torch.profiler: aten ops by count and shapes")
p.add_argument("--series", type=int, default=0)
a = p.parse_args()
d = torch.device("cuda:0")
m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz")); v, f = m[a.mesh + "_v"], m[a.mesh + "_f"]
rng = np.random.default_rng(0); B, T, G, H, Kh = a.B, 2, a.G, a.img, 15
N = B * T
step = MultiframeStep(torch.tensor(v, device=d), torch.tensor(f, device=d), torch.tensor(fps_lbs_logits(v, Kh), device=d),
                      num_training_frames=4 * N, img_size=H, num_guesses=G, num_lbs=Kh, scale_lr_decay=1.0,
                      prior_stream=os.environ.get("ACFM_PRIOR_STREAM", "0") != "0").to(d)
gt_cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
with torch.no_grad():
    gt, _ = step.renderer(step.solver.mean_v[None].repeat(N, 1, 1), step.faces1[None].expand(N, -1, -1), gt_cams)
    gt = (gt > 0.5).float()
batch = dict(masks=gt, edts_barrier=IU.compute_dt(gt, norm=False)[:, None].contiguous(),
             boundaries=IU.compute_boundaries(gt)[:, :1000].contiguous(),
             frames_idx=torch.arange(N, device=d).reshape(B, T), mirror_flag=torch.zeros(N, dtype=torch.long, device=d),
             transforms=torch.tensor([[1., 0, 0, 0]] * N, device=d), optical_flows=torch.randn(B, T, H, H, 2, device=d))
delta = (0.01 * torch.randn(N, Kh, 3, device=d)).requires_grad_(True)
tex = torch.rand(N, f.shape[0], 6, 6, 3, device=d, requires_grad=True) if a.tex else None
imgs = torch.rand(N, 3, H, H, device=d) if a.tex else None
if a.graph:
    from acfm_video_3d_reconstruction_amd.graphed import GraphedStep
    opt = torch.optim.Adam(list(step.parameters()), lr=1e-4, capturable=True, fused=True)
    inputs = dict(batch, delta=delta.detach())
    if a.tex: inputs.update(tex=tex.detach(), imgs=imgs)
    runner = GraphedStep(lambda i: step(i, i["delta"], textures=i.get("tex"), imgs=i.get("imgs"))[0], opt, inputs,
                         grad_inputs=("delta", "tex") if a.tex else ("delta",))
    def one():
        runner(inputs)
else:
    opt = torch.optim.Adam(list(step.parameters()) + [delta] + ([tex] if a.tex else []), lr=1e-4, fused=True)
    def one():
        opt.zero_grad(set_to_none=True)
        loss, _ = step(batch, delta, textures=tex, imgs=imgs)
        loss.backward(); opt.step()
for _ in range(3): one()
if getattr(a, "ops", False) and not a.graph:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as pr:
        one()
    rows = [e for e in pr.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and
            e.key.split("::")[1] in ("add", "add_", "mul", "mul_", "copy_", "clone", "fill_", "zero_", "zeros_like", "sum", "mean", "div", "neg", "repeat", "flip", "index_select", "embedding", "cat", "stack", "contiguous", "to", "_to_copy")]
    for e in sorted(rows, key=lambda e: (-e.count, e.key)):
        print("%3d x %-16s %s" % (e.count, e.key, str(e.input_shapes)[:150]))
    sys.exit(0)
if a.series:
    ts = []
    for _ in range(a.series):
        torch.cuda.synchronize(); t0 = time.perf_counter(); one(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("per-step ms:", " ".join("%.2f" % t for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): one()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
prof, ks = {}, float("nan")
if not a.graph:                          # the hipEvent brackets are not part of a captured graph
    lib = _lib.lib(); lib.acfm_prof_enable(1)
    for _ in range(a.iters): one()
    torch.cuda.synchronize()
    prof = _lib.prof_collect(); lib.acfm_prof_enable(0)
    ks = sum(ms for ms, _ in prof.values()) / a.iters
print(("hipGraph " if a.graph else "eager ") + "multiframe step B=%d T=2 G=%d (%d meshes) @%d, mesh=%s tex=%d: %.2f ms/step (%.0f clip-frames/s)%s" % (
    B, G, G * N, H, a.mesh, a.tex, 1e3 * dt, N / dt, "" if a.graph else ", HIP kernels %.2f ms" % ks))
for k, (ms, c) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("   %-24s %8.1f us/step  x%.0f" % (k, 1e3 * ms / a.iters, c / a.iters))

if a.torch_profile:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as pr:
        for _ in range(3): one()
        torch.cuda.synchronize()
    print(pr.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
