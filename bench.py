"""Headline benchmark: frames/s of differentiable render + backward, 642-vert bird template
@256x256, 64 frames per GPU (BASELINE.json configs[1]), synthetic data resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of frames:
deformation apply -> soft-silhouette render (K=20) -> fused L1/IoU/EDT losses -> boundary
loss -> atlas-texture render + MSE -> backward to handle offsets, cameras, shared mean shape
and atlas.  Frames are sharded over ranks (weak scaling: 64 frames per GPU); the only
exchange is one RCCL all-reduce of the shared-shape gradient.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant
kernel, algorithmic bytes vs 8 TB/s HBM, durations from hipEvents on the launch stream) and
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--config", type=int, default=2, choices=(2, 5),
                   help="BASELINE.json configs[1] (default: bird, 64 frames @256^2, fp32, drop-in API) or configs[4], one "
                        "GPU's share of it (5120-face subdivided horse, 16 frames @512^2, half storage with fp32 loss sums, "
                        "fused render+loss operators)")
    p.add_argument("--frames", type=int, default=None, help="frames per GPU (64; config 5: 16)")
    p.add_argument("--img", type=int, default=None, help="image size (256; config 5: 512)")
    p.add_argument("--handles", type=int, default=16)
    p.add_argument("--tex", type=int, default=1, help="include the atlas-texture render + loss")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-oracle time budget")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-lean", action="store_true", help="skip the all-slots-stored comparison run")
    p.add_argument("--headline-only", action="store_true",
                   help="only the headline step and its per-kernel timing (profiling runs: every launch of a kernel in the "
                        "trace then belongs to the same workload)")
    p.add_argument("--eager", action="store_true",
                   help="launch every kernel of the step from Python instead of replaying one captured hipGraph")
    p.add_argument("--learn-lbs", action="store_true",
                   help="handle weights (lbs) and mean shape are learned shared parameters: every step re-factorises the "
                        "deformation system and the ranks exchange the pre-solve sums [G = sum g delta^T | sum g | loss] "
                        "(~50 KB, sharding.SharedShapeExchange) instead of the mean-shape gradient alone; eager launches")
    p.add_argument("--tex-stream", type=int, default=0,
                   help="texture branch forked onto its own HIP stream behind the silhouette forward (graph edges in a capture); "
                        "measured: no gain, 0.693 vs 0.689 ms/step -- every kernel of either branch fills the chip")
    return p.parse_args()


def edt_and_boundaries(gt_mask):
    """set_input-style prep (multiframe/main.py:365-377, utils/image.py:94-146) on the device,
    once, before the timed region: un-normalised EDT of the GT mask and <= 1000 boundary points."""
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    edt = IU.compute_dt(gt_mask, norm=False)[:, None]
    bds = IU.compute_boundaries(gt_mask)
    if bds.shape[1] > 1000:  # keep the boundary loss's n_samples=1000 a pure permutation
        keep = torch.argsort(torch.rand(bds.shape[:2], device=bds.device) - bds[..., 2], dim=1)[:, :1000]
        bds = torch.gather(bds, 1, keep[..., None].expand(-1, -1, 3))
    if bds.shape[1] == 0:
        bds = torch.zeros(bds.shape[0], 1, 3, device=bds.device)
    return edt.contiguous(), bds.contiguous()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # ACFM_DIST_BACKEND=gloo + ACFM_ALL_RANKS_ON_GPU0=1: rehearsal of the multi-rank control flow
    # on a one-GPU box (collectives through the host); the real runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("ACFM_DIST_BACKEND", "nccl")
    if os.environ.get("ACFM_ALL_RANKS_ON_GPU0") == "1":
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from acfm_video_3d_reconstruction_amd import _lib
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    from acfm_video_3d_reconstruction_amd.sharding import SharedGradReducer
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams

    cfg5 = a.config == 5
    if a.headline_only:
        a.no_lean = a.no_cpu = True
    a.frames = a.frames or (16 if cfg5 else 64)
    a.img = a.img or (512 if cfg5 else 256)
    N, H, Kh = a.frames, a.img, a.handles
    m = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz"))
    v_np, f_np = m["bird_v"], m["bird_f"]
    if cfg5:   # one SubdivideMeshes pass of the horse template: 2562 verts / 5120 faces (the shim's restatement of pytorch3d.ops)
        from acfm_video_3d_reconstruction_amd.pytorch3d_shim.ops import SubdivideMeshes
        from acfm_video_3d_reconstruction_amd.pytorch3d_shim.structures import Meshes
        sub = SubdivideMeshes()(Meshes(verts=[torch.tensor(m["horse_v"])], faces=[torch.tensor(m["horse_f"])]))
        v_np, f_np = sub.verts_packed().numpy().astype(np.float32), sub.faces_packed().numpy().astype(np.int64)
    V, F = v_np.shape[0], f_np.shape[0]
    rng = np.random.default_rng(1000 + rank)
    ext = float(np.abs(v_np).max())

    # ---- synthetic inputs, resident in HBM before the timed region
    mean_v = torch.tensor(v_np, device=dev)
    faces = torch.tensor(f_np, device=dev)[None].repeat(N, 1, 1).contiguous()
    lbs_logits = torch.tensor(fps_lbs_logits(v_np, Kh), device=dev)
    solver = DeformSolver(mean_v, faces[0], lbs_logits)
    delta0 = torch.tensor(rng.normal(0, 0.02, (N, Kh, 3)).astype(np.float32), device=dev)
    cams0 = torch.tensor(make_cams(N, rng, extent=ext), device=dev)
    renderer = NeuralRenderer(H, storage="f16") if cfg5 else NeuralRenderer(H)
    with torch.no_grad():  # GT = own render of a differently perturbed pose, thresholded
        gt_delta = torch.tensor(rng.normal(0, 0.03, (N, Kh, 3)).astype(np.float32), device=dev)
        gt_cams = cams0.clone()
        gt_cams[:, 1:3] += torch.tensor(rng.uniform(-0.03, 0.03, (N, 2)).astype(np.float32), device=dev)
        gt_mask, _ = renderer(solver(gt_delta), faces, gt_cams)
        gt_mask = (gt_mask.float() > 0.5).float()
    edt, bds = edt_and_boundaries(gt_mask)
    imgs_gt = torch.tensor(rng.uniform(0, 1, (N, 3, H, H)).astype(np.float32), device=dev)
    R = 6
    atlas = torch.tensor(rng.uniform(0, 1, (N, F, R, R, 3)).astype(np.float32), device=dev, requires_grad=True)
    delta = delta0.clone().requires_grad_(True)
    cams = cams0.clone().requires_grad_(True)
    mean_p = mean_v.clone().requires_grad_(True)
    reducer = SharedGradReducer([mean_p])  # one flat fp32 all-reduce (RCCL) per step
    _flat, flat_views, flat_extra = reducer.packed(n_extra=1)

    params = [delta, cams, mean_p, atlas]

    side = torch.cuda.Stream(device=dev) if (a.tex and a.tex_stream) else None

    if cfg5:   # references and atlas held in half by the caller (what ACFM_STORE_F16 reads); the atlas gradient stays float
        gt_h, edt_h, imgs_h = gt_mask.half(), edt.half(), imgs_gt.half()

    seed = torch.ones((), device=dev)   # d total / d total, made once (autograd.grad would fill a fresh one every step)

    def compute(ren, fused=False):
        pred_v = solver(delta, mean_override=mean_p)                      # a8 (closed form)
        if cfg5:
            sil4, mask, p2f = ren.forward_silhouette_losses(pred_v, faces, cams, gt_h, edt_h, raw=True)
            bdt = L.bds_loss(ren.project_points(pred_v, cams), bds, faces, p2f, reduce=False)
            tmse = ren.forward_texture_mse(pred_v.detach(), faces, cams, atlas, imgs_h, gt_h)[0]
            total = L.combine_losses([sil4, bdt, tmse], [1.0, 0.0, 0.0, 0.1, 0.1, 0.5])
            g_delta, g_cams, g_mean, g_atlas = torch.autograd.grad(total, params, grad_outputs=seed)
            if world > 1:
                flat_views[0].copy_(g_mean)
                flat_extra.copy_(total.detach().reshape(1))
            return total.detach(), g_delta, g_cams, g_mean, g_atlas
        if fused:   # opt-in operator: the loss terms leave the raster kernel with the mask (acfm_sil_loss_*)
            sil4, mask, p2f = ren.forward_silhouette_losses(pred_v, faces, cams, gt_mask, edt, raw=True)
        else:
            mask, p2f = ren(pred_v, faces, cams)                             # a3
        tmse = None
        if side is not None:
            # the texture branch reads detached geometry only (main.py:627-636): forked onto its own HIP stream behind
            # the silhouette forward (whose face setup it takes over), it runs -- forward here, backward wherever autograd
            # reaches it -- beside the loss kernels and the silhouette backward; in a capture these are graph edges
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                tex, _, _ = ren(pred_v.detach(), faces, cams, textures=atlas)    # a4
                tmse = L.masked_texture_mse(tex, imgs_gt, gt_mask)               # main.py:655-662
        if not fused:
            sil4 = L.fused_silhouette_losses(mask, gt_mask, edt, raw=True)   # a10, a11: [N,4] = (l1, ., ., edt)
        proj = ren.project_points(pred_v, cams)                          # a2
        bdt = L.bds_loss(proj, bds, faces, p2f, reduce=False)            # a12
        # total = mean_n(l1 + 0.1 edt + 0.1 bds) [+ 0.5 mean_n(texture mse)]: one launch each way
        # (combine_losses) instead of ~13 elementwise launches on 64-element vectors
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            total = L.combine_losses([sil4, bdt, tmse], [1.0, 0.0, 0.0, 0.1, 0.1, 0.5])
        elif a.tex:
            if fused:
                tmse = ren.forward_texture_mse(pred_v.detach(), faces, cams, atlas, imgs_gt, gt_mask)[0]
            else:
                tex, _, _ = ren(pred_v.detach(), faces, cams, textures=atlas)        # a4
                tmse = L.masked_texture_mse(tex, imgs_gt, gt_mask)                   # main.py:655-662
            total = L.combine_losses([sil4, bdt, tmse], [1.0, 0.0, 0.0, 0.1, 0.1, 0.5])
        else:
            total = L.combine_losses([sil4, bdt], [1.0, 0.0, 0.0, 0.1, 0.1])
        # gradients of the per-frame parameters (handle offsets, cameras), the shared mean shape
        # and the atlas; autograd.grad hands the buffers over without AccumulateGrad's copies
        g_delta, g_cams, g_mean, g_atlas = torch.autograd.grad(total, params, grad_outputs=seed, allow_unused=not a.tex)
        if world > 1:
            # the step writes its shared gradient and loss scalar straight into the exchange buffer (the
            # last two nodes of the captured graph): the exchange itself is then one RCCL launch
            flat_views[0].copy_(g_mean)
            flat_extra.copy_(total.detach().reshape(1))
        return total.detach(), g_delta, g_cams, g_mean, g_atlas

    def exchange(total, g_mean):
        if world > 1:  # the one exchange: shared mean-shape gradient + loss scalar
            reducer.reduce_packed()
            mean_p.grad = flat_views[0]
        else:
            mean_p.grad = g_mean

    def step(ren=renderer, fused=False):
        total, g_delta, g_cams, g_mean, g_atlas = compute(ren, fused)
        exchange(total, g_mean)
        return total, g_delta, g_cams, g_atlas

    def graphed(ren, fused=False):
        """The same step with its ~60 launches (forward, backward, every gradient buffer) captured
        once into a hipGraph and replayed: shapes are static, every entry point of libacfm_hip.so is
        stream-ordered, so the step runs at the GPU's pace whatever the host's launch rate is (with
        8 ranks on one host the Python launch loop is the first thing to fall behind).  The all-reduce
        of the shared gradient stays an ordinary RCCL call on the replayed buffers."""
        cur = torch.cuda.current_stream(dev)
        s2 = torch.cuda.Stream(device=dev)
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            for _ in range(3):
                compute(ren, fused)
        cur.wait_stream(s2)
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of torch.distributed polls events while we capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            outs = compute(ren, fused)

        def replay(_ren=None):
            g.replay()
            exchange(outs[0], outs[3])
            return outs[0], outs[1], outs[2], outs[4]
        return replay

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graph_note = []

    def timed(ren, warmup, steps, use_graph=None, fused=False):
        use_graph = (not a.eager) if use_graph is None else use_graph
        fn = (lambda r: step(r, True)) if fused else step
        if use_graph:
            try:
                fn = graphed(ren, fused)
            except Exception as exc:   # never lose the measurement to a capture problem: fall back, and say so
                torch.cuda.synchronize()
                graph_note.append("capture failed (%s: %s), eager launch instead" % (type(exc).__name__, str(exc)[:200]))
        for _ in range(warmup):
            fn(ren)
        # the previous measurement's hipGraph (and its private memory pool) sits in a reference cycle: collect it
        # here, not in the middle of this one's timed steps (its hipFree stalls the host for tens of ms)
        gc.collect()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(ren)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- --learn-lbs: the exchange the north star names (shared mean-shape AND handle-weight gradients)
    lbs_info = None
    if a.learn_lbs:
        from acfm_video_3d_reconstruction_amd.sharding import SharedShapeExchange
        lbs_q = torch.nn.Parameter(lbs_logits.clone())
        mean_l = torch.nn.Parameter(mean_v.clone())
        solver_l = DeformSolver(mean_l, faces[0], lbs_q)
        ex = SharedShapeExchange(solver_l)

        def lbs_step(ren=renderer):
            pred_v = ex.apply(delta)                                       # one factorisation per step + deform apply
            mask, p2f = ren(pred_v, faces, cams)
            sil4 = L.fused_silhouette_losses(mask, gt_mask, edt, raw=True)
            bdt = L.bds_loss(ren.project_points(pred_v, cams), bds, faces, p2f, reduce=False)
            tex, _, _ = ren(pred_v.detach(), faces, cams, textures=atlas)
            tmse = L.masked_texture_mse(tex, imgs_gt, gt_mask)
            total = L.combine_losses([sil4, bdt, tmse], [1.0, 0.0, 0.0, 0.1, 0.1, 0.5])
            for p_ in (delta, cams, atlas):
                p_.grad = None
            total.backward()                                               # local: stops at the (P, mean) leaves
            ex.finish(extra_scalars=total.detach().reshape(1))             # ONE all-reduce + the solve's backward
            return total
        for _ in range(a.warmup):
            lbs_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            lbs_step()
        fence()
        dt_l = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_l], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_l = float(t.item())
        lbs_info = dict(value=round(world * N * a.steps / dt_l, 2), unit="frames/s", ms_per_step=round(1e3 * dt_l / a.steps, 4),
                        exchange_bytes=int(ex.bytes),
                        note="lbs and mean shape learned: per-step factorisation + pre-solve exchange [G | sum g | loss], eager launches")

    dt = timed(renderer, a.warmup, a.steps)          # the drop-in API: pix_to_face [N,H,W,20] int64
    ms_step = 1e3 * dt / a.steps
    value = world * N * a.steps / dt
    use_graph = not a.eager
    # the other launch mode of the same step, reported beside the headline (never instead of it)
    extras = not cfg5 and not a.headline_only
    dt_other = timed(renderer, max(2, a.warmup // 2), a.steps, use_graph=not use_graph) if side is None and extras else None
    # same step with every slot of pix_to_face [N,H,W,20] stored at render time (160 bytes per pixel nobody in the
    # step reads) instead of the default's nearest-face plane + the other planes on first use; beside the headline
    dt_lean = None
    if not a.no_lean and not cfg5:
        lean = NeuralRenderer(H, pix_to_face_slots=20)
        dt_lean = timed(lean, max(2, a.warmup // 2), a.steps)

    # same step through the opt-in fused render+loss operator (the silhouette losses leave the raster kernel with
    # the mask; no separate passes over the mask, no [N,H,W] mask gradient); beside the headline, never instead of it
    dt_fused = timed(renderer, max(2, a.warmup // 2), a.steps, fused=True) if side is None and extras else None
    # same step with the bit-reproducible silhouette backward (fixed-point accumulation, AcfmRasterTuning flag 1)
    dt_det = None
    if side is None and extras:
        with _lib.raster_tuning(deterministic=True):
            dt_det = timed(renderer, max(2, a.warmup // 2), a.steps)

    # ---- the metric string taken literally: silhouette render + backward alone (a3 fwd + bwd to vertices and
    # cameras, no losses, no texture branch); reported beside the headline step, never instead of it
    dt_render = None
    if extras:
        rv = solver(delta0).detach().requires_grad_(True)
        rc = cams0.clone().requires_grad_(True)
        rw = torch.randn(N, H, H, device=dev) / (H * H)

        def render_only():
            m, _ = renderer(rv, faces, rc)
            return torch.autograd.grad((m * rw).sum(), [rv, rc])
        for _ in range(max(2, a.warmup // 2)):
            render_only()
        dt_render = None
        for _round in range(2):   # eager launches with a fresh 671 MB output per call: one allocator hiccup would show
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                render_only()
            fence()
            dt = time.perf_counter() - t0
            dt_render = dt if dt_render is None else min(dt_render, dt)
        if world > 1:
            t = torch.tensor([dt_render], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_render = float(t.item())

    # ---- SURVEY section 8d's step, every row of it: the headline step + the per-optimiser-step factorisation
    # of the deformation system with learned handle weights (a8: cot Laplacian, fp64 Cholesky, lbs gradient)
    # + the mesh priors on the deformed shape (a14 locally_rigid_fn, a15 mesh_laplacian_smoothing 'cot');
    # reported beside the headline value, never instead of it
    dt_full = None
    if a.tex and side is None and extras:
        from acfm_video_3d_reconstruction_amd.pytorch3d_shim.loss import mesh_laplacian_smoothing
        from acfm_video_3d_reconstruction_amd.pytorch3d_shim.structures import Meshes
        lbs_p = torch.nn.Parameter(lbs_logits.clone())
        mean_q = torch.nn.Parameter(mean_v.clone())
        solver2 = DeformSolver(mean_q, faces[0], lbs_p)
        mesh_t = Meshes(verts=mean_v[None].repeat(N, 1, 1), faces=faces)
        params2 = [delta, cams, mean_q, lbs_p, atlas]
        prior_stream = torch.cuda.Stream(device=dev) if os.environ.get("ACFM_BENCH_PRIOR_STREAM", "1") != "0" else None

        def full_step():
            solver2.refresh()                                            # lbs / mean shape moved: one factorisation
            pred_v = solver2(delta)
            # the mesh priors (a dozen one-workgroup-per-mesh kernels, latency bound) on a second stream beside the
            # raster kernels: they need the deformed vertices only (os.environ ACFM_BENCH_PRIOR_STREAM=0: same stream)
            cur_s = torch.cuda.current_stream(dev)
            prior_s = prior_stream if prior_stream is not None else cur_s
            prior_s.wait_stream(cur_s)
            with torch.cuda.stream(prior_s):
                mesh = Meshes(verts=pred_v, faces=faces)
                prior = 0.1 * L.locally_rigid_fn(mesh, mesh_t) + 0.1 * mesh_laplacian_smoothing(mesh, method="cot")
            mask, p2f = renderer(pred_v, faces, cams)
            sil4 = L.fused_silhouette_losses(mask, gt_mask, edt, raw=True)
            bdt = L.bds_loss(renderer.project_points(pred_v, cams), bds, faces, p2f, reduce=False)
            tex, _, _ = renderer(pred_v.detach(), faces, cams, textures=atlas)
            tmse = L.masked_texture_mse(tex, imgs_gt, gt_mask)
            cur_s.wait_stream(prior_s)
            total = L.combine_losses([sil4, bdt, tmse], [1.0, 0.0, 0.0, 0.1, 0.1, 0.5]) + prior
            return torch.autograd.grad(total, params2, grad_outputs=seed)
        full_fn, full_mode = full_step, "eager launches"
        if not a.eager:
            try:                                   # the same step as one hipGraph (forward + backward, static shapes)
                cur = torch.cuda.current_stream(dev)
                s3 = torch.cuda.Stream(device=dev)
                s3.wait_stream(cur)
                with torch.cuda.stream(s3):
                    for _ in range(3):
                        full_step()
                cur.wait_stream(s3)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                    full_outs = full_step()
                full_fn, full_mode = g2.replay, "one hipGraph replay per step"
            except Exception as exc:
                torch.cuda.synchronize()
                graph_note.append("survey_8d_step: capture failed (%s), eager launch instead" % type(exc).__name__)
        for _ in range(max(2, a.warmup // 2)):
            full_fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            full_fn()
        fence()
        dt_full = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_full], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_full = float(t.item())

    # ---- per-kernel durations (hipEvents on the launch stream) over the same K steps
    roof = None
    kern = {}
    lib = _lib.lib()
    if rank == 0:
        lib.acfm_prof_enable(1)
    for _ in range(a.steps):  # every rank runs the same steps (the all-reduce is collective)
        step()
    fence()
    if rank == 0:
        ms = (ctypes.c_float * 24)()
        cnt = (ctypes.c_int * 24)()
        _lib.check(lib.acfm_prof_collect(ms, cnt, 24), "acfm_prof_collect")
        lib.acfm_prof_enable(0)
        for i in range(24):
            if cnt[i]:
                kern[lib.acfm_prof_name(i).decode()] = dict(avg_us=1e3 * ms[i] / cnt[i], launches=cnt[i],
                                                            us_per_step=1e3 * ms[i] / a.steps)
        dom = max(kern, key=lambda k: kern[k]["us_per_step"])
        # ALGORITHMIC bytes per launch, fp32:
        alg = {
            # read verts 12V + cam 28 (+ faces 12F shared, once per batch); write mask 4H^2 + nearest face id 8H^2
            "k_raster_fwd<K,soft>": N * (12 * H * H + 12 * V + 28) + 12 * F,
            # read grad 4H^2 + mask 4H^2 + K-th key 8H^2; write grad_verts 12V + grad_cam 28
            "k_sil_bwd": N * (16 * H * H + 12 * V + 28),
            # read atlas 12FR^2 ; write image 12H^2 + sil 4H^2 + face id 8H^2
            "k_raster_fwd<1,tex>": N * (24 * H * H + 12 * F * R * R + 12 * V + 28),
            "k_tex_bwd": N * (12 * H * H + 4 * H * H + 12 * F * R * R),
            "k_mask_losses": N * 12 * H * H, "k_mask_losses_bwd": N * 16 * H * H,
        }
        if cfg5:   # half storage, int32 nearest-face plane, fused losses (SURVEY 8d with the halved terms)
            alg = {
                # write mask 2H^2 + id 4H^2, read gt 2H^2 + edt 2H^2, verts 12V, cam 28 (+ faces 12F once)
                "k_raster_fwd<K,soft>": N * (10 * H * H + 12 * V + 28) + 12 * F,
                # read mask 2H^2 + gt 2H^2 + edt 2H^2 (no mask gradient: formed in the kernel), write grads 12V + 28
                "k_sil_bwd": N * (6 * H * H + 12 * V + 28),
                # read atlas 6FR^2 + reference image 6H^2 + mask 2H^2; write image 6H^2 + sil 2H^2 + id 4H^2
                "k_raster_fwd<1,tex>": N * (20 * H * H + 6 * F * R * R + 12 * V + 28),
                # read image 6H^2 + reference 6H^2 + mask 2H^2; write the float atlas gradient 12FR^2
                "k_tex_bwd": N * (14 * H * H + 12 * F * R * R),
            }
        ab = alg.get(dom, 0)
        ach = ab / (kern[dom]["avg_us"] * 1e-6) / 1e9 if ab else 0.0
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
        # the committed measurement of the same workload (rocprofv3 --pmc, separate passes) is quoted
        traffic, traffic_src, pmk = None, None, None
        for tag in ("r02", "r01"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)))
                w = pm["workload"]
                same = (w["frames"], w["img"], w["K"], w.get("mesh", "bird"), w.get("storage", "f32")) == \
                       (N, H, 20, "horse_subdiv1" if cfg5 else "bird", "f16" if cfg5 else "f32")
                if same and dom in pm["kernels"]:
                    pmk = pm["kernels"][dom]
                    traffic = pmk["fetch_bytes"] + pmk["write_bytes"]
                    traffic_src = "profiles/%s_pmc_traffic.json (FETCH_SIZE + WRITE_SIZE per launch, own --pmc passes)" % tag
                    break
            except (OSError, KeyError, ValueError):
                pass
        roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=ab, avg_launch_us=round(kern[dom]["avg_us"], 2))
        if pmk and pmk.get("valu_insts"):
            # What limits the kernel is VALU issue, not HBM.  Issue rates measured on this part with independent
            # instruction streams at 4 waves per SIMD (tools/ubench/valu_rates.hip, profiles/r02_valu_rates.txt):
            # fp32 add / mul / fma, integer add, and: 1.21 ns per wave-instruction and SIMD (844 G/s on 1024 SIMDs);
            # selects, compares, min / max / med3, shifts, 64-bit moves, DPP, packed fp32: 1.75-1.95 ns (525-585 G/s);
            # rcp / exp / sqrt: 3.5 ns (293 G/s).  The raster kernels' stream is mostly the second class (the sorted
            # insertion is one 64-bit compare + six selects per slot), so its ceiling lies between the two figures.
            rate = pmk["valu_insts"] / (kern[dom]["avg_us"] * 1e-6)
            v = dict(insts_per_launch=pmk["valu_insts"], achieved_ginst_s=round(rate / 1e9, 1),
                     full_rate_ginst_s=844.4, half_rate_class_ginst_s=572.7,
                     frac_of_full_rate=round(rate / 844.4e9, 4), frac_of_half_rate_class=round(rate / 572.7e9, 4),
                     source=traffic_src.split(" ")[0] + " (SQ_INSTS_VALU), profiles/r02_valu_rates.txt (rates)")
            if pmk.get("active_inst_valu") and pmk.get("grbm_gui_active"):
                # SQ_ACTIVE_INST_VALU counts quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs
                v["valu_busy"] = round(pmk["active_inst_valu"] * 4 / (1024 * pmk["grbm_gui_active"] / 8), 4)
            if pmk.get("thread_cycles_valu") and pmk.get("active_inst_valu"):
                v["lanes_active"] = round(pmk["thread_cycles_valu"] / (64 * pmk["active_inst_valu"]), 4)
            roof["valu"] = v
            v["valu_busy_note"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): an estimate, a few % above 1 when saturated"
            roof["limiter"] = {
                "k_sil_bwd": "VALU issue (per pixel x face pair: membership test key <= kth, exact edge distance again, its "
                             "backward to three vertices, 4 DPP row shifts per gradient component before the LDS accumulators)",
            }.get(dom, "VALU issue (selects / compares of the sorted K-nearest insertion and the exact per-pixel tests)")
        if traffic:   # what the kernel actually moves (the API's K int64 ids per pixel dominate): context, not `achieved`
            moved = traffic / (kern[dom]["avg_us"] * 1e-6) / 1e9
            roof.update(moved_gbs=round(moved, 1), moved_frac=round(moved / HBM_PEAK_GBS, 4))

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import oracle as O
        verts_np = solver(delta0).detach().cpu().numpy()
        cams_np = cams0.cpu().numpy()
        gmask = np.sign(np.random.default_rng(0).standard_normal((N, H, H))).astype(np.float32) / (H * H)
        # host-core share of a one-GPU box (the pool's guidance: 16 workers per GPU)
        cores = min(16, len(os.sched_getaffinity(0)))
        cores = O.set_threads(cores)

        def cpu_frames(k):
            t = time.perf_counter()
            O.sil_render_backward(verts_np[:k], f_np, cams_np[:k], H, gmask[:k])
            return time.perf_counter() - t
        t1 = cpu_frames(1)
        k = int(max(1, min(N, a.cpu_seconds / max(t1, 1e-3))))
        tk = cpu_frames(k)
        cpu = dict(value=round(k / tk, 3), unit="frames/s", cores=cores, kind="port",
                   sample="oracle silhouette render K=20 + backward, %d frame(s) @%dx%d, OpenMP over rows" % (k, H, H))

    if rank == 0:
        out = {
            "metric": "frames/s differentiable render+bwd, 5k-face mesh @512^2, 16 frames/GPU (config 5 shard)" if cfg5 else
                      "frames/s differentiable render+bwd, 642-vert mesh @256^2, batch=64",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (half storage)" if cfg5 else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE config 5, one GPU's share: subdivided horse template (%d v / %d f), %d frames/GPU "
                                    "@%dx%d, half storage with fp32 loss sums, deform apply + fused soft silhouette K=20 / L1 / "
                                    "IoU / EDT + boundary loss + fused atlas texture render / MSE, fwd+bwd" % (V, F, N, H, H))
                       if cfg5 else
                       "CUB bird template (642 v / 1280 f), %d frames/GPU @%dx%d, deform apply + "
                       "soft silhouette K=20 + L1/IoU/EDT + boundary loss%s, fwd+bwd" %
                       (N, H, H, " + atlas texture render/MSE" if a.tex else ""),
                       "frames_per_gpu": N, "img_size": H, "handles": Kh, "faces_per_pixel": 20,
                       "sharding": "frames over ranks; all-reduce of shared mean-shape grad"},
            "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
        }
        out["launch"] = "one hipGraph replay per step (+ the RCCL all-reduce)" if use_graph else "eager (one Python launch per kernel)"
        if graph_note:
            out["launch_note"] = graph_note
        if dt_other:
            out["hipgraph_replay" if not use_graph else "eager_launch"] = {
                "value": round(world * N * a.steps / dt_other, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_other / a.steps, 4)}
        if dt_render:
            out["render_only"] = {
                "value": round(world * N * a.steps / dt_render, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_render / a.steps, 4),
                "note": "soft-silhouette render K=20 (pix_to_face [N,H,W,20] as the default lazy tensor) + backward to "
                        "vertices and cameras only; eager launches"}
        if dt_full:
            out["survey_8d_step"] = {
                "value": round(world * N * a.steps / dt_full, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_full / a.steps, 4),
                "note": "headline step + per-step factorisation of the deformation system with learned handle weights "
                        "(cot Laplacian, fp64 Cholesky, lbs gradient) + locally_rigid_fn + mesh_laplacian_smoothing('cot'); "
                        + full_mode}
        if dt_fused:
            out["fused_render_loss"] = {
                "value": round(world * N * a.steps / dt_fused, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_fused / a.steps, 4),
                "note": "same step, silhouette losses and the masked texture MSE fused into the raster kernels "
                        "(NeuralRenderer.forward_silhouette_losses / forward_texture_mse: acfm_sil_loss_*, acfm_tex_mse_*)"}
        if lbs_info:
            out["learn_lbs_step"] = lbs_info
        if dt_det:
            out["deterministic_backward"] = {
                "value": round(world * N * a.steps / dt_det, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_det / a.steps, 4),
                "note": "same step with _lib.raster_tuning(deterministic=True): the silhouette backward accumulates in "
                        "2^-36 fixed point (int64 atomics), gradients bit-identical from run to run"}
        if dt_lean:
            out["all_slots_stored"] = {
                "value": round(world * N * a.steps / dt_lean, 2), "unit": "frames/s",
                "ms_per_step": round(1e3 * dt_lean / a.steps, 4),
                "note": "NeuralRenderer(pix_to_face_slots=20): all 20 planes of pix_to_face written by the render "
                        "(the default returns the same [N,H,W,20] int64 tensor lazily: nearest-face plane written, "
                        "the other planes rendered when first touched -- never, in this step)"}
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
