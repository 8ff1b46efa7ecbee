"""GPU parity of the COMPOSED paths against the oracle's restatement of the
reference's callers: the trainer's camera chain (multiframe/main.py:97-138, 551-584), every named term of
ShapeTrainer.forward (:523-765) and the first iteration of the test-time refinement loop
(nnutils/predictor.py:287-349) with its gradients."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _d():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_camera_pipeline_vs_oracle():
    """a20: decode + mirror_cameras + transform_cameras (one kernel each way) against the oracle's restatement
    of main.py:97-138, 564-584, forward 1e-6 and backward against float64 autograd of the oracle."""
    from acfm_video_3d_reconstruction_amd import harness, ops
    d = _d()
    g = torch.Generator().manual_seed(3)
    G, N = 5, 6
    emb = torch.randn(G, N, 7, generator=g)
    emb[..., 0] = torch.tensor([-30., -1., 0., 0.5, 3., -19.9])      # relu branch of the scale both ways (decay 0.05)
    mf = torch.tensor([1, 0, 1, 1, 0, 0])
    tr = torch.cat([torch.rand(N, 1, generator=g) + 0.5, torch.rand(N, 2, generator=g) - 0.5,
                    torch.tensor([1., 0., 1., 0., 1., 0.])[:, None]], 1)
    w = torch.randn(G * N, 7, generator=g)
    e64 = emb.double().requires_grad_(True)
    ref = O.camera_pipeline(e64, mf, tr.double(), 0.05)
    (ref * w.double()).sum().backward()
    e = emb.to(d).requires_grad_(True)
    out = ops.camera_pipeline(e, mf.to(d), tr.to(d), 0.05)
    (out * w.to(d)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(e.grad.cpu().numpy(), e64.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert _rel_l2(e.grad.cpu().numpy(), e64.grad.numpy()) < 1e-5
    # the torch formulation of the same chain (harness.py: the CPU / reference-shaped face) agrees too
    h = harness.decode_cameras(emb, 0.05).reshape(-1, 7)
    h = harness.transform_cameras(harness.mirror_cameras(h, None, mf.repeat(G)[:, None]), None, tr.repeat(G, 1))
    np.testing.assert_allclose(h.numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)


def _clip_batch(meshes, d, name="bird", B=2, T=2, G=3, H=64, Kh=8, R=2, seed=0):
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    from acfm_video_3d_reconstruction_amd.multiframe_step import MultiframeStep
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    v, f = meshes[name + "_v"], meshes[name + "_f"]
    N = B * T
    step = MultiframeStep(torch.tensor(v, device=d), torch.tensor(f, device=d),
                          torch.tensor(fps_lbs_logits(v, Kh), device=d), num_training_frames=12, img_size=H,
                          num_guesses=G, num_lbs=Kh, scale_lr_decay=0.05).to(d)
    ext = float(np.abs(v).max())
    with torch.no_grad():                       # camera embeddings that put the mesh in the frame, all different
        for g_, emb in enumerate(step.cameras):
            c = make_cams(12, rng, extent=ext)
            emb.weight[:, 0] = torch.tensor((c[:, 0] - 1.0) / 0.05, device=d)     # decoded scale = relu(0.05 e0 + 1)
            emb.weight[:, 1:3] = torch.tensor(c[:, 1:3], device=d)
            emb.weight[:, 3:] = torch.tensor(c[:, 3:] * rng.uniform(0.5, 2.0, (12, 1)), device=d).float()
    gt_cams = torch.tensor(make_cams(N, rng, extent=ext), device=d)
    with torch.no_grad():
        gt_mask, _ = step.renderer(step.solver.mean_v[None].repeat(N, 1, 1), step.faces1[None].expand(N, -1, -1),
                                   gt_cams)
        gt_mask = (gt_mask > 0.5).float()
    flows = torch.randn(B, T, H, H, 2, device=d)
    flows[:, :, : H // 4] = 0                                        # zero-flow pixels drop their vertices (loss_utils.py:461)
    batch = dict(masks=gt_mask, edts_barrier=IU.compute_dt(gt_mask, norm=False)[:, None].contiguous(),
                 boundaries=IU.compute_boundaries(gt_mask), frames_idx=torch.tensor([[0, 1], [7, 3]], device=d)[:B],
                 mirror_flag=torch.tensor([0, 1, 1, 0], device=d)[:N],
                 transforms=torch.tensor([[1.1, 0.03, -0.02, 1.], [1., 0, 0, 0], [0.9, -0.05, 0.04, 1.], [1., 0, 0, 0]],
                                         device=d)[:N],
                 optical_flows=flows)
    delta = (0.02 * torch.randn(N, Kh, 3, device=d))
    tex = torch.rand(N, f.shape[0], R, R, 3, device=d)
    imgs = torch.rand(N, 3, H, H, device=d)
    return step, batch, delta, tex, imgs


def _oracle_terms(step, batch, delta, tex, imgs, render_verts=None, render_cams=None):
    fi = batch["frames_idx"]
    cam_emb = torch.stack([e.weight.detach()[fi] for e in step.cameras]).reshape(len(step.cameras), -1, 7).cpu().numpy()
    o = step.opts
    return O.multiframe_forward_terms(
        cam_emb, batch["mirror_flag"].cpu().numpy(), batch["transforms"].cpu().numpy(),
        step.lbs.detach().cpu().numpy(), step.mean_v.detach().cpu().numpy(), step.faces1.cpu().numpy(),
        delta.detach().cpu().numpy(), batch["masks"].cpu().numpy(), batch["edts_barrier"].cpu().numpy(),
        batch["boundaries"].cpu().numpy(), batch["optical_flows"].cpu().numpy(),
        None if tex is None else tex.detach().cpu().numpy(), None if imgs is None else imgs.cpu().numpy(),
        render_verts=None if render_verts is None else render_verts.detach().cpu().numpy(),
        render_cams=None if render_cams is None else render_cams.detach().cpu().numpy(),
        num_frames=o.num_frames, of_loss_wt=o.of_loss_wt, mask_loss_wt=o.mask_loss_wt, rigid_wt=o.rigid_wt,
        deform_reg_wt=o.deform_reg_wt, handle_deform_reg_wt=o.handle_deform_reg_wt,
        boundaries_reg_wt=o.boundaries_reg_wt, edt_reg_wt=o.edt_reg_wt, bdt_reg_wt=o.bdt_reg_wt,
        triangle_reg_wt=o.triangle_reg_wt, tex_loss_wt=o.tex_loss_wt, scale_lr_decay=o.scale_lr_decay)


@pytest.mark.parametrize("name,handle_wt", [("bird", 0.0), ("horse", 0.3)])
def test_multiframe_forward_terms_vs_oracle(meshes, name, handle_wt):
    """f2 (+ a17, a20 in composition): every named term of MultiframeStep.forward against the oracle's
    composition of main.py:523-765 -- cameras, deformed vertices, per-hypothesis mask / silhouette-consistency
    / optical-flow / texture terms in their [G, B*T] layout (incl. of_loss.repeat, the mirrored texture pass,
    the literal texture-cycle reshape), hypothesis probabilities, priors and the total: 1e-5."""
    d = _d()
    step, batch, delta, tex, imgs = _clip_batch(meshes, d, name=name)
    step.opts.handle_deform_reg_wt = handle_wt
    with torch.no_grad():
        total, terms = step(batch, delta, textures=tex, imgs=imgs)
        again = step(batch, delta, textures=tex, imgs=imgs)[1]
    assert torch.equal(terms["pred_v"], again["pred_v"]) and torch.equal(terms["mask_loss"], again["mask_loss"])   # reproducible forward
    # the oracle's own float64 solve / camera chain are compared first (below); its rasteriser then gets the
    # product's float32 geometry (see oracle.multiframe_forward_terms: the render is discontinuous at depth swaps)
    ref = _oracle_terms(step, batch, delta, tex, imgs, render_verts=terms["pred_v"], render_cams=terms["cam_pred"])
    c = lambda t: t.detach().cpu().double().numpy()
    tol = dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c(terms["cam_pred"]), ref["cam_pred"].numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(c(terms["pred_v"]), ref["pred_v"].numpy(), rtol=0, atol=1e-5)
    for key, rkey in (("mask_loss", "mask_loss"), ("sil_cons_per_hyp", "sil_cons"), ("of_loss", "of_loss"),
                      ("tex_mse_per_hyp", "tex_mse"), ("total_per_hyp", "total_per_hyp"), ("probs", "probs")):
        np.testing.assert_allclose(c(terms[key]), ref[rkey].numpy(), err_msg=key, **tol)
    for key, rkey in (("rigid", "rigid"), ("triangle", "triangle"), ("cycle", "cycle"), ("weighted", "weighted"),
                      ("camera_loss", "camera_loss")):
        np.testing.assert_allclose(float(terms[key]), float(ref[rkey]), err_msg=key, **tol)
    np.testing.assert_allclose(float(total), float(ref["loss"]), **tol)
    # the silhouette-only call of the released training loop (train_utils.py:252): no textures, delta dropped
    with torch.no_grad():
        total2, terms2 = step(batch, delta, drop_deform=True, detach_camera=True)
    ref2 = _oracle_terms(step, batch, torch.zeros_like(delta), None, None, render_verts=terms2["pred_v"],
                         render_cams=terms2["cam_pred"])
    np.testing.assert_allclose(c(terms2["pred_v"]), ref2["pred_v"].numpy(), rtol=0, atol=1e-5)
    ref2_loss = ref2["loss"] - step.opts.handle_deform_reg_wt * ref2["handle"] \
        + step.opts.handle_deform_reg_wt * O.deform_l2reg(delta.cpu().double())     # handle term sees delta_v_res (:612)
    np.testing.assert_allclose(float(total2), float(ref2_loss), **tol)
    np.testing.assert_allclose(c(terms2["total_per_hyp"]), ref2["total_per_hyp"].numpy(), **tol)


def test_multiframe_warmup_vs_oracle(meshes):
    """ShapeTrainer.warmup (main.py:438-520): the mean shape under all G camera hypotheses; per-hypothesis total and
    the probabilities written to the embeddings, against the oracle composition with delta = 0, no priors."""
    d = _d()
    step, batch, delta, tex, imgs = _clip_batch(meshes, d, seed=4)
    with torch.no_grad():
        loss, probs = step.warmup(batch)
        cam = step.hypothesis_cameras(batch["frames_idx"], batch["mirror_flag"], batch["transforms"])
    ref = _oracle_terms(step, batch, torch.zeros_like(delta), None, None, render_verts=step.solver.mean_v[None].repeat(4, 1, 1),
                        render_cams=cam)
    np.testing.assert_allclose(cam.cpu().numpy(), ref["cam_pred"].numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(probs.cpu().double().numpy(), ref["probs"].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(loss), float(ref["total_per_hyp"].mean()), rtol=1e-5, atol=1e-5)
    pw = step.prob_embeddings.weight[batch["frames_idx"]]                      # [B,T,G] <- probs [G,B*T]
    np.testing.assert_allclose(pw.detach().cpu().double().numpy(),
                               ref["probs"].reshape(len(step.cameras), *batch["frames_idx"].shape).permute(1, 2, 0).numpy(),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("optimize_camera", [True, False])
def test_refinement_first_iteration_vs_oracle(meshes, optimize_camera):
    """f3: the loss of one refinement iteration (predictor.py:301-345) and its gradients with respect to the handle
    offsets and the raw camera parameters against the oracle (C raster backward + float64 autograd): values 1e-5,
    gradients 1e-4 of their scale and 1e-4 relative L2."""
    from acfm_video_3d_reconstruction_amd import image_utils as IU, ops
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    from acfm_video_3d_reconstruction_amd.refine import refine_clip, refine_total
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams
    d = _d()
    rng = np.random.default_rng(11)
    v, f = meshes["horse_v"], meshes["horse_f"]
    N, H, Kh = 4, 64, 8
    lbs = fps_lbs_logits(v, Kh)
    cams = make_cams(N, rng, extent=float(np.abs(v).max()))
    cams[:, 3:] *= rng.uniform(0.7, 1.5, (N, 1)).astype(np.float32)       # raw (non-unit) quaternions: the normalisation matters
    delta = rng.normal(0, 0.03, (N, Kh, 3)).astype(np.float32)
    faces = torch.tensor(f, device=d)[None].repeat(N, 1, 1).contiguous()
    solver = DeformSolver(torch.tensor(v, device=d), faces[0], torch.tensor(lbs, device=d))
    r = NeuralRenderer(H)
    with torch.no_grad():
        gt, _ = r(solver(torch.tensor(rng.normal(0, 0.05, (N, Kh, 3)).astype(np.float32), device=d)), faces,
                  ops.camera_normalize(torch.tensor(cams, device=d)))
        gt = (gt > 0.5).float()
    edt = IU.compute_dt_barrier(gt) if hasattr(IU, "compute_dt_barrier") else IU.compute_dt(gt, norm=False)
    edt = edt.reshape(N, 1, H, H).contiguous()
    bds = IU.compute_boundaries(gt)[:, :1000].contiguous()
    td = torch.tensor(delta, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    cam = ops.camera_normalize(tc) if optimize_camera else tc
    total, pred_v = refine_total(r, solver, td, cam, faces, gt, edt, bds)
    gd, gc = torch.autograd.grad(total, [td, tc])
    rt, rgd, rgc, rterms = O.refine_iteration(lbs, v, f, delta, cams, gt.cpu().numpy(), edt.cpu().numpy(),
                                              bds.cpu().numpy(), optimize_camera=optimize_camera,
                                              render_verts=pred_v.detach().cpu().numpy(),
                                              render_cams=cam.detach().cpu().numpy())
    np.testing.assert_allclose(pred_v.detach().cpu().numpy(), rterms["pred_v"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(float(total), float(rt), rtol=1e-5, atol=1e-6)
    for got, want, what in ((gd, rgd, "delta"), (gc, rgc, "cam")):
        got, want = got.cpu().numpy(), want.numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * np.abs(want).max(), err_msg=what)
        assert _rel_l2(got, want) < 1e-4, (what, _rel_l2(got, want))
    # and the loop itself takes exactly this step first: Adam's first update is -lr * sign(g) (|g| >> eps)
    pv, cam_out, d_out, hist = refine_clip(r, solver, torch.tensor(delta, device=d), torch.tensor(cams, device=d), faces,
                                           gt, edt, bds, num_optim_iter=1, optimize_camera=optimize_camera)
    np.testing.assert_allclose(hist[0], float(rt), rtol=1e-5, atol=1e-6)
    big = np.abs(rgd.numpy()) > 1e-3 * np.abs(rgd.numpy()).max()
    step = (d_out.cpu().numpy() - delta)[big]
    np.testing.assert_allclose(step, -5e-3 * np.sign(rgd.numpy()[big]), rtol=1e-3, atol=1e-6)


def _sharded_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acfm_video_3d_reconstruction_amd.sharding import frame_shard
    d = torch.device("cuda:0")
    m = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes.npz")))
    step, batch, delta, tex, imgs = _clip_batch(m, d, B=2, T=2, G=2, H=48, Kh=6, seed=9)
    step.opts.of_loss_wt = 0.0                                 # (see the test's docstring)
    T = step.opts.num_frames
    s, e = frame_shard(2, T, rank, world)                      # this rank's clips -> frames [s, e)
    loc = dict(masks=batch["masks"][s:e], edts_barrier=batch["edts_barrier"][s:e], boundaries=batch["boundaries"][s:e],
               frames_idx=batch["frames_idx"][s // T:e // T], mirror_flag=batch["mirror_flag"][s:e],
               transforms=batch["transforms"][s:e], optical_flows=batch["optical_flows"][s // T:e // T])
    ex = step.make_exchange(average=True)
    d_loc = delta[s:e].clone().requires_grad_(True)
    loss, _ = step(loc, d_loc, textures=tex[s:e], imgs=imgs[s:e], exchange=ex)
    loss.backward()
    tot = ex.finish(extra_scalars=loss.detach().reshape(1))
    if rank == 0:
        cam_g = torch.stack([e_.weight.grad for e_ in step.cameras])
        torch.save(dict(lbs=step.lbs.grad.cpu(), mean=step.mean_v.grad.cpu(), delta=d_loc.grad.cpu(), cams=cam_g.cpu(),
                        loss=(tot / world).cpu(), bytes=ex.bytes), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_multiframe_step_matches_the_full_batch(meshes, tmp_path):
    """SURVEY 8e on the real step: two ranks (gloo rehearsal, both on this GPU) take one clip each, run
    MultiframeStep.forward(exchange=...) and exchange [G = sum g delta^T | sum g | loss]; the handle-weight and
    mean-shape gradients equal the single-process full batch, per-frame gradients equal their rows of it.
    The optical-flow term is switched off here: the reference lays its per-clip values out with
    `of_loss.repeat(1, T)` (main.py:684-686), which for B > 1 hands frame n the loss of clip n % B -- a coupling
    ACROSS clips of the batch (reproduced literally, tests above) that no partition of the clips can preserve;
    every other term is per frame / per mesh and shards exactly (measured term by term: <= 2e-6)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "sh0.pt")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    d = _d()
    step, batch, delta, tex, imgs = _clip_batch(meshes, d, B=2, T=2, G=2, H=48, Kh=6, seed=9)
    step.opts.of_loss_wt = 0.0
    dl = delta.clone().requires_grad_(True)
    loss, _ = step(batch, dl, textures=tex, imgs=imgs)
    loss.backward()
    # d lbs = solve_backward(G) amplifies the rounding of G = sum g delta^T by the conditioning of the system (cond ~1e5,
    # SURVEY App-C): with G summed in float32 the two summation orders (per rank + all-reduce vs one batch) agreed to 5e-3
    # of the gradient's scale only.  G is now summed in double on every rank (acfm_deform_presolve_sums_f64), exchanged in
    # double and rounded to float32 ONCE behind the all-reduce -- as the single-process backward rounds its own double
    # sum -- so both paths hand the solve's backward the same bits: measured 2e-8 (lbs) and 6e-8 (mean shape)
    for key, ref, tol in (("lbs", step.lbs.grad, 2e-6), ("mean", step.mean_v.grad, 2e-6)):
        ref = ref.cpu().numpy()
        np.testing.assert_allclose(got[key].numpy(), ref, rtol=0, atol=tol * np.abs(ref).max(), err_msg=key)
        assert _rel_l2(got[key].numpy(), ref) < tol, (key, _rel_l2(got[key].numpy(), ref))
    np.testing.assert_allclose(float(got["loss"]), float(loss), rtol=1e-5)
    # per-frame parameters: rank 0's rows (frames 0, 1 of the full batch; its loss is the mean over HALF the frames)
    np.testing.assert_allclose(got["delta"].numpy() / 2, dl.grad[:2].cpu().numpy(), rtol=0,
                               atol=2e-4 * float(dl.grad.abs().max()))
    cam_ref = torch.stack([e_.weight.grad for e_ in step.cameras]).cpu().numpy()
    rows = batch["frames_idx"][0].cpu().numpy()
    np.testing.assert_allclose(got["cams"].numpy()[:, rows] / 2, cam_ref[:, rows], rtol=0, atol=2e-4 * np.abs(cam_ref).max())
    assert got["bytes"] == 8 * (642 * 6 + 3 * 642 + 1)


def test_exchange_buffer_written_in_place_with_direct_terms(meshes):
    """sharding.SharedShapeExchange on one process (world 1): the deformation apply's backward writes G = sum g delta^T and
    sum g in double straight into the head of the exchange buffer; a DIRECT term on the mean shape (a prior on the
    template), an extra shared parameter and two scalars ride in the same buffer.  finish() must give the gradients of
    plain autograd through the solver (lbs within the conditioning of the solve's backward, the rest to float rounding)."""
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.sharding import SharedShapeExchange
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _d()
    rng = np.random.default_rng(21)
    v, f = meshes["bird_v"], meshes["bird_f"]
    N, Kh = 5, 8
    delta = torch.tensor(rng.normal(0, 0.02, (N, Kh, 3)).astype(np.float32), device=d)
    w = torch.tensor(rng.normal(0, 1, (N, v.shape[0], 3)).astype(np.float32), device=d)

    def make():
        lbs = torch.nn.Parameter(torch.tensor(fps_lbs_logits(v, Kh), device=d))
        mean = torch.nn.Parameter(torch.tensor(v, device=d))
        extra = torch.nn.Parameter(torch.tensor([0.3, -0.2], device=d))
        return DeformSolver(mean, torch.tensor(f, device=d), lbs), extra

    def loss_of(pred_v, solver, extra):
        return (pred_v * w).sum() * (1.0 + extra[0]) + 0.5 * (solver.mean_v ** 2).sum() + extra[1] ** 2

    ref_s, ref_e = make()
    ref_s.refresh()
    loss_of(ref_s(delta), ref_s, ref_e).backward()

    s, e = make()
    ex = SharedShapeExchange(s, extra_params=[e])
    pred = ex.apply(delta)
    loss = loss_of(pred, s, e)
    loss.backward()                                    # (P, mean) leaves, the direct term on mean_v, the extra parameter
    assert ex._filled and ex._store is not None        # the backward's kernel wrote the doubles in place
    sc = ex.finish(extra_scalars=torch.stack([loss.detach(), loss.detach() * 2]))
    assert ex.bytes == 8 * (v.shape[0] * Kh + 3 * v.shape[0] + 2 + 2)
    np.testing.assert_allclose(sc.cpu().numpy(), [float(loss), 2 * float(loss)], rtol=1e-6)
    np.testing.assert_allclose(e.grad.cpu().numpy(), ref_e.grad.cpu().numpy(), rtol=1e-6)
    gm, rm = s.mean_v.grad.cpu().numpy(), ref_s.mean_v.grad.cpu().numpy()
    np.testing.assert_allclose(gm, rm, rtol=0, atol=2e-6 * np.abs(rm).max())
    gl, rl = s.lbs.grad.cpu().numpy(), ref_s.lbs.grad.cpu().numpy()
    np.testing.assert_allclose(gl, rl, rtol=0, atol=2e-6 * np.abs(rl).max())
