"""GPU parity tests for the loss surface (loss_utils drop-in) vs the reference's golden
outputs and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _d():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_mask_losses_vs_reference_golden():
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    g = load_golden("losses")
    d = _d()
    T = lambda k: torch.from_numpy(g[k]).to(d)
    pred, gt, edt = T("mask_pred"), T("mask_gt"), T("edt")
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(L.l1_loss(pred, gt, reduce=False).cpu(), g["l1"], **tol)
    np.testing.assert_allclose(L.l1_loss(pred, gt).cpu(), g["l1_r"], **tol)
    np.testing.assert_allclose(L.iou(pred, gt, reduce=False).cpu(), g["iou"], **tol)
    np.testing.assert_allclose(L.iou_loss(pred, gt, reduce=False).cpu(), g["iou_loss"], **tol)
    np.testing.assert_allclose(L.iou_loss(pred, gt).cpu(), g["iou_loss_r"], **tol)
    np.testing.assert_allclose(L.edt_loss(pred, edt, reduce=False).cpu(), g["edt_loss"], **tol)
    np.testing.assert_allclose(L.edt_loss(pred, edt).cpu(), g["edt_loss_r"], **tol)
    l1, iou, e = L.fused_silhouette_losses(pred, gt, edt)
    np.testing.assert_allclose(l1.cpu(), g["l1"], **tol)
    np.testing.assert_allclose(iou.cpu(), g["iou"], **tol)
    np.testing.assert_allclose(e.cpu(), g["edt_loss"], **tol)
    np.testing.assert_allclose(L.kp_l2_loss(T("kp_pred"), T("kp_gt"), "none").cpu(), g["kp_l2"], **tol)
    np.testing.assert_allclose(L.deform_l2reg(T("deform_in")).cpu(), g["deform_l2reg"], **tol)
    np.testing.assert_allclose(L.quat_loss_geodesic(T("q1"), T("q2")).cpu(), g["quat_geo"], **tol)
    c1 = torch.cat([torch.rand(4, 3, device=d), T("q1")], 1)
    c2 = torch.cat([torch.rand(4, 3, device=d), T("q2")], 1)
    np.testing.assert_allclose(L.camera_loss(c1, c2, 0.05).cpu(), O.camera_loss(c1.cpu(), c2.cpu(), 0.05),
                               **tol)


def test_mask_losses_backward():
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    d = _d()
    torch.manual_seed(0)
    N, H = 3, 96
    pred = torch.rand(N, H, H)
    gt = (torch.rand(N, H, H) > 0.5).float()
    edt = torch.rand(N, 1, H, H) * 4
    w = torch.rand(3, N)

    def total(fn_l1, fn_iou, fn_edt, p, g, e, w):
        return (w[0] * fn_l1(p, g, reduce=False) + w[1] * fn_iou(p, g, reduce=False) +
                w[2] * fn_edt(p, e, reduce=False)).sum()

    p_ref = pred.clone().double().requires_grad_(True)
    total(O.l1_loss, O.iou_loss, O.edt_loss, p_ref, gt.double(), edt.double(), w.double()).backward()
    p_gpu = pred.clone().to(d).requires_grad_(True)
    total(L.l1_loss, L.iou_loss, L.edt_loss, p_gpu, gt.to(d), edt.to(d), w.to(d)).backward()
    np.testing.assert_allclose(p_gpu.grad.cpu().numpy(), p_ref.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_bds_loss_golden_and_grad(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    g = load_golden("losses")
    d = _d()
    faces = torch.from_numpy(meshes["bird_f"])[None].repeat(4, 1, 1).to(d)
    v = torch.from_numpy(g["bds_verts"]).to(d).requires_grad_(True)
    out = L.bds_loss(v, torch.from_numpy(g["bds"]).to(d), faces, torch.from_numpy(g["bds_p2f"]).to(d),
                     reduce=False)
    np.testing.assert_allclose(out.detach().cpu(), g["bds_loss"], rtol=1e-5, atol=1e-5)
    out.sum().backward()
    vr = torch.from_numpy(g["bds_verts"]).double().requires_grad_(True)
    O.bds_loss(vr, torch.from_numpy(g["bds"]).double(), faces.cpu(), torch.from_numpy(g["bds_p2f"]),
               reduce=False).sum().backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), vr.grad.numpy(), rtol=1e-4, atol=1e-6)
    m = L.Boundaries_Loss()(v, torch.from_numpy(g["bds"]).to(d), faces, torch.from_numpy(g["bds_p2f"]).to(d))
    np.testing.assert_allclose(m.item(), g["bds_loss"].mean(), rtol=1e-5)


def test_optical_flow_loss_golden(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import OF_NeuralRenderer
    g = load_golden("losses")
    d = _d()
    faces = torch.from_numpy(meshes["bird_f"])[None, None].repeat(2, 2, 1, 1).to(d)
    ren = OF_NeuralRenderer(32)
    T = lambda k: torch.from_numpy(g[k]).to(d)
    loss, of_pred, vis, _, _ = L.optical_flow_loss(T("of_meshes"), faces, T("of_cams"), T("of_flows"),
                                                   ren, T("of_p2f"), reduce=False)
    np.testing.assert_array_equal(vis.cpu().numpy(), g["of_vis"])
    np.testing.assert_allclose(of_pred.cpu().numpy(), g["of_pred"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), g["of_loss"], rtol=1e-5, atol=1e-6)
    # pix_to_face=None: visibility from the HIP hard rasteriser == oracle's
    loss2, _, vis2, _, _ = L.optical_flow_loss(T("of_meshes"), faces, T("of_cams"), T("of_flows"), ren,
                                               None, reduce=False)
    ref_loss, _, ref_vis = O.optical_flow_loss(torch.from_numpy(g["of_meshes"]), faces.cpu(),
                                               torch.from_numpy(g["of_cams"]),
                                               torch.from_numpy(g["of_flows"]), None, reduce=False)
    np.testing.assert_array_equal(vis2.cpu().numpy(), ref_vis.numpy())
    np.testing.assert_allclose(loss2.cpu().numpy(), ref_loss.numpy(), rtol=1e-5, atol=1e-6)


def test_masked_texture_mse():
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    d = _d()
    torch.manual_seed(1)
    N, H = 3, 40
    tex = torch.rand(N, 3, H, H)
    img = torch.rand(N, 3, H, H)
    m = (torch.rand(N, H, H) > 0.4).float() * torch.rand(N, H, H)
    w = torch.rand(N)
    a = tex.clone().double().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(a * m[:, None].double(), img.double() * m[:, None].double(),
                                       reduction="none").mean((1, 2, 3))
    (ref * w.double()).sum().backward()
    b = tex.clone().to(d).requires_grad_(True)
    out = L.masked_texture_mse(b, img.to(d), m.to(d))
    (out * w.to(d)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-4, atol=1e-9)


def test_on_device_edt_and_boundaries(meshes):
    """set_input's scipy / skimage prep on the GPU, against scipy itself."""
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    d = _d()
    rng = np.random.default_rng(5)
    H = 96
    yy, xx = np.mgrid[:H, :H]
    masks = np.stack([((yy - 40) ** 2 / 900.0 + (xx - 50) ** 2 / 400.0 < 1).astype(np.float32),
                      (rng.uniform(size=(H, H)) > 0.97).astype(np.float32),
                      np.zeros((H, H), np.float32),                       # no foreground at all
                      np.ones((H, H), np.float32)])
    masks[0, :3, :] = 1                                                     # touches the image border
    tm = torch.from_numpy(masks).to(d)
    for norm in (False, True):
        ref = np.stack([O.compute_dt(m, norm=norm) for m in masks]).astype(np.float32)
        np.testing.assert_array_equal(IU.compute_dt(tm, norm=norm).cpu().numpy(), ref)
    refb = np.stack([O.compute_dt_barrier(m) for m in masks[:2]]).astype(np.float32)
    np.testing.assert_allclose(IU.compute_dt_barrier(tm[:2]).cpu().numpy(), refb, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(IU.compute_boundaries(tm).cpu().numpy(), O.compute_boundaries(masks))
    np.testing.assert_array_equal(IU.compute_dt(tm[0], norm=False).cpu().numpy(), ref_single(masks[0]))


def ref_single(m):
    return O.compute_dt(m, norm=False).astype(np.float32)


def test_deform_apply_mfma_kernels(meshes):
    """a8: verts = mean + P delta and its backward on the f32 matrix cores vs an fp64 evaluation."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _d()
    g = load_golden("solve")
    for (N, Kh, V) in ((5, 16, 642), (64, 32, 642), (3, 7, 50)):
        torch.manual_seed(N)
        mean, P, delta = torch.randn(V, 3), torch.randn(V, Kh), 0.1 * torch.randn(N, Kh, 3)
        w = torch.randn(N, V, 3)
        a = [t.clone().to(d).requires_grad_(True) for t in (mean, P, delta)]
        out = ops.deform_apply(*a)
        (out * w.to(d)).sum().backward()
        b = [t.clone().double().requires_grad_(True) for t in (mean, P, delta)]
        ref = b[0][None] + torch.matmul(b[1][None], b[2])
        (ref * w.double()).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
        for x, y in zip(a, b):
            np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.numpy(), rtol=1e-4, atol=1e-4)
    # through the solver, against the reference's own fp32 output and the fp64 formula
    v, f = torch.from_numpy(meshes["bird_v"]), torch.from_numpy(meshes["bird_f"])
    logits, dl = torch.from_numpy(g["bird_k16_logits"]), torch.from_numpy(g["bird_k16_delta"])
    solver = DeformSolver(v.to(d), f.to(d), logits.to(d))
    out = solver(dl.to(d)).cpu()
    truth = O.deform_solve(logits, v, dl, O.laplacian_cot(v.double(), f))
    assert float((out.double() - truth).abs().max()) < 1e-4
    assert np.abs(out.numpy() - g["bird_k16_pred_v"]).max() < 2e-4


def test_deform_solve_native(meshes):
    """a8: the blocked fp64 Cholesky P = (L^T L + A^T A)^-1 A^T (csrc/acfm_solve.hip) and its
    lbs gradient against an fp64 torch evaluation of the reference's expression."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _d()
    cases = [("bird", 16), ("horse", 15), ("cow", 32), ("horse", 3)]
    for name, Kh in cases:
        v, f = torch.from_numpy(meshes[name + "_v"]), torch.from_numpy(meshes[name + "_f"])
        L = O.laplacian_cot(v.double(), f).float()
        logits = torch.tensor(fps_lbs_logits(v.numpy(), Kh))
        torch.manual_seed(Kh)
        w = torch.randn(v.shape[0], Kh)
        lg = logits.clone().to(d).requires_grad_(True)
        P = ops.deform_solve(L.to(d), lg, check=True)
        (P * w.to(d)).sum().backward()
        l64 = logits.double().requires_grad_(True)
        A = torch.softmax(l64, dim=0).t()
        M = L.double().t() @ L.double() + A.t() @ A
        ref = torch.cholesky_solve(A.t(), torch.linalg.cholesky(M))
        (ref * w.double()).sum().backward()
        scale = float(ref.abs().max())
        assert float((P.detach().cpu().double() - ref.detach()).abs().max()) < 1e-5 * scale, (name, Kh)
        gs = float(l64.grad.abs().max())
        assert float((lg.grad.cpu().double() - l64.grad).abs().max()) < 1e-4 * gs, (name, Kh)
    # small sizes that are not a multiple of the tile: V = 50, 33, 32, 5
    for V, Kh in ((50, 7), (33, 4), (32, 2), (5, 1)):
        torch.manual_seed(V)
        Lr = torch.randn(V, V)
        logits = torch.randn(V, Kh)
        lg = logits.clone().to(d).requires_grad_(True)
        P = ops.deform_solve(Lr.to(d), lg, check=True)
        P.square().sum().backward()
        l64 = logits.double().requires_grad_(True)
        A = torch.softmax(l64, dim=0).t()
        M = Lr.double().t() @ Lr.double() + A.t() @ A
        ref = torch.cholesky_solve(A.t(), torch.linalg.cholesky(M))
        ref.square().sum().backward()
        np.testing.assert_allclose(P.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-6 * float(ref.abs().max()))
        np.testing.assert_allclose(lg.grad.cpu().numpy(), l64.grad.numpy(), rtol=1e-3, atol=1e-5 * float(l64.grad.abs().max()))
    # a matrix that is not positive definite is reported (the reference's torch.cholesky raises)
    with pytest.raises(RuntimeError):
        ops.deform_solve(torch.zeros(40, 40, device=d), torch.full((40, 2), float("nan"), device=d), check=True)


def test_deform_solve_single_launch(meshes):
    """The factorisation as ONE launch of ticketed tile jobs (k_chol_tiles, csrc/acfm_solve.hip): more jobs than the
    device holds at once (V = 2562: 6 804 jobs on 256 resident workgroups, so the ticket order is what keeps it
    moving), bit-identical repeats (fixed summation order whatever the order the jobs run in), other work in flight
    on a second stream, and replay from a hipGraph (the sentinel fill is part of the captured sequence)."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _d()
    v, f = meshes["horse_v"], meshes["horse_f"]
    v2, f2 = O.subdivide(v, f)
    v2, f2 = np.asarray(v2, dtype=np.float32), np.asarray(f2)
    assert v2.shape[0] == 2562
    L = O.laplacian_cot(torch.from_numpy(v2).double(), torch.from_numpy(f2)).float()
    logits = torch.tensor(fps_lbs_logits(v2, 12))
    lg = logits.clone().to(d).requires_grad_(True)
    P = ops.deform_solve(L.to(d), lg, check=True)
    P.square().sum().backward()
    l64 = logits.double().requires_grad_(True)
    A = torch.softmax(l64, dim=0).t()
    M = L.double().t() @ L.double() + A.t() @ A
    ref = torch.cholesky_solve(A.t(), torch.linalg.cholesky(M))
    ref.square().sum().backward()
    assert float((P.detach().cpu().double() - ref.detach()).abs().max()) < 1e-5 * float(ref.abs().max())
    assert float((lg.grad.cpu().double() - l64.grad).abs().max()) < 1e-4 * float(l64.grad.abs().max())
    # repeats, with a memory-bound kernel running beside them on another stream
    vh, fh = torch.from_numpy(v), torch.from_numpy(f)
    Lh = O.laplacian_cot(vh.double(), fh).float().to(d)
    lh = torch.tensor(fps_lbs_logits(v, 15), device=d)
    first = ops.deform_solve(Lh, lh).clone()
    side, big = torch.cuda.Stream(), torch.empty(64 << 20, device=d)
    torch.cuda.synchronize()
    for _ in range(10):
        with torch.cuda.stream(side):
            big.add_(1.0)
        assert torch.equal(ops.deform_solve(Lh, lh), first)
    torch.cuda.synchronize()
    # graph replay
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = ops.deform_solve(Lh, lh)  # warm-up on the capture stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = ops.deform_solve(Lh, lh)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first)


def test_mesh_priors_hip(meshes):
    """a9 / a14 / a15 on the GPU: cot Laplacian, Laplacian smoothing (cot + uniform) and edge rigidity
    through the reference-shaped API, against the reference's golden outputs and the oracle."""
    from acfm_video_3d_reconstruction_amd import pytorch3d_shim as p3d
    from acfm_video_3d_reconstruction_amd.nnutils import geom_utils as G, loss_utils as L
    from acfm_video_3d_reconstruction_amd.pytorch3d_shim.structures import Meshes
    d = _d()
    gl, g = load_golden("laplacian"), load_golden("losses")
    for name in ("bird", "horse", "cow"):
        v, f = torch.from_numpy(meshes[name + "_v"]).to(d), torch.from_numpy(meshes[name + "_f"]).to(d)
        Lm = G.mesh_laplacian(Meshes(verts=[v], faces=[f]), "cot").cpu().numpy()
        ref = np.zeros_like(Lm)
        ij = gl[name + "_ij"]
        ref[ij[:, 0], ij[:, 1]] = gl[name + "_val"]
        np.testing.assert_allclose(Lm, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    v, f = torch.from_numpy(meshes["horse_v"]), torch.from_numpy(meshes["horse_f"])
    torch.manual_seed(0)
    vb = v[None].repeat(3, 1, 1) + 0.01 * torch.randn(3, 642, 3)
    a = vb.clone().to(d).requires_grad_(True)
    ms = Meshes(verts=a, faces=f[None].repeat(3, 1, 1).to(d))
    la = p3d.loss.mesh_laplacian_smoothing(ms, "cot")
    br = vb.clone().double().requires_grad_(True)
    lb = O.laplacian_smoothing_cot(br, f)
    np.testing.assert_allclose(la.item(), lb.item(), rtol=1e-5)
    la.backward()
    lb.backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), br.grad.numpy(), rtol=2e-3, atol=1e-5)
    # uniform variant vs the host (torch-op) path of the same shim
    a2 = vb.clone().to(d).requires_grad_(True)
    lu = p3d.loss.mesh_laplacian_smoothing(Meshes(verts=a2, faces=f[None].repeat(3, 1, 1).to(d)), "uniform")
    c2 = vb.clone().requires_grad_(True)
    lc = p3d.loss.mesh_laplacian_smoothing(Meshes(verts=c2, faces=f[None].repeat(3, 1, 1)), "uniform")
    np.testing.assert_allclose(lu.item(), lc.item(), rtol=1e-5)
    lu.backward()
    lc.backward()
    np.testing.assert_allclose(a2.grad.cpu().numpy(), c2.grad.numpy(), rtol=2e-3, atol=1e-6)
    # edge rigidity vs the reference's golden value, gradient vs torch
    bv, bf = torch.from_numpy(meshes["bird_v"]), torch.from_numpy(meshes["bird_f"])
    faces4 = bf[None].repeat(4, 1, 1)
    dv = torch.from_numpy(g["rigid_v"]).to(d).requires_grad_(True)
    r = L.locally_rigid_fn(Meshes(verts=dv, faces=faces4.to(d)),
                           Meshes(verts=bv[None].repeat(4, 1, 1).to(d), faces=faces4.to(d)))
    np.testing.assert_allclose(r.item(), g["rigid"], rtol=1e-5)
    r.backward()
    rv = torch.from_numpy(g["rigid_v"]).double().requires_grad_(True)
    O.locally_rigid(rv, bv[None].repeat(4, 1, 1).double(), torch.from_numpy(O.edges_packed(meshes["bird_f"]))).backward()
    np.testing.assert_allclose(dv.grad.cpu().numpy(), rv.grad.numpy(), rtol=1e-3, atol=1e-6)


def test_multiframe_step_harness(meshes):
    """ShapeTrainer.warmup / forward (main.py:438-520, 523-765) replayed on the HIP
    ops with G camera hypotheses, embeddings and stub network outputs."""
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    from acfm_video_3d_reconstruction_amd.multiframe_step import MultiframeStep
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams
    d = _d()
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    v, f = meshes["bird_v"], meshes["bird_f"]
    B, T, G, H, Kh = 2, 2, 3, 64, 15
    N = B * T
    step = MultiframeStep(torch.tensor(v, device=d), torch.tensor(f, device=d),
                          torch.tensor(fps_lbs_logits(v, Kh), device=d), num_training_frames=10, img_size=H,
                          num_guesses=G, num_lbs=Kh, scale_lr_decay=1.0).to(d)
    gt_cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
    with torch.no_grad():
        gt_mask, _ = step.renderer(step.solver.mean_v[None].repeat(N, 1, 1), step.faces1[None].expand(N, -1, -1), gt_cams)
        gt_mask = (gt_mask > 0.5).float()
    batch = dict(masks=gt_mask, edts_barrier=IU.compute_dt(gt_mask, norm=False)[:, None].contiguous(),
                 boundaries=IU.compute_boundaries(gt_mask), frames_idx=torch.tensor([[0, 1], [4, 5]], device=d),
                 mirror_flag=torch.tensor([0, 0, 1, 1], device=d),
                 transforms=torch.tensor([[1., 0, 0, 0]] * N, device=d),
                 optical_flows=torch.randn(B, T, H, H, 2, device=d))
    # warm-up: gradients reach the camera embeddings only, probabilities are written back
    loss, probs = step.warmup(batch)
    loss.backward()
    assert all(e.weight.grad is not None and e.weight.grad.abs().sum() > 0 for e in step.cameras)
    assert step.lbs.grad is None
    assert probs.shape == (G, N) and torch.allclose(probs.sum(0), torch.ones(N, device=d), atol=1e-5)
    pw = step.prob_embeddings.weight[batch["frames_idx"]]
    assert torch.allclose(pw.sum(-1), torch.ones(B, T, device=d), atol=1e-5)
    opt = torch.optim.Adam([p for e in step.cameras for p in e.parameters()], lr=1e-2)
    first = None
    for _ in range(8):
        opt.zero_grad()
        loss, _ = step.warmup(batch)
        loss.backward()
        opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < first
    # full forward: textures + mirrored texture render, priors, hypothesis weighting
    step.zero_grad()
    delta = (0.01 * torch.randn(N, Kh, 3, device=d)).requires_grad_(True)
    tex = torch.rand(N, f.shape[0], 4, 4, 3, device=d, requires_grad=True)
    imgs = torch.rand(N, 3, H, H, device=d)
    total, terms = step(batch, delta, textures=tex, imgs=imgs)
    total.backward()
    assert torch.isfinite(total) and all(torch.isfinite(t).all() for t in (delta.grad, tex.grad, step.lbs.grad))
    assert delta.grad.abs().sum() > 0 and tex.grad.abs().sum() > 0 and step.lbs.grad.abs().sum() > 0
    assert set(["mask", "sil_cons", "rigid", "triangle", "camera_loss", "probs", "tex_mse"]) <= set(terms)
    total2, _ = step(batch, delta, drop_deform=True, detach_camera=True)     # train_utils.py:252's call
    assert torch.isfinite(total2)
    # hypothesis dropping (main.py:541-570, 737-742; train_utils.py:236-241): only the k most probable
    # camera embeddings of every frame are rendered, their probabilities are scattered back
    step.opts.drop_hypothesis = True
    step.set_num_guesses(2)
    before = step.prob_embeddings.weight[batch["frames_idx"]].clone()          # [B,T,G] from the warm-up
    top2 = before.topk(2, dim=-1)[1]
    step.zero_grad()
    total3, terms3 = step(batch, delta, textures=tex, imgs=imgs, predicted_camera=torch.randn(N, 7, device=d))
    total3.backward()
    assert torch.isfinite(total3) and terms3["probs"].shape == (2, N) and "cam_loss" in terms3
    after = step.prob_embeddings.weight[batch["frames_idx"]]
    assert torch.allclose(after.sum(-1), torch.ones(B, T, device=d), atol=1e-5)
    assert int((after > 0).sum()) <= 2 * N and torch.equal((after > 0) | (after == 0), torch.ones_like(after, dtype=torch.bool))
    picked = torch.zeros_like(after, dtype=torch.bool).scatter_(-1, top2, True)
    assert float(after.detach()[~picked].abs().sum()) == 0.0                            # dropped hypotheses carry no probability
    for g, emb in enumerate(step.cameras):                                     # gradients reach only rendered cameras
        rows = emb.weight.grad[batch["frames_idx"]].abs().sum(-1) > 0          # [B,T]
        assert torch.equal(rows, picked[..., g])
    # deformation embeddings instead of the encoder's offsets (main.py:531-539, 763-765)
    step.opts.optimize_deform = True
    total4, terms4 = step(batch, delta)
    assert torch.isfinite(total4) and "deform_loss" in terms4


def test_multiframe_step_hipgraph_matches_eager(meshes):
    """The whole optimiser step (forward, backward, Adam) captured as one hipGraph (graphed.py)
    follows the eager trajectory: same losses, same parameters after several steps on changing
    inputs (tolerance: float atomics in the backward kernels make neither run bit-reproducible)."""
    import copy
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    from acfm_video_3d_reconstruction_amd.graphed import GraphedStep
    from acfm_video_3d_reconstruction_amd.multiframe_step import MultiframeStep
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits, make_cams
    d = _d()
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    v, f = meshes["bird_v"], meshes["bird_f"]
    B, T, G, H, Kh = 2, 2, 3, 64, 15
    N = B * T
    step_e = MultiframeStep(torch.tensor(v, device=d), torch.tensor(f, device=d),
                            torch.tensor(fps_lbs_logits(v, Kh), device=d), num_training_frames=10, img_size=H,
                            num_guesses=G, num_lbs=Kh, scale_lr_decay=1.0).to(d)
    step_g = copy.deepcopy(step_e)
    gt_cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
    with torch.no_grad():
        gt_mask, _ = step_e.renderer(step_e.solver.mean_v[None].repeat(N, 1, 1),
                                     step_e.faces1[None].expand(N, -1, -1), gt_cams)
        gt_mask = (gt_mask > 0.5).float()

    def make_inputs(seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        return dict(masks=gt_mask, edts_barrier=IU.compute_dt(gt_mask, norm=False)[:, None].contiguous(),
                    boundaries=IU.compute_boundaries(gt_mask),
                    frames_idx=torch.randint(0, 10, (B, T), generator=g).to(d),
                    mirror_flag=torch.randint(0, 2, (N,), generator=g).to(d),
                    transforms=torch.tensor([[1., 0, 0, 0]] * N, device=d),
                    optical_flows=torch.randn(B, T, H, H, 2, generator=g).to(d),
                    delta=(0.01 * torch.randn(N, Kh, 3, generator=g)).to(d),
                    tex=torch.rand(N, f.shape[0], 4, 4, 3, generator=g).to(d),
                    imgs=torch.rand(N, 3, H, H, generator=g).to(d))

    def fn(step):
        return lambda i: step(i, i["delta"], textures=i["tex"], imgs=i["imgs"])[0]

    # SGD + momentum: the update is linear in the gradients, so the run-to-run noise of the float
    # atomics stays small (Adam turns a sign flip of a ~0 gradient into a full lr-sized step)
    opt_g = torch.optim.SGD(step_g.parameters(), lr=1e-3, momentum=0.9)
    p0 = [q.detach().clone() for q in step_e.parameters()]
    runner = GraphedStep(fn(step_g), opt_g, make_inputs(0), grad_inputs=("delta", "tex"))
    for pe, pg in zip(step_e.parameters(), step_g.parameters()):       # construction does not train
        assert torch.equal(pe, pg)
    opt_e = torch.optim.SGD(step_e.parameters(), lr=1e-3, momentum=0.9)
    for it in range(4):
        inp = make_inputs(it)
        loss_g = runner(inp).item()
        gd = runner.grads["delta"].clone()
        opt_e.zero_grad(set_to_none=True)
        ie = {k: (t.clone().requires_grad_(True) if k in ("delta", "tex") else t) for k, t in inp.items()}
        loss_e = fn(step_e)(ie)
        loss_e.backward()
        opt_e.step()
        assert abs(loss_g - loss_e.item()) < 2e-3 * abs(loss_e.item()), (it, loss_g, loss_e.item())
        np.testing.assert_allclose(gd.cpu().numpy(), ie["delta"].grad.cpu().numpy(), rtol=5e-2,
                                   atol=2e-3 * float(ie["delta"].grad.abs().max()))
    moved = 0
    for (name, pe), pg, q0 in zip(step_e.named_parameters(), step_g.parameters(), p0):
        de, dg = (pe.detach() - q0).cpu().numpy(), (pg.detach() - q0).cpu().numpy()
        moved += int(np.abs(de).max() > 0)
        np.testing.assert_allclose(dg, de, rtol=0, atol=2e-2 * float(np.abs(de).max()) + 1e-10, err_msg=name)
    assert moved >= 5


def test_camera_pipeline_fused_vs_harness():
    """a17: decode + mirror_cameras + transform_cameras in one kernel (csrc/acfm_camera.hip) against
    the torch composition of harness.py (itself checked against the reference's golden vectors in
    test_shim_and_harness.py), values and gradients, all flag combinations."""
    from acfm_video_3d_reconstruction_amd import harness, ops
    d = _d()
    torch.manual_seed(3)
    G, N = 5, 12
    for decay in (1.0, 0.05):
        emb = torch.randn(G, N, 7, device=d)
        emb[0, 0, 0] = -30.0                      # relu off
        emb[1, 1, 3:] = 0.0                       # zero quaternion: the eps branch of normalize
        mirror = torch.tensor([0, 1] * (N // 2), device=d)
        tr = torch.rand(N, 4, device=d)
        tr[:, 3] = torch.tensor([0, 0, 1, 1] * (N // 4), device=d).float()
        w = torch.randn(G * N, 7, device=d)
        a = emb.clone().requires_grad_(True)
        out = ops.camera_pipeline(a, mirror, tr, decay)
        (out * w).sum().backward()
        b = emb.clone().requires_grad_(True)
        ref = harness.decode_cameras(b, decay).reshape(-1, 7)
        ref = harness.mirror_cameras(ref, None, mirror.repeat(G)[:, None])
        ref = harness.transform_cameras(ref, None, tr.repeat(G, 1))
        (ref * w).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
        ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
        keep = np.ones((G, N), bool)
        keep[1, 1] = False                        # d normalize at 0: 1/eps-scaled, compared separately below
        np.testing.assert_allclose(ga[keep], gb[keep], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ga[1, 1, :3], gb[1, 1, :3], rtol=1e-4, atol=1e-5)


def test_optical_flow_loss_fused_vs_torch_path(meshes):
    """a13: the fused optical-flow loss kernel (ops.of_loss, loss_only=True) against the op-by-op
    torch formulation of the same function (which test_gpu_losses checks against the reference's
    golden vectors), value and gradients to vertices and cameras; T = 2 and T = 3."""
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import OF_NeuralRenderer
    from acfm_video_3d_reconstruction_amd.synthetic import make_cams
    d = _d()
    v, f = meshes["bird_v"], meshes["bird_f"]
    H = 64
    ren = OF_NeuralRenderer(H)
    for b, t in ((3, 2), (2, 3)):
        rng = np.random.default_rng(10 * b + t)
        verts = torch.tensor(v[None, None] + 0.01 * rng.standard_normal((b, t) + v.shape), dtype=torch.float32, device=d)
        cams = torch.tensor(make_cams(b * t, rng, extent=float(np.abs(v).max())), device=d)
        flows = torch.tensor(rng.standard_normal((b, t, H, H, 2)), dtype=torch.float32, device=d)
        flows[:, :, : H // 4] = 0.0                                   # a band without GT flow: dropped vertices
        faces = torch.from_numpy(f).to(d)[None, None].expand(b, t, -1, -1)
        w = torch.tensor(rng.uniform(0.5, 1.5, (b, t - 1)), dtype=torch.float32, device=d)
        va, ca = verts.clone().requires_grad_(True), cams.clone().requires_grad_(True)
        la = L.optical_flow_loss(va, faces, ca, flows, ren, None, reduce=False, loss_only=True)
        (la * w).sum().backward()
        vb, cb = verts.clone().requires_grad_(True), cams.clone().requires_grad_(True)
        lb = L.optical_flow_loss(vb, faces, cb, flows, ren, None, reduce=False)[0]
        (lb * w).sum().backward()
        assert la.shape == lb.shape == (b, t - 1)
        np.testing.assert_allclose(la.detach().cpu().numpy(), lb.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(va.grad.cpu().numpy(), vb.grad.cpu().numpy(), rtol=1e-4,
                                   atol=1e-5 * float(vb.grad.abs().max()))
        np.testing.assert_allclose(ca.grad.cpu().numpy(), cb.grad.cpu().numpy(), rtol=1e-3,
                                   atol=1e-4 * float(cb.grad.abs().max()))
        assert float(lb.abs().sum()) > 0
        # the data loader's flows, flipped in time, masked and shared by G hypotheses INSIDE the kernel
        # == main.py:676-686's flip / multiply / repeat(G) materialised, bit for bit
        G = 2
        masks = torch.tensor(rng.uniform(size=(b * t, H, H)) > 0.3, dtype=torch.float32, device=d)
        prep = (torch.flip(flows, dims=[1]) * masks.reshape(b, t, H, H)[..., None]).repeat(G, 1, 1, 1, 1)
        vg = verts.repeat(G, 1, 1, 1) + 0.001 * torch.randn(G * b, t, *verts.shape[2:], device=d)
        cg, fg = cams.repeat(G, 1), faces.repeat(G, 1, 1, 1)
        v1, c1 = vg.clone().requires_grad_(True), cg.clone().requires_grad_(True)
        l1 = L.optical_flow_loss(v1, fg, c1, flows, ren, None, reduce=False, loss_only=True, flow_masks=masks, flip_t=True)
        v2, c2 = vg.clone().requires_grad_(True), cg.clone().requires_grad_(True)
        l2 = L.optical_flow_loss(v2, fg, c2, prep, ren, None, reduce=False, loss_only=True)
        assert torch.equal(l1, l2) and l1.shape == (G * b, t - 1)
        l1.sum().backward()
        l2.sum().backward()
        np.testing.assert_allclose(v1.grad.cpu().numpy(), v2.grad.cpu().numpy(), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(c1.grad.cpu().numpy(), c2.grad.cpu().numpy(), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        L.optical_flow_loss(va, faces, ca, flows, ren, None, reduce=False, flip_t=True)


def test_correlation_cost_volume_vs_oracle():
    """the cost volume of the flow network's Correlation layer (the reference's one
    native extension) in MaskFlownet's configuration, against the fp64 oracle; odd sizes, all md."""
    from acfm_video_3d_reconstruction_amd.correlation import Correlation
    d = _d()
    rng = np.random.default_rng(5)
    for (N, C, H, W, md) in ((2, 16, 24, 40, 4), (1, 37, 17, 19, 2), (3, 8, 16, 16, 1), (1, 196, 12, 20, 3)):
        f1 = rng.standard_normal((N, C, H, W)).astype(np.float32)
        f2 = rng.standard_normal((N, C, H, W)).astype(np.float32)
        layer = Correlation(pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1, corr_multiply=1)
        with torch.no_grad():
            out = layer(torch.tensor(f1, device=d), torch.tensor(f2, device=d))
        ref = O.correlation(f1, f2, md)
        assert out.shape == ref.shape == (N, (2 * md + 1) ** 2, H, W)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        # the centre channel is the plain channel-mean product
        np.testing.assert_allclose(out[:, ((2 * md + 1) ** 2) // 2].cpu().numpy(), (f1 * f2).mean(1), rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        Correlation(pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2)


def test_losses_with_references_shared_by_hypotheses(meshes):
    """gt / edt / images / boundary points given once per frame for G hypotheses ([N/G,...] against N
    predictions, ref_batch in the C ABI) == the trainer's ref.repeat(G, ...), values and gradients."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    d = _d()
    torch.manual_seed(9)
    G, n0, H = 3, 4, 32
    N = G * n0
    mask = torch.rand(N, H, H, device=d)
    gt, edt = (torch.rand(n0, H, H, device=d) > 0.5).float(), torch.rand(n0, 1, H, H, device=d)
    w = torch.randn(N, 3, device=d)
    a = mask.clone().requires_grad_(True)
    la = torch.stack(L.fused_silhouette_losses(a, gt, edt), 1)
    (la * w).sum().backward()
    b = mask.clone().requires_grad_(True)
    lb = torch.stack(L.fused_silhouette_losses(b, gt.repeat(G, 1, 1), edt.repeat(G, 1, 1, 1)), 1)
    (lb * w).sum().backward()
    assert torch.equal(la, lb) and torch.equal(a.grad, b.grad)
    tex, img = torch.rand(N, 3, H, H, device=d), torch.rand(n0, 3, H, H, device=d)
    ta = tex.clone().requires_grad_(True)
    ma = L.masked_texture_mse(ta, img, gt)
    (ma * w[:, 0]).sum().backward()
    tb = tex.clone().requires_grad_(True)
    mb = L.masked_texture_mse(tb, img.repeat(G, 1, 1, 1), gt.repeat(G, 1, 1))
    (mb * w[:, 0]).sum().backward()
    assert torch.equal(ma, mb) and torch.equal(ta.grad, tb.grad)
    V, P = 50, 70
    xy = torch.rand(N, V, 2, device=d) * 2 - 1
    bds = torch.cat([torch.rand(n0, P, 2, device=d) * 2 - 1, (torch.rand(n0, P, 1, device=d) > 0.2).float()], -1)
    vis = (torch.rand(N, V, device=d) > 0.3).to(torch.uint8)
    xa = xy.clone().requires_grad_(True)
    ba = ops.bds_loss_per_mesh(xa, bds, vis)
    (ba * w[:, 1]).sum().backward()
    xb = xy.clone().requires_grad_(True)
    bb = ops.bds_loss_per_mesh(xb, bds.repeat(G, 1, 1), vis)
    (bb * w[:, 1]).sum().backward()
    assert torch.equal(ba, bb)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)   # float atomics
    with pytest.raises(ValueError):
        L.fused_silhouette_losses(mask[:10], gt, edt)                 # 10 predictions, 4 references


@pytest.mark.gpu
def test_camera_pipeline_from_the_embedding_tables():
    """ops.camera_pipeline_tables (look-ups of the G per-hypothesis tables + stack / top-k gather + decode + mirror +
    transform, one kernel each way) == nn.Embedding look-ups, torch.stack / gather and ops.camera_pipeline (itself
    compared with the oracle's restatement of main.py:97-138, 551-584 in test_gpu_composed), gradients included: dense
    [frames,7] per table, zero outside the rows the batch looked up."""
    from acfm_video_3d_reconstruction_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    Gall, F, B, T = 6, 40, 4, 2
    N = B * T
    tabs = [torch.randn(F, 7, generator=g).to(d).requires_grad_(True) for _ in range(Gall)]
    refs = [t.detach().clone().requires_grad_(True) for t in tabs]
    fi = torch.randperm(F, generator=g)[:N].reshape(B, T).to(d)
    mf = torch.tensor([0, 1, 0, 0, 1, 1, 0, 1], device=d)
    tr = torch.cat([torch.rand(N, 1, generator=g) + 0.5, torch.rand(N, 2, generator=g) - 0.5,
                    (torch.rand(N, 1, generator=g) > 0.5).float()], 1).to(d)
    wts = torch.randn(Gall * N, 7, generator=g).to(d)
    for k, sel in ((Gall, None), (3, torch.stack([torch.randperm(Gall, generator=g)[:3] for _ in range(N)], 1)
                                  .reshape(3, B, T).to(d))):
        for t in tabs + refs:
            t.grad = None
        got = ops.camera_pipeline_tables(tabs, fi, mf, tr, 0.05, num_guesses=k, selected=sel)
        cams = torch.stack([torch.nn.functional.embedding(fi, r) for r in refs])
        if sel is not None:
            cams = torch.gather(cams, 0, sel[..., None].expand(-1, -1, -1, 7))
        want = ops.camera_pipeline(cams.reshape(k, -1, 7), mf, tr, 0.05)
        assert torch.equal(got, want)
        (got * wts[:k * N]).sum().backward()
        (want * wts[:k * N]).sum().backward()
        for a, b in zip(tabs, refs):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
            else:
                np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        ops.camera_pipeline_tables(tabs[:2], fi, mf, tr, 0.05, num_guesses=3)
    # ids outside the tables (a data-loader bug; nn.Embedding raises): never dereferenced -- that row's camera is NaN, it
    # gets no gradient, every other row and every table cell is untouched; check=True raises on the host instead
    for bad_fi, bad_sel in ((fi.clone().index_put_((torch.tensor(1), torch.tensor(0)), torch.tensor(F + 100000, device=d)), None),
                            (fi.clone().index_put_((torch.tensor(2), torch.tensor(1)), torch.tensor(-3, device=d)), None),
                            (fi, torch.full((3, B, T), 0, device=d).index_put_(
                                (torch.tensor(1), torch.tensor(3), torch.tensor(0)), torch.tensor(Gall + 7, device=d)))):
        k = Gall if bad_sel is None else 3
        for t in tabs:
            t.grad = None
        good = ops.camera_pipeline_tables(tabs, fi, mf, tr, 0.05, num_guesses=k,
                                          selected=None if bad_sel is None else torch.zeros_like(bad_sel)).detach()
        out = ops.camera_pipeline_tables(tabs, bad_fi, mf, tr, 0.05, num_guesses=k, selected=bad_sel)
        nan_rows = torch.isnan(out).any(1)
        assert int(nan_rows.sum()) == (Gall if bad_sel is None else 1) and torch.isnan(out[nan_rows]).all()
        assert torch.equal(out[~nan_rows].detach(), good[~nan_rows])
        torch.nan_to_num(out, nan=0.0).sum().backward()
        torch.cuda.synchronize()
        assert all(t.grad is None or torch.isfinite(t.grad).all() for t in tabs)
        with pytest.raises(IndexError):
            ops.camera_pipeline_tables(tabs, bad_fi, mf, tr, 0.05, num_guesses=k, selected=bad_sel, check=True)
    ops.camera_pipeline_tables(tabs, fi, mf, tr, 0.05, check=True)
    # the mirrored pose of decoded cameras (what the texture branch renders under): one kernel == harness's chain of
    # pytorch3d.transforms calls (property-tested on the CPU, and against the oracle inside the composed tests)
    from acfm_video_3d_reconstruction_amd import harness
    cams = ops.camera_pipeline_tables(tabs, fi, mf, tr, 0.05).detach()
    cams[::3, 3:] *= -1.0
    np.testing.assert_allclose(ops.camera_mirror(cams).cpu().numpy(), harness._mirrored_pose(cams).cpu().numpy(),
                               rtol=0, atol=0)


@pytest.mark.gpu
def test_texture_cycle_term_vs_the_oracle_restatement():
    """ops.texture_cycle == oracle.texture_cycle_loss (main.py:705-711 as written: regroup, reshape to [-1,R,R], mean L2
    norm of neighbouring rows) in float64, value and gradient; repeated entries (zero norms) give no gradient."""
    from acfm_video_3d_reconstruction_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(4)
    for B, T, F, R in ((1, 2, 3, 2), (2, 2, 7, 6), (3, 3, 5, 4), (8, 2, 1280, 6)):
        x = torch.rand(B * T, F, R, R, 3, generator=g)
        x[0, 0] = 0.25                                   # a constant atlas face: some zero norms
        a = x.to(d).requires_grad_(True)
        la = ops.texture_cycle(a, T)
        b = x.double().requires_grad_(True)
        lb = O.texture_cycle_loss(b, T)
        np.testing.assert_allclose(la.item(), lb.item(), rtol=2e-6)
        (la * 3.0).backward()
        (lb * 3.0).backward()
        gb = torch.nan_to_num(b.grad, nan=0.0)           # (float64 torch.norm backward at 0: masked to 0 as well)
        np.testing.assert_allclose(a.grad.cpu().numpy(), gb.numpy(), rtol=1e-4, atol=1e-9)
        assert torch.equal(ops.texture_cycle(a, T), la)  # fixed summation order


@pytest.mark.gpu
def test_hypothesis_cameras_with_dropped_hypotheses(meshes):
    """MultiframeStep.hypothesis_cameras on the GPU (one fused kernel from the embedding tables) == the reference's chain
    written out with torch ops (harness.decode_cameras / mirror_cameras / transform_cameras on the stacked look-ups,
    main.py:551-584), with all hypotheses and with only the k most probable ones of every frame (main.py:541-548,
    568-570: drop_hypothesis), gradients to every embedding table included."""
    from acfm_video_3d_reconstruction_amd import harness
    from acfm_video_3d_reconstruction_amd.multiframe_step import MultiframeStep
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _d()
    v, f = meshes["bird_v"], meshes["bird_f"]
    B, T, G = 3, 2, 5
    N = B * T
    step = MultiframeStep(torch.tensor(v, device=d), torch.tensor(f, device=d), torch.tensor(fps_lbs_logits(v, 4), device=d),
                          num_training_frames=20, img_size=32, num_guesses=G, num_lbs=4, scale_lr_decay=0.05,
                          drop_hypothesis=True).to(d)
    g = torch.Generator(device="cpu").manual_seed(2)
    with torch.no_grad():
        for emb in step.cameras:
            emb.weight.add_(0.3 * torch.randn(emb.weight.shape, generator=g).to(d))
        step.prob_embeddings.weight.copy_(torch.rand(step.prob_embeddings.weight.shape, generator=g).to(d))
    fi = torch.randperm(20, generator=g)[:N].reshape(B, T).to(d)
    mf = torch.tensor([0, 1, 1, 0, 0, 1], device=d)
    tr = torch.cat([torch.rand(N, 1, generator=g) + 0.5, torch.rand(N, 2, generator=g) - 0.5,
                    (torch.rand(N, 1, generator=g) > 0.5).float()], 1).to(d)
    w = torch.randn(G * N, 7, generator=g).to(d)
    for k in (G, 2):
        step.set_num_guesses(k)
        sel = step.selected_hypotheses(fi)
        assert (sel is None) == (k == G)
        step.zero_grad(set_to_none=True)
        got = step.hypothesis_cameras(fi, mf, tr, selected=sel)
        (got * w[:k * N]).sum().backward()
        g_fused = [e.weight.grad.clone() for e in step.cameras]
        step.zero_grad(set_to_none=True)
        cams = torch.stack([emb(fi) for emb in step.cameras])
        if sel is not None:
            cams = torch.gather(cams, 0, sel[..., None].expand(-1, -1, -1, 7))
        want = harness.decode_cameras(cams.reshape(k, -1, 7), 0.05).reshape(-1, 7)
        want = harness.mirror_cameras(want, None, mf.repeat(k)[:, None])
        want = harness.transform_cameras(want, None, tr.repeat(k, 1))
        (want * w[:k * N]).sum().backward()
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        for a, e in zip(g_fused, step.cameras):
            np.testing.assert_allclose(a.cpu().numpy(), e.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_hypothesis_total_matches_the_torch_formula():
    """harness.hypothesis_total == the reference's per-hypothesis total + softmax weighting (multiframe/main.py:716-746:
    total = sum w_t T_t, probs = softmax(-total, 0).detach(), weighted = (total * probs).sum(0).mean()), the logged sums
    and the gradients; one launch each way on the GPU, the same torch formula on the CPU path."""
    from acfm_video_3d_reconstruction_amd import harness
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    for G, N in ((1, 1), (6, 16), (8, 300), (3, 2000)):
        raw = [torch.rand(G * N, generator=g) * s for s in (1.0, 0.1, 3.0, 0.5, 2.0)]
        w = [1.0, 0.25, 2.0, 0.0, 0.5]
        ag, aw = [-1, 0, 0, 1, 1], [0.0, 0.1, 2.0, 0.5, 0.5]
        ts = [t.to(d).requires_grad_(i != 3) for i, t in enumerate(raw)]
        weighted, total, probs, aux, means = harness.hypothesis_total(ts, w, G, N, ag, aw)
        rs = [t.double().reshape(G, N).requires_grad_(True) for t in raw]
        rt = sum(wi * t for wi, t in zip(w, rs))
        rp = torch.softmax(-rt, dim=0).detach()
        rw = (rt * rp).sum(0).mean()
        np.testing.assert_allclose(total.cpu().numpy(), rt.detach().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(probs.cpu().numpy(), rp.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(probs.sum(0).cpu().numpy(), 1.0, atol=1e-6)
        np.testing.assert_allclose(weighted.item(), rw.item(), rtol=1e-6)
        np.testing.assert_allclose(aux[0].cpu().numpy(), (0.1 * rs[1] + 2.0 * rs[2]).detach().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(aux[1].cpu().numpy(), (0.5 * rs[3] + 0.5 * rs[4]).detach().numpy(), rtol=1e-6, atol=1e-7)
        want = [rw.item(), rt.mean().item(), (0.1 * rs[1] + 2.0 * rs[2]).mean().item(),
                (0.5 * rs[3] + 0.5 * rs[4]).mean().item()] + [t.mean().item() for t in rs]
        np.testing.assert_allclose(means.cpu().numpy()[:9], want, rtol=2e-6)
        gs = torch.autograd.grad(weighted * 2.0, [ts[0], ts[1], ts[2], ts[4]])
        rg = torch.autograd.grad(rw * 2.0, [rs[0], rs[1], rs[2], rs[4]])
        for a, b in zip(gs, rg):
            np.testing.assert_allclose(a.cpu().numpy().reshape(G, N), b.numpy(), rtol=1e-5, atol=1e-9)
        # the CPU path of the same wrapper (torch ops) agrees
        cw, ct, cp, ca, cm = harness.hypothesis_total([t.reshape(G, N) for t in raw], w, G, N, ag, aw)
        np.testing.assert_allclose(cw.item(), rw.item(), rtol=1e-5)
        np.testing.assert_allclose(cm.numpy()[:9], want, rtol=1e-5)
    with pytest.raises(ValueError):
        harness.hypothesis_total([ts[0]], [1.0, 2.0], G, N)


@pytest.mark.gpu
def test_combine_losses_matches_the_torch_formula():
    """combine_losses([T0 [N,4], T1 [N], T2 [N]], w) == mean_n(sum of weighted columns), gradients
    included (multiframe/main.py:716-765's elementwise tail as one launch each way)."""
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    d = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for N in (1, 5, 64, 300):
        t0 = torch.rand(N, 4, generator=g).to(d).requires_grad_(True)
        t1 = torch.rand(N, generator=g).to(d).requires_grad_(True)
        t2 = torch.rand(N, generator=g).to(d)                      # no gradient asked for
        w = [1.0, 0.0, -0.5, 0.1, 0.1, 2.0]
        tot = L.combine_losses([t0, t1, t2], w)
        ref = ((t0 * torch.tensor(w[:4], device=d)).sum(1) + w[4] * t1 + w[5] * t2).mean()
        assert abs(tot.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
        g0, g1 = torch.autograd.grad(tot * 3.0, [t0, t1])
        r0, r1 = torch.autograd.grad(ref * 3.0, [t0, t1])
        np.testing.assert_allclose(g0.cpu().numpy(), r0.cpu().numpy(), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(g1.cpu().numpy(), r1.cpu().numpy(), rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):
        L.combine_losses([t0], [1.0, 2.0])
