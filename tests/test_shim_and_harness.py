"""CPU: host-side logic around the kernels -- PyTorch3D API shim, trainer harness helpers,
deformation solver -- against the oracle and the reference's golden outputs."""
import io
import os

import numpy as np
import torch

from conftest import load_golden
from oracle import oracle as O

from acfm_video_3d_reconstruction_amd import harness
from acfm_video_3d_reconstruction_amd import pytorch3d_shim as p3d
from acfm_video_3d_reconstruction_amd.deform import DeformSolver, deform_reference_formula
from acfm_video_3d_reconstruction_amd.nnutils import geom_utils, loss_utils
from acfm_video_3d_reconstruction_amd.pytorch3d_shim.structures import Meshes

T = torch.from_numpy


def test_meshes_packed_views(meshes):
    v, f = T(meshes["bird_v"]), T(meshes["bird_f"])
    ms = Meshes(verts=v[None].repeat(3, 1, 1), faces=f[None].repeat(3, 1, 1))
    assert len(ms) == 3 and not ms.isempty()
    assert ms.verts_packed().shape == (3 * 642, 3)
    fp = ms.faces_packed()
    assert fp.shape == (3 * 1280, 3) and fp[1280:2560].min() >= 642 and fp[2560:].max() == 3 * 642 - 1
    e1 = O.edges_packed(meshes["bird_f"])
    e = ms.edges_packed().numpy()
    assert e.shape == (3 * 1920, 2)
    np.testing.assert_array_equal(e[:1920], e1)
    np.testing.assert_array_equal(e[1920:3840], e1 + 642)
    ms2 = Meshes(verts=[v, v[:100]], faces=[f, f[:10] % 100])     # list form, unequal sizes
    assert ms2.verts_packed().shape[0] == 742 and ms2.verts_padded().shape == (2, 642, 3)
    assert list(ms2.num_verts_per_mesh()) == [642, 100]


def test_laplacian_smoothing_and_subdivide(meshes):
    v, f = T(meshes["horse_v"]), T(meshes["horse_f"])
    vb = (v[None].repeat(2, 1, 1) + 0.01 * torch.randn(2, 642, 3, generator=torch.Generator().manual_seed(0)))
    vb.requires_grad_(True)
    ms = Meshes(verts=vb, faces=f[None].repeat(2, 1, 1))
    a = p3d.loss.mesh_laplacian_smoothing(ms, "cot")
    vr = vb.detach().clone().requires_grad_(True)
    b = O.laplacian_smoothing_cot(vr, f)
    np.testing.assert_allclose(a.item(), b.item(), rtol=1e-5)
    a.backward()
    b.backward()
    np.testing.assert_allclose(vb.grad.numpy(), vr.grad.numpy(), rtol=1e-3, atol=1e-6)
    u = p3d.loss.mesh_laplacian_smoothing(ms, "uniform")
    assert u.item() > 0
    one = Meshes(verts=[v], faces=[f])
    out = p3d.ops.SubdivideMeshes(one)(one)
    ov, of = O.subdivide(meshes["horse_v"], meshes["horse_f"])
    np.testing.assert_allclose(out.verts_packed().numpy(), ov, atol=1e-7)
    np.testing.assert_array_equal(out.faces_packed().numpy(), of)


def test_load_obj_both_face_syntaxes():
    txt = "mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1/ 2/2/ 3/3/ \nf 1 3 4\nf 1 2 3 4\n"
    v, faces, aux = p3d.io.load_obj(io.StringIO(txt))
    assert v.shape == (4, 3) and faces.verts_idx.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3]]
    assert aux.verts_uvs.shape == (3, 2)


def test_transforms_and_camera_harness():
    g = torch.Generator().manual_seed(1)
    q = torch.nn.functional.normalize(torch.randn(6, 4, generator=g), dim=-1)
    R = p3d.transforms.quaternion_to_matrix(q)
    np.testing.assert_allclose(p3d.transforms.matrix_to_quaternion(R).numpy(),
                               p3d.transforms.standardize_quaternion(q).numpy(), atol=1e-5)
    a, b = q[:3, None], q[3:, None]
    np.testing.assert_allclose(p3d.transforms.quaternion_raw_multiply(a, b).numpy(),
                               geom_utils.hamilton_product(a, b).numpy(), atol=1e-7)
    # mirroring: projected x and z flip sign, y stays (main.py:113-125)
    cams = torch.cat([torch.rand(6, 1, generator=g) + 0.5, torch.rand(6, 2, generator=g) - 0.5, q], 1)
    X = torch.randn(6, 20, 3, generator=g)
    flag = torch.tensor([1, 0, 1, 1, 0, 1.])[:, None]
    cm = harness.mirror_cameras(cams, None, flag)
    p0, p1 = O.project_torch(X, cams), O.project_torch(X, cm)
    m = flag[:, :, None].bool().expand_as(p0)
    np.testing.assert_allclose(p1[..., 0][m[..., 0]].numpy(), -p0[..., 0][m[..., 0]].numpy(), atol=1e-5)
    np.testing.assert_allclose(p1[..., 1].numpy(), p0[..., 1].numpy(), atol=1e-5)
    np.testing.assert_allclose(p1[..., 2][m[..., 2]].numpy(), -p0[..., 2][m[..., 2]].numpy(), atol=1e-5)
    np.testing.assert_allclose(p1[~m].numpy(), p0[~m].numpy(), atol=0)
    # affine transform of the camera == affine transform of the projected points
    tr = torch.tensor([[1.3, 0.1, -0.2, 1.0]]).repeat(6, 1)
    ct = harness.transform_cameras(cams, None, tr)
    p2 = O.project_torch(X, ct)
    np.testing.assert_allclose(p2[..., :2].numpy(), (1.3 * p0[..., :2] + torch.tensor([0.1, -0.2])).numpy(), atol=1e-5)
    emb = torch.randn(4, 7, generator=g)
    dc = harness.decode_cameras(emb, 0.5)
    np.testing.assert_allclose(dc[:, 3:].norm(dim=-1).numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(dc[:, 0].numpy(), np.maximum(0.5 * emb[:, 0].numpy() + 1, 0) + 1e-12, atol=1e-7)
    L = torch.rand(5, 8, generator=g).requires_grad_(True)
    tot, probs, mean = harness.hypothesis_weighting(L)
    np.testing.assert_allclose(probs.sum(0).numpy(), 1.0, atol=1e-6)
    tot.backward()
    np.testing.assert_allclose(L.grad.numpy(), probs.numpy() / 8, atol=1e-7)   # weights are detached


def test_product_laplacian_rigid_and_solve_vs_reference_golden(meshes):
    g, gl, gs = load_golden("losses"), load_golden("laplacian"), load_golden("solve")
    for name in ("bird", "horse"):
        v, f = T(meshes[name + "_v"]), T(meshes[name + "_f"])
        L = geom_utils.mesh_laplacian(Meshes(verts=[v], faces=[f]), "cot").numpy()
        ref = np.zeros_like(L)
        ij = gl[name + "_ij"]
        ref[ij[:, 0], ij[:, 1]] = gl[name + "_val"]
        np.testing.assert_allclose(L, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    v, f = T(meshes["bird_v"]), T(meshes["bird_f"])
    faces4 = f[None].repeat(4, 1, 1)
    r = loss_utils.locally_rigid_fn(Meshes(verts=T(g["rigid_v"]), faces=faces4),
                                    Meshes(verts=v[None].repeat(4, 1, 1), faces=faces4))
    np.testing.assert_allclose(r.item(), g["rigid"], rtol=1e-5)
    for tag, tol_ref in (("bird_k16", 2e-4), ("horse_k16", 5e-3)):
        name = tag.split("_")[0]
        v, f = T(meshes[name + "_v"]), T(meshes[name + "_f"])
        logits, delta = T(gs[tag + "_logits"]), T(gs[tag + "_delta"])
        solver = DeformSolver(v, f, logits)
        out = solver(delta)
        assert np.abs(out.numpy() - gs[tag + "_pred_v"]).max() < tol_ref       # the reference's fp32 Cholesky
        L64 = O.laplacian_cot(v.double(), f)
        truth = O.deform_solve(logits, v, delta, L64).numpy()                  # fp64 formula = parity target
        assert np.abs(out.numpy() - truth).max() < 1e-4
        lit = deform_reference_formula(logits.double(), v.double(), delta.double(), L64)
        assert np.abs(lit.numpy() - truth).max() < 1e-9
        # gradients: d delta = P^T g, d mean = sum_n g, d lbs through the fp64 factorisation
        lg = torch.nn.Parameter(logits.clone())
        s2 = DeformSolver(v, f, lg)
        d2 = delta.clone().requires_grad_(True)
        mp = v.clone().requires_grad_(True)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
        (s2(d2, mean_override=mp) * w).sum().backward()
        lr = logits.double().clone().requires_grad_(True)
        dr = delta.double().clone().requires_grad_(True)
        (O.deform_solve(lr, v, dr, L64) * w.double()).sum().backward()
        np.testing.assert_allclose(d2.grad.numpy(), dr.grad.numpy(), rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(mp.grad.numpy(), w.sum(0).numpy(), rtol=1e-5, atol=1e-6)
        sc = np.abs(lr.grad.numpy()).max()
        np.testing.assert_allclose(lg.grad.numpy(), lr.grad.numpy(), rtol=1e-2, atol=1e-3 * sc)


def test_exported_never_called_surface_vs_reference(meshes):
    """thin wrappers the reference exports but never calls -- same outputs."""
    g, gl = load_golden("legacy"), load_golden("losses")
    tflow, images, dtf, vflow = T(g["tflow"]), T(g["images"]), T(g["dtf"]), T(g["vflow"])
    tol = dict(rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(geom_utils.sample_textures(tflow, images).numpy(), g["sample_textures"], **tol)
    np.testing.assert_allclose(geom_utils.sample_textures_v(vflow, images).numpy(), g["sample_textures_v"], **tol)
    np.testing.assert_allclose(loss_utils.texture_dt_loss(tflow, dtf).numpy(), g["texture_dt_loss"], **tol)
    np.testing.assert_allclose(loss_utils.texture_dt_loss_v(vflow, dtf, reduce=False).numpy(), g["texture_dt_loss_v"], **tol)
    np.testing.assert_allclose(loss_utils.texture_dt_loss_v(vflow, dtf).numpy(), g["texture_dt_loss_v_r"], **tol)
    np.testing.assert_allclose(loss_utils.mask_dt_loss(vflow, dtf).numpy(), g["mask_dt_loss"], **tol)
    np.testing.assert_allclose(loss_utils.triangle_loss(T(g["tri_v"]), T(g["tri_e2v"])).numpy(), g["triangle_loss"], **tol)
    np.testing.assert_allclose(loss_utils.entropy_loss(T(g["entropy_in"])).numpy(), g["entropy_loss"], **tol)
    np.testing.assert_allclose(loss_utils.texture_loss(images, images.flip(0), dtf[:, 0], dtf[:, 0].flip(0)).numpy(),
                               g["texture_loss"], **tol)
    v, f = T(meshes["bird_v"]), T(meshes["bird_f"])
    faces4 = f[None].repeat(4, 1, 1)
    te = loss_utils.template_edge_loss(Meshes(verts=T(gl["rigid_v"]), faces=faces4),
                                       Meshes(verts=v[None].repeat(4, 1, 1), faces=faces4))
    np.testing.assert_allclose(te.item(), g["template_edge_loss"], rtol=1e-5)
    loss, head = loss_utils.TexCycle()(tflow, torch.rand(2, 12, 2), torch.randint(-1, 12, (2, 8, 8)))
    assert loss.dim() == 0 and head.shape == (10, 2)


def test_edges_cache_survives_recycled_addresses(meshes):
    """The memoised edges_packed() of equal-sized batches is keyed on the faces tensor's address, strides,
    shape and version AND keeps that tensor alive: a second Meshes built from a same-shaped temporary with a
    different topology (the allocator is free to reuse an address once its tensor is gone) gets its own edges."""
    f0 = torch.from_numpy(meshes["bird_f"]).long()
    v = torch.from_numpy(meshes["bird_v"])[None].repeat(2, 1, 1)
    perm = torch.randperm(v.shape[1], generator=torch.Generator().manual_seed(5))
    want = {}
    for rnd in range(6):
        for name, f in (("a", f0), ("b", perm[f0])):           # same shape, different connectivity
            faces = f[None].repeat(2, 1, 1)                      # a temporary: freed at the end of the iteration
            e = Meshes(verts=v, faces=faces).edges_packed()
            ref = np.concatenate([O.edges_packed(f.numpy()), O.edges_packed(f.numpy()) + v.shape[1]], 0)
            np.testing.assert_array_equal(e.numpy(), ref)
            want[name] = e
            del faces, e
    assert not torch.equal(want["a"], want["b"])
    # a view of the same storage (what the trainer passes: faces1[None].expand(...)) hits the cache
    base = f0[None]
    e1 = Meshes(verts=v, faces=base.expand(2, -1, -1)).edges_packed()
    e2 = Meshes(verts=v, faces=base.expand(2, -1, -1)).edges_packed()
    assert e1 is e2


def test_rasterize_of_rejects_foreign_views():
    import pytest
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    r = NeuralRenderer(32)
    v, f = torch.zeros(1, 4, 3), torch.zeros(1, 2, 3, dtype=torch.int64)
    with pytest.raises(ValueError, match="R = diag"):
        r.rasterize_of(v, f, R=torch.eye(3)[None], T=torch.tensor([[0., 0., 2.732]]))
    with pytest.raises(ValueError, match="T = "):
        r.rasterize_of(v, f, R=torch.diag(torch.tensor([-1., 1., 1.]))[None], T=torch.tensor([[0., 0., 5.]]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # the reference's own view passes the check
        r.rasterize_of(v, f, R=torch.diag(torch.tensor([-1., 1., 1.]))[None], T=torch.tensor([[0., 0., 2.732]]))


def test_lazy_pix_to_face_dispatch():
    """ops.LazyPixToFace (the [N,H,W,K] pix_to_face whose slots beyond the nearest are produced on first use): `[..., 0]`,
    `[..., :1]` and detach() are views of the stored plane and never materialise; anything else builds the full tensor
    exactly once and then behaves like it.  Pure dispatch logic: host tensors stand in for the device ones."""
    import torch
    from acfm_video_3d_reconstruction_amd.ops import LazyPixToFace
    N, H, K = 2, 4, 5
    full = torch.arange(N * H * H * K, dtype=torch.int64).reshape(N, H, H, K)
    calls = []

    def make():
        calls.append(1)
        return full
    p = LazyPixToFace(full[..., :1].contiguous(), K, make)
    assert tuple(p.shape) == (N, H, H, K) and p.dtype == torch.int64 and not p.is_materialized
    assert torch.equal(p[..., 0], full[..., 0]) and torch.equal(p[..., :1], full[..., :1])
    assert torch.equal(p.detach()[..., 0], full[..., 0])
    assert not calls and not p.is_materialized
    assert torch.equal(p[..., 2], full[..., 2])              # another slot: the full tensor is built
    assert len(calls) == 1 and p.is_materialized
    assert torch.equal(p + 0, full) and int((p >= 0).sum()) == full.numel() and torch.equal(p[..., 0], full[..., 0])
    assert torch.equal(p.reshape(-1, K), full.reshape(-1, K))
    assert len(calls) == 1                                     # ... once


def test_lazy_grad_dispatch():
    """ops.LazyGrad (an image gradient a loss operator has not formed yet): take() hands the operator inputs to the
    render's backward only for the very image (storage, shape, version) and kind; detach() keeps it unformed; any
    other operation forms it exactly once and then behaves like the tensor."""
    import torch
    from acfm_video_3d_reconstruction_amd.ops import LazyGrad
    img = torch.rand(2, 4, 4)
    grad = torch.rand(2, 4, 4)
    calls = []

    def make():
        calls.append(1)
        return grad
    g = LazyGrad(img, "mask_losses", (img, None, None, 2, torch.ones(2, 4)), make)
    assert tuple(g.shape) == (2, 4, 4) and g.dtype == torch.float32 and not g.is_materialized
    assert g.take("tex_mse", img) is None and g.take("mask_losses", img.clone()) is None
    assert g.take("mask_losses", img.detach())[0] is img and g.detach() is g and not calls
    other = torch.rand(2, 4, 4)
    assert torch.equal(g + other, grad + other) and len(calls) == 1 and g.is_materialized     # e.g. autograd adding a second gradient
    assert g.take("mask_losses", img) is None                                                # formed: the plain path from now on
    assert torch.equal(g * 2, grad * 2) and torch.equal(g.reshape(2, -1), grad.reshape(2, -1)) and len(calls) == 1
    img2 = torch.rand(2, 4, 4)
    g2 = LazyGrad(img2, "mask_losses", (img2, None, None, 2, None), lambda: grad)
    img2.add_(1.0)                                                                            # the image changed under the gradient
    assert g2.take("mask_losses", img2) is not None          # (same version counter object: payload image IS img2) ...
    g3 = LazyGrad(img2, "mask_losses", (img2.clone(), None, None, 2, None), lambda: grad)
    assert g3.take("mask_losses", img2) is None              # ... a different storage is refused


def test_lazy_pix_to_face_survives_batch_splits_and_scatter():
    """What nn.DataParallel's scatter does to every tensor argument (main.py:326, 718: Boundaries_Loss gets pix_to_face
    through it, also on ONE device) is `chunk` along the batch dimension: the parts of an unformed LazyPixToFace stay
    unformed, carry their rows of the visibility bitmap, answer `[..., 0]` from their rows of the stored plane and
    form the parent -- once -- only when another slot is read."""
    import copy
    import torch
    from acfm_video_3d_reconstruction_amd.ops import LazyPixToFace
    N, H, K, V = 6, 4, 5, 7
    full = torch.arange(N * H * H * K, dtype=torch.int64).reshape(N, H, H, K)
    vis = torch.arange(N * V, dtype=torch.uint8).reshape(N, V)
    calls = []

    def make():
        calls.append(1)
        return full
    p = LazyPixToFace(full[..., :1].contiguous(), K, make, vis)
    parts = p.chunk(3, 0)
    assert len(parts) == 3 and all(isinstance(q, LazyPixToFace) and tuple(q.shape) == (2, H, H, K) for q in parts)
    assert torch.equal(parts[1][..., 0], full[2:4, ..., 0]) and torch.equal(parts[1]._acfm_vis, vis[2:4]) and not calls
    whole = p.chunk(1, 0)[0]                                                        # one device: all rows, still unformed
    assert isinstance(whole, LazyPixToFace) and tuple(whole.shape) == (N, H, H, K) and torch.equal(whole._acfm_vis, vis)
    assert isinstance(p.split(N, 0)[0], LazyPixToFace) and isinstance(p[0:N], LazyPixToFace) and not calls
    a, b = p.split([4, 2], 0)
    assert isinstance(b, LazyPixToFace) and torch.equal(b[..., :1], full[4:, ..., :1]) and torch.equal(b._acfm_vis, vis[4:])
    q = p[1:3]
    assert isinstance(q, LazyPixToFace) and torch.equal(q[..., 0], full[1:3, ..., 0])
    one = p[2]
    assert isinstance(one, LazyPixToFace) and tuple(one.shape) == (H, H, K) and torch.equal(one[..., 0], full[2, ..., 0])
    assert not calls and not p.is_materialized
    assert torch.equal(parts[2][..., 3], full[4:6, ..., 3]) and len(calls) == 1 and p.is_materialized   # parent formed once
    assert torch.equal(one[..., 1], full[2, ..., 1]) and len(calls) == 1
    c = copy.deepcopy(p)
    assert type(c) is torch.Tensor and torch.equal(c, full)


def test_lazy_grad_views_and_sums_stay_unformed():
    """The reference hands the silhouette-loss operators views of the rendered mask (`mask.view(N, -1)`, `mask[:, None]`,
    loss_utils.py:18-32, 72-77) and sums the gradients of l1_loss and edt_loss on the same mask (main.py:644, 716):
    the view backwards and that sum keep an ops.LazyGrad unformed; the merged gradient names both references and the
    sum of the upstream gradients, each operator's columns masked by the references it had."""
    import torch
    from acfm_video_3d_reconstruction_amd.ops import LazyGrad
    N, H = 2, 4
    img = torch.rand(N, H, H)
    gt, edt = torch.rand(N, H * H), torch.rand(N, H * H)
    ga, gb = torch.rand(N, H, H), torch.rand(N, H, H)
    calls = []
    go1 = torch.tensor([[1., 2., 3., 4.]] * N)
    go2 = torch.tensor([[10., 20., 30., 40.]] * N)
    l1 = LazyGrad(img, "mask_losses", (img, gt, None, N, go1), lambda: (calls.append("a"), ga)[1])
    ed = LazyGrad(img, "mask_losses", (img, None, edt, N, go2), lambda: (calls.append("b"), gb)[1])
    # shape-only views (what ViewBackward / UnsqueezeBackward do to the gradient)
    v = l1.view(N, -1)
    assert type(v) is LazyGrad and tuple(v.shape) == (N, H * H) and not calls
    back = v.view(N, H, H)
    assert back.take("mask_losses", img)[0] is img
    u = l1.unsqueeze(1)
    assert type(u) is LazyGrad and tuple(u.shape) == (N, 1, H, H) and type(u.squeeze(1)) is LazyGrad and not calls
    assert u.take("mask_losses", img) is not None          # same storage, element count and version
    # the sum of two operators' gradients on the same mask
    s = l1 + ed
    assert type(s) is LazyGrad and not s.is_materialized and not calls
    m, g, e, rb, go = s.take("mask_losses", img)
    assert m is img and g is gt and e is edt and rb == N
    assert torch.equal(go, torch.tensor([[1., 2., 3., 40.]] * N))       # l1's columns 0-2, edt's column 3
    assert torch.equal(s + 0, ga + gb) and sorted(calls) == ["a", "b"]   # formed after all: the plain sum
    # different masks / references do not merge: the plain path
    calls.clear()
    other = LazyGrad(img, "mask_losses", (img.clone(), None, edt, N, go2), lambda: gb)
    assert type(l1_ := LazyGrad(img, "mask_losses", (img, gt, None, N, go1), lambda: ga) + other) is torch.Tensor and torch.equal(l1_, ga + gb)
    x = LazyGrad(img, "mask_losses", (img, gt, None, N, go1), lambda: ga)
    y = LazyGrad(img, "mask_losses", (img, gt.clone(), None, N, go2), lambda: gb)
    assert type(x + y) is torch.Tensor
    # a reshaped view formed later has the view's shape
    z = LazyGrad(img, "mask_losses", (img, gt, None, N, go1), lambda: ga).view(N, -1)
    assert torch.equal(z * 1, ga.reshape(N, -1))


def test_perceptual_texture_loss_constructs_around_lpips(monkeypatch):
    """loss_utils.py:359-383: PerceptualTextureLoss_v2 is glue around the third-party `lpips` package.  predictor.py:108 and
    main.py:334 construct it: with the package present (a stand-in module here) construction and the call work and
    reproduce the reference's arithmetic; without it the constructor raises an ImportError that names the package."""
    import sys
    import types
    import pytest
    import torch
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils
    monkeypatch.setitem(sys.modules, "lpips", None)
    with pytest.raises(ImportError, match="lpips"):
        loss_utils.PerceptualTextureLoss_v2()

    class FakeLPIPS(torch.nn.Module):
        def __init__(self, net="alex", lpips=False, spatial=False):
            super().__init__()
            assert net == "alex" and spatial is True and lpips is False
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, a, b):
            return self.w * (a - b).abs().mean(1, keepdim=True)       # [B,1,H,W] "distance map"
    fake = types.ModuleType("lpips")
    fake.LPIPS = FakeLPIPS
    monkeypatch.setitem(sys.modules, "lpips", fake)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    fn = loss_utils.PerceptualTextureLoss_v2()
    g = torch.Generator().manual_seed(0)
    pred, img = torch.rand(3, 3, 8, 8, generator=g), torch.rand(3, 3, 8, 8, generator=g)
    mp, mg = torch.rand(3, 8, 8, generator=g), (torch.rand(3, 8, 8, generator=g) > 0.4).float()
    per = fn(pred, img, mp, mg, reduce=False)
    m = mg[:, None]
    want = ((2 * pred * m - 1) - (2 * img * m - 1)).abs().mean(1, keepdim=True) * m
    want = want.mean(-2, keepdim=True).mean(-1, keepdim=True).squeeze(-1).squeeze(-1).squeeze(-1)
    assert per.shape == (3,) and torch.allclose(per, want, atol=1e-6)
    assert torch.allclose(fn(pred, img, mp, mg), want.mean(), atol=1e-6)


def test_solve_status_word_decoding():
    """ops.decode_solve_info: the status word of acfm_deform_solve.  0 = ok; ACFM_SOLVE_INFO_HANDOFF (bit 30) = an expired
    hand-off wait of the single-launch factorisation -- reported as such, not as 'pivot row 1073741823' --; otherwise
    1 + the first row of the tile with a non-positive pivot (also when both are set: the hand-off message wins, the
    pivot may be a consequence of the NaNs)."""
    import re
    from acfm_video_3d_reconstruction_amd import ops
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "acfm_hip.h")).read()
    assert int(re.search(r"#define ACFM_SOLVE_INFO_HANDOFF (0x[0-9a-fA-F]+)", hdr).group(1), 16) == ops.SOLVE_INFO_HANDOFF
    assert ops.decode_solve_info(0) is None
    assert "row 64" in ops.decode_solve_info(65) and "positive definite" in ops.decode_solve_info(65)
    for word in (ops.SOLVE_INFO_HANDOFF, ops.SOLVE_INFO_HANDOFF | 33):
        msg = ops.decode_solve_info(word)
        assert "hand-off" in msg and "positive definite" not in msg and "1073741823" not in msg
