"""CPU: host-side logic around the kernels -- PyTorch3D API shim, trainer harness helpers,
deformation solver -- against the oracle and the reference's golden outputs."""
import io

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
