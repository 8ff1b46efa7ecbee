"""CPU: the oracle against golden vectors produced by the reference's own importable code
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import torch

from conftest import load_golden
from oracle import oracle as O


def test_projection_bit_exact():
    g = load_golden("projection")
    for name in ("bird", "horse"):
        X, cams = g[name + "_X"], g["cams"]
        # fp32 with separately rounded mul/add == the reference's chain of torch kernels
        np.testing.assert_array_equal(O.project(X, cams, 0.0), g[name + "_withz0"])
        np.testing.assert_array_equal(O.project(X, cams, 5.0), g[name + "_withz5"])
        np.testing.assert_array_equal(O.project(X, cams, 0.0)[..., :2], g[name + "_xy"])
        pt = O.project_torch(torch.from_numpy(X), torch.from_numpy(cams), 5.0).numpy()
        np.testing.assert_allclose(pt, g[name + "_withz5"], rtol=0, atol=1e-6)


def test_cot_laplacian(meshes):
    g = load_golden("laplacian")
    for name in ("bird", "horse", "cow"):
        v = torch.from_numpy(meshes[name + "_v"])
        f = torch.from_numpy(meshes[name + "_f"])
        L = O.laplacian_cot(v, f).numpy()
        ref = np.zeros_like(L)
        ij = g[name + "_ij"]
        ref[ij[:, 0], ij[:, 1]] = g[name + "_val"]
        scale = np.abs(ref).max()
        np.testing.assert_allclose(L, ref, rtol=0, atol=2e-5 * scale)


def test_deform_solve_vs_reference_fp32(meshes):
    g = load_golden("solve")
    # the reference's own fp32 Cholesky is only ~1e-3 accurate on the horse;
    # the fp64 oracle must agree with it to that level, and much better on the bird.
    for tag, tol in (("bird_k16", 2e-4), ("bird_k32", 2e-4), ("horse_k16", 5e-3)):
        name = tag.split("_")[0]
        v = torch.from_numpy(meshes[name + "_v"])
        f = torch.from_numpy(meshes[name + "_f"])
        L = O.laplacian_cot(v.double(), f)
        out = O.deform_solve(torch.from_numpy(g[tag + "_logits"]), v,
                             torch.from_numpy(g[tag + "_delta"]), L)
        err = np.abs(out.numpy() - g[tag + "_pred_v"]).max()
        assert err < tol, (tag, err)
        # closed form v = vbar + M^-1 A^T delta == as-written formula
        A = torch.softmax(torch.from_numpy(g[tag + "_logits"]).double(), 0).t()
        M = L.t() @ L + A.t() @ A
        P = torch.linalg.solve(M, A.t())
        closed = v.double()[None] + P[None] @ torch.from_numpy(g[tag + "_delta"]).double()
        assert (closed - out).abs().max() < 1e-8


def test_losses():
    g = load_golden("losses")
    T = torch.from_numpy
    pred, gt, edt = T(g["mask_pred"]), T(g["mask_gt"]), T(g["edt"])
    tol = dict(rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.l1_loss(pred, gt, reduce=False), g["l1"], **tol)
    np.testing.assert_allclose(O.l1_loss(pred, gt), g["l1_r"], **tol)
    np.testing.assert_allclose(O.iou(pred, gt, reduce=False), g["iou"], **tol)
    np.testing.assert_allclose(O.iou_loss(pred, gt, reduce=False), g["iou_loss"], **tol)
    np.testing.assert_allclose(O.iou_loss(pred, gt), g["iou_loss_r"], **tol)
    np.testing.assert_allclose(O.edt_loss(pred, edt, reduce=False), g["edt_loss"], **tol)
    np.testing.assert_allclose(O.edt_loss(pred, edt), g["edt_loss_r"], **tol)
    np.testing.assert_allclose(O.kp_l2_loss(T(g["kp_pred"]), T(g["kp_gt"]), "none"), g["kp_l2"], **tol)
    np.testing.assert_allclose(O.kp_l2_loss(T(g["kp_pred"]), T(g["kp_gt"])), g["kp_l2_r"], **tol)
    np.testing.assert_allclose(O.deform_l2reg(T(g["deform_in"])), g["deform_l2reg"], **tol)
    np.testing.assert_allclose(O.quat_loss_geodesic(T(g["q1"]), T(g["q2"])), g["quat_geo"], **tol)


def test_bds_and_rigid(meshes):
    g = load_golden("losses")
    T = torch.from_numpy
    faces = T(meshes["bird_f"])[None].repeat(4, 1, 1)
    out = O.bds_loss(T(g["bds_verts"]), T(g["bds"]), faces, T(g["bds_p2f"]), reduce=False)
    np.testing.assert_allclose(out, g["bds_loss"], rtol=1e-5, atol=1e-5)
    e = T(O.edges_packed(meshes["bird_f"]))
    assert e.shape == (1920, 2)
    vt = T(meshes["bird_v"])[None].repeat(4, 1, 1)
    np.testing.assert_allclose(O.locally_rigid(T(g["rigid_v"]), vt, e), g["rigid"], rtol=1e-5)


def test_optical_flow_loss(meshes):
    g = load_golden("losses")
    T = torch.from_numpy
    faces = T(meshes["bird_f"])[None, None].repeat(2, 2, 1, 1)
    loss, of_pred, vis = O.optical_flow_loss(T(g["of_meshes"]), faces, T(g["of_cams"]),
                                             T(g["of_flows"]), T(g["of_p2f"]), reduce=False)
    np.testing.assert_array_equal(vis.numpy(), g["of_vis"])
    np.testing.assert_allclose(of_pred.numpy(), g["of_pred"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss.numpy(), g["of_loss"], rtol=1e-5, atol=1e-6)


def test_subdivide_counts(meshes):
    v, f = O.subdivide(meshes["horse_v"], meshes["horse_f"])
    assert v.shape == (2562, 3) and f.shape == (5120, 3)
    assert O.edges_packed(f).shape[0] == 2562 + 5120 - 2
