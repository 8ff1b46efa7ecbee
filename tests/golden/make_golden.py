"""Generates the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

It imports the reference's pure-torch modules (multiframe/nnutils/geom_utils.py and
loss_utils.py with an in-memory stub for the absent ``lpips`` package), replays the
deformation-solve snippet of multiframe/main.py:600-608 verbatim on torch-CPU, and stores
inputs + outputs as small .npz files.  The fixtures are data only; no reference source is
copied.  PyTorch3D (the rasterizer) is not importable here, so there are no raster goldens:
see oracle/acfm_oracle.c header ("parity unpinned" for that part).
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/multiframe"
OUT = os.path.dirname(os.path.abspath(__file__))

warnings.filterwarnings("ignore")
sys.path.insert(0, REF)
sys.modules["lpips"] = types.ModuleType("lpips")
from nnutils import geom_utils, loss_utils  # noqa: E402  (the reference's own modules)


def load_obj(path):
    """Minimal OBJ reader: accepts 'f a b c' and 'f a/a/ b/b/ c/c/' (bird.obj)."""
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            f.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.int64)


class DuckMesh:
    """The 4-method subset of pytorch3d Meshes that geom_utils.mesh_laplacian touches."""

    def __init__(self, verts, faces, edges=None, n=1):
        self._v, self._f, self._e, self._n = verts, faces, edges, n
        self.device = verts.device

    def isempty(self):
        return False

    def verts_packed(self):
        return self._v

    def faces_packed(self):
        return self._f

    def edges_packed(self):
        return self._e

    def __len__(self):
        return self._n


def fps_lbs_logits(verts, k, pp=16):
    """Stand-in for mesh_net.py:523-544 (geodesic FPS needs the absent `gdist`):
    Euclidean farthest-point handles, weights 1/d^16, log."""
    idx = [int(np.argmax(np.linalg.norm(verts - verts.mean(0), axis=1)))]
    d = np.linalg.norm(verts - verts[idx[0]], axis=1)
    for _ in range(k - 1):
        idx.append(int(np.argmax(d)))
        d = np.minimum(d, np.linalg.norm(verts - verts[idx[-1]], axis=1))
    idx = np.sort(np.asarray(idx))
    dist = np.linalg.norm(verts[:, None] - verts[None, idx], axis=-1)
    with np.errstate(divide="ignore"):
        w = 1.0 / dist ** pp
    w[np.isinf(w)] = 0
    mx = w.max(0)
    for i, j in enumerate(idx):
        w[j, i] = mx[i]
    return np.log(np.clip(w, 1e-12, None)).astype(np.float32)


def main():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    meshes = {}
    for name in ("bird", "horse", "cow"):
        v, f = load_obj(os.path.join(REF, "meshes", name + ".obj"))
        assert v.shape == (642, 3) and f.shape == (1280, 3)
        meshes[name + "_v"], meshes[name + "_f"] = v, f
    np.savez_compressed(os.path.join(OUT, "meshes.npz"), **meshes)

    # ---- (1) projection: geom_utils.orthographic_proj{,_withz}, quat_rotate
    cams = np.concatenate([rng.uniform(0.3, 1.5, (16, 1)), rng.uniform(-0.3, 0.3, (16, 2)),
                           rng.normal(size=(16, 4))], 1).astype(np.float32)
    cams[:8, 3:] /= np.linalg.norm(cams[:8, 3:], axis=1, keepdims=True)  # half unit, half not
    g = dict(cams=cams)
    for name in ("bird", "horse"):
        X = torch.from_numpy(meshes[name + "_v"])[None].repeat(16, 1, 1)
        X = X + 0.01 * torch.randn_like(X)
        c = torch.from_numpy(cams)
        g[name + "_X"] = X.numpy()
        g[name + "_withz0"] = geom_utils.orthographic_proj_withz(X, c, offset_z=0.).numpy()
        g[name + "_withz5"] = geom_utils.orthographic_proj_withz(X, c, offset_z=5.).numpy()
        g[name + "_xy"] = geom_utils.orthographic_proj(X, c).numpy()
        g[name + "_rot"] = geom_utils.quat_rotate(X, c[:, 3:]).numpy()
    np.savez_compressed(os.path.join(OUT, "projection.npz"), **g)

    # ---- (2) cot Laplacian: geom_utils.mesh_laplacian(.., 'cot')   (stored sparse)
    g = {}
    Ls = {}
    for name in ("bird", "horse", "cow"):
        m = DuckMesh(torch.from_numpy(meshes[name + "_v"]), torch.from_numpy(meshes[name + "_f"]))
        L = geom_utils.mesh_laplacian(m, "cot")
        Ls[name] = L
        nz = L.nonzero()
        g[name + "_ij"] = nz.numpy().astype(np.int32)
        g[name + "_val"] = L[nz[:, 0], nz[:, 1]].numpy()
    np.savez_compressed(os.path.join(OUT, "laplacian.npz"), **g)

    # ---- (3) deformation solve, replay of multiframe/main.py:586-609 on torch-CPU fp32
    g = {}
    for name, kh in (("bird", 16), ("bird", 32), ("horse", 16)):
        v = torch.from_numpy(meshes[name + "_v"])
        logits = torch.from_numpy(fps_lbs_logits(meshes[name + "_v"], kh))
        nb = 4
        delta_res = 0.02 * torch.randn(nb, kh, 3)
        lbs = torch.nn.functional.softmax(logits, dim=0).permute(1, 0)  # get_lbs().permute(1,0)
        lbs = lbs[None].repeat(nb, 1, 1)
        mean_v = v[None].repeat(nb, 1, 1)
        delta_v = lbs.bmm(mean_v) + delta_res
        L = Ls[name].repeat(nb, 1, 1)
        delta = torch.bmm(L, mean_v)
        A = lbs
        A_augm = L.permute(0, 2, 1).matmul(L) + A.permute(0, 2, 1).matmul(A)
        b = L.permute(0, 2, 1) @ delta + A.permute(0, 2, 1) @ delta_v
        u = torch.cholesky(A_augm)
        pred_v = torch.cholesky_solve(b, u)
        tag = "%s_k%d" % (name, kh)
        g[tag + "_logits"] = logits.numpy()
        g[tag + "_delta"] = delta_res.numpy()
        g[tag + "_pred_v"] = pred_v.numpy()
    np.savez_compressed(os.path.join(OUT, "solve.npz"), **g)

    # ---- (4) losses: multiframe/nnutils/loss_utils.py on fixed random inputs
    g = {}
    N, H = 4, 32
    pred = torch.rand(N, H, H)
    tgt = (torch.rand(N, H, H) > 0.5).float()
    edt = torch.rand(N, 1, H, H) * 5
    g.update(mask_pred=pred.numpy(), mask_gt=tgt.numpy(), edt=edt.numpy())
    g["l1"] = loss_utils.l1_loss(pred, tgt, reduce=False).numpy()
    g["l1_r"] = loss_utils.l1_loss(pred, tgt).numpy()
    g["iou"] = loss_utils.iou(pred, tgt, reduce=False).numpy()
    g["iou_loss"] = loss_utils.iou_loss(pred, tgt, reduce=False).numpy()
    g["iou_loss_r"] = loss_utils.iou_loss(pred, tgt).numpy()
    g["edt"] = edt.numpy()
    g["edt_loss"] = loss_utils.edt_loss(pred, edt, reduce=False).numpy()
    g["edt_loss_r"] = loss_utils.edt_loss(pred, edt).numpy()

    # bds_loss with P <= 1000 (permutation-invariant) and a synthetic pix_to_face
    V, F, P = 642, 1280, 200
    faces = torch.from_numpy(meshes["bird_f"])[None].repeat(N, 1, 1)
    verts2 = torch.rand(N, V, 2) * 2 - 1
    bds = torch.cat([torch.rand(N, P, 2) * 2 - 1, (torch.rand(N, P, 1) > 0.2).float()], -1)
    p2f = torch.randint(-1, F, (N, H, H, 3))
    p2f = torch.where(p2f >= 0, p2f + torch.arange(N)[:, None, None, None] * F, p2f)
    g.update(bds_verts=verts2.numpy(), bds=bds.numpy(), bds_p2f=p2f.numpy())
    g["bds_loss"] = loss_utils.bds_loss(verts2, bds, faces, p2f, reduce=False).numpy()

    # locally_rigid_fn
    from_edges = np.unique(np.sort(np.concatenate(
        [meshes["bird_f"][:, [1, 2]], meshes["bird_f"][:, [2, 0]], meshes["bird_f"][:, [0, 1]]]), 1), axis=0)
    vt = torch.from_numpy(meshes["bird_v"])[None].repeat(N, 1, 1)
    vd = vt + 0.02 * torch.randn_like(vt)
    e = torch.from_numpy(from_edges)
    ep = torch.cat([e + n * V for n in range(N)], 0)
    g["rigid_v"] = vd.numpy()
    g["rigid"] = loss_utils.locally_rigid_fn(DuckMesh(vd.reshape(-1, 3), None, ep, N),
                                             DuckMesh(vt.reshape(-1, 3), None, ep, N)).numpy()

    kp_pred = torch.rand(N, 15, 2) * 2 - 1
    kp_gt = torch.cat([torch.rand(N, 15, 2) * 2 - 1, (torch.rand(N, 15, 1) > 0.3).float()], -1)
    g.update(kp_pred=kp_pred.numpy(), kp_gt=kp_gt.numpy())
    g["kp_l2"] = loss_utils.kp_l2_loss(kp_pred, kp_gt, reduction="none").numpy()
    g["kp_l2_r"] = loss_utils.kp_l2_loss(kp_pred, kp_gt).numpy()
    dv = torch.randn(N, 16, 3)
    g["deform_in"] = dv.numpy()
    g["deform_l2reg"] = loss_utils.deform_l2reg(dv).numpy()
    q1 = torch.nn.functional.normalize(torch.randn(N, 4), dim=-1)
    q2 = torch.nn.functional.normalize(torch.randn(N, 4), dim=-1)
    g.update(q1=q1.numpy(), q2=q2.numpy())
    g["quat_geo"] = loss_utils.quat_loss_geodesic(q1, q2).numpy()

    # optical_flow_loss with a supplied pix_to_face (renderer duck-typed: only proj_fn used)
    b, t = 2, 2
    ren = types.SimpleNamespace(proj_fn=geom_utils.orthographic_proj_withz)
    of_meshes = (torch.from_numpy(meshes["bird_v"])[None, None] +
                 0.02 * torch.randn(b, t, V, 3))
    of_cams = torch.from_numpy(cams[:b * t]).clone()
    of_cams[:, 3:] = torch.nn.functional.normalize(of_cams[:, 3:], dim=-1)
    of_cams[:, 0] = 1.2
    of_faces = torch.from_numpy(meshes["bird_f"])[None, None].repeat(b, t, 1, 1)
    flows = torch.randn(b, t, H, H, 2) * (torch.rand(b, t, H, H, 1) > 0.3).float()
    of_p2f = torch.randint(-1, F, (b * t, H, H, 2))
    of_p2f = torch.where(of_p2f >= 0, of_p2f + torch.arange(b * t)[:, None, None, None] * F, of_p2f)
    loss, of_pred, vis, _, _ = loss_utils.optical_flow_loss(of_meshes, of_faces, of_cams, flows,
                                                            ren, of_p2f, reduce=False)
    g.update(of_meshes=of_meshes.numpy(), of_cams=of_cams.numpy(), of_flows=flows.numpy(),
             of_p2f=of_p2f.numpy(), of_loss=loss.numpy(), of_pred=of_pred.numpy(),
             of_vis=vis.numpy())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **g)

    # ---- (5) exported-but-never-called surface
    g = {}
    Fn, Tt, Hh = 12, 3, 16
    tflow = torch.rand(2, Fn, Tt, Tt, 2) * 2 - 1
    images = torch.rand(2, 3, Hh, Hh)
    dtf = torch.rand(2, 1, Hh, Hh)
    vflow = torch.rand(2, 20, 2) * 2 - 1
    g.update(tflow=tflow.numpy(), images=images.numpy(), dtf=dtf.numpy(), vflow=vflow.numpy())
    g["sample_textures"] = geom_utils.sample_textures(tflow, images).numpy()
    g["sample_textures_v"] = geom_utils.sample_textures_v(vflow, images).numpy()
    g["texture_dt_loss"] = loss_utils.texture_dt_loss(tflow, dtf).numpy()
    g["texture_dt_loss_v"] = loss_utils.texture_dt_loss_v(vflow, dtf, reduce=False).numpy()
    g["texture_dt_loss_v_r"] = loss_utils.texture_dt_loss_v(vflow, dtf).numpy()
    g["mask_dt_loss"] = loss_utils.mask_dt_loss(vflow, dtf).numpy()
    tv = torch.rand(2, 9, 3)
    e2v = torch.randint(0, 9, (2, 7, 4))
    g.update(tri_v=tv.numpy(), tri_e2v=e2v.numpy())
    g["triangle_loss"] = loss_utils.triangle_loss(tv, e2v).numpy()
    pa = torch.softmax(torch.randn(5, 30), 1)
    g["entropy_in"] = pa.numpy()
    g["entropy_loss"] = loss_utils.entropy_loss(pa).numpy()
    g["template_edge_loss"] = loss_utils.template_edge_loss(DuckMesh(vd.reshape(-1, 3), None, ep, N),
                                                            DuckMesh(vt.reshape(-1, 3), None, ep, N)).numpy()
    g["texture_loss"] = loss_utils.texture_loss(images, images.flip(0), dtf[:, 0], dtf[:, 0].flip(0)).numpy()
    np.savez_compressed(os.path.join(OUT, "legacy.npz"), **g)
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
