"""CPU ORACLE -- test infrastructure, NOT product code.

Python face of ``acfm_oracle.c`` plus numpy/torch-CPU restatements of the reference's
Python-level hot-path functions.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``acfm_video_3d_reconstruction_amd`` never does.

Every function cites the reference lines it follows (paths relative to /root/reference).
Parity status: projection / losses / laplacian / solve are pinned by golden vectors made
from the importable reference modules (tests/golden/make_golden.py); the rasterizer,
blending and atlas sampling restate PyTorch3D 0.3.0 (absent here) and are *parity
unpinned* beyond analytic known-answer tests.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32P = ctypes.POINTER(ctypes.c_float)
I64P = ctypes.POINTER(ctypes.c_int64)
I32P = ctypes.POINTER(ctypes.c_int32)

EYE_Z = 2.732  # nmr.py:144  eye = (0, 0, -2.732)  ->  T = (0, 0, 2.732)
SIL_SIGMA = 1e-4  # nmr.py:153
SIL_K = 20  # nmr.py:158
SIL_BLUR = float(np.log(1.0 / 1e-4 - 1.0) * 1e-4)  # nmr.py:157


def build(force=False):
    so = os.path.join(_HERE, "libacfm_oracle.so")
    src = os.path.join(_HERE, "acfm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libacfm_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    """ACFM_ORACLE_SO: an alternative build of the same C file (the ASan/UBSan build of `make asan-test`)."""
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(os.environ.get("ACFM_ORACLE_SO") or build())
    return _LIB


def set_threads(n):
    """OpenMP threads for the C loops; returns the count in effect."""
    return int(lib().oracle_set_threads(int(n)))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t):
    return a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- projection
def project(verts, cams, offset_z=0.0):
    """geom_utils.orthographic_proj_withz (multiframe/nnutils/geom_utils.py:62-79)."""
    verts, cams = _f32(verts), _f32(cams)
    N, V, _ = verts.shape
    out = np.empty_like(verts)
    lib().oracle_project(_p(verts, F32P), _p(cams, F32P), N, V, ctypes.c_float(offset_z),
                         _p(out, F32P))
    return out


def project_torch(X, cam, offset_z=0.0):
    """Differentiable torch-CPU restatement of the same function (used for gradient
    oracles; any dtype).  geom_utils.py:62-79, 107-152."""
    q = cam[:, None, 3:7]
    qc = torch.cat([q[..., :1], -q[..., 1:]], -1)
    X4 = torch.cat([X[..., :1] * 0, X], -1)

    def ham(a, b):
        a0, a1, a2, a3 = a.unbind(-1)
        b0, b1, b2, b3 = b.unbind(-1)
        return torch.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                            a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                            a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
                            a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], -1)

    r = ham(q.expand(-1, X.shape[1], -1), ham(X4, qc.expand(-1, X.shape[1], -1)))[..., 1:]
    proj = cam[:, None, :1] * r
    xy = proj[..., :2] + cam[:, None, 1:3]
    z = proj[..., 2:] + offset_z
    return torch.cat([xy, z], -1)


def to_ndc(proj, flip_y=True):
    """Camera chain of NeuralRenderer.forward (nmr.py:145-149): y *= -1 (mask/texture
    branches only), view = X.diag(-1,1,1) + (0,0,2.732); orthographic projection is the
    identity on xy and MeshRasterizer keeps view-space z."""
    out = np.array(proj, dtype=np.float32, copy=True)
    out[..., 0] = -out[..., 0]
    if flip_y:
        out[..., 1] = -out[..., 1]
    out[..., 2] = out[..., 2] + np.float32(EYE_Z)
    return out


def face_verts_of(ndc, faces):
    """[N,V,3] x [N,F,3] -> [N*F,3,3] packed (Meshes.verts_packed()[faces_packed()])."""
    N = ndc.shape[0]
    faces = np.asarray(faces)
    if faces.ndim == 2:
        faces = np.broadcast_to(faces, (N,) + faces.shape)
    idx = np.arange(N)[:, None, None]
    return np.ascontiguousarray(ndc[idx, faces].reshape(-1, 3, 3).astype(np.float32))


# ----------------------------------------------------------------------------- rasterizer
def rasterize(face_verts, N, H, K, blur_radius, clip_bary=False):
    """PyTorch3D rasterize_meshes naive CPU path."""
    face_verts = _f32(face_verts)
    F = face_verts.shape[0] // N
    W = H
    p2f = np.empty((N, H, W, K), np.int64)
    zbuf = np.empty((N, H, W, K), np.float32)
    bary = np.empty((N, H, W, K, 3), np.float32)
    dists = np.empty((N, H, W, K), np.float32)
    lib().oracle_rasterize(_p(face_verts, F32P), N, F, H, W, K, ctypes.c_float(blur_radius),
                           int(bool(clip_bary)), _p(p2f, I64P), _p(zbuf, F32P), _p(bary, F32P),
                           _p(dists, F32P))
    return p2f, zbuf, bary, dists


def rasterize_backward_dists(face_verts, p2f, grad_dists):
    """RasterizeMeshesBackwardCpu, dists path only -> [N*F,3,3]."""
    face_verts = _f32(face_verts)
    N, H, W, K = p2f.shape
    F = face_verts.shape[0] // N
    p2f = np.ascontiguousarray(p2f, dtype=np.int64)
    grad_dists = _f32(grad_dists)
    out = np.empty_like(face_verts)
    lib().oracle_rasterize_backward_dists(_p(face_verts, F32P), _p(p2f, I64P),
                                          _p(grad_dists, F32P), N, F, H, W, K, _p(out, F32P))
    return out


def sigmoid_alpha_blend(p2f, dists, sigma=SIL_SIGMA):
    """SoftSilhouetteShader / sigmoid_alpha_blend -> mask [N,H,W]."""
    K = p2f.shape[-1]
    P = p2f.size // K
    mask = np.empty(p2f.shape[:-1], np.float32)
    lib().oracle_sigmoid_alpha_blend(_p(np.ascontiguousarray(p2f), I64P), _p(_f32(dists), F32P),
                                     ctypes.c_size_t(P), K, ctypes.c_float(sigma),
                                     _p(mask, F32P))
    return mask


def sigmoid_alpha_blend_backward(p2f, dists, grad_mask, sigma=SIL_SIGMA):
    K = p2f.shape[-1]
    P = p2f.size // K
    out = np.empty(p2f.shape, np.float32)
    lib().oracle_sigmoid_alpha_blend_backward(_p(np.ascontiguousarray(p2f), I64P),
                                              _p(_f32(dists), F32P), _p(_f32(grad_mask), F32P),
                                              ctypes.c_size_t(P), K, ctypes.c_float(sigma),
                                              _p(out, F32P))
    return out


# ----------------------------------------------------------------------------- renderers
def sil_render(verts, faces, cams, img_size, offset_z=0.0, K=SIL_K, sigma=SIL_SIGMA,
               blur=SIL_BLUR, return_aux=False):
    """NeuralRenderer.forward, mask branch (multiframe/nnutils/nmr.py:143-172)."""
    N = verts.shape[0]
    ndc = to_ndc(project(verts, cams, offset_z), flip_y=True)
    fv = face_verts_of(ndc, faces)
    p2f, zbuf, bary, dists = rasterize(fv, N, img_size, K, blur, clip_bary=False)
    mask = sigmoid_alpha_blend(p2f, dists, sigma)
    if return_aux:
        return mask, p2f, dict(ndc=ndc, face_verts=fv, zbuf=zbuf, bary=bary, dists=dists)
    return mask, p2f


def sil_render_backward(verts, faces, cams, img_size, grad_mask, offset_z=0.0, K=SIL_K,
                        sigma=SIL_SIGMA, blur=SIL_BLUR):
    """Gradient of sum(mask * grad_mask) wrt verts [N,V,3] and cams [N,7].  grad_mask: [N,H,W], or a callable
    mask -> [N,H,W] (a loss on the rendered mask: one render serves the loss and its backward).

    Raster + blend backward in C (App-A.4/A.5); the index gather and the projection chain
    go through torch-CPU autograd in float64 (the reference relies on autograd there)."""
    verts = _f32(verts)
    cams = _f32(cams)
    N, V, _ = verts.shape
    faces_np = np.asarray(faces)
    if faces_np.ndim == 2:
        faces_np = np.broadcast_to(faces_np, (N,) + faces_np.shape)
    mask, p2f, aux = sil_render(verts, faces, cams, img_size, offset_z, K, sigma, blur, True)
    if callable(grad_mask):
        grad_mask = grad_mask(mask, p2f)
    gd = sigmoid_alpha_blend_backward(p2f, aux["dists"], grad_mask, sigma)
    gfv = rasterize_backward_dists(aux["face_verts"], p2f, gd).reshape(N, -1, 3, 3)
    g_ndc = np.zeros((N, V, 3), np.float64)
    for n in range(N):
        np.add.at(g_ndc[n], faces_np[n].reshape(-1), gfv[n].reshape(-1, 3).astype(np.float64))
    tv = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    tc = torch.tensor(cams, dtype=torch.float64, requires_grad=True)
    proj = project_torch(tv, tc, offset_z)
    ndc = torch.stack([-proj[..., 0], -proj[..., 1], proj[..., 2] + EYE_Z], -1)
    (ndc * torch.from_numpy(g_ndc)).sum().backward()
    return tv.grad.numpy().astype(np.float32), tc.grad.numpy().astype(np.float32), mask, p2f


def headline_step(mean_v, P, delta, faces, cams, atlas, gt, edt, bds, img, img_size, weights=(1.0, 0.1, 0.1, 0.5),
                  render_verts=None):
    """The benchmark's headline step on the CPU, term for term (bench.py `compute`; BASELINE.md section 3's "full step"):
    deformation apply v = mean + P delta (main.py:586-609 collapsed) -> soft-silhouette render K=20 -> l1 + edt losses
    -> boundary loss -> atlas-texture render + masked MSE -> backward to handle offsets, cameras, mean shape and atlas.
    weights = (l1, edt, bds, texture mse); total = mean_n(l1 + w_e edt + w_b bds) + w_t mean_n(mse).
    numpy / torch-CPU float32 in, -> dict(total, g_delta, g_cams, g_mean, g_atlas, mask, p2f).
    render_verts: deformed vertices to render instead of this function's own mean + P delta (the render is
    discontinuous in its inputs: a parity test first compares the two deformations, then hands the product's float32
    geometry over so that everything downstream is compared without depending on a last-bit coincidence)."""
    w_l1, w_e, w_b, w_t = weights
    H = img_size
    mean_v, P, delta, cams = _f32(mean_v), _f32(P), _f32(delta), _f32(cams)
    N = delta.shape[0]
    verts = (mean_v[None] + np.einsum("vk,nkc->nvc", P, delta)).astype(np.float32)
    out = {"verts": verts}
    if render_verts is not None:
        verts = _f32(render_verts)
    gt, edt = _f32(gt), _f32(edt).reshape(N, H, H)

    def grad_of_mask(mask, p2f):
        out["l1"] = np.abs(mask - gt).reshape(N, -1).mean(1)
        out["edt"] = (edt * mask).reshape(N, -1).mean(1)
        return ((w_l1 * np.sign(mask - gt) + w_e * edt) / (N * H * H)).astype(np.float32)
    gv, gc, mask, p2f = sil_render_backward(verts, faces, cams, H, grad_of_mask)
    # boundary loss on the projected vertices (loss_utils.py:204-237), gradient by torch-CPU autograd
    tv = torch.tensor(verts, requires_grad=True)
    tc = torch.tensor(cams, requires_grad=True)
    faces_t = torch.as_tensor(np.broadcast_to(np.asarray(faces), (N,) + np.asarray(faces).shape[-2:]).copy())
    bdt = bds_loss(project_torch(tv, tc)[..., :2], torch.as_tensor(_f32(bds)), faces_t, torch.from_numpy(p2f), reduce=False)
    (w_b * bdt.mean()).backward()
    gv = gv + tv.grad.numpy()
    gc = gc + tc.grad.numpy()
    # texture branch on detached geometry (main.py:627-636, 655-662)
    imgs, _, _, tidx = tex_render(verts, faces, cams, atlas, H)
    m1 = gt[:, None]
    diff = imgs * m1 - _f32(img) * m1
    mse = (diff ** 2).reshape(N, -1).mean(1)
    g_img = (w_t / N) * 2.0 * diff * m1 / (3 * H * H)
    g_atlas = tex_render_backward_atlas(tidx, g_img, np.asarray(atlas).shape)
    out.update(total=float((w_l1 * out["l1"] + w_e * out["edt"] + w_b * bdt.detach().numpy()).mean() + w_t * mse.mean()),
               g_delta=np.einsum("vk,nvc->nkc", P, gv).astype(np.float32), g_cams=gc, g_mean=gv.sum(0), g_atlas=g_atlas,
               mask=mask, p2f=p2f, bdt=bdt.detach().numpy(), mse=mse)
    return out


def tex_render(verts, faces, cams, atlas, img_size, offset_z=0.0, sigma=1e-4, gamma=1e-4):
    """NeuralRenderer.forward, texture branch with atlas=True (nmr.py:173-200):
    hard raster K=1, blur 0, clip_barycentric_coords=True (nmr.py:87-89), TexturesAtlas,
    ambient-only Phong, softmax_rgb_blend.
    Returns imgs [N,3,H,W], sil [N,H,W], pix_to_face [N,H,W,1], texel_idx [N,H,W]."""
    N = verts.shape[0]
    H = img_size
    ndc = to_ndc(project(verts, cams, offset_z), flip_y=True)
    fv = face_verts_of(ndc, faces)
    p2f, zbuf, bary, dists = rasterize(fv, N, H, 1, 0.0, clip_bary=True)
    atlas = _f32(atlas)
    R = atlas.shape[2]
    P = N * H * H
    rgb = np.empty((N, H, H, 3), np.float32)
    sil = np.empty((N, H, H), np.float32)
    tidx = np.empty((N, H, H), np.int32)
    lib().oracle_atlas_shade(_p(np.ascontiguousarray(p2f), I64P), _p(zbuf, F32P), _p(bary, F32P),
                             _p(dists, F32P), _p(atlas, F32P), ctypes.c_size_t(P), R,
                             ctypes.c_float(sigma), ctypes.c_float(gamma), _p(rgb, F32P),
                             _p(sil, F32P), _p(tidx, I32P))
    return np.ascontiguousarray(rgb.transpose(0, 3, 1, 2)), sil, p2f, tidx


def vertex_color_render(verts, faces, cams, verts_rgb, img_size, offset_z=0.0):
    """NeuralRenderer.forward with atlas=False (nmr.py:177-179): Textures(verts_rgb) =
    barycentric (clipped, nmr.py:87-89) interpolation of vertex colours, same K=1 blend."""
    N = verts.shape[0]
    H = img_size
    ndc = to_ndc(project(verts, cams, offset_z), flip_y=True)
    fv = face_verts_of(ndc, faces)
    p2f, zbuf, bary, dists = rasterize(fv, N, H, 1, 0.0, clip_bary=True)
    faces = np.asarray(faces)
    if faces.ndim == 2:
        faces = np.broadcast_to(faces, (N,) + faces.shape)
    F = faces.shape[1]
    col = np.broadcast_to(np.asarray(verts_rgb, np.float32), (N, verts.shape[1], 3))
    imgs = np.zeros((N, H, H, 3), np.float32)
    for n in range(N):
        f = p2f[n, ..., 0]
        cov = f >= 0
        tri = faces[n][(f[cov] - n * F)]
        c = (bary[n, ..., 0, :][cov][:, :, None] * col[n][tri]).sum(1)
        imgs[n][cov] = c
    return np.ascontiguousarray(imgs.transpose(0, 3, 1, 2)), p2f


def tex_render_backward_atlas(tidx, grad_imgs, atlas_shape):
    """d/d atlas of sum(imgs * grad_imgs): rgb = w * texel / (w + delta) with the blend
    weight w = prob >= 0.5 and delta = 1e-10, i.e. d rgb / d texel = 1 to fp32 precision;
    integer texel indexing sends no gradient to geometry."""
    g = np.zeros((int(np.prod(atlas_shape[:-1])), 3), np.float64)
    gi = np.asarray(grad_imgs, np.float64).transpose(0, 2, 3, 1).reshape(-1, 3)
    t = np.asarray(tidx).reshape(-1)
    sel = t >= 0
    np.add.at(g, t[sel], gi[sel])
    return g.reshape(atlas_shape).astype(np.float32)


def of_raster(proj_verts, faces, img_size):
    """OF_NeuralRenderer.forward (nmr.py:224-238): hard raster K=1, blur 0, no y flip,
    verts already projected by proj_fn."""
    N = proj_verts.shape[0]
    ndc = to_ndc(_f32(proj_verts), flip_y=False)
    fv = face_verts_of(ndc, faces)
    p2f, _, _, _ = rasterize(fv, N, img_size, 1, 0.0, clip_bary=False)
    return p2f


# ----------------------------------------------------------------------------- losses
# torch-CPU restatements of multiframe/nnutils/loss_utils.py (tensor in, tensor out).
def l1_loss(pred, target, reduce=True):
    """loss_utils.l1_loss (loss_utils.py:72-77)."""
    loss = (pred - target).abs()
    return loss.mean() if reduce else loss.reshape(loss.shape[0], -1).mean(1)


def iou(pred, target, eps=1e-6, reduce=True):
    """loss_utils.iou (loss_utils.py:18-28)."""
    p = pred.reshape(pred.shape[0], -1)
    t = target.reshape(target.shape[0], -1)
    inter = (p * t).sum(1)
    union = (p + t - p * t).sum(1) + eps
    r = inter / union
    return r.sum() / r.numel() if reduce else r


def iou_loss(pred, target, reduce=True):
    """loss_utils.iou_loss (loss_utils.py:31-32)."""
    return 1 - iou(pred, target, reduce=reduce)


def edt_loss(mask, edt, reduce=True):
    """loss_utils.edt_loss (loss_utils.py:245-253)."""
    b = mask.shape[0]
    loss = (edt * mask[:, None]).reshape(b, -1).mean(-1)
    return loss.mean() if reduce else loss


def visible_vertices(faces, pix_to_face0, nv):
    """Visible-vertex mask shared by bds_loss (loss_utils.py:214-224) and
    optical_flow_loss (:432-443): verts of every face that appears in pix_to_face[...,0]."""
    bt = faces.shape[0]
    vis = torch.zeros(bt * nv)
    faces_ = (faces + torch.arange(bt)[:, None, None] * nv).reshape(-1, 3)
    fm = pix_to_face0.reshape(-1)
    fm = fm[fm >= 0].long()
    vis[faces_[fm].reshape(-1).unique()] = 1
    return vis.reshape(bt, nv)


def bds_loss(verts, bds, faces, pix_to_face, reduce=True):
    """loss_utils.bds_loss (loss_utils.py:204-237) with all boundary points kept
    (bds.shape[1] <= n_samples, so the randperm at :211 is a pure permutation)."""
    bt, nv, _ = verts.shape
    bds_v, bds_m = bds[..., :-1], bds[..., -1]
    vis = visible_vertices(faces, pix_to_face[..., 0], nv)
    dist = torch.cdist(bds_v, verts) ** 2
    dist = (1 - vis[:, None]) * 1000 + vis[:, None] * dist
    min_d = dist.min(-1)[0]
    loss = (min_d * bds_m).sum(-1)
    return loss.mean() if reduce else loss


def locally_rigid(verts, verts_t, edges):
    """loss_utils.locally_rigid_fn (loss_utils.py:150-164) on padded verts [N,V,3] with a
    shared edge list [E,2] (Meshes.edges_packed of N identical topologies)."""
    N = verts.shape[0]
    d = (verts[:, edges[:, 0]] - verts[:, edges[:, 1]]).norm(dim=-1)
    dt = (verts_t[:, edges[:, 0]] - verts_t[:, edges[:, 1]]).norm(dim=-1)
    return ((d - dt) ** 2).sum() / N


def kp_l2_loss(kp_pred, kp_gt, reduction="mean"):
    """loss_utils.kp_l2_loss (loss_utils.py:341-356)."""
    vis = (kp_gt[:, :, 2] > 0).float()
    loss = (kp_pred - kp_gt[:, :, :2]).abs().sum(-1) * vis
    loss = loss.mean(-1) / (vis.mean(-1) + 1e-4)
    return loss.mean() if reduction == "mean" else loss


def deform_l2reg(V):
    """loss_utils.deform_l2reg (loss_utils.py:322-327)."""
    return V.reshape(-1, V.shape[2]).norm(p=2, dim=1).mean()


def quat_loss_geodesic(q1, q2):
    """loss_utils.quat_loss_geodesic (loss_utils.py:262-277)."""
    a0, a1, a2, a3 = q1.unbind(-1)
    b0, b1, b2, b3 = q2[:, 0], -q2[:, 1], -q2[:, 2], -q2[:, 3]
    w = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3
    return (1 - w.abs())[:, None]


def camera_loss(cam_pred, cam_gt, margin):
    """loss_utils.camera_loss (loss_utils.py:280-289); hinge_loss (:256-259) restated
    without the hard-coded .cuda()."""
    rot = torch.clamp(quat_loss_geodesic(cam_pred[:, -4:], cam_gt[:, -4:]) - margin, min=0)
    st = torch.clamp(((cam_pred[:, :3] - cam_gt[:, :3]) ** 2).reshape(-1) - margin, min=0)
    return rot.mean() + st.mean()


def optical_flow_loss(meshes, faces, cams, flows, pix_to_face=None, reduce=True):
    """loss_utils.optical_flow_loss (loss_utils.py:419-474).  meshes [b,t,V,3],
    faces [b,t,F,3], cams [b*t,7], flows [b,t,H,W,2]; pix_to_face None -> hard raster of
    the projected verts through OF_NeuralRenderer (nmr.py:224-238)."""
    H, W = flows.shape[2:4]
    b, t, nv, _ = meshes.shape
    bt = b * t
    pts = project_torch(meshes.reshape(bt, nv, 3), cams.reshape(bt, -1))
    with torch.no_grad():
        f_ = faces.reshape(bt, -1, 3).long()
        if pix_to_face is None:
            p2f = torch.from_numpy(of_raster(pts.detach().float().numpy(), f_.numpy(), H))
        else:
            p2f = pix_to_face[..., :1].long()
        vis = visible_vertices(f_, p2f[..., 0], nv).reshape(b, t, nv)
    pxy = pts[..., :2]
    fl = flows.reshape(bt, H, W, 2).permute(0, 3, 1, 2)
    samp = torch.nn.functional.grid_sample(fl, pxy[:, :, None, :].to(fl.dtype),
                                           align_corners=False, mode="nearest")
    samp = samp[..., 0].permute(0, 2, 1).reshape(b, t, nv, 2)
    pix = W * (pxy.reshape(b, t, nv, 2) + 1) / 2
    of_pred = pix[:, :-1] - pix[:, 1:]
    vis = ((samp.abs().sum(-1) != 0) & vis.bool()).float()[:, 1:].detach()
    gt = vis[..., None] * samp[:, 1:]
    of_pred = vis[..., None] * of_pred
    loss = (gt[..., 0] - of_pred[..., 0]).abs().sum(-1) + (gt[..., 1] - of_pred[..., 1]).abs().sum(-1)
    loss = loss / H / (vis.sum(-1) + 1)
    if reduce:
        loss = loss.sum()
    return loss, of_pred, vis


# ----------------------------------------------------------------------------- mesh ops
def edges_packed(faces):
    """Meshes.edges_packed for one mesh: unique sorted (min,max) pairs."""
    f = np.asarray(faces)
    e = np.concatenate([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)
    e = np.sort(e, 1)
    return np.unique(e, axis=0)


def laplacian_cot(verts, faces):
    """geom_utils.mesh_laplacian(meshes, 'cot') (geom_utils.py:158-254, 257-324) for one
    mesh -> dense L [V,V] (torch, dtype of verts)."""
    V = verts.shape[0]
    fv = verts[faces]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    A = (v1 - v2).norm(dim=1)
    B = (v0 - v2).norm(dim=1)
    C = (v0 - v1).norm(dim=1)
    s = 0.5 * (A + B + C)
    area = (s * (s - A) * (s - B) * (s - C)).clamp(min=1e-12).sqrt()
    A2, B2, C2 = A * A, B * B, C * C
    cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], 1) / 4.0
    ii = faces[:, [1, 2, 0]].reshape(-1)
    jj = faces[:, [2, 0, 1]].reshape(-1)
    W = torch.zeros(V, V, dtype=verts.dtype)
    W.index_put_((ii, jj), cot.reshape(-1), accumulate=True)
    W = W + W.t()
    return W - torch.diag(W.sum(1))


def laplacian_smoothing_cot(verts, faces):
    """pytorch3d.loss.mesh_laplacian_smoothing(meshes, 'cot') for padded
    verts [N,V,3] sharing one face list [F,3]."""
    N, V, _ = verts.shape
    total = 0
    for n in range(N):
        with torch.no_grad():
            L = laplacian_cot(verts[n].detach(), faces)
            W = L - torch.diag(torch.diagonal(L))
            rs = W.sum(1)
            nw = torch.where(rs > 0, 1.0 / rs, torch.zeros_like(rs))
        lv = (W @ verts[n]) * nw[:, None] - verts[n]
        total = total + lv.norm(dim=1).sum() / V
    return total / N


def deform_solve(lbs_logits, mean_v, delta, L, dtype=torch.float64):
    """Deformation solve exactly as written in the reference (multiframe/main.py:586-609):
    A = softmax(lbs, 0)^T; delta_v = A v + delta; M = L^T L + A^T A;
    b = L^T (L v) + A^T delta_v; pred_v = cholesky_solve(b, cholesky(M)).
    Evaluated in ``dtype`` (float64 = the parity target)."""
    lbs_logits, mean_v, delta, L = (x.to(dtype) for x in (lbs_logits, mean_v, delta, L))
    A = torch.softmax(lbs_logits, dim=0).t()
    dv = (A @ mean_v)[None] + delta
    M = L.t() @ L + A.t() @ A
    b = (L.t() @ (L @ mean_v))[None] + A.t()[None] @ dv
    u = torch.linalg.cholesky(M)
    return torch.cholesky_solve(b, u[None].expand(b.shape[0], -1, -1))


def subdivide(verts, faces):
    """pytorch3d.ops.SubdivideMeshes for one mesh."""
    verts = np.asarray(verts)
    faces = np.asarray(faces)
    V = verts.shape[0]
    edges = edges_packed(faces)
    key = {(int(a), int(b)): i for i, (a, b) in enumerate(edges)}

    def eid(a, b):
        return key[(int(min(a, b)), int(max(a, b)))] + V

    new_v = np.concatenate([verts, 0.5 * (verts[edges[:, 0]] + verts[edges[:, 1]])], 0)
    f0, f1, f2, f3 = [], [], [], []
    for (a, b, c) in faces:
        e0, e1, e2 = eid(b, c), eid(a, c), eid(a, b)
        f0.append((a, e2, e1))
        f1.append((b, e0, e2))
        f2.append((c, e1, e0))
        f3.append((e0, e1, e2))
    return new_v.astype(verts.dtype), np.array(f0 + f1 + f2 + f3, dtype=faces.dtype)


# ----------------------------------------------------------------------------- input prep
def compute_dt(mask, norm=True):
    """multiframe/utils/image.py:94-102 verbatim semantics (scipy IS the reference here)."""
    from scipy.ndimage import distance_transform_edt
    dist = distance_transform_edt(1 - mask)
    if norm:
        dist = dist / max(mask.shape)
    return dist


def compute_dt_barrier(mask, k=50):
    """multiframe/utils/image.py:105-116."""
    from scipy.ndimage import distance_transform_edt
    diff = (distance_transform_edt(1 - mask) - distance_transform_edt(mask)) / max(mask.shape)
    return 1.0 / (1 + np.exp(k * -diff))


def find_boundaries(m):
    """skimage.segmentation.find_boundaries (0.18.1 pinned in environment.yml:192), mode='thick',
    connectivity=1: grey dilation != grey erosion over the 4-neighbourhood cross; skimage is not
    installed here, scipy.ndimage (what skimage calls underneath) is."""
    from scipy.ndimage import generate_binary_structure, grey_dilation, grey_erosion
    fp = generate_binary_structure(2, 1)
    return grey_dilation(m, footprint=fp) != grey_erosion(m, footprint=fp)


def compute_boundaries(masks):
    """multiframe/utils/image.py:122-146."""
    bds = [np.transpose(find_boundaries(m).nonzero()) for m in masks]
    max_bd = max(bd.shape[0] for bd in bds)
    out, flag = [], []
    for bd in bds:
        f = np.ones(max_bd)
        f[bd.shape[0]:] = 0
        flag.append(f)
        b = np.zeros((max_bd, 2))
        b[:bd.shape[0]] = bd
        out.append(b)
    out = np.array(out)
    out[..., 0] = (out[..., 0] / masks.shape[1] - 0.5) * 2
    out[..., 1] = (out[..., 1] / masks.shape[2] - 0.5) * 2
    out = out[..., ::-1].copy()
    return np.concatenate((out, np.array(flag)[:, :, None]), axis=-1).astype(np.float32)


def correlation(f1, f2, md):
    """Cost volume of the reference's correlation extension in MaskFlownet's configuration
    (correlation_cuda_kernel.cu:73-147 with pad = max_displacement = md, kernel 1, strides 1):
    out[n, (tj+md)(2md+1)+(ti+md), y, x] = mean_c f1[n,c,y,x] * f2[n,c,y+tj,x+ti], zero outside.
    Accumulated in float64 (the reference's float32 shuffle-reduction order is not reproduced)."""
    f1 = np.asarray(f1, np.float64)
    f2 = np.asarray(f2, np.float64)
    N, C, H, W = f1.shape
    D1 = 2 * md + 1
    pad = np.zeros((N, C, H + 2 * md, W + 2 * md))
    pad[:, :, md:md + H, md:md + W] = f2
    out = np.zeros((N, D1 * D1, H, W))
    for tj in range(-md, md + 1):
        for ti in range(-md, md + 1):
            out[:, (tj + md) * D1 + ti + md] = (f1 * pad[:, :, md + tj:md + tj + H, md + ti:md + ti + W]).sum(1) / C
    return out.astype(np.float32)


# ----------------------------------------------------------------------------- composed paths
# Restatements of the reference's CALLERS of the hot path, composed from the pieces above: the camera
# chain of the trainer (multiframe/main.py:97-138, 551-584), ShapeTrainer.forward's loss assembly
# (:586-765) and one iteration of the test-time refinement loop (nnutils/predictor.py:287-349).  torch-CPU
# float64 for everything smooth; the renders go through the C rasteriser (float32, like the product).
def standardize_quaternion(q):
    """pytorch3d.transforms.standardize_quaternion: non-negative real part."""
    return torch.where(q[..., :1] < 0, -q, q)


def quaternion_multiply(a, b):
    """pytorch3d.transforms.quaternion_multiply (0.3.0): Hamilton product (real first), then standardise."""
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    out = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                       aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)
    return standardize_quaternion(out)


def mirrored_pose(sfm_pose):
    """main.py:97-110 / 113-122 without the blend: (s, -tx, ty, q_mirror * standardize(q)).  q_mirror =
    matrix_to_quaternion(diag(-1, 1, -1)) = (0, 0, 1, 0) (0.5 sqrt(max(0, 1 - m00 + m11 - m22)) = 1 on y,
    the other three square roots vanish)."""
    q = standardize_quaternion(sfm_pose[:, -4:])
    qm = torch.tensor([0.0, 0.0, 1.0, 0.0], dtype=sfm_pose.dtype).expand_as(q)
    return torch.cat([sfm_pose[:, :1], -sfm_pose[:, 1:2], sfm_pose[:, 2:3], quaternion_multiply(qm, q)], -1)


def camera_pipeline(cam_emb, mirror_flag, transforms, scale_lr_decay=1.0):
    """Camera embeddings [G,N,7] -> cam_pred [G*N,7] (main.py:564-584 with :113-138):
    scales = relu(decay * e0 + 1) + 1e-12, quats normalised; mirror_cameras blended by the frame's flag
    (repeated over the G hypotheses); transform_cameras blended by the transform's flag."""
    G, N, _ = cam_emb.shape
    scales = torch.relu(scale_lr_decay * cam_emb[..., :1] + 1) + 1e-12
    quats = torch.nn.functional.normalize(cam_emb[..., 3:], dim=-1)
    cam = torch.cat([scales, cam_emb[..., 1:3], quats], dim=2).reshape(G * N, 7)
    mf = mirror_flag.repeat(G)[:, None].to(cam.dtype)
    cam = (1 - mf) * cam + mirrored_pose(cam) * mf
    tr = transforms.repeat(G, 1).to(cam.dtype)
    flag = tr[:, -1:]
    new = torch.cat([cam[:, :1] * tr[:, :1], cam[:, 1:2] * tr[:, :1] + tr[:, 1:2],
                     cam[:, 2:3] * tr[:, :1] + tr[:, 2:3], cam[:, -4:]], -1)
    return (1 - flag) * cam + new * flag


def texture_cycle_loss(textures, num_frames):
    """main.py:705-711, literally (the reshape to [-1,R,R] regroups the trailing (R,T,3) block as written)."""
    t_c = textures.reshape(-1, num_frames, *textures.shape[1:]).permute(0, 2, 3, 4, 1, 5)
    t_c = t_c.reshape(-1, t_c.shape[2], t_c.shape[3])
    return torch.norm(t_c[:, :-1] - t_c[:, 1:], p=2, dim=-1).mean()


def masked_texture_mse(tex, img, mask):
    """The MSE part of the texture loss (main.py:655-662); the LPIPS part is out of scope."""
    m = mask[:, None]
    return ((tex * m - img * m) ** 2).mean((1, 2, 3))


DEFAULT_OPTS = dict(num_frames=2, kp_loss_wt=0., of_loss_wt=1., mask_loss_wt=1., rigid_wt=0.5, deform_reg_wt=1.,
                    handle_deform_reg_wt=0., boundaries_reg_wt=1., edt_reg_wt=0.1, bdt_reg_wt=2.,
                    triangle_reg_wt=0.1, tex_loss_wt=.5, scale_lr_decay=0.05)   # main.py:55-89


def _f64(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def multiframe_forward_terms(cam_emb, mirror_flag, transforms, lbs_logits, mean_v, faces, delta, masks,
                             edts_barrier, boundaries, optical_flows=None, textures=None, imgs=None,
                             render_verts=None, render_cams=None, **opts):
    """ShapeTrainer.forward (main.py:523-765) between the network heads and the loss scalar, every named term.
    cam_emb [G,N,7] (camera embeddings of the N = B*T frames), mirror_flag [N], transforms [N,4],
    lbs_logits [V,Kh], mean_v [V,3], faces [F,3], delta [N,Kh,3] (= delta_v_res; drop_deform: zeros),
    masks [N,H,H], edts_barrier [N,1,H,H], boundaries [N,P,3], optical_flows [B,T,H,H,2],
    textures [N,F,R,R,3], imgs [N,3,H,H].  Returns a dict of float64 tensors / numpy arrays.
    render_verts [N,V,3] / render_cams [G*N,7] (float32, optional): geometry handed to the rasteriser and to
    everything downstream of it INSTEAD of this function's own float64 solve / camera chain (which are still
    computed and returned as pred_v / cam_pred for comparison).  The render is discontinuous in its inputs -- a
    depth swap at the K-th slot of one pixel moves a mask value by up to ~0.1 -- so a last-bit difference between
    a float32 product solve and the float64 solve here can flip such an event; parity tests compare the solve and
    the camera chain first (1e-5 / 1e-6) and then feed the product's float32 values to the oracle's downstream."""
    o = dict(DEFAULT_OPTS)
    o.update(opts)
    T = o["num_frames"]
    cam_emb, transforms, delta, masks, edts, bds = map(_f64, (cam_emb, transforms, delta, masks, edts_barrier,
                                                              boundaries))
    mirror_flag = torch.as_tensor(np.asarray(mirror_flag)).long()
    lbs_logits, mean_v = _f64(lbs_logits), _f64(mean_v)
    faces_t = torch.as_tensor(np.asarray(faces)).long()
    G, N, _ = cam_emb.shape
    H = masks.shape[-1]
    B = N // T
    out = {}
    cam = camera_pipeline(cam_emb, mirror_flag, transforms, o["scale_lr_decay"])          # :564-584
    out["cam_pred"] = cam
    L = laplacian_cot(mean_v, faces_t)                                                      # :600-601
    pred_v1 = deform_solve(lbs_logits, mean_v, delta, L)                                    # :586-609
    out["pred_v"] = pred_v1
    if render_verts is not None:
        pred_v1 = _f64(render_verts)
    if render_cams is not None:
        cam = _f64(render_cams)
    pred_v = pred_v1.repeat(G, 1, 1)                                                        # :610
    pv32, cam32 = pred_v.float().numpy(), cam.float().numpy()
    faces_np = faces_t.numpy()
    mask_pred, p2f = sil_render(pv32, faces_np, cam32, H)                                   # :637-640
    mp = torch.from_numpy(mask_pred).double()
    out["mask_pred"], out["pix_to_face"] = mask_pred, p2f
    mask_loss = l1_loss(mp, masks.repeat(G, 1, 1), reduce=False).reshape(G, N)              # :644-645
    total = o["mask_loss_wt"] * mask_loss
    if o["of_loss_wt"] > 0 and optical_flows is not None:                                   # :664-688
        flows = _f64(optical_flows)
        masks_of = masks.reshape(B, T, H, H)
        flows_f = (torch.flip(flows, dims=[1]) * masks_of[..., None]).repeat(G, 1, 1, 1, 1)
        faces_of = faces_t[None, None].expand(G * B, T, -1, -1)
        of_loss = optical_flow_loss(pred_v.reshape(G * B, T, -1, 3), faces_of, cam, flows_f, None, reduce=False)[0]
        of_loss = of_loss.reshape(G, -1).repeat(1, T).reshape(G, -1)                        # :684-686
    else:
        of_loss = torch.zeros(1, dtype=torch.float64)
    out["of_loss"] = of_loss
    total = total + o["of_loss_wt"] * of_loss
    proj = project_torch(pred_v, cam)[..., :2]                                              # :714
    edt = edt_loss(mp, edts.repeat(G, 1, 1, 1), reduce=False).reshape(G, N)                 # :715-716
    bdt = bds_loss(proj, bds.repeat(G, 1, 1), faces_t[None].expand(G * N, -1, -1),
                   torch.from_numpy(p2f), reduce=False).reshape(G, N)                       # :717-719
    sil_cons = o["edt_reg_wt"] * edt + o["bdt_reg_wt"] * bdt                                # :721
    total = total + o["boundaries_reg_wt"] * sil_cons
    out.update(mask_loss=mask_loss, edt_loss=edt, bdt_loss=bdt, sil_cons=sil_cons)
    cycle = torch.zeros((), dtype=torch.float64)
    if textures is not None and imgs is not None:                                           # :616-636, 647-662
        tex, im = _f64(textures), _f64(imgs)
        atl_rep = np.ascontiguousarray(np.tile(tex.float().numpy(), (G, 1, 1, 1, 1)))
        tp, _, _, _ = tex_render(pv32, faces_np, cam32, atl_rep, H)
        cam_f = mirrored_pose(cam)
        tpf, _, _, _ = tex_render(pv32, faces_np, cam_f.float().numpy(), atl_rep, H)
        im_f, m_f = torch.flip(im, dims=(3,)), torch.flip(masks, dims=(2,))
        tex_mse = 0.5 * (masked_texture_mse(torch.from_numpy(tp).double(), im.repeat(G, 1, 1, 1), masks.repeat(G, 1, 1))
                         + masked_texture_mse(torch.from_numpy(tpf).double(), im_f.repeat(G, 1, 1, 1),
                                              m_f.repeat(G, 1, 1)))
        tex_mse = tex_mse.reshape(G, N)
        total = total + o["tex_loss_wt"] * tex_mse
        cycle = texture_cycle_loss(tex, T)                                                  # :705-711
        out.update(tex_mse=tex_mse, texture_pred=tp, texture_pred_flip=tpf, cycle=cycle)
    out["total_per_hyp"] = total
    out["camera_loss"] = total.mean()                                                       # :734
    probs = torch.softmax(-total, dim=0)                                                    # :735
    weighted = (total * probs).sum(0).mean()                                                # :743-745
    edges = torch.from_numpy(edges_packed(faces_np))
    rigid = locally_rigid(pred_v, mean_v[None].repeat(G * N, 1, 1), edges)                  # :713 (mean over meshes)
    triangle = laplacian_smoothing_cot(pred_v, faces_t)                                     # :702-703
    handle = deform_l2reg(delta)                                                            # :612
    loss = weighted + o["rigid_wt"] * rigid + o["triangle_reg_wt"] * triangle \
        + o["deform_reg_wt"] * cycle + o["handle_deform_reg_wt"] * handle                   # :747-751
    out.update(probs=probs, weighted=weighted, rigid=rigid, triangle=triangle, handle=handle, loss=loss)
    return out


def refine_iteration(lbs_logits, mean_v, faces, delta, cam_raw, masks, edts_barrier, boundaries, mask_loss_wt=1.0,
                     boundaries_reg_wt=1.0, edt_reg_wt=0.1, bdt_reg_wt=0.1, optimize_camera=True,
                     render_verts=None, render_cams=None):
    """One iteration of the post-processing loop (predictor.py:301-345), loss and gradients:
    cam = (s, t, normalize(q)); pred_v = solve(delta); mask render; total = mask_wt * l1 + bds_wt *
    (bdt_reg_wt * edt_loss + edt_reg_wt * bdt_loss) (sic, :322).  Returns (total, d total / d delta,
    d total / d cam_raw, dict of the terms).  The raster / blend backward is the C oracle's, the rest
    float64 autograd.  render_verts / render_cams: float32 geometry for the rasteriser (see
    multiframe_forward_terms); the differentiable chain stays this function's own float64 one."""
    lbs_logits, mean_v, masks, edts, bds = map(_f64, (lbs_logits, mean_v, masks, edts_barrier, boundaries))
    faces_t = torch.as_tensor(np.asarray(faces)).long()
    delta = _f64(delta).clone().requires_grad_(True)
    cam_raw = _f64(cam_raw).clone().requires_grad_(True)
    N, H = masks.shape[0], masks.shape[-1]
    cam = torch.cat([cam_raw[:, :3], torch.nn.functional.normalize(cam_raw[:, 3:], dim=-1)], 1) \
        if optimize_camera else cam_raw
    L = laplacian_cot(mean_v, faces_t)
    pred_v = deform_solve(lbs_logits, mean_v, delta, L)
    pv32, cam32, faces_np = pred_v.detach().float().numpy(), cam.detach().float().numpy(), faces_t.numpy()
    if render_verts is not None:
        pv32 = np.ascontiguousarray(render_verts, dtype=np.float32)
    if render_cams is not None:
        cam32 = np.ascontiguousarray(render_cams, dtype=np.float32)
    mask_pred, p2f = sil_render(pv32, faces_np, cam32, H)
    mp = torch.from_numpy(mask_pred).double().requires_grad_(True)
    mask_loss = l1_loss(mp, masks)
    edt = edt_loss(mp, edts)
    proj = project_torch(pred_v, cam)[..., :2]
    bdt = bds_loss(proj, bds, faces_t[None].expand(N, -1, -1), torch.from_numpy(p2f))
    mask_terms = mask_loss_wt * mask_loss + boundaries_reg_wt * bdt_reg_wt * edt
    total = mask_terms + boundaries_reg_wt * edt_reg_wt * bdt
    gm, = torch.autograd.grad(mask_terms, mp)
    gv, gc, _, _ = sil_render_backward(pv32, faces_np, cam32, H, gm.float().numpy())
    # the mask terms reach delta / cam_raw through the render's vertex and camera gradients (gv, gc);
    # the boundary term through the projection
    surrogate = boundaries_reg_wt * edt_reg_wt * bdt + (pred_v * torch.from_numpy(gv).double()).sum() \
        + (cam * torch.from_numpy(gc).double()).sum()
    g_delta, g_cam = torch.autograd.grad(surrogate, [delta, cam_raw])
    return total.detach(), g_delta, g_cam, dict(mask_loss=mask_loss.detach(), edt_loss=edt.detach(),
                                                bdt_loss=bdt.detach(), pred_v=pred_v.detach(), mask_pred=mask_pred)
