"""Drop-in for multiframe/nnutils/loss_utils.py: same function names, arguments, reduce
flags and error behaviour; the per-pixel / per-point work runs in libacfm_hip.so.

HIP-backed: l1_loss, iou, iou_loss, edt_loss (one fused pass, ops.mask_losses), bds_loss
(ops.visible_vertices + ops.bds_loss_per_mesh), optical_flow_loss (ops.project +
ops.hard_raster + ops.visible_vertices).  The remaining functions are O(N*V) or smaller
reductions on device tensors.  LPIPS (PerceptualTextureLoss_v2, loss_utils.py:359-383) is a
third-party AlexNet and is out of scope."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib, ops
from . import geom_utils


def _is_mask_batch(predict, target):
    return predict.is_cuda and predict.dim() >= 2 and predict.shape == target.shape


def iou(predict, target, eps=1e-6, reduce=True):
    """loss_utils.py:18-28."""
    # (the prediction goes in as it is, not as a [N, H W] view: see l1_loss)
    out = ops.mask_losses(predict, target.reshape(target.shape[0], -1))
    r = out[:, 1] / (out[:, 2] + eps)
    if reduce:
        return r.sum() / r.nelement()
    return r


def iou_loss(predict, target, reduce=True):
    """loss_utils.py:31-32."""
    return 1 - iou(predict[:, None], target[:, None], reduce=reduce)


def l1_loss(predict, target, reduce=True):
    """loss_utils.py:72-77."""
    # the rendered mask goes in as it is: its gradient then reaches the render's backward unformed (ops.LazyGrad) -- and
    # summed, still unformed, with edt_loss's on the same mask (main.py:644, 716) -- and that kernel forms it per pixel
    out = ops.mask_losses(predict, target.reshape(target.shape[0], -1))
    if reduce:
        return out[:, 0].mean()  # equal element counts per row: mean of row means == global mean
    return out[:, 0]


def edt_loss(mask_rendered, edt, reduce=True):
    """loss_utils.py:245-253.  edt [B,1,H,W]."""
    bsize = mask_rendered.shape[0]
    if edt.shape[1] != 1:
        # the reference broadcasts mask[:, None] over C channels and averages everything
        err = (edt * mask_rendered[:, None]).reshape(bsize, -1).mean(-1)
        return err.mean() if reduce else err
    out = ops.mask_losses(mask_rendered, None, edt.reshape(edt.shape[0], -1))
    return out[:, 3].mean() if reduce else out[:, 3]


def fused_silhouette_losses(mask_pred, mask_gt, edt, eps=1e-6, raw=False):
    """One pass over the rendered mask for all three silhouette terms
    -> (l1 [N], iou [N], edt [N]); equals l1_loss / iou / edt_loss with reduce=False.
    raw=True: the kernel's [N,4] output (mean|m-gt|, sum m*gt, sum(m+gt-m*gt), mean edt*m) as it is,
    for combine_losses (no column views, no IoU division: nothing but the one launch)."""
    N = mask_pred.shape[0]
    # mask_gt / edt may be [N/G, ...]: the ground truth of a frame shared by its G hypotheses
    # (the mask goes in as it is, not as a [N, H W] view: its gradient then reaches the render's backward unformed --
    # ops.LazyGrad -- and that kernel forms it per pixel instead of reading a [N,H,W] image of it)
    out = ops.mask_losses(mask_pred, mask_gt.reshape(mask_gt.shape[0], -1), edt.reshape(edt.shape[0], -1))
    if raw:
        return out
    return out[:, 0], out[:, 1] / (out[:, 2] + eps), out[:, 3]


def combine_losses(terms, weights):
    """Weighted total of per-mesh loss terms and its mean over the batch, (1/N) sum_n sum_t w_t * term_t[n]
    (the elementwise tail of multiframe/main.py:716-765), as one launch forward and one backward.
    terms: up to 4 tensors [N] or [N,C<=4]; weights: one float per column, term by term."""
    return ops.combine_losses(terms, weights)


_CONJ_SIGN = (1.0, -1.0, -1.0, -1.0)


def quat_conj(q):
    """loss_utils.py:35-36: (w, x, y, z) -> (w, -x, -y, -z) on [B,N,4]."""
    return q * _lib.const(_CONJ_SIGN, q.device, q.dtype)


def quat2ang(q):
    """loss_utils.py:39-42: rotation angle 2*acos(w), w clamped away from +-1; [B,N,4] -> [B,N,1]."""
    w = q[..., 0].clamp(-1 + 1e-6, 1 - 1e-6)
    return (2.0 * torch.acos(w))[..., None]


hamilton_product = geom_utils.hamilton_product


def _edge_lengths(meshes):
    edges = meshes.edges_packed()
    ve = meshes.verts_packed()[edges]
    v0, v1 = ve.unbind(1)
    return v0, v1


def template_edge_loss(meshes, template_mesh):
    """loss_utils.py:80-114 (unused by main.py)."""
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
    N = len(meshes)
    v0, v1 = _edge_lengths(meshes)
    t0, t1 = _edge_lengths(template_mesh)
    e = (v0 - v1).norm(dim=1, p=2) ** 2.0
    te = (t0 - t1).norm(dim=1, p=2) ** 2.0
    return (e - te).norm(p=2) / N


def mask_dt_loss(proj_verts, dist_transf):
    """loss_utils.py:117-129."""
    grid = proj_verts.unsqueeze(1)
    d = F.grid_sample(dist_transf, grid, padding_mode='border', align_corners=True)
    return d.mean()


def texture_dt_loss(texture_flow, dist_transf, vis_rend=None, cams=None, verts=None, tex_pred=None):
    """loss_utils.py:132-147."""
    T = texture_flow.size(-2)
    Fn = texture_flow.size(1)
    grid = texture_flow.view(-1, Fn, T * T, 2)
    return F.grid_sample(dist_transf, grid, align_corners=True).mean()


def locally_rigid_fn(meshes, mesh_template):
    """loss_utils.py:150-164."""
    N = len(meshes)
    vp = meshes.verts_packed()
    if vp.is_cuda:
        equal = getattr(meshes, "_equal_sized", None)
        vpm = meshes.verts_list()[0].shape[0] if (equal is not None and equal()) else 0
        return ops.edge_rigidity_sum(vp, meshes.edges_packed(), mesh_template.verts_packed(),
                                     mesh_template.edges_packed(), vpm) / N
    v0, v1 = _edge_lengths(meshes)
    t0, t1 = _edge_lengths(mesh_template)
    loss = ((v0 - v1).norm(dim=1, p=2) - (t0 - t1).norm(dim=1, p=2)) ** 2
    return loss.sum() / N


class Locally_Rigid(nn.Module):
    def forward(self, meshes, mesh_template):
        return locally_rigid_fn(meshes, mesh_template)


def texture_dt_loss_v(texture_flow, dist_transf, vis_rend=None, cams=None, verts=None, tex_pred=None,
                      reduce=True):
    """loss_utils.py:172-191."""
    V = texture_flow.size(1)
    grid = texture_flow.view(-1, V, 1, 2)
    d = F.grid_sample(dist_transf, grid, align_corners=True)
    if reduce:
        return d.mean()
    return d.mean(-1).mean(-1).squeeze(1)


def texture_loss(img_pred, img_gt, mask_pred, mask_gt):
    """loss_utils.py:194-201."""
    return F.l1_loss(img_pred * mask_pred.unsqueeze(1), img_gt * mask_gt.unsqueeze(1))


def masked_texture_mse(texture_pred, imgs, masks):
    """The texture MSE term the reference writes inline (multiframe/main.py:655-662):
    F.mse_loss(texture_pred * masks[:, None], imgs * masks[:, None], reduction='none').mean((1, 2, 3))
    -> [N], fused into one pass (imgs / masks are constants)."""
    return ops.tex_mse(texture_pred, imgs, masks)


def bds_loss(verts, bds, faces, pix_to_face, reduce=True, n_samples=1000, k=1):
    """loss_utils.py:204-237.  verts [B,V,2] projected vertices, bds [B,P,3] = (x, y, valid),
    pix_to_face [B,H,W,K] (slot 0 = nearest face)."""
    if k != 1:
        raise NotImplementedError("bds_loss: only k=1 (what the reference uses) is built")
    bt, nv, _ = verts.shape
    indices = torch.randperm(bds.shape[1])[:n_samples]  # CPU generator, like the reference (:211)
    if bds.shape[1] > n_samples:
        bds = bds[..., indices.to(bds.device), :]
    vis = ops.visible_vertices(pix_to_face, faces, nv)
    loss = ops.bds_loss_per_mesh(verts, bds, vis)
    if reduce:
        return loss.mean()
    return loss


class Boundaries_Loss(nn.Module):
    def forward(self, verts, bds, faces, pix_to_face, reduce=True, n_samples=1000):
        return bds_loss(verts, bds, faces, pix_to_face, reduce=reduce, n_samples=n_samples)


def hinge_loss(loss, margin):
    """loss_utils.py:256-259 (the reference hard-codes .cuda(); here: the input's device)."""
    zeros = torch.zeros(1, device=loss.device, dtype=loss.dtype)
    return torch.max(loss - margin, zeros)


def quat_loss_geodesic(q1, q2):
    """loss_utils.py:262-277."""
    # only the real part of q1 (x) conj(q2) is needed: w1 w2 + x1 x2 + y1 y2 + z1 z2 in the
    # reference's operation order (products subtracted from the first with the conjugate's signs)
    a0, a1, a2, a3 = q1.unbind(-1)
    b0, b1, b2, b3 = q2[:, 0], -1 * q2[:, 1], -1 * q2[:, 2], -1 * q2[:, 3]
    real = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3
    return (1 - real.abs())[:, None]


def camera_loss(cam_pred, cam_gt, margin):
    """loss_utils.py:280-289."""
    rot_loss = hinge_loss(quat_loss_geodesic(cam_pred[:, -4:], cam_gt[:, -4:]), margin)
    st_loss = hinge_loss(((cam_pred[:, :3] - cam_gt[:, :3]) ** 2).view(-1), margin)
    return rot_loss.mean() + st_loss.mean()


def triangle_loss(verts, edge2verts):
    """loss_utils.py:292-319 (unused by main.py)."""
    idx = torch.stack([edge2verts, edge2verts, edge2verts], dim=2)
    vA = torch.gather(verts, 1, idx[:, :, :, 0])
    vB = torch.gather(verts, 1, idx[:, :, :, 1])
    vC = torch.gather(verts, 1, idx[:, :, :, 2])
    vD = torch.gather(verts, 1, idx[:, :, :, 3])
    n1 = F.normalize(geom_utils.cross_product(vD - vA, vB - vA), dim=2)
    n2 = F.normalize(geom_utils.cross_product(vB - vA, vC - vA), dim=2)
    return ((1 - (n1 * n2).sum(2)) ** 2).mean()


def deform_l2reg(V):
    """loss_utils.py:322-327."""
    return V.reshape(-1, V.shape[2]).norm(p=2, dim=1).mean()


def entropy_loss(A):
    """loss_utils.py:330-338."""
    return torch.mean(-torch.sum(A * torch.log(A), 1))


def kp_l2_loss(kp_pred, kp_gt, reduction='mean'):
    """loss_utils.py:341-356."""
    visible = kp_gt[..., 2].gt(0).to(kp_pred.dtype)
    per_kp = (kp_pred - kp_gt[..., :2]).abs().sum(-1) * visible          # L1, despite the name
    per_img = per_kp.mean(-1) / (visible.mean(-1) + 1e-4)
    return per_img.mean() if reduction == 'mean' else per_img


class PerceptualTextureLoss_v2(object):
    """loss_utils.py:359-383: LPIPS (AlexNet features, spatial map) between the masked rendered texture and the masked
    image, times the ground-truth mask, mean over the pixels.  The network is the third-party `lpips` package, kept as
    the torch module it is and imported when the loss is constructed: without the package the
    constructor raises an ImportError naming it, with it predictor.py:103-108 and main.py:333-335 construct and call
    this class unchanged.  The module lives on the GPU the process uses (the reference: `.cuda()` + nn.DataParallel)."""

    def __init__(self, net='alex', lpips_f=False):
        try:
            import lpips
        except ImportError as exc:
            raise ImportError("PerceptualTextureLoss_v2 needs the third-party `lpips` package (pip install lpips; the "
                              "reference pins none, docs/install.md): it is the AlexNet perceptual metric, not part of "
                              "the MI355X hot path -- every other loss of loss_utils works without it") from exc
        fn = lpips.LPIPS(net=net, lpips=lpips_f, spatial=True)
        if torch.cuda.is_available():
            fn = fn.cuda()
        self.loss_fn_alex = nn.DataParallel(fn)

    def __call__(self, img_pred, img_gt, mask_pred, mask_gt, reduce=True):
        """img_pred, img_gt [B,3,H,W]; mask_pred (unused, as in the reference), mask_gt [B,H,W] -> scalar or [B]."""
        mask_gt = mask_gt.unsqueeze(1)
        pred = 2 * (img_pred * mask_gt) - 1
        target = 2 * (img_gt * mask_gt) - 1
        dist = self.loss_fn_alex(pred, target) * mask_gt
        dist = dist.mean(-2).mean(-1).squeeze(-1)          # [B,1,H,W] -> [B], rows first as in the reference
        return dist.mean() if reduce else dist


class TexCycle(nn.Module):
    """loss_utils.py:386-416 (exported, never called by main.py): the learned texture flow of a
    face should average to the face's projected position, on visible faces only."""

    def __init__(self, im_size=256, nf=1280, eps=1e-12):
        super(TexCycle, self).__init__()

    def forward(self, flow, prob, aggr_info):
        batch, n_faces = flow.shape[:2]
        mean_flow = flow.reshape(batch, n_faces, -1, 2).mean(2)
        seen = torch.zeros(batch, n_faces, 1, device=flow.device, dtype=flow.dtype)
        for i in range(batch):
            ids = aggr_info[i].reshape(-1).long()
            seen[i, ids[ids >= 0].unique()] = 1
        return F.mse_loss(mean_flow * seen, prob * seen), mean_flow[0, :10]


def optical_flow_loss(meshes, faces, cams, flows, renderer, pix_to_face, reduce=True, loss_only=False,
                      flow_masks=None, flip_t=False):
    """loss_utils.py:419-474.  meshes [b,t,V,3], faces [b,t,F,3], cams [b*t,7],
    flows [b,t,H,W,2]; renderer: an OF_NeuralRenderer-like object (proj_fn + __call__).
    loss_only=True (not in the reference): return just the loss, from one fused kernel
    (ops.of_loss), instead of the 5-tuple with the per-vertex intermediates; with it, flows may be the data loader's
    [clips,t,H,W,2] for b = G*clips rendered clips, flipped in time (flip_t) and masked (flow_masks [clips*t,H,W]) inside
    the kernel instead of by main.py:676-686's flip / multiply / repeat(G)."""
    H, W = flows.shape[2:4]
    b, t, nv, _ = meshes.shape
    bt = b * t
    predicted_points = renderer.proj_fn(meshes.reshape(bt, nv, -1), cams.reshape(bt, -1))
    with torch.no_grad():
        faces_bt = faces.reshape(bt, faces.shape[2], 3).long()
        if pix_to_face is None:
            pix_to_face = renderer(predicted_points.reshape(bt, nv, 3), faces_bt)
        elif pix_to_face.shape[-1] != 1:
            pix_to_face = pix_to_face[..., :1]
        visible_vertices = ops.visible_vertices(pix_to_face, faces_bt, nv).reshape(b, t, nv)
    if loss_only and predicted_points.is_cuda:
        loss = ops.of_loss(predicted_points.reshape(bt, nv, 3), flows.reshape(-1, H, W, 2), visible_vertices, b, t,
                           masks=flow_masks, flip_t=flip_t)
        return loss.sum() if reduce else loss
    if flow_masks is not None or flip_t or flows.shape[0] != b:
        raise ValueError("optical_flow_loss: shared / flipped / masked flows need loss_only=True on the GPU")

    xy = predicted_points[..., :2]                                            # [bt, V, 2] in [-1, 1]
    # GT flow at each projected vertex: nearest pixel (:449-452)
    flow_img = flows.reshape(bt, H, W, -1).permute(0, 3, 1, 2)
    gt_flow = F.grid_sample(flow_img, xy[:, :, None, :], align_corners=False, mode='nearest')
    gt_flow = gt_flow.squeeze(-1).transpose(1, 2).reshape(b, t, nv, -1)
    # predicted flow: pixel displacement of the same vertex between frame k and k+1 (:455-459)
    xy_bt = xy.reshape(b, t, nv, 2)
    pix = W * (xy_bt + 1) / 2
    flow_pred = pix[:, :-1] - pix[:, 1:]
    # vertices that are visible AND land on a pixel with a non-zero GT flow, frames 1.. (:462-465)
    keep = (gt_flow.abs().sum(-1) != 0) & visible_vertices.bool()
    keep = keep[:, 1:].to(flow_pred.dtype).detach()
    gt_kept = keep[..., None] * gt_flow[:, 1:]
    flow_pred = keep[..., None] * flow_pred
    loss = (gt_kept - flow_pred).abs().sum(dim=(-1, -2))                      # L1 over x and y (:468-469)
    loss = loss / H / (keep.sum(-1) + 1)
    return (loss.sum() if reduce else loss), flow_pred, keep, xy_bt, gt_kept


class Optical_Flow_Loss(nn.Module):
    def forward(self, meshes, faces, cams, flows, renderer, pix_to_face, reduce=True):
        return optical_flow_loss(meshes, faces, cams, flows, renderer, pix_to_face, reduce=reduce)
