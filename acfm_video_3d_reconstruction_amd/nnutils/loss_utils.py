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

from .. import ops
from . import geom_utils


def _is_mask_batch(predict, target):
    return predict.is_cuda and predict.dim() >= 2 and predict.shape == target.shape


def iou(predict, target, eps=1e-6, reduce=True):
    """loss_utils.py:18-28."""
    out = ops.mask_losses(predict.reshape(predict.shape[0], -1), target.reshape(target.shape[0], -1))
    r = out[:, 1] / (out[:, 2] + eps)
    if reduce:
        return r.sum() / r.nelement()
    return r


def iou_loss(predict, target, reduce=True):
    """loss_utils.py:31-32."""
    return 1 - iou(predict[:, None], target[:, None], reduce=reduce)


def l1_loss(predict, target, reduce=True):
    """loss_utils.py:72-77."""
    out = ops.mask_losses(predict.reshape(predict.shape[0], -1), target.reshape(target.shape[0], -1))
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
    out = ops.mask_losses(mask_rendered.reshape(bsize, -1), None, edt.reshape(bsize, -1))
    return out[:, 3].mean() if reduce else out[:, 3]


def fused_silhouette_losses(mask_pred, mask_gt, edt, eps=1e-6):
    """One pass over the rendered mask for all three silhouette terms
    -> (l1 [N], iou [N], edt [N]); equals l1_loss / iou / edt_loss with reduce=False."""
    N = mask_pred.shape[0]
    out = ops.mask_losses(mask_pred.reshape(N, -1), mask_gt.reshape(N, -1), edt.reshape(N, -1))
    return out[:, 0], out[:, 1] / (out[:, 2] + eps), out[:, 3]


def quat_conj(q):
    return torch.cat([q[:, :, [0]], -1 * q[:, :, 1:4]], dim=-1)


def quat2ang(q):
    ang = 2 * torch.acos(torch.clamp(q[:, :, 0], min=-1 + 1E-6, max=1 - 1E-6))
    return ang.unsqueeze(-1)


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
    q1 = torch.unsqueeze(q1, 1)
    q2 = torch.unsqueeze(q2, 1)
    q2_conj = torch.cat([q2[:, :, [0]], -1 * q2[:, :, 1:4]], dim=-1)
    q_rel = geom_utils.hamilton_product(q1, q2_conj)
    return 1 - torch.abs(q_rel[:, :, 0])


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
    V = V.view(-1, V.size(2))
    return torch.mean(torch.norm(V, p=2, dim=1))


def entropy_loss(A):
    """loss_utils.py:330-338."""
    return torch.mean(-torch.sum(A * torch.log(A), 1))


def kp_l2_loss(kp_pred, kp_gt, reduction='mean'):
    """loss_utils.py:341-356."""
    vis = (kp_gt[:, :, 2] > 0).float()
    loss = torch.nn.L1Loss(reduction='none')(kp_pred, kp_gt[:, :, :2]).sum(-1) * vis
    loss = loss.mean(-1) / (vis.mean(-1) + 1e-4)
    if reduction == 'mean':
        return loss.mean()
    return loss


class PerceptualTextureLoss_v2(object):
    """loss_utils.py:359-383 wraps the third-party `lpips` AlexNet; out of scope here."""

    def __init__(self, net='alex', lpips_f=False):
        raise NotImplementedError("LPIPS is a third-party network (lpips package) and is not part of "
                                  "the MI355X hot path; plug the reference's module in unchanged")


class TexCycle(nn.Module):
    """loss_utils.py:386-416 (exported, never called by main.py): the learned texture flow of a
    face should average to the face's projected position, on visible faces only."""

    def __init__(self, im_size=256, nf=1280, eps=1e-12):
        super(TexCycle, self).__init__()

    def forward(self, flow, prob, aggr_info):
        nb, nf, nr, _, _ = flow.size()
        avg_flow = torch.mean(flow.view(nb, nf, -1, 2), dim=2)
        mask = torch.zeros(avg_flow.size(), device=avg_flow.device)
        for cnt in range(nb):
            fids = torch.unique(aggr_info[cnt]).long()
            mask[cnt, fids[fids >= 0], :] = 1
        loss = torch.nn.MSELoss()(avg_flow * mask, prob * mask)
        return loss, avg_flow[0, 0:10, :]


def optical_flow_loss(meshes, faces, cams, flows, renderer, pix_to_face, reduce=True):
    """loss_utils.py:419-474.  meshes [b,t,V,3], faces [b,t,F,3], cams [b*t,7],
    flows [b,t,H,W,2]; renderer: an OF_NeuralRenderer-like object (proj_fn + __call__)."""
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

    pts = predicted_points[:, :, None, :2]
    fl = flows.reshape(bt, H, W, -1).permute(0, 3, 1, 2)
    samples_ofs_gt = F.grid_sample(fl, pts, align_corners=False, mode='nearest')
    samples_ofs_gt = samples_ofs_gt[..., 0].permute(0, 2, 1).reshape(b, t, nv, -1)

    predicted_points = pts.reshape(b, t, nv, -1)
    predicted_points_ = W * (predicted_points + 1) / 2
    of_pred = predicted_points_[:, :-1] - predicted_points_[:, 1:]

    visible_vertices = (samples_ofs_gt.abs().sum(-1) != 0).bool() * visible_vertices.bool()
    visible_vertices = visible_vertices.float()[:, 1:].detach()
    samples_ofs_gt = visible_vertices[..., None] * samples_ofs_gt[:, 1:]
    of_pred = visible_vertices[..., None] * of_pred
    loss = torch.norm(samples_ofs_gt[..., 0] - of_pred[..., 0], p=1, dim=-1) + torch.norm(
        samples_ofs_gt[..., 1] - of_pred[..., 1], p=1, dim=-1)
    loss = loss / H / (visible_vertices.sum(-1) + 1)
    if reduce:
        loss = loss.sum()
    return loss, of_pred, visible_vertices, predicted_points, samples_ofs_gt


class Optical_Flow_Loss(nn.Module):
    def forward(self, meshes, faces, cams, flows, renderer, pix_to_face, reduce=True):
        return optical_flow_loss(meshes, faces, cams, flows, renderer, pix_to_face, reduce=reduce)
