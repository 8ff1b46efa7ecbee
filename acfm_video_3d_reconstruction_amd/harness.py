"""Caller-side pieces of the hot path that the reference keeps inline in its trainer
(SURVEY section 8 rows a17, a20): camera decode / mirror / affine transforms and the
hypothesis weighting.  Small differentiable device ops (the camera embeddings are optimised
through them during pose warm-up, train_utils.py:186-213)."""
import torch
import torch.nn.functional as F

from .pytorch3d_shim.transforms import matrix_to_quaternion, quaternion_multiply, standardize_quaternion


def decode_cameras(cam_emb, scale_lr_decay=1.0):
    """multiframe/main.py:572-577 (== :452-457, predictor.py:248-252): embedding [...,7] ->
    (s = relu(decay*e0 + 1) + 1e-12, t = e1..2, q = normalize(e3..6))."""
    scales = F.relu(scale_lr_decay * cam_emb[..., :1] + 1) + 1e-12
    quats = F.normalize(cam_emb[..., 3:], dim=-1)
    return torch.cat([scales, cam_emb[..., 1:3], quats], dim=-1)


_MIRROR_Q = {}


def _mirror_rotation(device):
    """Quaternion of diag(-1, 1, -1) (main.py:104); built once per device, outside any capture."""
    key = str(device)
    if key not in _MIRROR_Q:
        diag = torch.diag(torch.tensor([-1., 1., -1.]))[None]
        _MIRROR_Q[key] = matrix_to_quaternion(diag).to(device)
    return _MIRROR_Q[key]


def _mirror_quat(q):
    return quaternion_multiply(_mirror_rotation(q.device), standardize_quaternion(q))


def _mirrored_pose(sfm_pose):
    return torch.cat([sfm_pose[:, :1], -sfm_pose[:, 1:2], sfm_pose[:, 2:3], _mirror_quat(sfm_pose[:, -4:])],
                     dim=-1)


def mirror_cameras(sfm_pose, img_shape, mirror_flag):
    """multiframe/main.py:113-125: pose of the horizontally flipped image, blended by flag."""
    m = mirror_flag.float()
    return (1 - m) * sfm_pose + _mirrored_pose(sfm_pose) * m


def mirror_sample(img, sfm_pose, mask_pred, mask):
    """multiframe/main.py:97-110."""
    return (torch.flip(img, dims=(3,)), _mirrored_pose(sfm_pose), torch.flip(mask_pred, dims=(2,)),
            torch.flip(mask, dims=(2,)))


def transform_cameras(sfm_pose, im_shape, transforms):
    """multiframe/main.py:128-138: crop/scale augmentation applied to the camera
    (transforms [...,4] = (a, dx, dy, flag))."""
    flag = transforms[..., -1].unsqueeze(-1).float()
    scale = sfm_pose[:, :1] * transforms[..., :1]
    tx = sfm_pose[:, 1:2] * transforms[..., :1] + transforms[..., 1:2]
    ty = sfm_pose[:, 2:3] * transforms[..., :1] + transforms[..., 2:3]
    new = torch.cat([scale, tx, ty, sfm_pose[:, -4:]], dim=-1)
    return (1 - flag) * sfm_pose + new * flag


def hypothesis_total(terms, weights, G, N, aux_group=None, aux_weights=None):
    """multiframe/main.py:716-746 in one launch each way on the GPU (ops.hypothesis_total); on the CPU the same from
    torch ops.  Returns (weighted, total [G,N], probs [G,N], aux [2,G,N], means [12])."""
    ts = [t.reshape(G, N) for t in terms]
    if ts[0].is_cuda:
        from . import ops
        return ops.hypothesis_total(ts, weights, G, N, aux_group, aux_weights)
    total = sum(float(w) * t for w, t in zip(weights, ts))
    weighted, probs, mean_total = hypothesis_weighting(total)
    aux = torch.zeros(2, G, N, dtype=total.dtype)
    if aux_group is not None:
        for t, k, w in zip(ts, aux_group, aux_weights):
            if k >= 0:
                aux[k] = aux[k] + float(w) * t.detach()
    means = torch.stack([weighted.detach(), mean_total.detach(), aux[0].mean(), aux[1].mean()]
                        + [t.detach().mean() for t in ts] + [total.new_zeros(())] * (8 - len(ts)))
    return weighted, total.detach(), probs, aux, means


def hypothesis_weighting(loss_per_hyp):
    """multiframe/main.py:735-745: loss [G, B*T] -> (total = mean_n sum_g w*L,
    w = softmax(-L, 0) detached, mean loss that the reference logs as `camera_loss`)."""
    probs = torch.softmax(-loss_per_hyp, dim=0).detach()
    return (loss_per_hyp * probs).sum(0).mean(), probs, loss_per_hyp.mean()
