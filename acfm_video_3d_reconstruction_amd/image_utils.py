"""On-device versions of the mask preparation of multiframe/utils/image.py (same function
names; masks stay on the GPU instead of the reference's numpy / device->host round trip,
multiframe/main.py:365-377)."""
import torch

from . import _lib


def _masks3(mask):
    m = mask.detach().to(torch.float32)
    if m.dim() == 2:
        m = m[None]
    return m.contiguous()


def compute_dt(mask, norm=True):
    """image.py:94-102: distance_transform_edt(1 - mask), optionally / max(H, W).
    mask [H,W] or [N,H,W] (0/1) on the GPU -> same shape, float32."""
    _lib.require_gpu(mask)
    m = _masks3(mask)
    N, H, W = m.shape
    out = torch.empty_like(m)
    nb = _lib.lib().acfm_edt_workspace_bytes(N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=m.device)
    divisor = max(H, W) if norm else 1
    with torch.cuda.device(m.device):
        _lib.check(_lib.lib().acfm_edt(_lib.ptr(m), N, H, W, int(divisor), _lib.ptr(out), _lib.ptr(ws), nb,
                                       _lib.cur_stream(m.device)), "acfm_edt")
    return out[0] if mask.dim() == 2 else out


def compute_dt_barrier(mask, k=50):
    """image.py:105-116: sigmoid(k * (dt_out - dt_in) / max(H, W))."""
    m = _masks3(mask)
    H, W = m.shape[-2:]
    diff = (compute_dt(m, norm=False) - compute_dt(1.0 - m, norm=False)) / max(H, W)
    out = 1.0 / (1.0 + torch.exp(k * -diff))
    return out[0] if mask.dim() == 2 else out


def compute_boundaries(masks):
    """image.py:122-146: masks [N,H,W] -> float32 [N, max_count, 3] = (x, y, valid) of the
    boundary pixels (find_boundaries, mode='thick'), padded to the longest list of the batch.
    One small device->host read (the counts) sizes the result, like the reference's max()."""
    _lib.require_gpu(masks)
    m = _masks3(masks)
    N, H, W = m.shape
    cap = H * W
    out = torch.empty((N, cap, 3), dtype=torch.float32, device=m.device)
    counts = torch.empty((N,), dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        _lib.check(_lib.lib().acfm_boundaries(_lib.ptr(m), N, H, W, cap, _lib.ptr(out), _lib.ptr(counts),
                                              _lib.cur_stream(m.device)), "acfm_boundaries")
    max_bd = int(counts.max().item())
    return out[:, :max_bd].contiguous()
