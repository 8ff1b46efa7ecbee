"""pytorch3d.loss.mesh_laplacian_smoothing (called at multiframe/main.py:703 with 'cot',
monocular/main.py:276 with 'uniform'); semantics: SURVEY App-A.7."""
import torch


def _cot_weights(verts, faces):
    """Per-face cotangents/4 exactly as geom_utils.laplacian_cot (geom_utils.py:272-298)."""
    fv = verts[faces]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    A = (v1 - v2).norm(dim=1)
    B = (v0 - v2).norm(dim=1)
    C = (v0 - v1).norm(dim=1)
    s = 0.5 * (A + B + C)
    area = (s * (s - A) * (s - B) * (s - C)).clamp_(min=1e-12).sqrt()
    A2, B2, C2 = A * A, B * B, C * C
    cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], dim=1)
    return cot / 4.0


def mesh_laplacian_smoothing(meshes, method: str = "uniform"):
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
    N = len(meshes)
    verts = meshes.verts_packed()
    faces = meshes.faces_packed()
    weights = meshes.inv_num_verts_packed()
    V = verts.shape[0]
    if verts.is_cuda and method in ("cot", "uniform"):
        from .. import ops  # fused gfx950 kernels (csrc/acfm_mesh.hip); torch ops below = host tensors
        conn = faces if method == "cot" else meshes.edges_packed()
        vpm = fpm = 0
        if method == "cot" and meshes._equal_sized():    # mesh m = verts [m V, (m+1) V), faces [m F, (m+1) F)
            vpm, fpm = meshes.verts_list()[0].shape[0], meshes.faces_list()[0].shape[0]
        return ops.laplacian_smoothing_sum(verts, conn, weights, 0 if method == "cot" else 1, vpm, fpm) / N
    if method == "uniform":
        L = meshes.laplacian_packed()
        loss = torch.sparse.mm(L, verts)
    elif method in ("cot", "cotcurv"):
        with torch.no_grad():
            cot = _cot_weights(verts.detach(), faces)                 # weights are constants
            ii = faces[:, [1, 2, 0]].reshape(-1)
            jj = faces[:, [2, 0, 1]].reshape(-1)
            w = cot.reshape(-1)
            rows = torch.cat([ii, jj])
            cols = torch.cat([jj, ii])
            ww = torch.cat([w, w])
            rowsum = torch.zeros(V, dtype=verts.dtype, device=verts.device).index_add_(0, rows, ww)
            if method == "cot":
                norm_w = torch.where(rowsum > 0, 1.0 / rowsum, rowsum).view(-1, 1)
            else:
                area = torch.zeros(V, dtype=verts.dtype, device=verts.device)
                fv = verts.detach()[faces]
                a = 0.5 * torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1).norm(dim=1)
                area.index_add_(0, faces.reshape(-1), a.repeat_interleave(3))
                norm_w = 0.25 * torch.where(area > 0, 1.0 / area, area).view(-1, 1)
        Lv = torch.zeros_like(verts).index_add_(0, rows, ww[:, None] * verts[cols])   # (W v)
        loss = Lv * norm_w - verts if method == "cot" else (Lv - verts) * norm_w
    else:
        raise ValueError("Method should be one of {uniform, cot, cotcurv}")
    loss = loss.norm(dim=1) * weights
    return loss.sum() / N
