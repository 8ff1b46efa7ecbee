"""Drop-in for multiframe/nnutils/geom_utils.py.  Projection runs in the HIP library; the
tiny quaternion helpers and the (no-grad, once-per-step) cot Laplacian are torch device ops."""
import torch

from .. import ops


def orthographic_proj_withz(X, cam, offset_z=0.):
    """geom_utils.py:62-79.  X [B,N,3], cam [B,7] = (s, tx, ty, quat wxyz) -> [B,N,3]."""
    return ops.project(X, cam, offset_z)


def orthographic_proj(X, cam):
    """geom_utils.py:48-59 -> [B,N,2]."""
    return ops.project_xy(X, cam, 0.)


def quat_rotate(X, q):
    """geom_utils.py:134-152 (q is used as given, not normalised)."""
    cam = torch.cat([torch.ones_like(q[:, :1]), torch.zeros_like(q[:, :2]), q], 1)
    return ops.project(X, cam, 0.)


def cross_product(qa, qb):
    """geom_utils.py:82-104."""
    a0, a1, a2 = qa.unbind(-1)
    b0, b1, b2 = qb.unbind(-1)
    return torch.stack([a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0], dim=-1)


def hamilton_product(qa, qb):
    """geom_utils.py:107-131."""
    a0, a1, a2, a3 = qa.unbind(-1)
    b0, b1, b2, b3 = qb.unbind(-1)
    return torch.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                        a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                        a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
                        a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], dim=-1)


def sample_textures(texture_flow, images):
    """geom_utils.py:11-28 (exported by the reference, never called on the hot path)."""
    T = texture_flow.size(-2)
    F = texture_flow.size(1)
    grid = texture_flow.view(-1, F, T * T, 2)
    s = torch.nn.functional.grid_sample(images, grid, align_corners=True)
    return s.view(-1, 3, F, T, T).permute(0, 2, 3, 4, 1)


def sample_textures_v(texture_flow, images):
    """geom_utils.py:31-45."""
    V = texture_flow.size(-2)
    grid = texture_flow.view(-1, V, 1, 2)
    s = torch.nn.functional.grid_sample(images, grid, align_corners=True)
    return s.squeeze(-1).permute(0, 2, 1)


def laplacian_cot(meshes):
    """geom_utils.py:257-324 -> (dense W [V,V] with W[i,j] = cot a_ij + cot b_ij, inv_areas [V,1]).
    The reference returns W sparse; callers only use it through mesh_laplacian."""
    verts = meshes.verts_packed()
    faces = meshes.faces_packed()
    V = verts.shape[0]
    fv = verts[faces]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    A = (v1 - v2).norm(dim=1)
    B = (v0 - v2).norm(dim=1)
    C = (v0 - v1).norm(dim=1)
    s = 0.5 * (A + B + C)
    area = (s * (s - A) * (s - B) * (s - C)).clamp_(min=1e-12).sqrt()
    A2, B2, C2 = A * A, B * B, C * C
    cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], dim=1)
    cot /= 4.0
    ii = faces[:, [1, 2, 0]].reshape(-1)
    jj = faces[:, [2, 0, 1]].reshape(-1)
    W = torch.zeros(V, V, dtype=verts.dtype, device=verts.device)
    W.index_put_((ii, jj), cot.reshape(-1), accumulate=True)
    W = W + W.t()
    inv_areas = torch.zeros(V, dtype=verts.dtype, device=verts.device)
    inv_areas.scatter_add_(0, faces.reshape(-1), torch.stack([area] * 3, dim=1).reshape(-1))
    nz = inv_areas > 0
    inv_areas[nz] = 1.0 / inv_areas[nz]
    return W, inv_areas.view(-1, 1)


def mesh_laplacian(meshes, method="uniform"):
    """geom_utils.py:158-254: dense Laplacian L [V,V] (constant: computed under no_grad)."""
    if meshes.isempty():
        return torch.tensor([0.0], dtype=torch.float32, device=meshes.device, requires_grad=True)
    with torch.no_grad():
        if method == "uniform":
            L = meshes.laplacian_packed().to_dense()
        elif method in ["cot"]:
            verts = meshes.verts_packed()
            if verts.is_cuda:
                L = ops.cot_laplacian(verts, meshes.faces_packed())   # one fused kernel
            else:
                W, _ = laplacian_cot(meshes)
                L = W - torch.diag(W.sum(dim=1))
        else:
            raise ValueError("method should be one of {uniform, cot}")
    return L
