"""Template deformation solve.

The reference (multiframe/main.py:586-609, nnutils/predictor.py:260-276, 313-315) builds, for
every frame, M = L^T L + A^T A and b = L^T L v + A^T (A v + delta) and calls torch.cholesky +
cholesky_solve on B*T identical 642x642 systems.  Algebraically b = M v + A^T delta, so

        pred_v_n = v + P delta_n,      P = M^-1 A^T   (V x K_h)

P depends only on the handle weights (lbs) and on the no-grad Laplacian of the mean shape; both
are learned (mesh_net.py:480, 543), so it is rebuilt ONCE per optimiser step: on the GPU by the
blocked fp64 Cholesky of csrc/acfm_solve.hip (ops.deform_solve: more accurate than the
reference's own fp32 Cholesky), and applied per frame as a [V x K_h] . [K_h x 3]
product (ops.deform_apply).  Gradients: d delta_n = P^T g_n, d v = sum_n g_n (L is a constant,
geom_utils.py:245), d lbs through the saved factor (acfm_deform_solve_backward).
"""
import torch
from torch import nn

from .nnutils import geom_utils


class _OneMesh:
    """The 4-method view of a single mesh that geom_utils.mesh_laplacian needs."""

    def __init__(self, verts, faces):
        self._v, self._f, self.device = verts, faces, verts.device

    def isempty(self):
        return False

    def verts_packed(self):
        return self._v

    def faces_packed(self):
        return self._f


def handle_matrix(lbs_logits):
    """MeshNet.get_lbs().permute(1, 0) (mesh_net.py:597-599, main.py:586): A [K_h, V]."""
    return torch.softmax(lbs_logits, dim=0).permute(1, 0)


def solve_matrix(L, A):
    """P = (L^T L + A^T A)^-1 A^T in fp64 via torch.linalg (host tensors / cross-check of the
    native kernel; main.py:605-608 collapsed)."""
    L64, A64 = L.double(), A.double()
    M = L64.t() @ L64 + A64.t() @ A64
    u = torch.linalg.cholesky(M)
    return torch.cholesky_solve(A64.t(), u)


def _solve(L, lbs_logits):
    if lbs_logits.is_cuda:
        from . import ops
        return ops.deform_solve(L, lbs_logits)
    return solve_matrix(L, handle_matrix(lbs_logits)).float()


class DeformSolver(nn.Module):
    """mean_v / lbs_logits may be nn.Parameters (learned, as in the reference) or plain tensors."""

    def __init__(self, mean_v, faces, lbs_logits, method="cot"):
        super().__init__()
        self.method = method
        if isinstance(mean_v, nn.Parameter):
            self.mean_v = mean_v
        else:
            self.register_buffer("mean_v", mean_v.detach().clone())
        self.register_buffer("faces", faces.detach().clone().long())
        if isinstance(lbs_logits, nn.Parameter):
            self.lbs = lbs_logits
        else:
            self.register_buffer("lbs", lbs_logits.detach().clone())
        self._P = None
        self._L = None

    def laplacian(self, mean_v=None):
        v = (self.mean_v if mean_v is None else mean_v).detach()
        return geom_utils.mesh_laplacian(_OneMesh(v, self.faces), self.method)

    def refresh(self, mean_v=None):
        """Call after lbs or the mean shape (which defines L) changed, i.e. once per optimiser
        step: the next solve_matrix() rebuilds L and re-factorises."""
        self._P = None
        self._L = self.laplacian(mean_v)

    def solve_matrix(self):
        """P [V,K_h]; one factorisation per refresh().  With lbs requiring grad (and grad mode on)
        the factorisation is part of the autograd graph of this step."""
        if self._L is None:
            self._L = self.laplacian()
        if self.lbs.requires_grad and torch.is_grad_enabled():
            P = _solve(self._L, self.lbs)
            self._P = P.detach()
            return P
        if self._P is None:
            with torch.no_grad():
                self._P = _solve(self._L, self.lbs)
        return self._P

    def forward(self, delta, mean_override=None):
        """delta [N,K_h,3] handle offsets -> deformed verts [N,V,3] (delta = 0: the mean shape)."""
        mean = self.mean_v if mean_override is None else mean_override
        P = self.solve_matrix()
        if delta.is_cuda:
            from . import ops
            return ops.deform_apply(mean, P, delta)   # f32 MFMA kernels (csrc/acfm_deform.hip)
        return mean[None] + torch.matmul(P[None], delta)  # host tensors: CPU-side tests of the algebra


def deform_reference_formula(lbs_logits, mean_v, delta, L):
    """The reference's expression, term by term (main.py:586-609), on device tensors;
    for tests and for callers that want the literal formulation."""
    A = handle_matrix(lbs_logits)
    nb = delta.shape[0]
    A_b = A[None].repeat(nb, 1, 1)
    mean_b = mean_v[None].repeat(nb, 1, 1)
    delta_v = A_b.bmm(mean_b) + delta
    L_b = L[None].repeat(nb, 1, 1)
    d = torch.bmm(L_b, mean_b)
    A_augm = L_b.permute(0, 2, 1).matmul(L_b) + A_b.permute(0, 2, 1).matmul(A_b)
    b = L_b.permute(0, 2, 1) @ d + A_b.permute(0, 2, 1) @ delta_v
    u = torch.linalg.cholesky(A_augm)
    return torch.cholesky_solve(b, u)
