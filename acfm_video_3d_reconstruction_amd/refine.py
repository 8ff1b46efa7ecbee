"""Test-time refinement of a clip: the post-processing Adam loop of
multiframe/nnutils/predictor.py:287-349 (BASELINE.json configs[2], "multiframe camera+deform
solve") on the MI355X kernels.

Differences from the reference loop are formulation only: the reference re-factorises the
CONSTANT 642x642 system `cholesky(A_augm)` inside every iteration (predictor.py:313-315); here
P = M^-1 A^T is factorised once (deform.DeformSolver) and every iteration applies
v = v_mean + P (delta).  Same losses, same weights, same optimiser (Adam, lr 5e-3)."""
import torch

from . import ops
from .nnutils import loss_utils


def refine_total(renderer, solver, delta, cam, faces, masks, edts_barrier, boundaries, mask_loss_wt=1.0,
                 boundaries_reg_wt=1.0, edt_reg_wt=0.1, bdt_reg_wt=0.1, of_loss_wt=0.0, optical_flows=None,
                 of_renderer=None, num_frames=2):
    """The loss of one refinement iteration (predictor.py:310-345) for handle offsets `delta` [N,K_h,3] and
    cameras `cam` [N,7]: -> (total, pred_v).  refine_clip() differentiates exactly this."""
    pred_v = solver(delta)                                                    # predictor.py:310-315
    mask_pred, pix_to_face = renderer(pred_v, faces, cam)                     # :317
    pred_proj = renderer.project_points(pred_v, cam)                          # :319
    # mask_loss = l1.mean() (:318), edt_loss = edt.mean() (:320), bdt_loss = per-frame boundary term .mean() (:321);
    # the reference pairs bdt_reg_wt with the EDT term and edt_reg_wt with the boundary term (:322); total as at
    # :343-344.  On the GPU the three weighted per-frame means are ONE launch each way (loss_utils.combine_losses on
    # the raw [N,4] output of the silhouette-loss kernel + the per-frame boundary term) instead of ~20 elementwise /
    # reduction launches on N-element vectors.
    if mask_pred.is_cuda:
        raw = loss_utils.fused_silhouette_losses(mask_pred, masks, edts_barrier, raw=True)   # (l1, ., ., edt) per frame
        bdt = loss_utils.bds_loss(pred_proj, boundaries, faces, pix_to_face, reduce=False)   # :321
        total = loss_utils.combine_losses([raw, bdt], [mask_loss_wt, 0.0, 0.0, boundaries_reg_wt * bdt_reg_wt,
                                                       boundaries_reg_wt * edt_reg_wt])
    else:
        l1, _, edt = loss_utils.fused_silhouette_losses(mask_pred, masks, edts_barrier)
        bdt_loss = loss_utils.bds_loss(pred_proj, boundaries, faces, pix_to_face)  # :321
        per_frame = mask_loss_wt * l1 + (boundaries_reg_wt * bdt_reg_wt) * edt
        total = per_frame.mean() + (boundaries_reg_wt * edt_reg_wt) * bdt_loss
    if of_loss_wt > 0 and optical_flows is not None:
        b = optical_flows.shape[0]
        masks_of = masks.reshape(b, num_frames, masks.shape[1], masks.shape[2])
        pred_v_of = pred_v.reshape(b, num_frames, pred_v.shape[1], pred_v.shape[2])
        faces_of = faces.reshape(b, num_frames, faces.shape[1], 3)
        flows_f = torch.flip(optical_flows, dims=[1]) * masks_of[..., None]
        of_loss, _, _, _, _ = loss_utils.optical_flow_loss(pred_v_of, faces_of, cam, flows_f,
                                                           of_renderer, pix_to_face)   # :334-339
        total = total + of_loss_wt * of_loss
    return total, pred_v


class ClipRefiner:
    """The refinement loop as an object: `step()` performs one Adam iteration (render, losses, backward, update);
    `capture()` records one iteration into a hipGraph after `n_eager` eager ones (Adam's moment estimates exist, the
    allocator is warm) and `step()` replays it from then on.  refine_clip() drives it; bench.py --config 3 times
    `step()`.  State: `delta` [N,K_h,3] (handle offsets), `cam_raw` [N,7] (if the cameras are optimised), `history`."""

    def __init__(self, renderer, solver, delta_v_res, cam_pred, faces, masks, edts_barrier, boundaries,
                 optimize_camera=False, mask_loss_wt=1.0, boundaries_reg_wt=1.0, edt_reg_wt=0.1, bdt_reg_wt=0.1,
                 of_loss_wt=0.0, optical_flows=None, of_renderer=None, num_frames=2, lr=5e-3, capturable=False,
                 log_len=1):
        self.args = (renderer, solver, faces, masks, edts_barrier, boundaries)
        self.kw = (mask_loss_wt, boundaries_reg_wt, edt_reg_wt, bdt_reg_wt, of_loss_wt, optical_flows, of_renderer, num_frames)
        self.delta = delta_v_res.clone().detach().requires_grad_(True)
        params = [self.delta]
        self.optimize_camera = optimize_camera
        self.cam_raw = None
        if optimize_camera:
            # the reference optimises scale, trans and quat as three tensors (predictor.py:296-300); Adam is
            # element-wise, so one [N,7] leaf is the same optimisation with a third of the launches
            self.cam_raw = cam_pred.clone().detach().requires_grad_(True)
            params.append(self.cam_raw)
        self.capturable = bool(capturable and self.delta.is_cuda)
        # one fused multi-tensor kernel per step instead of ~a dozen foreach launches (the loop is
        # launch-latency-bound: ~30 short kernels of this library per iteration)
        self.opt = torch.optim.Adam(params, lr=lr, capturable=self.capturable, fused=bool(self.delta.is_cuda))
        self.cam_fixed = cam_pred.detach()
        self.cam_out, self.pred_v = self.cam_fixed, None
        self.hist_buf = torch.zeros(max(int(log_len), 1), device=self.delta.device)
        self.it_idx = torch.zeros((), dtype=torch.long, device=self.delta.device)
        self.graph = None
        self.iterations = 0

    def _iteration(self):
        renderer, solver, faces, masks, edts_barrier, boundaries = self.args
        cam = self.cam_fixed
        if self.optimize_camera:
            if self.cam_raw.is_cuda:
                cam = ops.camera_normalize(self.cam_raw)
            else:
                cam = torch.cat([self.cam_raw[:, :3], torch.nn.functional.normalize(self.cam_raw[:, 3:], dim=-1)], dim=1)
        total, pred_v = refine_total(renderer, solver, self.delta, cam, faces, masks, edts_barrier, boundaries, *self.kw)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        self.hist_buf.index_put_((self.it_idx.clamp(max=self.hist_buf.numel() - 1),), total.detach())   # the loss log stays on the device
        self.it_idx.add_(1)
        self.cam_out, self.pred_v = cam.detach(), pred_v.detach()

    def capture(self, n_eager=3):
        """n_eager eager iterations on a side stream, then one iteration recorded (not executed) into a hipGraph."""
        if not self.capturable:
            raise RuntimeError("ClipRefiner(capturable=True) on device tensors is needed to capture the iteration")
        dev = self.delta.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(n_eager):
                self._iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.iterations += n_eager
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()
        return n_eager

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._iteration()
        self.iterations += 1

    def history(self):
        n = min(self.iterations, self.hist_buf.numel())
        if self.graph is not None:
            ops.invalidate_setups()          # the replays rewrote the loop's tensors without version bumps
        return self.hist_buf[:n].tolist()


def refine_clip(renderer, solver, delta_v_res, cam_pred, faces, masks, edts_barrier, boundaries,
                num_optim_iter=20, optimize_camera=False, mask_loss_wt=1.0, boundaries_reg_wt=1.0,
                edt_reg_wt=0.1, bdt_reg_wt=0.1, of_loss_wt=0.0, optical_flows=None, of_renderer=None,
                num_frames=2, lr=5e-3, use_graph=False):
    """delta_v_res [N,K_h,3] predicted handle offsets, cam_pred [N,7], masks [N,H,W],
    edts_barrier [N,1,H,W], boundaries [N,P,3]; optional optical_flows [b,T,H,W,2].
    use_graph=True captures one whole iteration (render, losses, backward, Adam update) into a
    hipGraph after three eager iterations and replays it for the rest: the loop is launch-bound
    (~30 short kernels per iteration) and every entry point is stream-ordered, so the replayed
    iterations perform exactly the eager sequence of updates without returning to Python.
    Returns (pred_v, cam, delta, history of total losses)."""
    graphable = use_graph and delta_v_res.is_cuda
    r = ClipRefiner(renderer, solver, delta_v_res, cam_pred, faces, masks, edts_barrier, boundaries,
                    optimize_camera=optimize_camera, mask_loss_wt=mask_loss_wt, boundaries_reg_wt=boundaries_reg_wt,
                    edt_reg_wt=edt_reg_wt, bdt_reg_wt=bdt_reg_wt, of_loss_wt=of_loss_wt, optical_flows=optical_flows,
                    of_renderer=of_renderer, num_frames=num_frames, lr=lr, capturable=graphable, log_len=num_optim_iter)
    done = 0
    if graphable and num_optim_iter > 3:
        done = r.capture(3)
    for _ in range(num_optim_iter - done):
        r.step()
    history = r.history()
    pred_v = r.pred_v
    return (pred_v.clone() if pred_v is not None else None), r.cam_out.clone(), r.delta.detach().clone(), history
