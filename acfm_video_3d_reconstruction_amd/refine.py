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
    delta = delta_v_res.clone().detach().requires_grad_(True)
    params = [delta]
    if optimize_camera:
        # the reference optimises scale, trans and quat as three tensors (predictor.py:296-300); Adam is
        # element-wise, so one [N,7] leaf is the same optimisation with a third of the launches
        cam_raw = cam_pred.clone().detach().requires_grad_(True)
        params.append(cam_raw)
    graphable = use_graph and delta.is_cuda
    # one fused multi-tensor kernel per step instead of ~a dozen foreach launches (the loop is
    # launch-latency-bound: ~30 short kernels of this library per iteration)
    opt = torch.optim.Adam(params, lr=lr, capturable=graphable, fused=bool(delta.is_cuda))
    state = {"cam": cam_pred.detach(), "pred_v": None}
    hist_buf = torch.zeros(max(num_optim_iter, 1), device=delta.device)
    it_idx = torch.zeros((), dtype=torch.long, device=delta.device)

    def iteration():
        cam = state["cam"]
        if optimize_camera:
            if cam_raw.is_cuda:
                from . import ops
                cam = ops.camera_normalize(cam_raw)
            else:
                cam = torch.cat([cam_raw[:, :3], torch.nn.functional.normalize(cam_raw[:, 3:], dim=-1)], dim=1)
        total, pred_v = refine_total(renderer, solver, delta, cam, faces, masks, edts_barrier, boundaries,
                                     mask_loss_wt, boundaries_reg_wt, edt_reg_wt, bdt_reg_wt, of_loss_wt,
                                     optical_flows, of_renderer, num_frames)
        opt.zero_grad(set_to_none=True)
        total.backward()
        opt.step()
        hist_buf.index_put_((it_idx,), total.detach())      # loss log stays on the device
        it_idx.add_(1)
        state["cam_out"], state["pred_v"] = cam.detach(), pred_v.detach()

    n_eager = num_optim_iter if not graphable else min(3, num_optim_iter)
    if graphable and num_optim_iter > n_eager:
        side = torch.cuda.Stream(device=delta.device)
        side.wait_stream(torch.cuda.current_stream(delta.device))
        with torch.cuda.stream(side):
            for _ in range(n_eager):
                iteration()
        torch.cuda.current_stream(delta.device).wait_stream(side)
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            iteration()                      # recorded, not executed
        for _ in range(num_optim_iter - n_eager):
            graph.replay()
        ops.invalidate_setups()              # the replays rewrote the loop's tensors without version bumps
    else:
        for _ in range(num_optim_iter):
            iteration()
    history = hist_buf[:num_optim_iter].tolist()
    pred_v = state["pred_v"]
    return (pred_v.clone() if pred_v is not None else None), state.get("cam_out", state["cam"]).clone(), \
        delta.detach().clone(), history
