"""The hot part of ACFM's multiframe training step on the MI355X ops.

`MultiframeStep.warmup` / `.forward` replay the call sequence of `ShapeTrainer.warmup`
(multiframe/main.py:438-520) and `ShapeTrainer.forward` (:523-765) for everything that sits
between the network heads and the loss scalar: per-frame camera / probability / deformation
embeddings with G hypotheses (mesh_net.py:404-451), camera decode + mirror + affine transforms,
the deformation solve, silhouette / texture renders, silhouette + boundary + optical-flow +
keypoint losses, the hypothesis softmax weighting and the shape priors.  The learned networks
(ResNet encoder, texture and camera heads, mesh_net.py) are out of scope: their outputs
(`delta_v_res`, `textures`) are inputs here.

Tensors follow the reference's shapes: B clips of T frames, N = B*T, G hypotheses, G*N meshes."""
import math
import contextlib
from types import SimpleNamespace

import torch
from torch import nn

from . import harness
from .deform import DeformSolver
from .nnutils import loss_utils
from .nnutils.nmr import NeuralRenderer, OF_NeuralRenderer
from .pytorch3d_shim.loss import mesh_laplacian_smoothing
from .pytorch3d_shim.structures import Meshes

# defaults of the reference's flags (multiframe/main.py:55-89)
DEFAULTS = dict(num_frames=2, num_guesses=8, num_lbs=15, kp_loss_wt=0., of_loss_wt=1., mask_loss_wt=1.,
                rigid_wt=0.5, deform_reg_wt=1., handle_deform_reg_wt=0., boundaries_reg_wt=1., edt_reg_wt=0.1,
                bdt_reg_wt=2., triangle_reg_wt=0.1, tex_loss_wt=.5, scale_lr_decay=0.05, optimize_deform=False,
                optimize_deform_lr=100., drop_hypothesis=False, texture=False, cam_loss_wt=2., deform_loss_wt=2.)


def texture_cycle_loss(textures, num_frames):
    """The texture temporal-consistency term of ShapeTrainer.forward, literally (multiframe/main.py:705-711):
    the per-frame atlases [B*T,F,R,R,3] are regrouped as [B,F,R,R,T,3] and then RESHAPED to [-1,R,R] -- the
    reference's own regrouping of the trailing (R, T, 3) block, reproduced as written, not "fixed" -- and the
    loss is the mean L2 norm (over the last axis) of the differences of neighbours along axis 1."""
    if textures.is_cuda:      # the same regrouping and norms in two launches forward, one backward (csrc/acfm_loss.hip)
        from . import ops
        return ops.texture_cycle(textures, num_frames)
    t_c = textures.reshape(-1, num_frames, *textures.shape[1:]).permute(0, 2, 3, 4, 1, 5)
    t_c = t_c.reshape(-1, t_c.shape[2], t_c.shape[3])
    return torch.norm(t_c[:, :-1] - t_c[:, 1:], p=2, dim=-1).mean()


def _y_rotation_quats(num_guesses):
    """mesh_net.py:424-434: hypothesis g starts as a rotation of 360*g/(G-1) degrees about +y."""
    ang = torch.linspace(0, 360, num_guesses) * math.pi / 180.0
    return torch.stack([torch.cos(ang / 2), torch.zeros_like(ang), torch.sin(ang / 2), torch.zeros_like(ang)], 1)


class MultiframeStep(nn.Module):
    def __init__(self, mean_v, faces, lbs_logits, num_training_frames, img_size=256, vert2kp=None, prior_stream=False,
                 **opts):
        super().__init__()
        o = dict(DEFAULTS)
        o.update(opts)
        self.opts = SimpleNamespace(**o)
        self.prior_stream, self._prior_s = bool(prior_stream), None   # mesh priors beside the raster kernels (forward)
        G = self.opts.num_guesses
        self.num_cameras = G          # embeddings; opts.num_guesses may later drop below it (train_utils.py:236-241)
        q0 = _y_rotation_quats(G)
        self.cameras = nn.ModuleList()
        for g in range(G):                                       # mesh_net.py:436-444
            emb = nn.Embedding(num_training_frames, 7)
            with torch.no_grad():
                emb.weight[:, 0] = 0
                emb.weight[:, 1:3] = (torch.rand(2) - 0.5) * 0.1
                emb.weight[:, 3:] = q0[g] + 0.1 * torch.rand(4)
            self.cameras.append(emb)
        self.prob_embeddings = nn.Embedding(num_training_frames, G)
        self.prob_embeddings.weight.data.fill_(1)               # mesh_net.py:445-446
        self.deform_emb = nn.Embedding(num_training_frames, self.opts.num_lbs * 3)
        self.deform_mirror_emb = nn.Embedding(num_training_frames, self.opts.num_lbs * 3)
        self.deform_emb.weight.data.zero_()                     # mesh_net.py:448-451
        self.deform_mirror_emb.weight.data.zero_()
        self.lbs = nn.Parameter(lbs_logits.detach().clone())    # mesh_net.py:543-544
        self.mean_v = nn.Parameter(mean_v.detach().clone())     # mesh_net.py:480 (symmetrisation: identity here)
        self.register_buffer("faces1", faces.detach().clone().long())
        self.solver = DeformSolver(self.mean_v, faces, self.lbs)
        self.vert2kp = vert2kp
        self.renderer = NeuralRenderer(img_size)
        self.tex_renderer = NeuralRenderer(img_size)
        self.of_renderer = OF_NeuralRenderer(img_size)

    def make_exchange(self, group=None, average=True):
        """The frame-sharded step's one collective: pre-solve sums of the shared shape parameters (lbs,
        mean shape) + any other learned shared tensor (vert2kp); per-frame embeddings stay on the owning rank.
        average=True: ranks hold equal shards and per-rank losses are means over them."""
        from .sharding import SharedShapeExchange
        extra = [self.vert2kp] if isinstance(self.vert2kp, nn.Parameter) else []
        return SharedShapeExchange(self.solver, extra_params=extra, group=group, average=average)

    def set_num_guesses(self, k):
        """train_utils.py:236-241: after the first epochs only the k most probable hypotheses of
        every frame are rendered (opts.drop_hypothesis)."""
        self.opts.num_guesses = max(1, min(int(k), self.num_cameras))

    def selected_hypotheses(self, frames_idx):
        """main.py:541-548: indices [k,B,T] of the k most probable camera embeddings of each frame,
        or None when all of them are active."""
        if not self.opts.drop_hypothesis or self.opts.num_guesses >= self.num_cameras:
            return None
        w = self.prob_embeddings.weight.data[frames_idx]                       # [B,T,G_all]
        return w.topk(self.opts.num_guesses, largest=True, dim=-1, sorted=True)[1].permute(2, 0, 1)

    # ------------------------------------------------------------------ cameras (main.py:551-584)
    def hypothesis_cameras(self, frames_idx, mirror_flag, transforms, detach=False, selected=None):
        G = self.opts.num_guesses
        w0 = self.cameras[0].weight
        if w0.is_cuda:   # look-ups + stack / top-k gather + decode + mirror + transform fused (csrc/acfm_camera.hip)
            from . import ops
            cam_pred = ops.camera_pipeline_tables([emb.weight for emb in self.cameras], frames_idx, mirror_flag,
                                                  transforms, self.opts.scale_lr_decay, num_guesses=G,
                                                  selected=selected)
            return cam_pred.detach() if detach else cam_pred
        cams = torch.stack([emb(frames_idx) for emb in self.cameras])          # [G_all,B,T,7]
        if selected is not None:                                               # main.py:568-570
            cams = torch.gather(cams, 0, selected[..., None].expand(-1, -1, -1, 7))
        cams = cams.reshape(G, -1, 7)          # (host tensors: the reference's chain of torch ops, harness.py)
        cam_pred = harness.decode_cameras(cams, self.opts.scale_lr_decay).reshape(-1, 7)
        cam_pred = harness.mirror_cameras(cam_pred, None, mirror_flag.repeat(G)[:, None])
        cam_pred = harness.transform_cameras(cam_pred, None, transforms.repeat(G, 1))
        return cam_pred.detach() if detach else cam_pred

    def _silhouette_terms(self, pred_v, faces, cam, batch, G, parts=False):
        o = self.opts
        mask_pred, pix_to_face = self.renderer(pred_v, faces, cam)
        # the frame's ground truth is shared by its G hypotheses (the loss kernels index it n % N: the
        # reference's masks / edts / boundaries .repeat(G, ...) copies, main.py:472-479, are not made)
        l1, _, edt = loss_utils.fused_silhouette_losses(mask_pred, batch["masks"], batch["edts_barrier"])
        pred_proj = self.renderer.project_points(pred_v, cam)
        bdt = loss_utils.bds_loss(pred_proj, batch["boundaries"], faces, pix_to_face, reduce=False)
        if parts:
            return mask_pred, l1, edt, bdt
        sil_cons = o.edt_reg_wt * edt + o.bdt_reg_wt * bdt
        return mask_pred, l1, sil_cons

    def _flow_term(self, pred_v, cam, batch, G):
        o = self.opts
        if not (o.of_loss_wt > 0) or "optical_flows" not in batch:
            return torch.zeros(1, device=pred_v.device)
        T = o.num_frames
        B = batch["masks"].shape[0] // T
        faces_of = self.faces1[None, None].expand(G * B, T, -1, -1)
        if pred_v.is_cuda:    # main.py:676-686's flip / mask / repeat(G) of the flow images happen inside the loss kernel
            of_loss = loss_utils.optical_flow_loss(pred_v.reshape(G * B, T, -1, 3), faces_of, cam, batch["optical_flows"],
                                                   self.of_renderer, pix_to_face=None, reduce=False, loss_only=True,
                                                   flow_masks=batch["masks"], flip_t=True)
            return of_loss.reshape(G, -1).repeat(1, T).reshape(G, -1)    # main.py:684-686
        masks_of = batch["masks"].reshape(B, T, *batch["masks"].shape[1:])
        flows = (torch.flip(batch["optical_flows"], dims=[1]) * masks_of[..., None]).repeat(G, 1, 1, 1, 1)
        of_loss = loss_utils.optical_flow_loss(pred_v.reshape(G * B, T, -1, 3), faces_of, cam, flows,
                                               self.of_renderer, pix_to_face=None, reduce=False, loss_only=True)
        return of_loss.reshape(G, -1).repeat(1, T).reshape(G, -1)    # main.py:684-686

    # ------------------------------------------------------------------ main.py:438-520
    def warmup(self, batch):
        """Pose warm-up: render the undeformed mean shape under all G camera hypotheses; only the
        camera embeddings receive gradients; writes the hypothesis probabilities."""
        o = self.opts
        G = o.num_guesses
        N = batch["masks"].shape[0]
        cam = self.hypothesis_cameras(batch["frames_idx"], batch["mirror_flag"], batch["transforms"])
        mean_v = self.solver.mean_v[None].repeat(G * N, 1, 1)
        faces = self.faces1[None].expand(G * N, -1, -1)
        _, mask_loss, sil_cons = self._silhouette_terms(mean_v, faces, cam, batch, G)
        total = o.mask_loss_wt * mask_loss.reshape(G, N) + o.of_loss_wt * self._flow_term(mean_v, cam, batch, G) \
            + o.boundaries_reg_wt * sil_cons.reshape(G, N)
        if o.kp_loss_wt > 0 and self.vert2kp is not None:
            kp_v = torch.matmul(torch.softmax(self.vert2kp, dim=1), mean_v)
            kp = loss_utils.kp_l2_loss(self.renderer.project_points(kp_v, cam), batch["kps"].repeat(G, 1, 1),
                                       reduction="none")
            total = total + o.kp_loss_wt * kp.reshape(G, N)
        probs = torch.softmax(-total, dim=0).detach()
        with torch.no_grad():                                    # main.py:517-519
            self.prob_embeddings.weight[batch["frames_idx"]] = \
                probs.reshape(G, *batch["frames_idx"].shape).permute(1, 2, 0)
        return total.mean(), probs

    # ------------------------------------------------------------------ main.py:523-765
    def forward(self, batch, delta_v_res, textures=None, imgs=None, detach_camera=False, drop_deform=False,
                predicted_camera=None, exchange=None):
        """delta_v_res [N,K_h,3]: handle offsets predicted by the (out-of-scope) encoder head;
        predicted_camera [N,7] (optional): output of its camera head, pulled towards the most
        probable hypothesis (main.py:753-762).  Returns (total_loss, dict of the reference's named terms)."""
        o = self.opts
        G, T = o.num_guesses, o.num_frames
        N = delta_v_res.shape[0]
        selected = self.selected_hypotheses(batch["frames_idx"])
        cam = self.hypothesis_cameras(batch["frames_idx"], batch["mirror_flag"], batch["transforms"],
                                      detach=detach_camera, selected=selected)
        # deformation (main.py:531-539, 586-609): per-frame embeddings when optimize_deform, else the
        # encoder's prediction; delta = 0 when drop_deform
        deforms = None
        if o.optimize_deform:
            flag = batch["mirror_flag"][:, None, None].float()
            d0 = self.deform_emb(batch["frames_idx"]).reshape(-1, o.num_lbs, 3)
            d1 = self.deform_mirror_emb(batch["frames_idx"]).reshape(-1, o.num_lbs, 3)
            deforms = ((1 - flag) * d0 + flag * d1) * o.optimize_deform_lr
        if drop_deform:
            delta = torch.zeros_like(delta_v_res)
        elif o.optimize_deform:
            delta = deforms
        else:
            delta = delta_v_res
        if exchange is not None:
            # frame-sharded step (sharding.SharedShapeExchange over self.solver; `batch` holds this rank's clips,
            # sharding.frame_shard): the local backward stops at the (P, mean) leaves, exchange.finish() after
            # loss.backward() all-reduces the pre-solve sums and finishes d lbs identically on every rank
            pred_v1 = exchange.apply(delta)
        else:
            self.solver.refresh()   # lbs / mean shape moved in the last optimiser step: one factorisation
            pred_v1 = self.solver(delta)                         # [N,V,3]
        pred_v = pred_v1.repeat(G, 1, 1)
        faces = self.faces1[None].expand(G * N, -1, -1)
        terms = {}
        # priors on the deformed shape (main.py:698-714, 748-751).  They need the deformed vertices only and are a
        # dozen one-workgroup-per-mesh kernels (latency bound); prior_stream=True issues them on a second stream
        # beside the raster kernels (a fork/join when captured into a hipGraph), the streams meet again before the
        # total is formed.  Off by default: measured on the 96-mesh step it LOSES (2.48 -> 2.73 ms as one hipGraph),
        # while the simpler 64-frame step of bench.py (survey_8d_step) gains 3.5 % from the same fork.
        faces_n = self.faces1[None].expand(N * G, -1, -1)
        cur_s = side_s = None
        if self.prior_stream and pred_v.is_cuda:
            cur_s = torch.cuda.current_stream(pred_v.device)
            if self._prior_s is None or self._prior_s.device != pred_v.device:
                self._prior_s = torch.cuda.Stream(device=pred_v.device)
            side_s = self._prior_s
            side_s.wait_stream(cur_s)
            pred_v.record_stream(side_s)
        with (torch.cuda.stream(side_s) if side_s is not None else contextlib.nullcontext()):
            mesh_3d = Meshes(verts=pred_v, faces=faces_n)
            mesh_t = Meshes(verts=self.solver.mean_v[None].repeat(G * N, 1, 1), faces=faces_n)
            triangle = mesh_laplacian_smoothing(mesh_3d, method="cot")
            rigid = loss_utils.locally_rigid_fn(mesh_3d, mesh_t)
        # The per-hypothesis total (main.py:716-734: weight * term + ...) and its weighting over the hypotheses
        # (:735-746) are formed from the terms by ONE operator each way (harness.hypothesis_total), which also
        # returns the sums the reference logs; the terms themselves come from the kernels above.
        mask_pred, mask_loss, edt, bdt = self._silhouette_terms(pred_v, faces, cam, batch, G, parts=True)
        of_term = self._flow_term(pred_v, cam, batch, G)
        tl, tw, tg, ta = [mask_loss, edt, bdt], [o.mask_loss_wt, o.boundaries_reg_wt * o.edt_reg_wt,
                                                  o.boundaries_reg_wt * o.bdt_reg_wt], [-1, 0, 0], \
            [0.0, o.edt_reg_wt, o.bdt_reg_wt]
        if o.of_loss_wt > 0 and "optical_flows" in batch:     # (else _flow_term returned main.py:688's zeros(1) placeholder)
            if of_term.numel() != G * N:
                # main.py:684-686 lays the per-clip loss out with repeat(1, T): G*B*(T-1)*T elements, = G*N only for the
                # T = 2 clips ACFM trains on; the reference's `total_loss += of_loss_wt * of_loss` fails to broadcast otherwise
                raise ValueError("optical-flow term has %d elements for %d hypotheses x %d frames (num_frames must be 2, "
                                 "multiframe/main.py:684-686)" % (of_term.numel(), G, N))
            tl.append(of_term); tw.append(o.of_loss_wt); tg.append(-1); ta.append(0.0)
        if o.kp_loss_wt > 0 and self.vert2kp is not None:
            kp_v = torch.matmul(torch.softmax(self.vert2kp, dim=1), pred_v)
            kp = loss_utils.kp_l2_loss(self.renderer.project_points(kp_v, cam), batch["kps"].repeat(G, 1, 1),
                                       reduction="none")
            tl.append(kp); tw.append(o.kp_loss_wt); tg.append(-1); ta.append(0.0)
        have_tex = textures is not None and imgs is not None
        if have_tex:
            # texture branch on detached geometry, original + mirrored camera (main.py:627-636, 655-662;
            # the LPIPS part of the reference's texture loss is out of scope)
            tex = textures    # [N,...] shared by the G hypotheses of a frame: the op indexes n % N (= repeat(G))
            tex_pred, _, _ = self.tex_renderer(pred_v.detach(), faces, cam, textures=tex)
            # main.py:97-110 (mirror_sample) without flipping the rendered masks nobody reads
            imgs_f, masks_f = torch.flip(imgs, dims=(3,)), torch.flip(batch["masks"], dims=(2,))
            if cam.is_cuda:   # one kernel instead of ~35 quaternion launches; the texture render sends its cameras no gradient
                from . import ops
                cam_f = ops.camera_mirror(cam)
            else:
                cam_f = harness._mirrored_pose(cam)
            tex_pred_f, _, _ = self.tex_renderer(pred_v.detach(), faces, cam_f, textures=tex)
            tl += [loss_utils.masked_texture_mse(tex_pred, imgs, batch["masks"]),
                   loss_utils.masked_texture_mse(tex_pred_f, imgs_f, masks_f)]          # mse = their mean
            tw += [0.5 * o.tex_loss_wt] * 2; tg += [1, 1]; ta += [0.5, 0.5]
            cycle = texture_cycle_loss(textures, T)              # main.py:705-711, added at :749
        weighted, total, probs, aux, means = harness.hypothesis_total(tl, tw, G, N, tg, ta)
        cam_loss = means[1]
        terms.update(cam_pred=cam.detach(), pred_v=pred_v1.detach(), mask_loss=mask_loss.reshape(G, N).detach(),
                     sil_cons_per_hyp=aux[0], of_loss=of_term.detach(), total_per_hyp=total, weighted=weighted.detach())
        if have_tex:
            terms["tex_mse"] = means[3]
            terms["tex_mse_per_hyp"] = aux[1]
        if selected is not None:                   # probabilities go back to the embeddings they came from (:737-742)
            with torch.no_grad():
                fi = batch["frames_idx"]
                pw = self.prob_embeddings.weight
                cur = torch.zeros_like(pw[fi]).permute(2, 0, 1)                # [G_all,B,T]
                pw[fi] = torch.scatter(cur, 0, selected, probs.reshape(G, *fi.shape)).permute(1, 2, 0)
        if side_s is not None:
            cur_s.wait_stream(side_s)
            triangle.record_stream(cur_s)
            rigid.record_stream(cur_s)
        handle = loss_utils.deform_l2reg(delta_v_res)
        loss = weighted + o.rigid_wt * rigid + o.triangle_reg_wt * triangle + o.handle_deform_reg_wt * handle
        if textures is not None and imgs is not None:
            loss = loss + o.deform_reg_wt * cycle                # main.py:749 (deform_reg_wt weighs the texture cycle term)
            terms["cycle"] = cycle.detach()
        if predicted_camera is not None:           # main.py:753-762: camera head vs the most probable hypothesis
            best = probs.reshape(G, N).argmax(dim=0)
            cam_sel = cam.reshape(G, N, 7)[best, torch.arange(N, device=cam.device)]
            cam_head = loss_utils.camera_loss(predicted_camera, cam_sel.detach(), 0)
            loss = loss + o.cam_loss_wt * cam_head
            terms["cam_loss"] = cam_head.detach()
        if o.optimize_deform:                      # main.py:763-765: encoder head vs the per-frame deformation embedding
            deform_loss = torch.nn.functional.mse_loss(delta_v_res, deforms.detach())
            loss = loss + o.deform_loss_wt * deform_loss
            terms["deform_loss"] = deform_loss.detach()
        terms.update(mask=means[4], sil_cons=means[2], rigid=rigid.detach(),
                     triangle=triangle.detach(), camera_loss=cam_loss.detach(), probs=probs)
        return loss, terms
