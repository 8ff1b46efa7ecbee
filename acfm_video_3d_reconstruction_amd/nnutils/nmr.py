"""Drop-in for the reference's multiframe/nnutils/nmr.py (same classes, same forward
signatures, same return shapes/dtypes) running on the gfx950 kernels of libacfm_hip.so.

reference                                   here
NeuralRenderer.forward (nmr.py:143-200)     ops.sil_render / ops.tex_render
NeuralRenderer.project_points (:127-129)    ops.project
OF_NeuralRenderer.forward (:224-238)        ops.hard_raster
"""
import math
import weakref

import torch

from .. import ops
from . import geom_utils


_LAST_PROJ = weakref.WeakKeyDictionary()   # renderer -> (key, projection of its last silhouette render): see project_points


class NeuralRenderer(torch.nn.Module):
    """Soft-silhouette (K=20, blur ln(9999)*1e-4, sigma 1e-4) and atlas-texture (K=1)
    renderer with a weak-perspective camera.  Stateless and replicable (the reference wraps
    it in nn.DataParallel, main.py:183-193); honours the input tensors' device and the
    current HIP stream."""

    def __init__(self, img_size=256, faces_per_pixel=20, sigma=1e-4, gamma=1e-4, pix_to_face_slots=None, storage="f32"):
        """pix_to_face_slots: None -> pix_to_face [N,H,W,faces_per_pixel] int64 like the reference's, as an
        ops.LazyPixToFace: the render writes the nearest-face plane (all the reference's callers read:
        `pix_to_face[..., 0]` loss_utils.py:214, `[..., :1]` :431 -- both are views of it) and the other planes are
        rendered when something first touches them; faces_per_pixel -> every slot is stored at render time
        (160 bytes per pixel at K = 20); 1 -> only the nearest-face plane [N,H,W,1] exists.
        The mask always blends the faces_per_pixel nearest faces."""
        super().__init__()
        self.img_size = img_size
        # storage="f16" (BASELINE config 5, "fp16 render with fp32 loss accumulate"): rendered masks / images and the
        # atlas are held in float16 and pix_to_face is the int32 nearest-face plane [N,H,W,1]; every decision of the
        # rasteriser and every loss sum stays float32 (ids identical to the float32 renderer)
        self.storage = storage
        self.pix_to_face_slots = 1 if storage == "f16" else ("lazy" if pix_to_face_slots is None else pix_to_face_slots)
        self.faces_per_pixel = faces_per_pixel          # nmr.py:158
        self.sigma = sigma                              # nmr.py:153
        self.gamma = gamma
        self.blur_radius = math.log(1. / 1e-4 - 1.) * sigma  # nmr.py:157
        self.proj_fn = geom_utils.orthographic_proj_withz    # nmr.py:117
        self.offset_z = 0.                                   # nmr.py:119 (monocular: 5.)

    def ambient_light_only(self):  # nmr.py:121 (no-op in the reference too)
        return

    def set_bgcolor(self, color):  # nmr.py:124
        return

    def project_points(self, verts, cams):  # nmr.py:127-129
        if self.proj_fn is geom_utils.orthographic_proj_withz:   # the default: (x, y) straight from the kernel
            # the trainer projects the prediction it has just rendered (main.py:620 / :715, predictor.py:317 / :319):
            # the silhouette render's face setup produced exactly these (x, y) -- same function, same bits -- as an
            # output of its own autograd node, so the boundary loss's gradient returns through that render's ONE
            # projection backward (no second projection kernel, no sum of two vertex / camera gradients).  Handed out
            # once, and only for the very tensors of the render (storage, version, shape).
            hit = _LAST_PROJ.get(self)
            if hit is not None and verts.dtype == torch.float32 and hit[0] == ops._proj_key(verts, cams):
                del _LAST_PROJ[self]
                return hit[1]
            return ops.project_xy(verts, cams, 0.)
        return self.proj_fn(verts, cams)[:, :, :2]

    def _remember_proj(self, pix_to_face):
        # (kept beside the module, not in it: the projection is a non-leaf tensor, which copy.deepcopy / pickling of
        # the module must never meet)
        proj = getattr(pix_to_face, "_acfm_proj", None) if self.proj_fn is geom_utils.orthographic_proj_withz else None
        if proj is not None:
            _LAST_PROJ[self] = proj
        else:
            _LAST_PROJ.pop(self, None)

    def rasterize_of(self, verts, faces, R=None, T=None):
        """nmr.py:131-141: hard K=1 raster of already-projected verts.  The reference passes
        the look_at R/T of OF_NeuralRenderer (R = diag(-1, 1, 1), T = (0, 0, 2.732), nmr.py:224-231);
        only that fixed view is built into the kernels: any other R / T raises instead of being ignored."""
        if R is not None:
            Rm = torch.as_tensor(R, dtype=torch.float32).reshape(-1, 3, 3).cpu()
            if not torch.allclose(Rm, torch.diag(torch.tensor([-1., 1., 1.]))[None].expand_as(Rm), atol=1e-6):
                raise ValueError("rasterize_of: only the reference's view R = diag(-1, 1, 1) is supported")
        if T is not None:
            Tm = torch.as_tensor(T, dtype=torch.float32).reshape(-1, 3).cpu()
            if not torch.allclose(Tm, torch.tensor([0., 0., 2.732])[None].expand_as(Tm), atol=1e-6):
                raise ValueError("rasterize_of: only the reference's view T = (0, 0, 2.732) is supported")
        return ops.hard_raster(verts, faces, self.img_size)

    def forward(self, vertices, faces, cams, textures=None, atlas=True):
        if textures is None:
            self.mask_only = True
            masks, pix_to_face = ops.sil_render(vertices, faces, cams, self.img_size,
                                                K=self.faces_per_pixel, blur=self.blur_radius,
                                                sigma=self.sigma, offset_z=self.offset_z,
                                                k_out=self.pix_to_face_slots, storage=self.storage)
            self._remember_proj(pix_to_face)
            return masks, pix_to_face
        self.mask_only = False
        if not atlas:  # nmr.py:177-179: Textures(verts_rgb), visualisation only (no gradients)
            if textures.ndim == 2:
                textures = textures[None]
            return ops.vertex_color_render(vertices, faces, cams, textures.to(vertices.device),
                                           self.img_size, sigma=1e-4, gamma=1e-4, offset_z=self.offset_z)
        imgs, sil, pix_to_face = ops.tex_render(vertices, faces, cams, textures.to(vertices.device),
                                                self.img_size, sigma=1e-4, gamma=1e-4,
                                                offset_z=self.offset_z, storage=self.storage)
        return imgs, sil, pix_to_face


    def forward_silhouette_losses(self, vertices, faces, cams, mask_gt, edt, eps=1e-6, raw=False):
        """Opt-in fused operator (no reference counterpart as ONE call: it is `forward` followed by
        loss_utils.l1_loss / iou / edt_loss, multiframe/main.py:637-645, 715-716): the loss terms leave the raster
        kernel with the mask, the backward needs no [N,H,W] mask gradient.
        -> ((l1 [N], iou [N], edt [N]) or the raw [N,4] vector, mask_pred (no gradient), pix_to_face)."""
        out, masks, pix_to_face = ops.sil_render_losses(vertices, faces, cams, self.img_size, mask_gt, edt,
                                                        K=self.faces_per_pixel, blur=self.blur_radius,
                                                        sigma=self.sigma, offset_z=self.offset_z,
                                                        k_out=self.pix_to_face_slots, storage=self.storage)
        self._remember_proj(pix_to_face)
        if raw:
            return out, masks, pix_to_face
        return (out[:, 0], out[:, 1] / (out[:, 2] + eps), out[:, 3]), masks, pix_to_face


    def forward_texture_mse(self, vertices, faces, cams, textures, imgs, masks):
        """Opt-in fused operator: `forward(..., textures=...)` followed by the masked MSE of multiframe/main.py:655-662,
        F.mse_loss(texture_pred * masks, imgs * masks, reduction='none').mean((1, 2, 3)), as one call.
        -> (mse [N], texture_pred (no gradient), sil, pix_to_face)."""
        return ops.tex_render_mse(vertices, faces, cams, textures.to(vertices.device), imgs, masks, self.img_size,
                                  sigma=1e-4, gamma=1e-4, offset_z=self.offset_z, storage=self.storage)


class OF_NeuralRenderer(torch.nn.Module):
    """nmr.py:203-238: visibility rasteriser for the optical-flow loss."""

    def __init__(self, img_size=256):
        super().__init__()
        self.img_size = img_size
        self.proj_fn = geom_utils.orthographic_proj_withz
        self.offset_z = 5.

    def project_points(self, verts, cams):
        if self.proj_fn is geom_utils.orthographic_proj_withz:
            return ops.project_xy(verts, cams, 0.)
        return self.proj_fn(verts, cams)[:, :, :2]

    def forward(self, verts, faces):
        return ops.hard_raster(verts, faces, self.img_size)
