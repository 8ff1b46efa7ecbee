"""ctypes binding of libacfm_hip.so (C ABI declared in include/acfm_hip.h).

This is the only place the shared library is loaded.  It fails loudly: a missing library or
a CPU tensor is an error, never a silent fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACFM_LIB: an alternative build of the same C ABI (the diagnostic build libacfm_hip_diag.so, or a kernel
# variant under test in tools/variants.py); never a fallback -- the path must exist.
SO_PATH = os.environ.get("ACFM_LIB") or os.path.join(_HERE, "libacfm_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/acfm_hip.h
SIGNATURES = {
    "acfm_version": (_i, []),
    "acfm_arch": (ctypes.c_char_p, []),
    "acfm_prof_enable": (_i, [_i]),
    "acfm_prof_collect": (_i, [_vp, _vp, _i]),
    "acfm_prof_name": (ctypes.c_char_p, [_i]),
    "acfm_project": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_project_backward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "acfm_project_xy": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_project_xy_backward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "acfm_deform_apply": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_deform_apply_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "acfm_deform_presolve_sums_f64": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "acfm_correlation_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_of_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_of_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_of_loss_shared": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_of_loss_shared_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_camera_pipeline": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_camera_pipeline_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_camera_mirror": (_i, [_vp, _i, _vp, _vp]),
    "acfm_camera_pipeline_tables": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_camera_pipeline_tables_backward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "acfm_camera_normalize": (_i, [_vp, _i, _vp, _vp]),
    "acfm_camera_normalize_backward": (_i, [_vp, _vp, _i, _vp, _vp]),
    "acfm_deform_solve_workspace_bytes": (_sz, [_i, _i]),
    "acfm_deform_solve": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "acfm_deform_solve_backward": (_i, [_vp, _i, _i, _vp, _sz, _vp, _vp]),
    "acfm_deform_solve_info": (_i, [_vp, _sz, _i, _vp, _vp]),
    "acfm_deform_solve_info_offset": (_sz, [_i]),
    "acfm_raster_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "acfm_stream_capture_id": (_i, [_vp, _vp]),
    "acfm_sil_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                              _sz, _vp, _vp]),
    "acfm_sil_forward_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                 _sz, _vp, _vp, _vp]),
    "acfm_sil_backward_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                                  _sz, _i, _vp, _vp, _vp]),
    "acfm_sil_loss_forward_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "acfm_sil_loss_backward_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _f, _vp,
                                       _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "acfm_sil_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                               _sz, _i, _vp, _vp]),
    "acfm_sil_loss_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _sz, _vp, _vp]),
    "acfm_sil_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _f, _vp,
                                    _vp, _vp, _sz, _i, _vp, _vp]),
    "acfm_hard_raster": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "acfm_tex_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp,
                              _vp, _sz, _i, _f, _i, _vp, _vp]),
    "acfm_vertex_color_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _sz,
                                       _i, _f, _vp, _vp]),
    "acfm_tex_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_tex_backward_faces": (_i, [_vp, _vp, _vp, _sz, _f, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_tex_mse_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _sz, _i, _f, _i, _vp, _vp]),
    "acfm_tex_mse_backward_faces": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _f, _i, _i, _i, _i, _i, _i, _vp, _vp,
                                         _vp]),
    "acfm_combine_losses": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "acfm_combine_losses_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "acfm_hypothesis_total": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "acfm_hypothesis_total_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_texture_cycle_scratch_floats": (_sz, [_i, _i, _i, _i]),
    "acfm_texture_cycle": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_texture_cycle_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "acfm_mask_losses": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_mask_losses_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_tex_mse": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_tex_mse_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "acfm_cot_laplacian": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "acfm_laplacian_smoothing_state_floats": (_sz, [_i, _i]),
    "acfm_laplacian_smoothing": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_laplacian_smoothing_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_edge_rigidity": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "acfm_edge_rigidity_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "acfm_edt_workspace_bytes": (_sz, [_i, _i, _i]),
    "acfm_edt": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "acfm_boundaries": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_visible_vertices": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "acfm_bds_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "acfm_bds_loss_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
}

_ERR = {1: "ACFM_E_BADARG (shape/parameter outside what the kernels support)",
        2: "ACFM_E_LAUNCH (HIP launch failed)", 3: "ACFM_E_WORKSPACE (workspace too small)"}

_LIB = None


def build(verbose=False):
    """Compile libacfm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    return SO_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libacfm_hip.so is missing (%s). Build it with `python -c \"import __graft_entry__ "
                "as g; g.build()\"` or `make -C %s`. There is no CPU fallback." % (SO_PATH, CSRC))
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


PROF_NKERNELS = 24  # ACFM_PROF_NKERNELS


def prof_collect():
    """-> {kernel name: (total ms, launches)} since acfm_prof_enable(1) / the last collect."""
    ms = (ctypes.c_float * PROF_NKERNELS)()
    cnt = (ctypes.c_int * PROF_NKERNELS)()
    check(lib().acfm_prof_collect(ms, cnt, PROF_NKERNELS), "acfm_prof_collect")
    return {lib().acfm_prof_name(i).decode(): (ms[i], cnt[i]) for i in range(PROF_NKERNELS) if cnt[i]}


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cur_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "error %d" % rc)))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("acfm_video_3d_reconstruction_amd ops run on the GPU only "
                               "(got a %s tensor); there is no CPU fallback" % t.device)


class RasterTuning(ctypes.Structure):
    """AcfmRasterTuning of include/acfm_hip.h (per call; results never depend on it)."""
    _fields_ = [("split_mode", _i), ("grid_div", _i * 3), ("flags", _i)]


_TUNE = __import__("threading").local()


class raster_tuning:
    """with raster_tuning(split=0, grid_div=(1, 0, 0)): ...   -- every raster op called inside the block
    (on this thread) passes this tuning to the C ABI; outside any block the library defaults apply.
    Tests use it to force the split / unsplit kernels and every grid divisor; the library itself keeps
    no tuning state."""

    def __init__(self, split=-5, grid_div=(0, 0, 0), deterministic=False, record_cover=None):
        """deterministic=True: the silhouette backward accumulates in fixed point (bit-reproducible run to run).
        record_cover: True / False forces ACFM_RECORD_COVER (the silhouette render leaves the nearest covering face
        per pixel for a texture render that takes its workspace over) on / off; None lets ops.py decide -- on while
        texture renders do follow the silhouette renders on this device.
        (Bit 1 of the flags, half storage, is not set here: it changes buffer types, so the ops that support it take
        storage="f16" and set it themselves -- with_f16() below.)"""
        self.t = RasterTuning(int(split), (_i * 3)(*[int(d) for d in grid_div]), 1 if deterministic else 0)
        self.t.record_cover = record_cover   # (a Python attribute: not part of the C structure)

    def __enter__(self):
        self.prev = getattr(_TUNE, "cur", None)
        _TUNE.cur = self.t
        return self

    def __exit__(self, *exc):
        _TUNE.cur = self.prev


def tuning():
    """-> (ctypes pointer or None, the structure to keep alive / to save for the backward)."""
    t = getattr(_TUNE, "cur", None)
    return (ctypes.byref(t) if t is not None else None), t


def tuning_ptr(t):
    return ctypes.byref(t) if t is not None else None


def with_f16(t, on):
    """The tuning in effect with flags bit 1 (ACFM_STORE_F16) set or cleared: a fresh structure."""
    if t is None:
        return RasterTuning(-5, (_i * 3)(0, 0, 0), 2) if on else None
    return RasterTuning(t.split_mode, (_i * 3)(*t.grid_div), (t.flags & ~2) | (2 if on else 0))


def with_cover(t, on):
    """The tuning in effect with flags bit 2 (ACFM_RECORD_COVER) set or cleared: a fresh structure (None = defaults)."""
    if t is None:
        return RasterTuning(-5, (_i * 3)(0, 0, 0), 4) if on else None
    if bool(t.flags & 4) == bool(on):
        return t
    return RasterTuning(t.split_mode, (_i * 3)(*t.grid_div), (t.flags & ~4) | (4 if on else 0))


class SilExtras(ctypes.Structure):
    """AcfmSilExtras of include/acfm_hip.h (every field optional)."""
    _fields_ = [("proj_xy", _vp), ("grad_proj_xy", _vp), ("tex_imgs", _vp), ("tex_sil", _vp), ("tex_pix_to_face", _vp),
                ("tex_texel_idx", _vp)]


def sil_extras(proj_xy=None, grad_proj_xy=None, prefill=None):
    """-> (ctypes pointer, the structure to keep alive) for the `_ex` silhouette entry points; tensors or None."""
    p = lambda t: None if t is None else t.data_ptr()
    pf = prefill or (None, None, None, None)
    e = SilExtras(p(proj_xy), p(grad_proj_xy), p(pf[0]), p(pf[1]), p(pf[2]), p(pf[3]))
    return ctypes.byref(e), e


_CONSTS = {}


def const(values, device, dtype=None):
    """Small constant tensor on `device`, uploaded once per (values, device, dtype): a fresh
    torch.tensor(..., device=cuda) is a pageable host-to-device copy, which is not allowed while
    a stream is being captured into a hipGraph."""
    import torch
    dtype = dtype or torch.float32
    key = (tuple(values), str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t
