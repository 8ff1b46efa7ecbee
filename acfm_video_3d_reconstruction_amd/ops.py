"""torch.autograd bindings of the C ABI (include/acfm_hip.h).

torch is used here for device memory, streams and autograd bookkeeping only; every op body
is one or more stream-ordered calls into libacfm_hip.so."""
import ctypes
import math
import threading

import torch

from . import _lib

SIL_K = 20  # nmr.py:158
SIL_SIGMA = 1e-4  # nmr.py:153
SIL_BLUR = math.log(1.0 / 1e-4 - 1.0) * 1e-4  # nmr.py:157


def _f32c(t):
    if type(t) is LazyGrad:          # (defined below) a gradient another operator of this module has not formed yet
        t = t.materialize()
    return t.detach().to(torch.float32).contiguous()


def _real(t, f16):
    """Storage tensor of the raster ops: float32, or float16 with storage="f16" (ACFM_STORE_F16)."""
    return t.detach().to(torch.float16 if f16 else torch.float32).contiguous()


def _is_f16(storage):
    if storage not in ("f32", "f16"):
        raise ValueError("storage must be 'f32' or 'f16', got %r" % (storage,))
    return storage == "f16"


_FACES = {}   # (storage ptr, offset, strides, version, shape, N) -> (source kept alive, contiguous int64 [N,F,3])


def expand_faces(faces, N):
    """faces [F,3] / [1,F,3] / [N,F,3] -> contiguous int64 [N,F,3] on the same device.  The
    reference passes one face list broadcast over the batch (`faces[None].expand(N, -1, -1)`) to
    every render call; its materialised copy is memoised on the source tensor (storage, version)."""
    if faces.dim() == 2:
        faces = faces[None]
    if faces.shape[0] != N:
        if faces.shape[0] != 1:
            raise ValueError("faces batch %d does not match %d meshes" % (faces.shape[0], N))
        faces = faces.expand(N, -1, -1)
    if faces.dtype == torch.int64 and faces.is_contiguous():
        return faces
    key = (faces.data_ptr(), faces.storage_offset(), faces.stride(), faces._version, tuple(faces.shape),
           str(faces.dtype), str(faces.device))
    with _LOCK:
        hit = _FACES.get(key)
        if hit is None:
            if len(_FACES) > 8:
                _FACES.clear()
            hit = _FACES[key] = (faces, faces.to(torch.int64).contiguous())
    return hit[1]


def _workspace(N, V, F, H, device):
    nbytes = _lib.lib().acfm_raster_workspace_bytes(N, V, F, H)
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


# The face setup (projection, face records, coarse masks, schedule) of the last silhouette render
# per device.  The reference renders the texture of the same prediction right after its silhouette
# (multiframe/main.py:616-636): when verts / cams / faces are the same storage at the same version,
# the texture render takes the workspace over instead of setting up again (acfm_tex_forward,
# ws_ready).  The entry keeps its tensors alive, so an address cannot be recycled under it.
#
# What makes a take-over safe -- all of it is part of the key:
#   * same tensors: storage address, shape and autograd version of verts / cams / faces;
#   * same HIP stream (the texture render is ordered behind the silhouette render that filled ws);
#   * same capture: both calls are eager, or both are recorded into the SAME hipGraph capture (id from
#     hipStreamGetCaptureInfo) -- a workspace never crosses a graph boundary;
#   * same epoch: invalidate_setups() ends every sharing.  A hipGraph REPLAY rewrites the tensors its capture wrote
#     without touching their version counters, so whoever replays a graph and then hands such tensors to the renderers
#     eagerly announces it with invalidate_setups() (graphed.GraphedStep.replay and bench.py do; torch itself is not
#     edited -- round 2 wrapped torch.cuda.CUDAGraph.replay process-wide for this).  Writes that bypass the version
#     counters in other ways (`t.data` in-place ops, foreign kernels on raw pointers) are the caller's to announce
#     the same way; share_setup(False) turns the take-over off.
# The caches below (_SETUP, _COVER, _FACES) are shared by the threads nn.DataParallel runs its replicas on
# (main.py:183-193): every read-modify-write of them happens under _LOCK.
_SETUP = {}
_EPOCH = [0]
_SHARE = [True]
_LOCK = threading.RLock()


def invalidate_setups():
    """Forget every cached face setup: call after writing to verts / cams / faces behind torch's back, and after a
    hipGraph replay whose outputs are then rendered eagerly (see above)."""
    with _LOCK:
        _EPOCH[0] += 1
        _SETUP.clear()


def share_setup(on):
    """Enable / disable the silhouette -> texture workspace take-over (default on).  Returns the old value."""
    with _LOCK:
        old, _SHARE[0] = _SHARE[0], bool(on)
        if not on:
            _SETUP.clear()
    return old


def graph_replay(graph):
    """graph.replay() + invalidate_setups(): the replay of a torch.cuda.CUDAGraph whose tensors are also handed to
    the renderers outside the graph."""
    invalidate_setups()
    return graph.replay()


def _capture_id(device):
    """0 when the current stream of `device` is not capturing, else the id of the capture."""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    cid = ctypes.c_ulonglong(0)
    _lib.check(_lib.lib().acfm_stream_capture_id(_lib.cur_stream(device), ctypes.byref(cid)), "acfm_stream_capture_id")
    return int(cid.value) or -1


def _setup_key(v, c, f, H, offset_z):
    return (v.data_ptr(), v._version, tuple(v.shape), c.data_ptr(), c._version, f.data_ptr(), f._version,
            tuple(f.shape), int(H), float(offset_z), torch.cuda.current_stream(v.device).cuda_stream,
            _capture_id(v.device), _EPOCH[0])


PREFILL_TEX = [True]    # the silhouette render pre-fills the following texture render's empty blocks (see _SilRender.forward)


# ACFM_RECORD_COVER: the silhouette render can note the nearest COVERING face of every pixel on its way (a few
# instructions per covering pair), and the texture render that takes its workspace over then shades from that plane
# instead of binning and walking the faces again (70 -> ~30 us per 64 frames @256^2).  Worth it only when a texture
# render does follow, which the renderer cannot know: per device, the flag is on while the previous silhouette
# render's plane was used, and goes off when a plane was left unread.  Pure speed: outputs are bit-identical.
_COVER = {}


def _cover_tuning(device, tune):
    """Tuning for a silhouette render on `device`: `tune` with the cover flag decided."""
    forced = getattr(tune, "record_cover", None) if tune is not None else None
    if forced is not None:
        return _lib.with_cover(tune, bool(forced))
    with _LOCK:
        st = _COVER.setdefault(device, {"on": False, "pending": False})
        if st["pending"]:          # the last plane was never read: stop recording
            st["on"] = False
        st["pending"] = on = st["on"]
    return _lib.with_cover(tune, on)


def _cover_taken(device, tune):
    """A texture render took a silhouette render's workspace over: -> the ws_ready value for the C ABI."""
    with _LOCK:
        st = _COVER.setdefault(device, {"on": False, "pending": False})
        st["on"], st["pending"] = True, False
    return 2 if (tune is not None and tune.flags & 4) else 1


def _remember_setup(v, c, f, H, offset_z, ws, nb, blur, tune, prefill=None):
    if _SHARE[0]:
        ent = (_setup_key(v, c, f, H, offset_z), ws, nb, float(blur), (v, c, f), tune, [prefill])
        with _LOCK:
            _SETUP[v.device] = ent


def _take_prefill(holder, N, H):
    """The texture-output buffers (imgs, sil, pix_to_face, texel_idx) that the silhouette render of a shared setup
    pre-filled on the empty blocks -- handed out once -- or None."""
    with _LOCK:
        pf, holder[0] = holder[0], None
    return pf if (pf is not None and tuple(pf[0].shape) == (N, 3, H, H)) else None


def _shared_setup(v, c, f, H, offset_z):
    """-> (ws, nbytes, blur, tuning, prefill holder) of a silhouette render of exactly these inputs, or None."""
    with _LOCK:
        ent = _SETUP.get(v.device) if _SHARE[0] else None
    if ent is not None and ent[0] == _setup_key(v, c, f, H, offset_z):
        return ent[1], ent[2], ent[3], ent[5], ent[6]
    return None


# ------------------------------------------------------------------------------ projection
class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, cams, offset_z):
        _lib.require_gpu(verts, cams)
        v, c = _f32c(verts), _f32c(cams)
        N, V, _ = v.shape
        out = torch.empty_like(v)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_project(_lib.ptr(v), _lib.ptr(c), N, V, float(offset_z),
                                               _lib.ptr(out), _lib.cur_stream(v.device)), "acfm_project")
        ctx.save_for_backward(v, c)
        return out

    @staticmethod
    def backward(ctx, g):
        v, c = ctx.saved_tensors
        N, V, _ = v.shape
        g = _f32c(g)
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_project_backward(_lib.ptr(v), _lib.ptr(c), _lib.ptr(g), N, V,
                                                        _lib.ptr(gv), _lib.ptr(gc),
                                                        _lib.cur_stream(v.device)),
                       "acfm_project_backward")
        return gv, gc, None


class _ProjectXY(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, cams, offset_z):
        _lib.require_gpu(verts, cams)
        v, c = _f32c(verts), _f32c(cams)
        N, V, _ = v.shape
        out = torch.empty((N, V, 2), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_project_xy(_lib.ptr(v), _lib.ptr(c), N, V, float(offset_z),
                                                  _lib.ptr(out), _lib.cur_stream(v.device)), "acfm_project_xy")
        ctx.save_for_backward(v, c)
        return out

    @staticmethod
    def backward(ctx, g):
        v, c = ctx.saved_tensors
        N, V, _ = v.shape
        g = _f32c(g)
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_project_xy_backward(_lib.ptr(v), _lib.ptr(c), _lib.ptr(g), N, V,
                                                           _lib.ptr(gv), _lib.ptr(gc),
                                                           _lib.cur_stream(v.device)),
                       "acfm_project_xy_backward")
        return gv, gc, None


def project_xy(verts, cams, offset_z=0.0):
    """The (x, y) part of orthographic_proj_withz, [N,V,2]: bit-identical to project(...)[..., :2],
    one launch each way (no slice copy forward, no zero-padded [N,V,3] gradient backward)."""
    return _ProjectXY.apply(verts, cams, offset_z)


def project(verts, cams, offset_z=0.0):
    """[N,V,3] x [N,7] -> [N,V,3]; geom_utils.orthographic_proj_withz semantics."""
    return _Project.apply(verts, cams, offset_z)


# ------------------------------------------------------------------------------ deformation
class _DeformApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean_v, P, delta):
        _lib.require_gpu(mean_v, P, delta)
        m, p, d = _f32c(mean_v), _f32c(P), _f32c(delta)
        N, Kh, _ = d.shape
        V = m.shape[0]
        if p.shape != (V, Kh):
            raise ValueError("P must be [V,K_h] = [%d,%d], got %s" % (V, Kh, tuple(p.shape)))
        out = torch.empty((N, V, 3), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            _lib.check(_lib.lib().acfm_deform_apply(_lib.ptr(m), _lib.ptr(p), _lib.ptr(d), N, V, Kh,
                                                    _lib.ptr(out), _lib.cur_stream(m.device)),
                       "acfm_deform_apply")
        ctx.save_for_backward(p, d)
        return out

    @staticmethod
    def backward(ctx, g):
        p, d = ctx.saved_tensors
        N, Kh, _ = d.shape
        V = p.shape[0]
        g = _f32c(g)
        gm = torch.empty((V, 3), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        gd = torch.empty_like(d) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_deform_apply_backward(
                _lib.ptr(p), _lib.ptr(d), _lib.ptr(g), N, V, Kh, _lib.ptr(gd), _lib.ptr(gm), None,
                _lib.cur_stream(g.device)), "acfm_deform_apply_backward")
            if gp is not None:
                # dL/dP = sum_n g_n delta_n^T feeds the solve's backward, which amplifies its rounding by the conditioning
                # of the system: summed in double and rounded once (acfm_deform_presolve_sums_f64), so that the value does
                # not depend on how the frames are grouped; a frame-sharded step (sharding.SharedShapeExchange) takes the
                # unrounded doubles for its exchange buffer
                sink = _PRESOLVE_SINKS.get(p.data_ptr())
                if sink is not None:       # (G [V,K_h], sum g [V,3]): float64 views INSIDE the exchange buffer
                    g64, m64 = sink.presolve_buffer(V, Kh, g.device)
                else:
                    g64, m64 = torch.empty((V, Kh), dtype=torch.float64, device=g.device), None
                _lib.check(_lib.lib().acfm_deform_presolve_sums_f64(
                    _lib.ptr(d), _lib.ptr(g), N, V, Kh, _lib.ptr(g64), _lib.ptr(m64), _lib.ptr(gp), _lib.cur_stream(g.device)),
                    "acfm_deform_presolve_sums_f64")
        return gm, gp, gd


# P leaf (data_ptr) -> the object that wants the unrounded double sums of its backward (presolve_buffer(V, Kh, device)
# -> persistent float64 tensors ([V,K_h], [V,3]) it will read after the backward).  One entry per live exchange.
_PRESOLVE_SINKS = {}


def register_presolve_sink(P, sink, previous_key=None):
    """sharding.SharedShapeExchange.apply(): the backward of deform_apply(., P, .) writes sum_n g_n delta_n^T in double
    into sink.presolve_buffer(...).  -> the key to hand back as previous_key next time (or to drop_presolve_sink)."""
    with _LOCK:
        if previous_key is not None:
            _PRESOLVE_SINKS.pop(previous_key, None)
        key = P.data_ptr()
        _PRESOLVE_SINKS[key] = sink
    return key


def drop_presolve_sink(key):
    with _LOCK:
        _PRESOLVE_SINKS.pop(key, None)


def deform_apply(mean_v, P, delta):
    """verts[n] = mean_v + P @ delta[n]  (mean_v [V,3], P [V,K_h], delta [N,K_h,3])."""
    return _DeformApply.apply(mean_v, P, delta)


SOLVE_INFO_HANDOFF = 0x40000000   # ACFM_SOLVE_INFO_HANDOFF (include/acfm_hip.h)


def decode_solve_info(info):
    """Status word of acfm_deform_solve -> None (ok) or the message of the error it stands for."""
    info = int(info)
    if info == 0:
        return None
    if info & SOLVE_INFO_HANDOFF:
        return ("deform_solve: a hand-off wait of the single-launch factorisation expired (a wave of k_chol_tiles was "
                "starved for > 50 ms, e.g. by time-slicing or side-stream kernels on its CU): P holds NaNs -- run the "
                "solve again")
    return ("deform_solve: L^T L + A^T A is not positive definite (pivot tile starting at row %d)"
            % ((info & (SOLVE_INFO_HANDOFF - 1)) - 1))


# Status words of the solves launched with check=False (DeformSolver's per-step factorisations), copied to pinned host
# memory without blocking: (event, pinned int32).  The next deform_solve / solve_status() call reads those whose copy
# has completed and raises for a failed one -- a step late, but never silently and never with a synchronisation.
_SOLVE_PENDING = []


def solve_status(wait=False):
    """Raise RuntimeError for any earlier deform_solve(check=False) whose status word has arrived and reports a failure
    (a non-positive pivot, an expired hand-off).  wait=True: first wait for the outstanding ones (synchronises)."""
    with _LOCK:
        pending, _SOLVE_PENDING[:] = list(_SOLVE_PENDING), []
    msg, keep = None, []
    for ev, host in pending:
        if wait:
            ev.synchronize()
        if ev.query():
            msg = msg or decode_solve_info(host.item())
        else:
            keep.append((ev, host))
    with _LOCK:
        _SOLVE_PENDING[:0] = keep
    if msg:
        raise RuntimeError(msg + " [reported by an earlier deform_solve(check=False)]")


class _DeformSolve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L, lbs, check):
        _lib.require_gpu(L, lbs)
        l, b = _f32c(L.detach()), _f32c(lbs)
        V, Kh = b.shape
        if l.shape != (V, V):
            raise ValueError("L must be [V,V] = [%d,%d], got %s" % (V, V, tuple(l.shape)))
        if Kh > 32:
            raise ValueError("at most 32 handles (got %d)" % Kh)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            solve_status()
        lib = _lib.lib()
        nbytes = lib.acfm_deform_solve_workspace_bytes(V, Kh)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
        P = torch.empty((V, Kh), dtype=torch.float32, device=b.device)
        with torch.cuda.device(b.device):
            st = _lib.cur_stream(b.device)
            _lib.check(lib.acfm_deform_solve(_lib.ptr(l), _lib.ptr(b), V, Kh, _lib.ptr(P), _lib.ptr(ws), nbytes, st),
                       "acfm_deform_solve")
            if check:
                info = ctypes.c_int(0)
                _lib.check(lib.acfm_deform_solve_info(_lib.ptr(ws), nbytes, V, ctypes.byref(info), st),
                           "acfm_deform_solve_info")
                msg = decode_solve_info(info.value)
                if msg:
                    raise RuntimeError(msg)
            elif not capturing:
                off = lib.acfm_deform_solve_info_offset(V)
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(ws[off:off + 4].view(torch.int32), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(b.device))
                with _LOCK:
                    _SOLVE_PENDING.append((ev, host))
                    del _SOLVE_PENDING[:-8]
        ctx.ws, ctx.dims = ws, (V, Kh)
        return P

    @staticmethod
    def backward(ctx, g):
        V, Kh = ctx.dims
        if not ctx.needs_input_grad[1]:
            return None, None, None
        g = _f32c(g)
        gl = torch.empty((V, Kh), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_deform_solve_backward(_lib.ptr(g), V, Kh, _lib.ptr(ctx.ws), ctx.ws.numel(),
                                                             _lib.ptr(gl), _lib.cur_stream(g.device)),
                       "acfm_deform_solve_backward")
        return None, gl, None


def deform_solve(L, lbs_logits, check=False):
    """P [V,K_h] = (L^T L + A^T A)^-1 A^T with A = softmax(lbs_logits, dim 0)^T: the reference's
    per-frame Cholesky solve (multiframe/main.py:586-609) collapsed to one fp64 factorisation per
    step.  L [V,V] dense Laplacian (no gradient), lbs_logits [V,K_h] (gradient supported).
    check=True synchronises and raises if the matrix is not positive definite or a hand-off of the single-launch
    factorisation timed out (decode_solve_info); check=False (the per-step path) copies the status word to the host
    without blocking and the NEXT deform_solve / solve_status() call raises for it."""
    return _DeformSolve.apply(L, lbs_logits, bool(check))


# ------------------------------------------------------------------------------ cameras
class _CameraPipeline(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, mirror_flag, transforms, decay):
        _lib.require_gpu(emb, mirror_flag, transforms)
        e = _f32c(emb)
        R = e.numel() // 7
        mf = mirror_flag.detach().reshape(-1).to(torch.int64).contiguous()
        tr = _f32c(transforms).reshape(-1, 4)
        N = mf.numel()
        if tr.shape[0] != N or R % N != 0:
            raise ValueError("camera rows %d must be a multiple of the %d frames (transforms %s)"
                             % (R, N, tuple(tr.shape)))
        out = torch.empty((R, 7), dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().acfm_camera_pipeline(_lib.ptr(e), _lib.ptr(mf), _lib.ptr(tr), R, N, float(decay),
                                                       _lib.ptr(out), _lib.cur_stream(e.device)),
                       "acfm_camera_pipeline")
        ctx.save_for_backward(e, mf, tr)
        ctx.decay = float(decay)
        return out

    @staticmethod
    def backward(ctx, g):
        e, mf, tr = ctx.saved_tensors
        R, N = e.numel() // 7, mf.numel()
        g = _f32c(g)
        ge = torch.empty_like(e)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().acfm_camera_pipeline_backward(
                _lib.ptr(e), _lib.ptr(mf), _lib.ptr(tr), _lib.ptr(g), R, N, ctx.decay, _lib.ptr(ge),
                _lib.cur_stream(e.device)), "acfm_camera_pipeline_backward")
        return ge, None, None, None


def camera_pipeline(cam_emb, mirror_flag, transforms, scale_lr_decay=1.0):
    """Camera embeddings [G,N,7] (or [R,7], row r of frame r % N) -> cameras [R,7]: decode, mirror by
    the per-frame flag [N], crop/scale transform [N,4] (multiframe/main.py:551-584) in one kernel."""
    return _CameraPipeline.apply(cam_emb, mirror_flag, transforms, scale_lr_decay)


class _CameraPipelineTables(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frames_idx, selected, mirror_flag, transforms, decay, G, *tables):
        _lib.require_gpu(frames_idx, mirror_flag, transforms, *tables)
        ts = [_f32c(t) for t in tables]
        F = ts[0].shape[0]
        if not 1 <= len(ts) <= 32 or any(t.dim() != 2 or tuple(t.shape) != (F, 7) for t in ts):
            raise ValueError("camera_pipeline_tables: 1..32 embedding tables [frames, 7] of one size")
        fi = frames_idx.detach().reshape(-1).to(torch.int64).contiguous()
        mf = mirror_flag.detach().reshape(-1).to(torch.int64).contiguous()
        tr = _f32c(transforms).reshape(-1, 4)
        N = fi.numel()
        R = int(G) * N
        sel = None if selected is None else selected.detach().reshape(-1).to(torch.int64).contiguous()
        if mf.numel() != N or tr.shape[0] != N or (sel is not None and sel.numel() != R) or \
                (sel is None and int(G) > len(ts)):
            raise ValueError("camera_pipeline_tables: %d frames, mirror flags %d, transforms %s, G = %d, %d tables"
                             % (N, mf.numel(), tuple(tr.shape), G, len(ts)))
        out = torch.empty((R, 7), dtype=torch.float32, device=ts[0].device)
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().acfm_camera_pipeline_tables(
                ptrs, len(ts), F, _lib.ptr(fi), _lib.ptr(sel) if sel is not None else None, _lib.ptr(mf), _lib.ptr(tr),
                R, N, float(decay), _lib.ptr(out), _lib.cur_stream(out.device)), "acfm_camera_pipeline_tables")
        ctx.save_for_backward(fi, mf, tr, *ts) if sel is None else ctx.save_for_backward(fi, mf, tr, sel, *ts)
        ctx.cfg = (float(decay), R, N, F, len(ts), sel is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        decay, R, N, F, nt, has_sel = ctx.cfg
        saved = ctx.saved_tensors
        fi, mf, tr = saved[:3]
        sel = saved[3] if has_sel else None
        ts = saved[4:] if has_sel else saved[3:]
        g = _f32c(g)
        need = ctx.needs_input_grad[6:]
        outs = [torch.empty((F, 7), dtype=torch.float32, device=g.device) if need[i] else None for i in range(nt)]
        ptrs = (ctypes.c_void_p * nt)(*[t.data_ptr() for t in ts])
        gptrs = (ctypes.c_void_p * nt)(*[(o.data_ptr() if o is not None else None) for o in outs])
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_camera_pipeline_tables_backward(
                ptrs, nt, F, _lib.ptr(fi), _lib.ptr(sel) if sel is not None else None, _lib.ptr(mf), _lib.ptr(tr),
                _lib.ptr(g), R, N, decay, gptrs, _lib.cur_stream(g.device)), "acfm_camera_pipeline_tables_backward")
        return (None,) * 6 + tuple(outs)


def camera_pipeline_tables(tables, frames_idx, mirror_flag, transforms, scale_lr_decay=1.0, num_guesses=None,
                           selected=None, check=False):
    """Cameras [G*N,7] of all hypotheses straight from the per-hypothesis embedding tables ([frames,7] each,
    mesh_net.py:436-444): the look-ups, the stack / top-k gather (main.py:551-570) and the decode / mirror / transform
    chain (:572-584) in one kernel each way.  Row g*N + n reads tables[selected[g, n] if given else g][frames_idx[n]];
    the backward returns dense [frames,7] gradients like nn.Embedding's.
    A frames_idx outside [0, frames) or a `selected` outside [0, len(tables)) -- where nn.Embedding raises -- gives that
    row a NaN camera and no gradient (never an out-of-bounds access); check=True validates both on the host first
    (one synchronisation: for data-loader batches, not inside a captured step) and raises IndexError."""
    G = len(tables) if num_guesses is None else int(num_guesses)
    if check:
        nf = tables[0].shape[0]
        lo, hi = int(frames_idx.min()), int(frames_idx.max())
        if lo < 0 or hi >= nf:
            raise IndexError("camera_pipeline_tables: frames_idx in [%d, %d] outside the %d rows of the embedding tables"
                             % (lo, hi, nf))
        if selected is not None and (int(selected.min()) < 0 or int(selected.max()) >= len(tables)):
            raise IndexError("camera_pipeline_tables: selected hypothesis outside the %d tables" % len(tables))
    return _CameraPipelineTables.apply(frames_idx, selected, mirror_flag, transforms, scale_lr_decay, G, *tables)


def camera_mirror(cams):
    """Decoded cameras [R,7] -> the pose of the horizontally flipped image (multiframe/main.py:97-125 with the flag
    set), one kernel; no gradient (the texture render it feeds sends none to its cameras)."""
    _lib.require_gpu(cams)
    c = _f32c(cams.detach()).reshape(-1, 7)
    out = torch.empty_like(c)
    with torch.cuda.device(c.device):
        _lib.check(_lib.lib().acfm_camera_mirror(_lib.ptr(c), c.shape[0], _lib.ptr(out), _lib.cur_stream(c.device)),
                   "acfm_camera_mirror")
    return out


class _CameraNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw):
        _lib.require_gpu(raw)
        r = _f32c(raw)
        N = r.numel() // 7
        out = torch.empty_like(r)
        with torch.cuda.device(r.device):
            _lib.check(_lib.lib().acfm_camera_normalize(_lib.ptr(r), N, _lib.ptr(out), _lib.cur_stream(r.device)),
                       "acfm_camera_normalize")
        ctx.save_for_backward(r)
        return out

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        g = _f32c(g)
        gr = torch.empty_like(r)
        with torch.cuda.device(r.device):
            _lib.check(_lib.lib().acfm_camera_normalize_backward(_lib.ptr(r), _lib.ptr(g), r.numel() // 7, _lib.ptr(gr),
                                                                 _lib.cur_stream(r.device)),
                       "acfm_camera_normalize_backward")
        return gr


def camera_normalize(cam_raw):
    """[N,7] (s, tx, ty, q) -> (s, tx, ty, q/|q|): torch.cat([scale, trans, F.normalize(quat)]) in one kernel."""
    return _CameraNormalize.apply(cam_raw)


# ------------------------------------------------------------------------------ optical flow
class _OFLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj, flows, vis, B, T, masks, flip_t):
        _lib.require_gpu(proj, flows, vis)
        p, fl = _f32c(proj), _f32c(flows)
        V = p.shape[-2]
        H, W = fl.shape[-3], fl.shape[-2]
        vi = vis.detach().reshape(-1, V).to(torch.uint8).contiguous()
        clips = fl.numel() // (T * H * W * 2)
        mk = None if masks is None else _f32c(masks)
        if p.numel() != B * T * V * 3 or clips < 1 or fl.numel() != clips * T * H * W * 2 or B % clips != 0 or \
                vi.shape[0] != B * T or (mk is not None and mk.numel() != clips * T * H * W):
            raise ValueError("of_loss: proj %s flows %s vis %s masks %s do not match B=%d T=%d"
                             % (tuple(p.shape), tuple(fl.shape), tuple(vi.shape),
                                None if mk is None else tuple(mk.shape), B, T))
        loss = torch.empty((B, T - 1), dtype=torch.float32, device=p.device)
        cnt = torch.empty((B, T - 1), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().acfm_of_loss_shared(_lib.ptr(p), _lib.ptr(fl), _lib.ptr(mk) if mk is not None else None,
                                                      _lib.ptr(vi), B, T, V, H, W, clips, int(bool(flip_t)),
                                                      _lib.ptr(loss), _lib.ptr(cnt), _lib.cur_stream(p.device)),
                       "acfm_of_loss_shared")
        ctx.save_for_backward(p, fl, vi, cnt) if mk is None else ctx.save_for_backward(p, fl, vi, cnt, mk)
        ctx.dims = (B, T, V, H, W, clips, int(bool(flip_t)), mk is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        B, T, V, H, W, clips, flip_t, has_m = ctx.dims
        p, fl, vi, cnt = ctx.saved_tensors[:4]
        mk = ctx.saved_tensors[4] if has_m else None
        g = _f32c(g)
        gp = torch.empty_like(p)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().acfm_of_loss_shared_backward(
                _lib.ptr(p), _lib.ptr(fl), _lib.ptr(mk) if mk is not None else None, _lib.ptr(vi), _lib.ptr(cnt),
                _lib.ptr(g), B, T, V, H, W, clips, flip_t, _lib.ptr(gp), _lib.cur_stream(p.device)),
                "acfm_of_loss_shared_backward")
        return gp, None, None, None, None, None, None


def of_loss(proj, flows, vis, B, T, masks=None, flip_t=False):
    """Optical-flow loss per clip and frame pair [B,T-1] from projected vertices [B*T,V,3], GT flow
    images and the visible-vertex bitmap [B*T,V] (loss_utils.py:445-474, one kernel).  flows: [B*T,H,W,2], or
    [clips*T,H,W,2] with B a multiple of clips (rendered clip c reads clip c % clips: the hypotheses of a clip share its
    data); flip_t: frame k reads frame T-1-k; masks [clips*T,H,W]: the flow is multiplied by the mask of frame k at the
    sampled pixel -- i.e. main.py:676-686's flip / mask / repeat(G) of the flow images without materialising them."""
    return _OFLoss.apply(proj, flows, vis, int(B), int(T), masks, bool(flip_t))


# ------------------------------------------------------------------------------ correlation
def correlation(f1, f2, max_displacement):
    """Cost volume of MaskFlownet's Correlation layer (pad = max_displacement = md, kernel 1, strides 1):
    f1, f2 [N,C,H,W] -> [N,(2md+1)^2,H,W].  Forward only (the flow network is frozen in ACFM)."""
    _lib.require_gpu(f1, f2)
    if f1.requires_grad or f2.requires_grad:
        if torch.is_grad_enabled():
            raise RuntimeError("correlation: forward only (frozen flow network); wrap the call in torch.no_grad()")
    a, b = _f32c(f1), _f32c(f2)
    if a.dim() != 4 or a.shape != b.shape:
        raise ValueError("correlation: two [N,C,H,W] tensors of equal shape, got %s and %s" % (tuple(a.shape), tuple(b.shape)))
    N, C, H, W = a.shape
    md = int(max_displacement)
    out = torch.empty((N, (2 * md + 1) ** 2, H, W), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().acfm_correlation_forward(_lib.ptr(a), _lib.ptr(b), N, C, H, W, md, _lib.ptr(out),
                                                       _lib.cur_stream(a.device)), "acfm_correlation_forward")
    return out


# ------------------------------------------------------------------------------ lazy pix_to_face
class LazyPixToFace(torch.Tensor):
    """`pix_to_face [N,H,W,K]` int64 as the reference's renderer returns it, with the K-1 planes nobody in the training
    path reads produced on first use.  The render writes the nearest-face plane (all that loss_utils.py:214 `[..., 0]`
    and :431 `[..., :1]` read: 8 bytes per pixel instead of 8 K); `p[..., 0]` and `p[..., :1]` are views of it; splitting
    along the batch dimension (`p[a:b]`, `p[n]`, `chunk` / `split` on dim 0 -- what nn.DataParallel's scatter does to
    every tensor argument, main.py:326, 718, also on one device) gives lazy tensors over the parts; ANY other
    operation (indexing another slot, .cpu(), comparisons, printing ...) first renders the full tensor -- the same
    kernel once more with all K slots stored, from the very tensors of the original call -- and then behaves like the
    plain tensor.  If those inputs changed meanwhile (in-place write, invalidate_setups()) it raises instead of
    returning ids of other geometry: NeuralRenderer(pix_to_face_slots=K) stores all K slots at render time."""

    @staticmethod
    def __new__(cls, plane0, K, make_full, vis=None):
        shape = tuple(plane0.shape[:-1]) + (int(K),)
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=plane0.dtype, device=plane0.device)
        r._plane0, r._make_full, r._full = plane0, make_full, None
        r._acfm_vis = vis
        return r

    @property
    def is_materialized(self):
        return self._full is not None

    def materialize(self):
        if self._full is None:
            self._full = self._make_full()
            self._make_full = None
        return self._full

    def __repr__(self):
        return "LazyPixToFace(shape=%s, materialized=%s)" % (tuple(self.shape), self.is_materialized)

    def __deepcopy__(self, memo):
        return self.materialize().clone()

    def _rows(self, start, end, squeeze=False):
        """Lazy tensor over meshes start..end-1 (squeeze: the single mesh `start`, batch dimension dropped)."""
        parent = self
        pl = self._plane0[start] if squeeze else self._plane0[start:end]
        vis = self._acfm_vis
        if vis is not None:
            vis = None if squeeze else vis[start:end]
        return LazyPixToFace(pl, self.shape[-1], (lambda: parent.materialize()[start] if squeeze
                                                  else parent.materialize()[start:end]), vis)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        a0 = args[0] if args else None
        if isinstance(a0, LazyPixToFace) and a0._full is None:
            nd = a0.dim()
            aten = torch.ops.aten
            dim = args[1] if len(args) > 1 and isinstance(args[1], int) else None
            last = dim in (nd - 1, -1) and nd > 1
            first = dim in (0, -nd) and nd == 4
            if func is aten.select.int and last and args[2] == 0:
                return a0._plane0.select(nd - 1, 0)
            if func is aten.slice.Tensor and last and len(args) >= 4 and args[2] in (0, None) and args[3] == 1 and \
                    (len(args) < 5 or args[4] == 1):
                return a0._plane0
            if func in (aten.detach.default, aten.alias.default):
                return a0
            N = a0.shape[0]
            if func is aten.select.int and first:
                i = args[2] + N if args[2] < 0 else args[2]
                if 0 <= i < N:
                    return a0._rows(i, i + 1, squeeze=True)
            if func is aten.slice.Tensor and (first or (len(args) == 1 and nd == 4)) and (len(args) < 5 or args[4] == 1):
                lo = args[2] if len(args) > 2 and args[2] is not None else 0
                hi = args[3] if len(args) > 3 and args[3] is not None else N
                lo = max(0, lo + N if lo < 0 else lo)
                hi = min(N, hi + N if hi < 0 else hi)
                if lo < hi:
                    return a0._rows(lo, hi)
            if func is aten.split.Tensor and (len(args) < 3 or args[2] in (0, -nd)) and nd == 4 and args[1] > 0:
                return [a0._rows(lo, min(N, lo + args[1])) for lo in range(0, N, args[1])]
            if func is aten.split_with_sizes.default and (len(args) < 3 or args[2] in (0, -nd)) and nd == 4 and \
                    sum(args[1]) == N and all(x > 0 for x in args[1]):
                out, lo = [], 0
                for x in args[1]:
                    out.append(a0._rows(lo, lo + x))
                    lo += x
                return out
        from torch.utils._pytree import tree_map
        unwrap = lambda x: x.materialize() if isinstance(x, LazyPixToFace) else x
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs))


def _lazy_pix_to_face(plane0, vis, v, f, c, H, K, blur, sigma, offset_z):
    """Wraps the nearest-face plane of a silhouette render (see LazyPixToFace)."""
    versions = (v._version, f._version, c._version, _EPOCH[0])
    tune = _lib.tuning()[1]
    born_in = _capture_id(v.device)

    def make_full():
        if (v._version, f._version, c._version, _EPOCH[0]) != versions:
            raise RuntimeError("pix_to_face: slots beyond [..., 0] are rendered on first use, but the vertices / faces / "
                               "cameras of that render have been modified since (in-place write, or a hipGraph replay "
                               "announced with invalidate_setups()); "
                               "use NeuralRenderer(pix_to_face_slots=faces_per_pixel) to store every slot at render time")
        if _capture_id(v.device) != born_in:
            # made while a hipGraph was being captured: its inputs are the capture's buffers, which hold nothing until a
            # replay and whatever the last replay left afterwards -- there is no "the render's inputs" to re-render from
            raise RuntimeError("pix_to_face: this tensor was rendered inside a hipGraph capture; its slots beyond [..., 0] "
                               "can only be formed inside that same capture.  Use NeuralRenderer(pix_to_face_slots="
                               "faces_per_pixel) in captured steps that read them")
        N, V, _ = v.shape
        F = f.shape[1]
        mask = torch.empty((N, H, H), dtype=torch.float32, device=v.device)
        full = torch.empty((N, H, H, K), dtype=torch.int64, device=v.device)
        kth = torch.empty((N, H, H), dtype=torch.int64, device=v.device)
        vis2 = torch.empty((N, V), dtype=torch.uint8, device=v.device)
        ws, nb = _workspace(N, V, F, H, v.device)          # its own workspace: the render's is still the backward's
        with torch.no_grad(), torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_sil_forward(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), N, V, F, H, K, K, float(blur), float(sigma), float(offset_z),
                _lib.ptr(mask), _lib.ptr(full), _lib.ptr(kth), _lib.ptr(vis2), _lib.ptr(ws), nb,
                _lib.tuning_ptr(tune), _lib.cur_stream(v.device)), "acfm_sil_forward")
        return full

    return LazyPixToFace(plane0, K, make_full, vis)


# The reference computes its losses on the rendered images with separate operators (main.py:644-662, 716-717), so the
# gradient of a loss with respect to a whole image -- [N,H,W] for the silhouette terms, [N,3,H,W] for the texture
# MSE -- is written by one kernel only to be read once by the render's backward.  Both backward kernels can form that
# gradient per pixel themselves (acfm_sil_loss_backward, acfm_tex_mse_backward_faces: the opt-in fused operators).
# LazyGrad lets the drop-in operators use them too: the loss operator's backward hands autograd a tensor of the right
# shape that merely REMEMBERS how the gradient is defined (the operator's inputs and its upstream gradient); the render's
# backward recognises it -- same image, untouched -- and calls the fused kernel.  Two things autograd does to such a
# gradient on the reference's own call sequence stay lazy: shape-only views (l1_loss / iou hand the operator
# `mask.view(N, -1)` / `mask[:, None]`, loss_utils.py:18-32, 72-77: the view's backward reshapes the gradient) and the
# SUM of the gradients of two silhouette-loss operators on the same mask (l1_loss and edt_loss, main.py:644 and :716:
# the sum is the gradient of one operator with both references and the two upstream gradients added).  ANY other use
# (a hook, another consumer's gradient added, torch.autograd.grad with respect to the image itself) forms the gradient
# with the loss operator's own backward kernel first and then behaves like the plain tensor.
LAZY_GRADS = [True]


def _same_tensor(a, b):
    return a is b or (a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape
                      and a._version == b._version and a.dtype == b.dtype)


class LazyGrad(torch.Tensor):
    @staticmethod
    def __new__(cls, like, tag, payload, make_full, shape=None):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(like.shape) if shape is None else tuple(shape),
                                                dtype=like.dtype, device=like.device)
        r._tag, r._payload, r._make_full, r._full = tag, payload, make_full, None
        return r

    @property
    def is_materialized(self):
        return self._full is not None

    def materialize(self):
        if self._full is None:
            self._full = self._make_full().reshape(self.shape)
            self._make_full = self._payload = None
        return self._full

    def take(self, tag, image):
        """The operator inputs behind this gradient if it is still unformed, of kind `tag`, and a gradient with
        respect to exactly `image` (same storage, element count and version); else None."""
        if self._full is not None or self._tag != tag:
            return None
        img = self._payload[0]
        if img.data_ptr() != image.data_ptr() or img.numel() != image.numel() or self.numel() != image.numel() or \
                img._version != image._version:
            return None
        return self._payload

    def _reshaped(self, shape):
        parent = self
        return LazyGrad(self, self._tag, self._payload, lambda: parent.materialize(), shape=shape)

    def _merged(self, other):
        """self + other for two unformed silhouette-loss gradients of the same mask, itself unformed; else None."""
        if self._full is not None or other._full is not None or self._tag != "mask_losses" or other._tag != "mask_losses" \
                or self.shape != other.shape:
            return None
        m1, g1, e1, rb1, go1 = self._payload
        m2, g2, e2, rb2, go2 = other._payload
        if not _same_tensor(m1, m2):
            return None
        if (g1 is not None and g2 is not None and not _same_tensor(g1, g2)) or \
                (e1 is not None and e2 is not None and not _same_tensor(e1, e2)):
            return None
        refs = [r for r in (g1, g2, e1, e2) if r is not None]
        if any(r.shape[0] != refs[0].shape[0] for r in refs):
            return None
        # a term whose reference an operator did not have contributes nothing to that operator's gradient
        # (acfm_mask_losses_backward skips it): mask its columns before the two upstream gradients are added
        def cols(go, g, e):
            if g is not None and e is not None:
                return go
            return go * _lib.const((1.0, 1.0, 1.0, 0.0) if e is None else (0.0, 0.0, 0.0, 1.0), go.device)
        go = cols(go1, g1, e1) + cols(go2, g2, e2)
        g, e = (g1 if g1 is not None else g2), (e1 if e1 is not None else e2)
        rb = refs[0].shape[0] if refs else m1.shape[0]
        a, b = self, other

        def make():
            return a.materialize() + b.materialize()
        return LazyGrad(self, "mask_losses", (m1, g, e, rb, go), make)

    def __repr__(self):
        return "LazyGrad(%s, shape=%s, materialized=%s)" % (self._tag, tuple(self.shape), self.is_materialized)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        aten = torch.ops.aten
        a0 = args[0] if args else None
        if type(a0) is LazyGrad and a0._full is None:
            if func in (aten.detach.default, aten.alias.default):
                return a0
            if func in (aten.view.default, aten.reshape.default, aten._unsafe_view.default):
                shape = list(args[1])
                if -1 in shape:
                    known = 1
                    for x in shape:
                        known *= x if x != -1 else 1
                    shape[shape.index(-1)] = a0.numel() // max(known, 1)
                n = 1
                for x in shape:
                    n *= x
                if n == a0.numel():
                    return a0._reshaped(shape)
            if func is aten.squeeze.dim and a0.shape[args[1]] == 1:
                shape = list(a0.shape)
                del shape[args[1]]
                return a0._reshaped(shape)
            if func is aten.unsqueeze.default:
                shape = list(a0.shape)
                shape.insert(args[1] if args[1] >= 0 else args[1] + a0.dim() + 1, 1)
                return a0._reshaped(shape)
            if func is aten.add.Tensor and len(args) == 2 and type(args[1]) is LazyGrad and \
                    kwargs.get("alpha", 1) == 1:
                m = a0._merged(args[1])
                if m is not None:
                    return m
        from torch.utils._pytree import tree_map
        unwrap = lambda x: x.materialize() if type(x) is LazyGrad else x
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs))


# ------------------------------------------------------------------------------ silhouette
class _SilRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, cams, img_size, K, blur, sigma, offset_z, k_out, f16=False):
        _lib.require_gpu(verts, faces, cams)
        v, c = _f32c(verts), _f32c(cams)
        N, V, _ = v.shape
        f = expand_faces(faces, N)
        F, H = f.shape[1], int(img_size)
        if f16 and k_out != 1:
            raise ValueError("storage='f16' writes the int32 nearest-face plane only (k_out = 1)")
        mask = torch.empty((N, H, H), dtype=torch.float16 if f16 else torch.float32, device=v.device)
        p2f = torch.empty((N, H, H, k_out), dtype=torch.int32 if f16 else torch.int64, device=v.device)
        kth = torch.empty((N, H, H), dtype=torch.int64, device=v.device)  # u64 keys, opaque
        vis = torch.empty((N, V), dtype=torch.uint8, device=v.device)
        ws, nb = _workspace(N, V, F, H, v.device)
        tune = _cover_tuning(v.device, _lib.tuning()[1])
        if f16:
            tune = _lib.with_f16(tune, True)
        tp = _lib.tuning_ptr(tune)
        # a texture render of this prediction is expected (the cover flag is on while they do follow): its constant
        # outputs on the empty blocks are stored by THIS kernel, behind the walk, into buffers that render then adopts
        # (acfm_sil_forward_prefill / acfm_tex_forward ws_ready = 3: -20 us of the texture kernel's 36)
        prefill = None
        if PREFILL_TEX[0] and tune is not None and (tune.flags & 4) and not f16 and _SHARE[0]:
            prefill = (torch.empty((N, 3, H, H), dtype=torch.float32, device=v.device),
                       torch.empty((N, H, H), dtype=torch.float32, device=v.device),
                       torch.empty((N, H, H, 1), dtype=torch.int64, device=v.device),
                       torch.empty((N, H, H), dtype=torch.int32, device=v.device))
        # NeuralRenderer.project_points of these very vertices and cameras (main.py:715, predictor.py:319: the boundary
        # loss's input) comes out of the face setup (AcfmSilExtras.proj_xy) and its gradient goes back through THIS
        # operator's one projection backward: no second projection kernel either way, no sum of two vertex / camera
        # gradients afterwards (k_project, k_project_bwd and two torch adds less per step)
        proj = torch.empty((N, V, 2), dtype=torch.float32, device=v.device)
        exp, _ex_keep = _lib.sil_extras(proj_xy=proj, prefill=prefill)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_sil_forward_ex(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), N, V, F, H, K, int(k_out), float(blur), float(sigma),
                float(offset_z), _lib.ptr(mask), _lib.ptr(p2f), _lib.ptr(kth), _lib.ptr(vis),
                _lib.ptr(ws), nb, tp, exp, _lib.cur_stream(v.device)), "acfm_sil_forward_ex")
        _remember_setup(v, c, f, H, offset_z, ws, nb, blur, tune, prefill)
        ctx.save_for_backward(v, f, c, mask, kth)
        ctx.cfg = (H, float(blur), float(sigma), float(offset_z))
        ctx.ws = (ws, nb, tune)  # face records + tile schedule: reused by backward (no second setup)
        ctx.mark_non_differentiable(p2f, vis)
        ctx.set_materialize_grads(False)  # no zero-filled [N,H,H,K] int64 "gradient" for pix_to_face
        return mask, p2f, vis, proj

    @staticmethod
    def backward(ctx, gmask, _gp2f, _gvis, gproj):
        v, f, c, mask, kth = ctx.saved_tensors
        H, blur, sigma, offset_z = ctx.cfg
        N, V, _ = v.shape
        F = f.shape[1]
        if gmask is None and gproj is None:
            return (None,) * 10
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[2] else None
        gp = _f32c(gproj) if gproj is not None else None
        if gmask is None:        # only the projection was used: its own backward
            with torch.cuda.device(v.device):
                _lib.check(_lib.lib().acfm_project_xy_backward(_lib.ptr(v), _lib.ptr(c), _lib.ptr(gp), N, V, _lib.ptr(gv),
                                                               _lib.ptr(gc), _lib.cur_stream(v.device)),
                           "acfm_project_xy_backward")
            return (gv, None, gc) + (None,) * 7
        ws, nb, tune = ctx.ws
        exp, _ex_keep = _lib.sil_extras(grad_proj_xy=gp)
        pay = gmask.take("mask_losses", mask) if type(gmask) is LazyGrad else None
        if pay is not None:      # the silhouette losses' gradient, still unformed: the backward kernel forms it per pixel
            _m, lg, le, rb, go = pay
            with torch.cuda.device(v.device):
                _lib.check(_lib.lib().acfm_sil_loss_backward_ex(
                    _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(mask), _lib.ptr(kth), _lib.ptr(lg), _lib.ptr(le), rb,
                    _lib.ptr(go), N, V, F, H, blur, sigma, offset_z, _lib.ptr(gv), _lib.ptr(gc), _lib.ptr(ws), nb, 1,
                    _lib.tuning_ptr(tune), exp, _lib.cur_stream(v.device)), "acfm_sil_loss_backward_ex")
            return (gv, None, gc) + (None,) * 7
        g = _f32c(gmask)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_sil_backward_ex(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(mask), _lib.ptr(kth), _lib.ptr(g), N,
                V, F, H, blur, sigma, offset_z, _lib.ptr(gv), _lib.ptr(gc), _lib.ptr(ws), nb, 1,
                _lib.tuning_ptr(tune), exp, _lib.cur_stream(v.device)), "acfm_sil_backward_ex")
        return (gv, None, gc) + (None,) * 7


def _proj_key(verts, cams):
    return (verts.data_ptr(), verts._version, tuple(verts.shape), str(verts.dtype), cams.data_ptr(), cams._version,
            tuple(cams.shape), str(cams.dtype))


def sil_render(verts, faces, cams, img_size, K=SIL_K, blur=SIL_BLUR, sigma=SIL_SIGMA, offset_z=0.0,
               k_out=None, storage="f32"):
    """Soft silhouette: -> (mask [N,H,H] f32, pix_to_face [N,H,H,k_out] i64), k_out = K (default,
    what PyTorch3D returns) or 1 (nearest-face plane only; K faces are still blended).
    The visible-vertex bitmap the raster kernel produces on the side (vertices of every
    nearest face, = what bds_loss / optical_flow_loss derive from pix_to_face[..., 0]) rides
    along on the pix_to_face tensor object as `._acfm_vis`."""
    f16 = _is_f16(storage)   # "f16": mask [N,H,H] float16, pix_to_face [N,H,H,1] int32 (BASELINE config 5); fp32 arithmetic
    lazy = k_out == "lazy" and not f16 and K > 1
    mask, p2f, vis, proj = _SilRender.apply(verts, faces, cams, img_size, K, blur, sigma, offset_z,
                                            1 if (f16 or lazy) else (K if k_out in (None, "lazy") else int(k_out)), f16)
    if lazy:   # k_out="lazy": [N,H,H,K] whose slots 1.. are rendered on first use (LazyPixToFace)
        p2f = _lazy_pix_to_face(p2f, vis, _f32c(verts), expand_faces(faces, verts.shape[0]), _f32c(cams),
                                int(img_size), int(K), blur, sigma, offset_z)
    else:
        p2f._acfm_vis = vis
    # the (x, y) projection of these vertices under these cameras, an output of the render's autograd node: rides on
    # the pix_to_face object like the visibility bitmap (NeuralRenderer.project_points picks it up)
    p2f._acfm_proj = (_proj_key(verts, cams), proj)
    return mask, p2f


class _SilRenderLosses(torch.autograd.Function):
    """acfm_sil_loss_forward / _backward: soft silhouette render + the [N,4] silhouette-loss vector as one operator."""

    @staticmethod
    def forward(ctx, verts, faces, cams, gt, edt, img_size, K, blur, sigma, offset_z, k_out, f16=False):
        _lib.require_gpu(verts, faces, cams, gt, edt)
        v, c = _f32c(verts), _f32c(cams)
        N, V, _ = v.shape
        f = expand_faces(faces, N)
        F, H = f.shape[1], int(img_size)
        if f16 and k_out != 1:
            raise ValueError("storage='f16' writes the int32 nearest-face plane only (k_out = 1)")
        g = _real(gt, f16).reshape(-1, H, H) if gt is not None else None
        e = _real(edt, f16).reshape(-1, H, H) if edt is not None else None
        RB = N
        for r in (g, e):
            if r is not None:
                RB = _ref_batch(N, r, "sil_render_losses")
        if g is not None and e is not None and g.shape[0] != e.shape[0]:
            raise ValueError("sil_render_losses: gt and edt must have the same batch")
        mask = torch.empty((N, H, H), dtype=torch.float16 if f16 else torch.float32, device=v.device)
        p2f = torch.empty((N, H, H, k_out), dtype=torch.int32 if f16 else torch.int64, device=v.device)
        kth = torch.empty((N, H, H), dtype=torch.int64, device=v.device)
        vis = torch.empty((N, V), dtype=torch.uint8, device=v.device)
        losses = torch.empty((N, 4), dtype=torch.float32, device=v.device)
        ws, nb = _workspace(N, V, F, H, v.device)
        tune = _cover_tuning(v.device, _lib.tuning()[1])
        if f16:
            tune = _lib.with_f16(tune, True)
        tp = _lib.tuning_ptr(tune)
        proj = torch.empty((N, V, 2), dtype=torch.float32, device=v.device)      # (see _SilRender.forward)
        exp, _ex_keep = _lib.sil_extras(proj_xy=proj)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_sil_loss_forward_ex(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(g), _lib.ptr(e), RB, N, V, F, H, K, int(k_out),
                float(blur), float(sigma), float(offset_z), _lib.ptr(mask), _lib.ptr(p2f), _lib.ptr(kth),
                _lib.ptr(vis), _lib.ptr(losses), _lib.ptr(ws), nb, tp, exp, _lib.cur_stream(v.device)),
                "acfm_sil_loss_forward_ex")
        _remember_setup(v, c, f, H, offset_z, ws, nb, blur, tune)
        ctx.save_for_backward(v, f, c, mask, kth, g, e)
        ctx.cfg = (H, float(blur), float(sigma), float(offset_z), RB)
        ctx.ws = (ws, nb, tune)
        ctx.mark_non_differentiable(mask, p2f, vis)
        ctx.set_materialize_grads(False)
        return losses, mask, p2f, vis, proj

    @staticmethod
    def backward(ctx, glosses, _gm, _gp, _gv, gproj):
        v, f, c, mask, kth, g, e = ctx.saved_tensors
        H, blur, sigma, offset_z, RB = ctx.cfg
        N, V, _ = v.shape
        F = f.shape[1]
        none = (None,) * 12
        if glosses is None and gproj is None:
            return none
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c) if ctx.needs_input_grad[2] else None
        gp = _f32c(gproj) if gproj is not None else None
        if glosses is None:      # only the projection was used: its own backward
            with torch.cuda.device(v.device):
                _lib.check(_lib.lib().acfm_project_xy_backward(_lib.ptr(v), _lib.ptr(c), _lib.ptr(gp), N, V, _lib.ptr(gv),
                                                               _lib.ptr(gc), _lib.cur_stream(v.device)),
                           "acfm_project_xy_backward")
            return (gv, None, gc) + none[3:]
        go = _f32c(glosses)
        ws, nb, tune = ctx.ws
        exp, _ex_keep = _lib.sil_extras(grad_proj_xy=gp)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_sil_loss_backward_ex(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(mask), _lib.ptr(kth), _lib.ptr(g), _lib.ptr(e), RB,
                _lib.ptr(go), N, V, F, H, blur, sigma, offset_z, _lib.ptr(gv), _lib.ptr(gc), _lib.ptr(ws), nb, 1,
                _lib.tuning_ptr(tune), exp, _lib.cur_stream(v.device)), "acfm_sil_loss_backward_ex")
        return (gv, None, gc) + none[3:]


def sil_render_losses(verts, faces, cams, img_size, gt=None, edt=None, K=SIL_K, blur=SIL_BLUR, sigma=SIL_SIGMA,
                      offset_z=0.0, k_out=None, storage="f32"):
    """Soft-silhouette render and its silhouette losses as ONE operator (opt-in; the drop-in pair is
    sil_render + mask_losses): -> (losses [N,4] = (mean|m-gt|, sum m*gt, sum(m+gt-m*gt), mean edt*m), mask [N,H,H],
    pix_to_face [N,H,H,k_out]).  Gradients flow from `losses` to verts / cams; `mask` is returned for inspection and
    carries none (use sil_render when the mask itself feeds further differentiable code).  gt / edt: [N,...] or
    [N/G,...] shared by the G hypotheses of a frame."""
    f16 = _is_f16(storage)   # "f16": mask and the references are held in float16, the loss sums stay float32
    lazy = k_out == "lazy" and not f16 and K > 1
    losses, mask, p2f, vis, proj = _SilRenderLosses.apply(verts, faces, cams, gt, edt, img_size, K, blur, sigma, offset_z,
                                                          1 if (f16 or lazy) else (K if k_out in (None, "lazy") else int(k_out)),
                                                          f16)
    if lazy:
        p2f = _lazy_pix_to_face(p2f, vis, _f32c(verts), expand_faces(faces, verts.shape[0]), _f32c(cams),
                                int(img_size), int(K), blur, sigma, offset_z)
    else:
        p2f._acfm_vis = vis
    p2f._acfm_proj = (_proj_key(verts, cams), proj)
    return losses, mask, p2f


# ------------------------------------------------------------------------------ hard raster
def hard_raster(verts_proj, faces, img_size):
    """OF_NeuralRenderer.forward: pre-projected verts -> pix_to_face [N,H,H,1] i64."""
    _lib.require_gpu(verts_proj, faces)
    v = _f32c(verts_proj)
    N, V, _ = v.shape
    f = expand_faces(faces, N)
    F, H = f.shape[1], int(img_size)
    p2f = torch.empty((N, H, H, 1), dtype=torch.int64, device=v.device)
    vis = torch.empty((N, V), dtype=torch.uint8, device=v.device)
    ws, nb = _workspace(N, V, F, H, v.device)
    with torch.cuda.device(v.device):
        _lib.check(_lib.lib().acfm_hard_raster(_lib.ptr(v), _lib.ptr(f), N, V, F, H, _lib.ptr(p2f),
                                               _lib.ptr(vis), _lib.ptr(ws), nb, _lib.tuning()[0],
                                               _lib.cur_stream(v.device)), "acfm_hard_raster")
    p2f._acfm_vis = vis
    return p2f


# ------------------------------------------------------------------------------ texture
# atlas gradient: gather per face over the pixels of its box (acfm_tex_backward_faces, R <= 8) instead
# of one global float atomic per pixel and channel (acfm_tex_backward); both are kept and tested
TEX_BWD_GATHER = True


class _TexRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, cams, atlas, img_size, sigma, gamma, offset_z, f16=False):
        _lib.require_gpu(verts, faces, cams, atlas)
        v, c, a = _f32c(verts), _f32c(cams), _real(atlas, f16)   # (a float32 atlas is cast per call: hold it in half to spare that)
        N, V, _ = v.shape
        f = expand_faces(faces, N)
        F, H = f.shape[1], int(img_size)
        NA = a.shape[0] if a.dim() == 5 else 0
        if a.dim() != 5 or NA == 0 or N % NA != 0 or a.shape[1] != F or a.shape[2] != a.shape[3] or a.shape[4] != 3:
            raise ValueError("atlas must be [N,F,R,R,3] (or [N/G,F,R,R,3], shared by G hypotheses), got %s for "
                             "N=%d F=%d" % (tuple(a.shape), N, F))
        R = a.shape[2]
        rdt = torch.float16 if f16 else torch.float32
        shared = _shared_setup(v, c, f, H, offset_z)
        holder = None
        if shared is not None:      # the workspace (and the tuning it was carved with) of the silhouette render
            ws, nb, ws_blur, tune, holder = shared
        else:
            ws, nb = _workspace(N, V, F, H, v.device)
            ws_blur, tune = 0.0, _lib.tuning()[1]
        ws_ready = _cover_taken(v.device, tune) if shared is not None else 0
        pf = _take_prefill(holder, N, H) if (ws_ready == 2 and not f16) else None
        if pf is not None:
            imgs, sil, p2f, tidx = pf          # their empty blocks were stored by the silhouette render
            ws_ready = 3
        else:
            imgs = torch.empty((N, 3, H, H), dtype=rdt, device=v.device)
            sil = torch.empty((N, H, H), dtype=rdt, device=v.device)
            p2f = torch.empty((N, H, H, 1), dtype=torch.int32 if f16 else torch.int64, device=v.device)
            tidx = torch.empty((N, H, H), dtype=torch.int32, device=v.device)
        tune = _lib.with_f16(tune, f16)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_tex_forward(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(a), N, V, F, H, R, float(sigma),
                float(gamma), float(offset_z), _lib.ptr(imgs), _lib.ptr(sil), _lib.ptr(p2f),
                _lib.ptr(tidx), _lib.ptr(ws), nb, ws_ready, float(ws_blur), NA,
                _lib.tuning_ptr(tune), _lib.cur_stream(v.device)), "acfm_tex_forward")
        ctx.save_for_backward(tidx, imgs)   # (imgs: to recognise a LazyGrad of this very image, at this version, in the backward)
        ctx.cfg = (N, F, H, R, NA, V)
        ctx.adt = atlas.dtype
        ctx.tune, ctx.f16 = tune, f16
        ctx.ws = (ws, nb, float(ws_blur))   # face boxes: the gather form of the atlas gradient walks them
        ctx.mark_non_differentiable(sil, p2f)
        ctx.set_materialize_grads(False)
        return imgs, sil, p2f

    @staticmethod
    def backward(ctx, gimgs, _gs, _gp):
        tidx, imgs = ctx.saved_tensors
        N, F, H, R, NA, V = ctx.cfg
        ga = None
        pay = None
        if ctx.needs_input_grad[3] and type(gimgs) is LazyGrad and TEX_BWD_GATHER and R <= 8 and not ctx.f16:
            pay = gimgs.take("tex_mse", imgs)
        if pay is not None:      # the texture MSE's gradient, still unformed: the atlas-gradient kernel forms it per pixel
            t0, ri, rm, rb, go = pay
            ga = torch.empty((NA, F, R, R, 3), dtype=torch.float32, device=t0.device)
            ws, nb, ws_blur = ctx.ws
            with torch.cuda.device(t0.device):
                _lib.check(_lib.lib().acfm_tex_mse_backward_faces(
                    _lib.ptr(t0), _lib.ptr(ri), _lib.ptr(rm), rb, _lib.ptr(go), _lib.ptr(tidx), _lib.ptr(ws), nb,
                    ws_blur, N, V, F, H, R, NA, _lib.ptr(ga), _lib.tuning_ptr(ctx.tune), _lib.cur_stream(t0.device)),
                    "acfm_tex_mse_backward_faces")
        elif ctx.needs_input_grad[3] and gimgs is not None:
            g = _f32c(gimgs)
            ga = torch.empty((NA, F, R, R, 3), dtype=torch.float32, device=g.device)
            ws, nb, ws_blur = ctx.ws
            with torch.cuda.device(g.device):
                if TEX_BWD_GATHER and R <= 8:
                    _lib.check(_lib.lib().acfm_tex_backward_faces(
                        _lib.ptr(g), _lib.ptr(tidx), _lib.ptr(ws), nb, ws_blur, N, V, F, H, R, NA,
                        _lib.ptr(ga), _lib.cur_stream(g.device)), "acfm_tex_backward_faces")
                else:
                    _lib.check(_lib.lib().acfm_tex_backward(_lib.ptr(g), _lib.ptr(tidx), N, F, H, R, NA,
                                                            _lib.ptr(ga), _lib.cur_stream(g.device)),
                               "acfm_tex_backward")
        # geometry / camera: integer texel lookup and K=1 blending send (numerically) no
        # gradient -- |d rgb / d dist| <= 1e-6 |texel| from the delta=1e-10 term.
        if ga is not None and ctx.adt != torch.float32:
            ga = ga.to(ctx.adt)
        return None, None, None, ga, None, None, None, None, None


def tex_render(verts, faces, cams, atlas, img_size, sigma=1e-4, gamma=1e-4, offset_z=0.0, storage="f32"):
    """Atlas-textured hard render: -> (imgs [N,3,H,H], sil [N,H,H], pix_to_face [N,H,H,1]).
    atlas [N,F,R,R,3], or [N/G,F,R,R,3] when G hypotheses of every frame share the frame's
    texture (mesh n samples atlas n % (N/G); equivalent to atlas.repeat(G,1,1,1,1) without the
    copies, gradients of the G renders summed)."""
    return _TexRender.apply(verts, faces, cams, atlas, img_size, sigma, gamma, offset_z, _is_f16(storage))


class _TexRenderMSE(torch.autograd.Function):
    """acfm_tex_mse_forward / acfm_tex_mse_backward_faces: atlas render + masked MSE against reference images as one op."""

    @staticmethod
    def forward(ctx, verts, faces, cams, atlas, ref_img, ref_mask, img_size, sigma, gamma, offset_z, f16=False):
        _lib.require_gpu(verts, faces, cams, atlas, ref_img, ref_mask)
        v, c, a = _f32c(verts), _f32c(cams), _real(atlas, f16)
        ri, rm = _real(ref_img, f16), _real(ref_mask, f16)
        N, V, _ = v.shape
        f = expand_faces(faces, N)
        F, H = f.shape[1], int(img_size)
        NA = a.shape[0] if a.dim() == 5 else 0
        if a.dim() != 5 or NA == 0 or N % NA != 0 or a.shape[1] != F or a.shape[2] != a.shape[3] or a.shape[4] != 3:
            raise ValueError("atlas must be [N,F,R,R,3] (or [N/G,F,R,R,3]), got %s for N=%d F=%d" % (tuple(a.shape), N, F))
        R = a.shape[2]
        if R > 8:
            raise ValueError("tex_render_mse: atlas resolution R <= 8 (use tex_render + tex_mse beyond)")
        RB = _ref_batch(N, ri, "tex_render_mse")
        if ri.shape[1:] != (3, H, H) or rm.reshape(-1, H, H).shape[0] != RB:
            raise ValueError("ref_img [N or N/G,3,H,H] and ref_mask [same batch,H,H]")
        rm = rm.reshape(RB, H, H)
        rdt = torch.float16 if f16 else torch.float32
        loss = torch.empty((N,), dtype=torch.float32, device=v.device)
        shared = _shared_setup(v, c, f, H, offset_z)
        holder = None
        if shared is not None:
            ws, nb, ws_blur, tune, holder = shared
        else:
            ws, nb = _workspace(N, V, F, H, v.device)
            ws_blur, tune = 0.0, _lib.tuning()[1]
        ws_ready = _cover_taken(v.device, tune) if shared is not None else 0
        pf = _take_prefill(holder, N, H) if (ws_ready == 2 and not f16) else None
        if pf is not None:
            imgs, sil, p2f, tidx = pf
            ws_ready = 3
        else:
            imgs = torch.empty((N, 3, H, H), dtype=rdt, device=v.device)
            sil = torch.empty((N, H, H), dtype=rdt, device=v.device)
            p2f = torch.empty((N, H, H, 1), dtype=torch.int32 if f16 else torch.int64, device=v.device)
            tidx = torch.empty((N, H, H), dtype=torch.int32, device=v.device)
        tune = _lib.with_f16(tune, f16)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_tex_mse_forward(
                _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(a), _lib.ptr(ri), _lib.ptr(rm), RB, N, V, F, H, R,
                float(sigma), float(gamma), float(offset_z), _lib.ptr(imgs), _lib.ptr(sil), _lib.ptr(p2f),
                _lib.ptr(tidx), _lib.ptr(loss), _lib.ptr(ws), nb, ws_ready, float(ws_blur), NA,
                _lib.tuning_ptr(tune), _lib.cur_stream(v.device)), "acfm_tex_mse_forward")
        ctx.save_for_backward(tidx, imgs, ri, rm)
        ctx.cfg = (N, F, H, R, NA, V, RB)
        ctx.adt, ctx.tune = atlas.dtype, tune
        ctx.ws = (ws, nb, float(ws_blur))
        ctx.mark_non_differentiable(imgs, sil, p2f)
        ctx.set_materialize_grads(False)
        return loss, imgs, sil, p2f

    @staticmethod
    def backward(ctx, gloss, _gi, _gs, _gp):
        tidx, imgs, ri, rm = ctx.saved_tensors
        N, F, H, R, NA, V, RB = ctx.cfg
        ga = None
        if ctx.needs_input_grad[3] and gloss is not None:
            g = _f32c(gloss)
            ga = torch.empty((NA, F, R, R, 3), dtype=torch.float32, device=g.device)
            ws, nb, ws_blur = ctx.ws
            with torch.cuda.device(g.device):
                _lib.check(_lib.lib().acfm_tex_mse_backward_faces(
                    _lib.ptr(imgs), _lib.ptr(ri), _lib.ptr(rm), RB, _lib.ptr(g), _lib.ptr(tidx), _lib.ptr(ws), nb,
                    ws_blur, N, V, F, H, R, NA, _lib.ptr(ga), _lib.tuning_ptr(ctx.tune), _lib.cur_stream(g.device)),
                    "acfm_tex_mse_backward_faces")
            if ctx.adt != torch.float32:
                ga = ga.to(ctx.adt)
        return (None, None, None, ga) + (None,) * 7


def tex_render_mse(verts, faces, cams, atlas, ref_img, ref_mask, img_size, sigma=1e-4, gamma=1e-4, offset_z=0.0,
                   storage="f32"):
    """Atlas-textured render and its masked MSE against reference images as ONE operator (opt-in; the drop-in pair is
    tex_render + tex_mse): -> (loss [N] = mean over (3,H,W) of (tex*mask - img*mask)^2, imgs [N,3,H,H] (no gradient),
    sil, pix_to_face).  Gradient flows from `loss` to the atlas.  ref_img / ref_mask: [N,...] or [N/G,...]."""
    return _TexRenderMSE.apply(verts, faces, cams, atlas, ref_img, ref_mask, img_size, sigma, gamma, offset_z,
                               _is_f16(storage))


def vertex_color_render(verts, faces, cams, verts_rgb, img_size, sigma=1e-4, gamma=1e-4, offset_z=0.0):
    """atlas=False path (per-vertex RGB, visualisation): forward only, no gradients."""
    _lib.require_gpu(verts, faces, cams, verts_rgb)
    v, c, col = _f32c(verts), _f32c(cams), _f32c(verts_rgb)
    N, V, _ = v.shape
    f = expand_faces(faces, N)
    F, H = f.shape[1], int(img_size)
    if col.dim() == 2:
        col = col[None]
    col = col.expand(N, V, 3).contiguous()
    imgs = torch.empty((N, 3, H, H), dtype=torch.float32, device=v.device)
    sil = torch.empty((N, H, H), dtype=torch.float32, device=v.device)
    p2f = torch.empty((N, H, H, 1), dtype=torch.int64, device=v.device)
    nb = _lib.lib().acfm_raster_workspace_bytes(N, V, F, H) + 4 * N * H * H
    ws = torch.empty(nb, dtype=torch.uint8, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(_lib.lib().acfm_vertex_color_forward(
            _lib.ptr(v), _lib.ptr(f), _lib.ptr(c), _lib.ptr(col), N, V, F, H, float(sigma), float(gamma),
            float(offset_z), _lib.ptr(imgs), _lib.ptr(sil), _lib.ptr(p2f), _lib.ptr(ws), nb, 0, 0.0,
            _lib.tuning()[0], _lib.cur_stream(v.device)), "acfm_vertex_color_forward")
    return imgs, sil, p2f


# ------------------------------------------------------------------------------ mask losses
def _ref_batch(N, ref, what):
    """References (ground-truth masks, images, boundary points) may be given once per frame for the G
    hypotheses rendered of it: [N/G, ...] against N predictions, prediction n <-> reference n % (N/G)
    (= the trainer's ref.repeat(G, ...) without the copies)."""
    RB = ref.shape[0]
    if RB <= 0 or N % RB != 0:
        raise ValueError("%s: %d references for %d predictions (must divide)" % (what, RB, N))
    return RB


class _MaskLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, gt, edt):
        _lib.require_gpu(mask, gt, edt)
        m = _f32c(mask)
        N = m.shape[0]
        HW = m[0].numel()
        g = _f32c(gt).reshape(-1, HW) if gt is not None else None
        e = _f32c(edt).reshape(-1, HW) if edt is not None else None
        RB = N
        for r in (g, e):
            if r is not None:
                RB = _ref_batch(N, r, "mask_losses")
        if g is not None and e is not None and g.shape[0] != e.shape[0]:
            raise ValueError("mask_losses: gt and edt must have the same batch")
        out = torch.empty((N, 4), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            _lib.check(_lib.lib().acfm_mask_losses(_lib.ptr(m), _lib.ptr(g), _lib.ptr(e), N, HW, RB,
                                                   _lib.ptr(out), _lib.cur_stream(m.device)),
                       "acfm_mask_losses")
        ctx.save_for_backward(m, g, e)
        ctx.rb = RB
        return out

    @staticmethod
    def backward(ctx, gout):
        m, g, e = ctx.saved_tensors
        N = m.shape[0]
        HW = m[0].numel()
        go = _f32c(gout)
        rb = ctx.rb

        def make():
            gm = torch.empty_like(m)
            with torch.cuda.device(m.device):
                _lib.check(_lib.lib().acfm_mask_losses_backward(_lib.ptr(m), _lib.ptr(g), _lib.ptr(e),
                                                                _lib.ptr(go), N, HW, rb, _lib.ptr(gm),
                                                                _lib.cur_stream(m.device)),
                           "acfm_mask_losses_backward")
            return gm
        if LAZY_GRADS[0] and m.dim() >= 2:
            return LazyGrad(m, "mask_losses", (m, g, e, rb, go), make), None, None
        return make(), None, None


def mask_losses(mask, gt=None, edt=None):
    """One pass over the mask -> [N,4] = (mean|m-gt|, sum m*gt, sum(m+gt-m*gt), mean edt*m).
    gt / edt: [N,...] or [N/G,...] shared by G hypotheses per frame (see _ref_batch)."""
    return _MaskLosses.apply(mask, gt, edt)


class _TexMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, img, mask):
        _lib.require_gpu(tex, img, mask)
        t, i, m = _f32c(tex), _f32c(img), _f32c(mask)
        N = t.shape[0]
        HW = m[0].numel()
        RB = _ref_batch(N, i, "tex_mse")
        if t.shape[1:] != i.shape[1:] or t.shape[1] != 3 or t[0, 0].numel() != HW or m.shape[0] != RB:
            raise ValueError("tex [N,3,H,W], img [N or N/G,3,H,W] and mask [same batch as img,H,W]")
        out = torch.empty((N,), dtype=torch.float32, device=t.device)
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib().acfm_tex_mse(_lib.ptr(t), _lib.ptr(i), _lib.ptr(m), N, HW, RB, _lib.ptr(out),
                                               _lib.cur_stream(t.device)), "acfm_tex_mse")
        ctx.save_for_backward(t, i, m)
        ctx.rb = RB
        return out

    @staticmethod
    def backward(ctx, go):
        t, i, m = ctx.saved_tensors
        N = t.shape[0]
        HW = m[0].numel()
        g = _f32c(go)
        rb = ctx.rb

        def make():
            gt = torch.empty_like(t)
            with torch.cuda.device(t.device):
                _lib.check(_lib.lib().acfm_tex_mse_backward(_lib.ptr(t), _lib.ptr(i), _lib.ptr(m), _lib.ptr(g),
                                                            N, HW, rb, _lib.ptr(gt), _lib.cur_stream(t.device)),
                           "acfm_tex_mse_backward")
            return gt
        if LAZY_GRADS[0]:
            return LazyGrad(t, "tex_mse", (t, i, m, rb, g), make), None, None
        return make(), None, None


def tex_mse(tex, img, mask):
    """mean over (3,H,W) of (tex*mask - img*mask)^2 per mesh -> [N]; gradient to tex only."""
    return _TexMSE.apply(tex, img, mask)


# ------------------------------------------------------------------------------ loss combination
class _Combine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        _lib.require_gpu(*terms)
        ts = [_f32c(t if t.dim() == 2 else t.reshape(t.shape[0], -1)) for t in terms]
        N = ts[0].shape[0]
        cols = [int(t.shape[1]) for t in ts]
        if not 1 <= len(ts) <= 4 or any(t.shape[0] != N for t in ts) or any(c < 1 or c > 4 for c in cols) \
                or sum(cols) != len(weights):
            raise ValueError("combine_losses: up to 4 terms [N] or [N,C<=4] with one weight per column")
        total = torch.empty((), dtype=torch.float32, device=ts[0].device)
        ctx.args = ((ctypes.c_int * len(cols))(*cols), (ctypes.c_float * len(weights))(*[float(w) for w in weights]),
                    len(ts), N, [t.shape for t in terms])
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(total.device):
            _lib.check(_lib.lib().acfm_combine_losses(ptrs, ctx.args[0], ctx.args[1], len(ts), N, _lib.ptr(total),
                                                      _lib.cur_stream(total.device)), "acfm_combine_losses")
        return total

    @staticmethod
    def backward(ctx, go):
        cols, w, nt, N, shapes = ctx.args
        g = _f32c(go).reshape(1)
        need = ctx.needs_input_grad[1:]
        outs = [torch.empty((N, cols[i]), dtype=torch.float32, device=g.device) if need[i] else None
                for i in range(nt)]
        ptrs = (ctypes.c_void_p * nt)(*[(o.data_ptr() if o is not None else None) for o in outs])
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_combine_losses_backward(_lib.ptr(g), ptrs, cols, w, nt, N,
                                                               _lib.cur_stream(g.device)),
                       "acfm_combine_losses_backward")
        return (None,) + tuple(o.reshape(shapes[i]) if o is not None else None for i, o in enumerate(outs))


def combine_losses(terms, weights):
    """(1/N) sum_n sum_t sum_c w[t][c] * terms[t][n, c] -> scalar: the weighted total of per-mesh loss
    vectors and its batch mean (multiframe/main.py:716-765) as one launch each way.  terms: up to 4
    tensors [N] or [N, C<=4] (e.g. the [N,4] output of mask_losses as it is); weights: one float per
    column, flattened term by term."""
    return _Combine.apply(tuple(float(w) for w in weights), *terms)


class _HypTotal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, aux_group, aux_weights, G, N, *terms):
        _lib.require_gpu(*terms)
        ts = [_f32c(t).reshape(-1) for t in terms]
        nt = len(ts)
        if not 1 <= nt <= 8 or any(t.numel() != G * N for t in ts) or len(weights) != nt:
            raise ValueError("hypothesis_total: 1..8 terms of G*N elements, one weight each")
        dev = ts[0].device
        total = torch.empty((G, N), dtype=torch.float32, device=dev)
        probs = torch.empty((G, N), dtype=torch.float32, device=dev)
        aux = torch.empty((2, G, N), dtype=torch.float32, device=dev)
        out = torch.empty(12, dtype=torch.float32, device=dev)
        w = (ctypes.c_float * nt)(*[float(x) for x in weights])
        ag = (ctypes.c_int * nt)(*[int(x) for x in aux_group])
        aw = (ctypes.c_float * nt)(*[float(x) for x in aux_weights])
        ptrs = (ctypes.c_void_p * nt)(*[t.data_ptr() for t in ts])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().acfm_hypothesis_total(ptrs, w, ag, aw, nt, G, N, _lib.ptr(total), _lib.ptr(probs),
                                                        _lib.ptr(aux[0]), _lib.ptr(aux[1]), _lib.ptr(out),
                                                        _lib.cur_stream(dev)), "acfm_hypothesis_total")
        ctx.args = (w, nt, G, N, [t.shape for t in terms])
        ctx.save_for_backward(probs)
        ctx.mark_non_differentiable(total, probs, aux, out)
        return out[0], total, probs, aux, out

    @staticmethod
    def backward(ctx, go, *_):
        w, nt, G, N, shapes = ctx.args
        probs, = ctx.saved_tensors
        g = _f32c(go).reshape(1)
        need = ctx.needs_input_grad[5:]
        outs = [torch.empty((G, N), dtype=torch.float32, device=g.device) if need[i] else None for i in range(nt)]
        ptrs = (ctypes.c_void_p * nt)(*[(o.data_ptr() if o is not None else None) for o in outs])
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_hypothesis_total_backward(_lib.ptr(g), _lib.ptr(probs), w, nt, G, N, ptrs,
                                                                 _lib.cur_stream(g.device)),
                       "acfm_hypothesis_total_backward")
        return (None,) * 5 + tuple(o.reshape(shapes[i]) if o is not None else None for i, o in enumerate(outs))


def hypothesis_total(terms, weights, G, N, aux_group=None, aux_weights=None):
    """Per-hypothesis total, its softmax weighting and the weighted mean (multiframe/main.py:716-746) as one launch
    each way.  terms: 1..8 tensors of G*N elements (row g*N + n); weights: one float each.
    Returns (weighted, total [G,N], probs [G,N], aux [2,G,N], means [12]): weighted = (1/N) sum_n sum_g probs total with
    probs = softmax(-total, dim 0) carrying no gradient; aux[k] = sum of aux_weights[t] * terms[t] over the terms with
    aux_group[t] == k; means = (weighted, mean total, mean aux0, mean aux1, mean of each term ...).  Only `weighted`
    is differentiable."""
    nt = len(terms)
    ag = tuple(aux_group) if aux_group is not None else (-1,) * nt
    aw = tuple(aux_weights) if aux_weights is not None else (0.0,) * nt
    return _HypTotal.apply(tuple(float(w) for w in weights), ag, aw, int(G), int(N), *terms)


class _TexCycle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, textures, T):
        _lib.require_gpu(textures)
        x = _f32c(textures)
        if x.dim() != 5 or x.shape[2] != x.shape[3] or x.shape[4] != 3 or x.shape[0] % int(T) != 0 or x.shape[2] < 2:
            raise ValueError("texture_cycle: atlases [B*T,F,R,R,3] with R >= 2, got %s (T = %d)" % (tuple(x.shape), T))
        B, F, R = x.shape[0] // int(T), x.shape[1], x.shape[2]
        lib = _lib.lib()
        scratch = torch.empty(lib.acfm_texture_cycle_scratch_floats(B, int(T), F, R), dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.acfm_texture_cycle(_lib.ptr(x), B, int(T), F, R, _lib.ptr(scratch), _lib.ptr(loss),
                                              _lib.cur_stream(x.device)), "acfm_texture_cycle")
        ctx.save_for_backward(x, scratch)
        ctx.cfg = (B, int(T), F, R, textures.dtype)
        return loss

    @staticmethod
    def backward(ctx, go):
        x, scratch = ctx.saved_tensors
        B, T, F, R, dt = ctx.cfg
        g = _f32c(go).reshape(1)
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().acfm_texture_cycle_backward(_lib.ptr(x), _lib.ptr(scratch), _lib.ptr(g), B, T, F, R,
                                                              _lib.ptr(gx), _lib.cur_stream(x.device)),
                       "acfm_texture_cycle_backward")
        return (gx if dt == torch.float32 else gx.to(dt)), None


def texture_cycle(textures, num_frames):
    """The texture temporal-consistency term of ShapeTrainer.forward as written there (multiframe/main.py:705-711):
    atlases [B*T,F,R,R,3] -> scalar; two launches forward, one backward (fixed summation order)."""
    return _TexCycle.apply(textures, int(num_frames))


# ------------------------------------------------------------------------------ boundary loss
def visible_vertices(pix_to_face, faces, nv):
    """[N,H,W,K] i64 (slot 0 read) x faces [N,F,3] -> uint8 [N,nv].  A pix_to_face tensor that
    comes straight from sil_render / hard_raster carries the bitmap already (fused into the
    raster kernel); any other tensor goes through the stand-alone kernel."""
    fused = getattr(pix_to_face, "_acfm_vis", None)
    if fused is not None and fused.shape == (pix_to_face.shape[0], nv):
        return fused
    if isinstance(pix_to_face, LazyPixToFace) and not pix_to_face.is_materialized:
        pix_to_face = pix_to_face[..., :1]          # the stand-alone kernel reads slot 0 only
    _lib.require_gpu(pix_to_face, faces)
    p = pix_to_face.detach().to(torch.int64).contiguous()
    N, K = p.shape[0], p.shape[-1]
    HW = p[0].numel() // K
    f = expand_faces(faces, N)
    vis = torch.empty((N, nv), dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().acfm_visible_vertices(_lib.ptr(p), _lib.ptr(f), N, nv, f.shape[1], HW, K,
                                                    _lib.ptr(vis), _lib.cur_stream(p.device)),
                   "acfm_visible_vertices")
    return vis


class _BdsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts_xy, bds, vis):
        _lib.require_gpu(verts_xy, bds, vis)
        v, b = _f32c(verts_xy), _f32c(bds)
        N, V, _ = v.shape
        P = b.shape[1]
        RB = _ref_batch(N, b, "bds_loss")
        loss = torch.empty((N,), dtype=torch.float32, device=v.device)
        arg = torch.empty((N, P), dtype=torch.int32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_bds_loss(_lib.ptr(v), _lib.ptr(b), _lib.ptr(vis.contiguous()), N,
                                                V, P, RB, _lib.ptr(loss), _lib.ptr(arg),
                                                _lib.cur_stream(v.device)), "acfm_bds_loss")
        ctx.save_for_backward(v, b, arg)
        ctx.rb = RB
        return loss

    @staticmethod
    def backward(ctx, gl):
        v, b, arg = ctx.saved_tensors
        N, V, _ = v.shape
        P = b.shape[1]
        g = _f32c(gl)
        gv = torch.empty_like(v)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_bds_loss_backward(_lib.ptr(v), _lib.ptr(b), _lib.ptr(arg),
                                                         _lib.ptr(g), N, V, P, ctx.rb, _lib.ptr(gv),
                                                         _lib.cur_stream(v.device)),
                       "acfm_bds_loss_backward")
        return gv, None, None


def bds_loss_per_mesh(verts_xy, bds, vis):
    """[N,V,2] x [N,P,3] x uint8 [N,V] -> [N] (sum over boundary points)."""
    return _BdsLoss.apply(verts_xy, bds, vis)


# ------------------------------------------------------------------------------ mesh priors
def cot_laplacian(verts, faces):
    """Dense cot Laplacian L [V,V] of one mesh (verts [V,3], faces [F,3]); constant (no grad)."""
    _lib.require_gpu(verts, faces)
    v = _f32c(verts)
    f = faces.detach().to(torch.int64).contiguous()
    V, F = v.shape[0], f.shape[0]
    L = torch.empty((V, V), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(_lib.lib().acfm_cot_laplacian(_lib.ptr(v), _lib.ptr(f), V, F, _lib.ptr(L),
                                                 _lib.cur_stream(v.device)), "acfm_cot_laplacian")
    return L


class _LaplacianSmoothing(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts_packed, conn, vweight, method, vpm=0, fpm=0):
        _lib.require_gpu(verts_packed, conn, vweight)
        v, w = _f32c(verts_packed), _f32c(vweight)
        c = conn.detach().to(torch.int64).contiguous()
        P, F = v.shape[0], c.shape[0]
        n = _lib.lib().acfm_laplacian_smoothing_state_floats(P, F)
        state = torch.empty(n, dtype=torch.float32, device=v.device)
        loss = torch.empty((), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_laplacian_smoothing(_lib.ptr(v), _lib.ptr(c), _lib.ptr(w), P, F,
                                                           int(method), int(vpm), int(fpm), _lib.ptr(loss),
                                                           _lib.ptr(state), _lib.cur_stream(v.device)),
                       "acfm_laplacian_smoothing")
        ctx.save_for_backward(c, state)
        ctx.cfg = (P, F, int(method), int(vpm), int(fpm))
        return loss

    @staticmethod
    def backward(ctx, go):
        c, state = ctx.saved_tensors
        P, F, method, vpm, fpm = ctx.cfg
        g = _f32c(go).reshape(1)
        gv = torch.empty((P, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_laplacian_smoothing_backward(_lib.ptr(c), _lib.ptr(state), _lib.ptr(g),
                                                                    P, F, method, vpm, fpm, _lib.ptr(gv),
                                                                    _lib.cur_stream(g.device)),
                       "acfm_laplacian_smoothing_backward")
        return gv, None, None, None, None, None


def laplacian_smoothing_sum(verts_packed, conn, vweight, method, verts_per_mesh=0, faces_per_mesh=0):
    """sum_v vweight[v] * |L v|_v on packed meshes; method 0 = cot (conn = faces), 1 = uniform
    (conn = unique edges).  verts_per_mesh / faces_per_mesh: the packed arrays are equal-sized meshes
    one after the other (Meshes built from padded [N,V,3] / [N,F,3] tensors): per-mesh LDS kernels."""
    return _LaplacianSmoothing.apply(verts_packed, conn, vweight, method, verts_per_mesh, faces_per_mesh)


class _EdgeRigidity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, edges, verts_t, edges_t, vpm=0):
        _lib.require_gpu(verts, edges, verts_t, edges_t)
        v, vt = _f32c(verts), _f32c(verts_t)
        e = edges.detach().to(torch.int64).contiguous()
        et = edges_t.detach().to(torch.int64).contiguous()
        if e.shape != et.shape:
            raise ValueError("meshes and template must have the same number of edges")
        loss = torch.empty((), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            _lib.check(_lib.lib().acfm_edge_rigidity(_lib.ptr(v), _lib.ptr(e), _lib.ptr(vt), _lib.ptr(et),
                                                     e.shape[0], _lib.ptr(loss), _lib.cur_stream(v.device)),
                       "acfm_edge_rigidity")
        ctx.save_for_backward(v, e, vt, et)
        ctx.vpm = int(vpm)
        return loss

    @staticmethod
    def backward(ctx, go):
        v, e, vt, et = ctx.saved_tensors
        g = _f32c(go).reshape(1)
        gv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        gvt = torch.empty_like(vt) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().acfm_edge_rigidity_backward(
                _lib.ptr(v), _lib.ptr(e), _lib.ptr(vt), _lib.ptr(et), e.shape[0], v.shape[0], vt.shape[0],
                ctx.vpm, _lib.ptr(g), _lib.ptr(gv), _lib.ptr(gvt), _lib.cur_stream(g.device)),
                "acfm_edge_rigidity_backward")
        return gv, None, gvt, None, None


def edge_rigidity_sum(verts_packed, edges, verts_t_packed, edges_t, verts_per_mesh=0):
    """sum_e (|v[e0]-v[e1]| - |vt[et0]-vt[et1]|)^2 on packed meshes.  verts_per_mesh > 0: equal-sized
    meshes and `edges` sorted by its first vertex (Meshes.edges_packed()): the backward accumulates per
    mesh in LDS instead of with global atomics."""
    return _EdgeRigidity.apply(verts_packed, edges, verts_t_packed, edges_t, verts_per_mesh)
