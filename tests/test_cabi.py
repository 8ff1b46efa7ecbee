"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/acfm_hip.h declares; the Python layer refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "acfm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(acfm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from acfm_video_3d_reconstruction_amd import _lib
    _lib.build()
    assert os.path.exists(_lib.SO_PATH)
    raw = ctypes.CDLL(_lib.SO_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 15
    for name in syms:
        assert hasattr(raw, name), "libacfm_hip.so does not export %s" % name
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    h = _lib.lib()
    assert h.acfm_arch() == b"gfx950"
    assert h.acfm_raster_workspace_bytes(64, 642, 1280, 256) > 0
    assert h.acfm_raster_workspace_bytes(0, 642, 1280, 256) == 0


def test_library_exports_nothing_undeclared():
    """No hidden knobs: every acfm_* symbol the shipping library exports is declared in the public
    header (the acfm_debug_* diagnostics exist only in the DIAG build, tuning is per call)."""
    import subprocess
    from acfm_video_3d_reconstruction_amd import _lib
    _lib.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.SO_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("acfm_")}
    assert exported == set(_declared_symbols()), sorted(exported ^ set(_declared_symbols()))
    assert not any("debug" in s for s in exported)


def test_no_cpu_fallback():
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    v = torch.zeros(1, 4, 3)
    f = torch.zeros(1, 2, 3, dtype=torch.int64)
    c = torch.zeros(1, 7)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NeuralRenderer(32)(v, f, c)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.project(v, c)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "acfm_video_3d_reconstruction_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "libacfm_oracle" not in src, fn
                assert '#include "../../oracle' not in src and "oracle/acfm_oracle" not in src.replace(
                    "oracle/acfm_oracle.c is the bit-level spec", ""), fn


def test_cover_flag_policy_and_tuning_helpers():
    """ACFM_RECORD_COVER is decided on the host (ops._cover_tuning / _cover_taken): off until a texture render has taken
    a silhouette render's workspace over, off again once a recorded plane went unread; raster_tuning(record_cover=...)
    forces it; the helpers never mutate the caller's structure and keep the other flag bits."""
    import torch
    from acfm_video_3d_reconstruction_amd import _lib, ops
    dev = torch.device("cpu")          # the policy is pure host logic: any hashable device key
    ops._COVER.pop(dev, None)
    flag = lambda t: bool(t is not None and t.flags & 4)
    assert not flag(ops._cover_tuning(dev, None))                 # nothing seen yet
    assert ops._cover_taken(dev, None) == 1                       # taken over, but no plane was recorded
    t = ops._cover_tuning(dev, None)
    assert flag(t) and ops._cover_taken(dev, t) == 2              # recorded and read
    assert flag(ops._cover_tuning(dev, None))
    assert not flag(ops._cover_tuning(dev, None))                 # the previous plane was never read
    assert not flag(ops._cover_tuning(dev, None))
    # forced either way, whatever the state; the deterministic / f16 bits survive
    with _lib.raster_tuning(deterministic=True, record_cover=True) as rt:
        t = ops._cover_tuning(dev, rt.t)
        assert t.flags == 5 and rt.t.flags == 1
        t16 = _lib.with_f16(t, True)
        assert t16.flags == 7 and t.flags == 5
    with _lib.raster_tuning(record_cover=False) as rt:
        ops._COVER[dev] = {"on": True, "pending": False}
        assert not flag(ops._cover_tuning(dev, rt.t))
    assert _lib.with_cover(None, False) is None and _lib.with_cover(None, True).flags == 4
    ops._COVER.pop(dev, None)
