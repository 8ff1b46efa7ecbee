"""MI355X-native (gfx950) differentiable-render hot path of ACFM.

Drop-in operator surface (same names / signatures as the reference's
multiframe/nnutils/{nmr,geom_utils,loss_utils}.py) on top of hand-written HIP kernels in
``libacfm_hip.so`` (C ABI: include/acfm_hip.h).  There is no CPU fallback: every op raises
if the HIP library is missing or the tensors are not on a GPU.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["_lib", "ops"]
__version__ = "0.1.0"
