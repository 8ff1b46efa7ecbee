"""Operator surface with the reference's module / function names
(multiframe/nnutils/{nmr,geom_utils,loss_utils}.py)."""
from . import geom_utils, loss_utils, nmr  # noqa: F401
