"""The slice of the PyTorch3D 0.3.0 API that the reference's callers import
(multiframe/main.py:29-38, nnutils/predictor.py:9,21,64, nnutils/mesh_net.py:16):
`structures.Meshes`, `loss.mesh_laplacian_smoothing`, `ops.SubdivideMeshes`, `io.load_obj`,
`transforms.{standardize_quaternion, quaternion_multiply, matrix_to_quaternion, ...}`.

It is a shape-compatible shim (same names, arguments, return types), not PyTorch3D: the
renderer classes are NOT here -- rendering goes through nnutils.nmr on the HIP kernels.
`install()` registers the shim as the `pytorch3d` package so `from pytorch3d.structures import
Meshes` in unmodified caller code resolves to it when the real package is absent."""
import sys

from . import io, loss, ops, structures, transforms  # noqa: F401


def install(force=False):
    if "pytorch3d" in sys.modules and not force:
        return sys.modules["pytorch3d"]
    me = sys.modules[__name__]
    sys.modules["pytorch3d"] = me
    for name in ("io", "loss", "ops", "structures", "transforms"):
        sys.modules["pytorch3d." + name] = getattr(me, name)
    return me
