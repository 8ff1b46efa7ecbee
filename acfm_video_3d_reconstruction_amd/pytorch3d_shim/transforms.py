"""pytorch3d.transforms functions used by the reference's camera mirroring
(multiframe/main.py:97-125): real-first quaternions."""
import torch
from .. import _lib


def standardize_quaternion(quaternions):
    """Flip the sign so the real part is non-negative."""
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a, b):
    """Composition of two rotations as the versor with non-negative real part."""
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def quaternion_invert(quaternion):
    return quaternion * _lib.const((1.0, -1.0, -1.0, -1.0), quaternion.device, quaternion.dtype)


def _copysign(a, b):
    return torch.where((a < 0) != (b < 0), -a, a)


def _sqrt_positive_part(x):
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def matrix_to_quaternion(matrix):
    """PyTorch3D 0.3.0 formula (sqrt-positive-part + copysign)."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError("Invalid rotation matrix  shape f{matrix.shape}.")
    m00, m11, m22 = matrix[..., 0, 0], matrix[..., 1, 1], matrix[..., 2, 2]
    o0 = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    o1 = _copysign(x, matrix[..., 2, 1] - matrix[..., 1, 2])
    o2 = _copysign(y, matrix[..., 0, 2] - matrix[..., 2, 0])
    o3 = _copysign(z, matrix[..., 1, 0] - matrix[..., 0, 1])
    return torch.stack((o0, o1, o2, o3), -1)


def quaternion_to_matrix(quaternions):
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))
