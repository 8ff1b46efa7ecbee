"""Shared synthetic-input builders for the parity tests."""
import numpy as np


def make_cams(n, rng, scale=(0.6, 0.9), extent=1.0, trans=0.1):
    """Weak-perspective cameras [n,7] = (s, tx, ty, quat wxyz): azimuth about y composed
    with a small elevation about x, normalised."""
    s = rng.uniform(*scale, size=n) / extent
    t = rng.uniform(-trans, trans, size=(n, 2))
    az = rng.uniform(0, 2 * np.pi, size=n)
    el = rng.normal(0, np.deg2rad(15), size=n)
    qy = np.stack([np.cos(az / 2), 0 * az, np.sin(az / 2), 0 * az], 1)
    qx = np.stack([np.cos(el / 2), np.sin(el / 2), 0 * el, 0 * el], 1)

    def ham(a, b):
        a0, a1, a2, a3 = a.T
        b0, b1, b2, b3 = b.T
        return np.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                         a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], 1)

    q = ham(qx, qy)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.concatenate([s[:, None], t, q], 1).astype(np.float32)


def batch_verts(v, n, rng, noise=0.01):
    return (v[None] + noise * rng.standard_normal((n,) + v.shape)).astype(np.float32)
