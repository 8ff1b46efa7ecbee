"""Synthetic inputs of the benchmark / parity tests.  Host-side numpy,
run once before any timed region."""
import numpy as np


def make_cams(n, rng, scale=(0.6, 0.9), extent=1.0, trans=0.1):
    """Weak-perspective cameras [n,7] = (s, tx, ty, quat wxyz): azimuth ~U(0,2pi) about y
    composed with elevation ~N(0,15deg) about x, normalised."""
    s = rng.uniform(*scale, size=n) / extent
    t = rng.uniform(-trans, trans, size=(n, 2))
    az = rng.uniform(0, 2 * np.pi, size=n)
    el = rng.normal(0, np.deg2rad(15), size=n)
    qy = np.stack([np.cos(az / 2), 0 * az, np.sin(az / 2), 0 * az], 1)
    qx = np.stack([np.cos(el / 2), np.sin(el / 2), 0 * el, 0 * el], 1)
    a0, a1, a2, a3 = qx.T
    b0, b1, b2, b3 = qy.T
    q = np.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                  a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.concatenate([s[:, None], t, q], 1).astype(np.float32)


def batch_verts(v, n, rng, noise=0.01):
    return (v[None] + noise * rng.standard_normal((n,) + v.shape)).astype(np.float32)


def fps_lbs_logits(verts, k, pp=16):
    """Handle-weight logits like MeshNet's init (mesh_net.py:523-544) with Euclidean instead
    of geodesic farthest-point sampling (`gdist` is not available): 1/d^16, log."""
    idx = [int(np.argmax(np.linalg.norm(verts - verts.mean(0), axis=1)))]
    d = np.linalg.norm(verts - verts[idx[0]], axis=1)
    for _ in range(k - 1):
        idx.append(int(np.argmax(d)))
        d = np.minimum(d, np.linalg.norm(verts - verts[idx[-1]], axis=1))
    idx = np.sort(np.asarray(idx))
    dist = np.linalg.norm(verts[:, None] - verts[None, idx], axis=-1)
    with np.errstate(divide="ignore"):
        w = 1.0 / dist ** pp
    w[np.isinf(w)] = 0
    mx = w.max(0)
    for i, j in enumerate(idx):
        w[j, i] = mx[i]
    return np.log(np.clip(w, 1e-12, None)).astype(np.float32)
