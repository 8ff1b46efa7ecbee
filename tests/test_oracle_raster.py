"""CPU: analytic known-answer tests that anchor the rasteriser part of the oracle (PyTorch3D is
not importable here and the reference ships no raster fixtures: 'parity unpinned', see
oracle/acfm_oracle.c).  Geometry follows SURVEY.md App-A."""
import numpy as np

from oracle import oracle as O


def _tri(z=1.0, s=0.5):
    return np.array([[[-s, -s, z], [s, -s, z], [0.0, s, z]]], np.float32)


def _pix(H):
    i = np.arange(H)
    c = -1.0 + (2.0 * (H - 1 - i) + 1.0) / H      # PixToNdc(H-1-i): +x left, +y up
    return np.meshgrid(c, c, indexing="ij")        # yf[yi,xi], xf[yi,xi]


def _dist2_to_tri(px, py, tri):
    def seg(ax, ay, bx, by):
        bax, bay = bx - ax, by - ay
        t = np.clip(((px - ax) * bax + (py - ay) * bay) / (bax * bax + bay * bay), 0, 1)
        return (ax + t * bax - px) ** 2 + (ay + t * bay - py) ** 2
    (x0, y0, _), (x1, y1, _), (x2, y2, _) = tri
    return np.minimum(np.minimum(seg(x0, y0, x1, y1), seg(x0, y0, x2, y2)), seg(x1, y1, x2, y2))


def _inside(px, py, tri):
    (x0, y0, _), (x1, y1, _), (x2, y2, _) = tri
    e = lambda ax, ay, bx, by: (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    a = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
    return (e(x1, y1, x2, y2) / a > 0) & (e(x2, y2, x0, y0) / a > 0) & (e(x0, y0, x1, y1) / a > 0)


def test_single_triangle_coverage_and_distances():
    H, blur = 32, 0.01
    fv = _tri()
    p2f, zbuf, bary, dists = O.rasterize(fv, 1, H, 2, blur)
    yf, xf = _pix(H)
    ins = _inside(xf.astype(np.float64), yf.astype(np.float64), fv[0].astype(np.float64))
    d2 = _dist2_to_tri(xf.astype(np.float64), yf.astype(np.float64), fv[0].astype(np.float64))
    expect = ins | (d2 < blur)
    np.testing.assert_array_equal(p2f[0, ..., 0] >= 0, expect)
    assert (p2f[0, ..., 1] == -1).all()                      # one face only
    got = dists[0, ..., 0][expect]
    np.testing.assert_allclose(got, np.where(ins, -d2, d2)[expect], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(zbuf[0, ..., 0][expect], 1.0, rtol=1e-6)
    b = bary[0, ..., 0, :][ins]
    np.testing.assert_allclose(b.sum(-1), 1.0, atol=1e-5)
    assert (b > 0).all()
    # empty slots are -1 everywhere
    assert (zbuf[0, ..., 0][~expect] == -1).all() and (dists[0, ..., 0][~expect] == -1).all()


def test_image_orientation():
    """+x_ndc is LEFT and +y_ndc is UP in the output image (App-A.0)."""
    H = 16
    fv = np.array([[[0.9, 0.9, 1], [0.6, 0.9, 1], [0.9, 0.6, 1]]], np.float32)  # top-left corner
    p2f, _, _, _ = O.rasterize(fv, 1, H, 1, 0.0)
    ys, xs = np.nonzero(p2f[0, ..., 0] >= 0)
    assert len(ys) > 0 and ys.max() < H // 4 and xs.max() < H // 4


def test_topk_depth_order_truncation_and_tie_break():
    H = 8
    zs = [3.0, 1.0, 2.0, 1.0]                                  # faces 1 and 3 tie in depth
    fv = np.concatenate([_tri(z, 0.9) for z in zs]).astype(np.float32)
    p2f, zbuf, _, _ = O.rasterize(fv, 1, H, 3, 0.0)
    c = H // 2
    assert list(p2f[0, c, c]) == [1, 3, 2]                     # ascending z, tie -> smaller id, farthest dropped
    np.testing.assert_allclose(zbuf[0, c, c], [1.0, 1.0, 2.0], rtol=1e-6)
    # packed ids: second mesh is offset by F
    fv2 = np.concatenate([fv, fv])
    p2f2, _, _, _ = O.rasterize(fv2, 2, H, 3, 0.0)
    assert list(p2f2[1, c, c]) == [5, 7, 6]


def test_behind_camera_and_degenerate_faces_skipped():
    H = 8
    fv = np.concatenate([_tri(-1.0), np.zeros((1, 3, 3), np.float32)])
    p2f, _, _, _ = O.rasterize(fv, 1, H, 2, 0.01)
    assert (p2f == -1).all()


def test_sigmoid_alpha_blend_shared_edge():
    """Exactly on an edge shared by two front faces the mask is 1 - 0.5*0.5 = 0.75 (App-A.5)."""
    p2f = np.array([[[[0, 1, -1]]]], np.int64)
    d = np.array([[[[0.0, 0.0, -1.0]]]], np.float32)
    np.testing.assert_allclose(O.sigmoid_alpha_blend(p2f, d), 0.75, atol=1e-7)
    p2f = np.array([[[[-1, -1, -1]]]], np.int64)
    assert O.sigmoid_alpha_blend(p2f, d)[0, 0, 0] == 0.0


def test_backward_matches_finite_differences():
    rng = np.random.default_rng(0)
    H, K = 24, 4
    verts = rng.uniform(-0.6, 0.6, (1, 6, 3)).astype(np.float32)
    verts[..., 2] = rng.uniform(-0.2, 0.2, (1, 6))
    faces = np.array([[0, 1, 2], [3, 4, 5], [0, 2, 4], [1, 3, 5]], np.int64)
    cams = np.array([[0.9, 0.02, -0.03, 0.96, 0.1, 0.2, 0.15]], np.float32)
    w = rng.standard_normal((1, H, H)).astype(np.float32)
    sigma, blur = 3e-3, float(np.log(9999.0) * 3e-3)          # wide blur: smooth enough for FD
    gv, gc, _, _ = O.sil_render_backward(verts, faces, cams, H, w, K=K, sigma=sigma, blur=blur)

    def f(v, c):
        m, _ = O.sil_render(v, faces, c, H, K=K, sigma=sigma, blur=blur)
        return float((m.astype(np.float64) * w).sum())
    eps = 2e-3
    num_c = np.zeros(7)
    for i in range(7):
        cp, cm = cams.copy(), cams.copy()
        cp[0, i] += eps
        cm[0, i] -= eps
        num_c[i] = (f(verts, cp) - f(verts, cm)) / (2 * eps)
    # the rasteriser gradient holds the clamped projection parameter and the top-K set fixed, so
    # agreement with finite differences is approximate by construction
    assert np.corrcoef(num_c, gc[0])[0, 1] > 0.98
    np.testing.assert_allclose(gc[0], num_c, rtol=0.25, atol=0.05 * np.abs(num_c).max())
    assert np.abs(gv).max() > 0


def test_texture_branch_known_answers():
    H, R = 16, 4
    verts = np.array([[[-0.8, -0.8, 0.0], [0.8, -0.8, 0.0], [0.0, 0.8, 0.0]]], np.float32)
    faces = np.array([[0, 1, 2]], np.int64)
    cams = np.array([[1.0, 0, 0, 1, 0, 0, 0]], np.float32)
    atlas = np.random.default_rng(1).uniform(0.1, 1, (1, 1, R, R, 3)).astype(np.float32)
    imgs, sil, p2f, tidx = O.tex_render(verts, faces, cams, atlas, H)
    cov = p2f[0, ..., 0] >= 0
    assert cov.sum() > 20
    assert (imgs[0][:, ~cov] == 0).all() and (sil[0][~cov] == 0).all() and (tidx[0][~cov] == -1).all()
    flat = atlas.reshape(-1, 3)
    np.testing.assert_allclose(imgs[0][:, cov].T, flat[tidx[0][cov]], rtol=1e-6)   # colour == texel
    assert ((sil[0][cov] >= 0.5) & (sil[0][cov] <= 1.0)).all()
    g = np.ones_like(imgs)
    ga = O.tex_render_backward_atlas(tidx, g, atlas.shape)
    assert abs(ga.sum() - 3 * cov.sum()) < 1e-3


def test_oracle_correlation_against_direct_loops():
    """The correlation oracle (oracle.correlation) against the definition written out with loops
    (correlation_cuda_kernel.cu:73-147 for pad = md, kernel 1, strides 1)."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    N, C, H, W, md = 1, 3, 5, 6, 2
    f1, f2 = rng.standard_normal((N, C, H, W)), rng.standard_normal((N, C, H, W))
    got = O.correlation(f1, f2, md)
    D1 = 2 * md + 1
    for tj in range(-md, md + 1):
        for ti in range(-md, md + 1):
            for y in range(H):
                for x in range(W):
                    s = 0.0
                    if 0 <= y + tj < H and 0 <= x + ti < W:
                        s = float((f1[0, :, y, x] * f2[0, :, y + tj, x + ti]).sum()) / C
                    assert abs(got[0, (tj + md) * D1 + ti + md, y, x] - s) < 1e-6
