"""GPU parity tests (run with -m gpu on the MI355X box): HIP kernels through the C ABI /
Python operator surface vs the CPU oracle on identical seeded inputs.
Bars: pix_to_face bit-exact; masks 1e-6; gradients 1e-4."""
import numpy as np
import pytest
import torch

from helpers import batch_verts, make_cams  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _setup(meshes, name, n, seed, noise=0.01):
    rng = np.random.default_rng(seed)
    v, f = meshes[name + "_v"], meshes[name + "_f"]
    verts = batch_verts(v, n, rng, noise)
    cams = make_cams(n, rng, extent=float(np.abs(v).max()))
    return verts, f, cams


def test_projection_bit_exact(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils import geom_utils
    from conftest import load_golden
    g = load_golden("projection")
    d = _dev()
    X, cams = torch.from_numpy(g["bird_X"]).to(d), torch.from_numpy(g["cams"]).to(d)
    out = geom_utils.orthographic_proj_withz(X, cams, offset_z=5.).cpu().numpy()
    np.testing.assert_array_equal(out, g["bird_withz5"])           # the reference's own output
    np.testing.assert_array_equal(out, O.project(g["bird_X"], g["cams"], 5.0))
    np.testing.assert_array_equal(geom_utils.orthographic_proj(X, cams).cpu().numpy(), g["bird_xy"])
    np.testing.assert_allclose(geom_utils.quat_rotate(X, cams[:, 3:]).cpu().numpy(), g["bird_rot"],
                               rtol=0, atol=1e-6)


def test_projection_backward(meshes):
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 3, 1)
    cams[:, 3:] *= 1.3  # non-unit quaternion: proj_fn does not normalise
    rng = np.random.default_rng(2)
    g = rng.standard_normal(verts.shape).astype(np.float32)
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    (ops.project(tv, tc, 0.5) * torch.tensor(g, device=d)).sum().backward()
    rv = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    rc = torch.tensor(cams, dtype=torch.float64, requires_grad=True)
    (O.project_torch(rv, rc, 0.5) * torch.from_numpy(g).double()).sum().backward()
    np.testing.assert_allclose(tv.grad.cpu().numpy(), rv.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), rc.grad.numpy(), rtol=1e-4, atol=1e-3)
    # the (x, y)-only form (orthographic_proj / project_points): same bits forward, same gradients as the
    # slice of the full projection
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    tv2 = torch.tensor(verts, device=d, requires_grad=True)
    tc2 = torch.tensor(cams, device=d, requires_grad=True)
    xy = NeuralRenderer(64).project_points(tv2, tc2)
    assert xy.shape == verts.shape[:2] + (2,) and xy.is_contiguous()
    assert torch.equal(xy, ops.project(tv2, tc2, 0.0)[..., :2])
    (xy * torch.tensor(g[..., :2].copy(), device=d)).sum().backward()
    rv2 = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    rc2 = torch.tensor(cams, dtype=torch.float64, requires_grad=True)
    (O.project_torch(rv2, rc2, 0.0)[..., :2] * torch.from_numpy(g[..., :2].copy()).double()).sum().backward()
    np.testing.assert_allclose(tv2.grad.cpu().numpy(), rv2.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tc2.grad.cpu().numpy(), rc2.grad.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("name,n,H,seed", [("bird", 4, 128, 0), ("horse", 2, 256, 3), ("cow", 2, 64, 5)])
def test_silhouette_forward(meshes, name, n, H, seed):
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    d = _dev()
    verts, f, cams = _setup(meshes, name, n, seed)
    ref_mask, ref_p2f = O.sil_render(verts, f, cams, H)
    r = NeuralRenderer(H)
    faces = torch.from_numpy(f)[None].repeat(n, 1, 1).to(d)
    mask, p2f = r(torch.from_numpy(verts).to(d), faces, torch.from_numpy(cams).to(d))
    assert mask.shape == (n, H, H) and mask.dtype == torch.float32
    assert p2f.shape == (n, H, H, 20) and p2f.dtype == torch.int64
    np.testing.assert_array_equal(p2f.cpu().numpy(), ref_p2f)            # bit-exact face ids
    np.testing.assert_allclose(mask.cpu().numpy(), ref_mask, rtol=0, atol=1e-6)
    assert (ref_p2f[..., 0] >= 0).mean() > 0.02                          # the test is not vacuous
    # IoU drift vs the oracle (north star: < 1e-4)
    gt = (ref_mask > 0.5).astype(np.float32)
    iou_a = (mask.cpu().numpy() * gt).sum() / (mask.cpu().numpy() + gt - mask.cpu().numpy() * gt).sum()
    iou_b = (ref_mask * gt).sum() / (ref_mask + gt - ref_mask * gt).sum()
    assert abs(iou_a - iou_b) < 1e-6
    # visible-vertex bitmap fused into the raster == the stand-alone kernel == the oracle
    from acfm_video_3d_reconstruction_amd import ops
    V = verts.shape[1]
    fused = p2f._acfm_vis.cpu().numpy()
    alone = ops.visible_vertices(p2f.clone(), faces, V).cpu().numpy()
    ref_vis = O.visible_vertices(torch.from_numpy(f)[None].repeat(n, 1, 1), torch.from_numpy(ref_p2f[..., 0]), V)
    np.testing.assert_array_equal(fused, alone)
    np.testing.assert_array_equal(fused, ref_vis.numpy().astype(np.uint8))


def test_silhouette_overflow_truncation(meshes):
    """More than K faces within the blur radius of a pixel (tiny scale: whole mesh in a few
    pixels) -> the K nearest-depth faces must be kept, like the oracle."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 2, 7)
    cams[:, 0] *= 0.08
    H = 64
    ref_mask, ref_p2f = O.sil_render(verts, f, cams, H)
    assert (ref_p2f[..., 19] >= 0).sum() > 10
    mask, p2f = ops.sil_render(torch.from_numpy(verts).to(d), torch.from_numpy(f).to(d),
                               torch.from_numpy(cams).to(d), H)
    np.testing.assert_array_equal(p2f.cpu().numpy(), ref_p2f)
    np.testing.assert_allclose(mask.cpu().numpy(), ref_mask, rtol=0, atol=1e-6)


def test_silhouette_small_K(meshes):
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 2, 11)
    H = 64
    for K in (2, 4, 10):
        ref_mask, ref_p2f = O.sil_render(verts, f, cams, H, K=K)
        mask, p2f = ops.sil_render(torch.from_numpy(verts).to(d), torch.from_numpy(f).to(d),
                                   torch.from_numpy(cams).to(d), H, K=K)
        np.testing.assert_array_equal(p2f.cpu().numpy(), ref_p2f)
        np.testing.assert_allclose(mask.cpu().numpy(), ref_mask, rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,n,H,seed", [("bird", 3, 128, 21), ("horse", 2, 64, 22)])
def test_silhouette_backward(meshes, name, n, H, seed):
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    verts, f, cams = _setup(meshes, name, n, seed)
    rng = np.random.default_rng(seed + 100)
    gmask = (rng.standard_normal((n, H, H)) / (H * H)).astype(np.float32)
    gv_ref, gc_ref, _, _ = O.sil_render_backward(verts, f, cams, H, gmask)
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    mask, _ = ops.sil_render(tv, torch.from_numpy(f).to(d), tc, H)
    (mask * torch.tensor(gmask, device=d)).sum().backward()
    sv, sc = np.abs(gv_ref).max(), np.abs(gc_ref).max()
    assert sv > 0 and sc > 0
    np.testing.assert_allclose(tv.grad.cpu().numpy(), gv_ref, rtol=1e-4, atol=1e-4 * sv)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), gc_ref, rtol=1e-4, atol=1e-4 * sc)


def test_hard_raster_of(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import OF_NeuralRenderer
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 4, 31)
    H = 128
    proj = O.project(verts, cams, 0.0)
    ref = O.of_raster(proj, f, H)
    r = OF_NeuralRenderer(H)
    p2f = r(torch.from_numpy(proj).to(d), torch.from_numpy(f)[None].repeat(4, 1, 1).to(d))
    assert p2f.shape == (4, H, H, 1) and p2f.dtype == torch.int64
    np.testing.assert_array_equal(p2f.cpu().numpy(), ref)
    assert (ref >= 0).mean() > 0.02


def test_texture_forward_backward(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    d = _dev()
    n, H, R = 3, 128, 6
    verts, f, cams = _setup(meshes, "bird", n, 41)
    rng = np.random.default_rng(42)
    atlas = rng.uniform(0, 1, (n, f.shape[0], R, R, 3)).astype(np.float32)
    imgs_ref, sil_ref, p2f_ref, tidx_ref = O.tex_render(verts, f, cams, atlas, H)
    r = NeuralRenderer(H)
    ta = torch.tensor(atlas, device=d, requires_grad=True)
    imgs, sil, p2f = r(torch.from_numpy(verts).to(d), torch.from_numpy(f)[None].repeat(n, 1, 1).to(d),
                       torch.from_numpy(cams).to(d), textures=ta)
    assert imgs.shape == (n, 3, H, H) and sil.shape == (n, H, H) and p2f.shape == (n, H, H, 1)
    np.testing.assert_array_equal(p2f.cpu().numpy(), p2f_ref)
    np.testing.assert_allclose(imgs.detach().cpu().numpy(), imgs_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(sil.cpu().numpy(), sil_ref, rtol=0, atol=1e-6)
    g = rng.standard_normal(imgs_ref.shape).astype(np.float32)
    (imgs * torch.tensor(g, device=d)).sum().backward()
    ga_ref = O.tex_render_backward_atlas(tidx_ref, g, atlas.shape)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ga_ref, rtol=1e-5, atol=1e-5)


def test_texture_render_takes_over_the_silhouette_setup(meshes):
    """A texture render of the same verts / cams / faces right after a silhouette render reuses that
    render's face setup (ops._SETUP, acfm_tex_forward ws_ready): identical output to a stand-alone
    texture render and to the oracle; any in-place change of an input ends the sharing."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    for name, n, H in (("bird", 3, 128), ("cow", 2, 64)):
        verts, f, cams = _setup(meshes, name, n, 43)
        rng = np.random.default_rng(44)
        atlas = torch.tensor(rng.uniform(0, 1, (n, f.shape[0], 4, 4, 3)).astype(np.float32), device=d)
        tv, tc = torch.tensor(verts, device=d, requires_grad=True), torch.tensor(cams, device=d)
        faces = torch.from_numpy(f)[None].to(d).expand(n, -1, -1)       # broadcast view, as the reference passes it
        ops._SETUP.clear()
        alone = ops.tex_render(tv.detach(), faces, tc, atlas, H)
        mask, _ = ops.sil_render(tv, faces, tc, H)
        hit = ops._shared_setup(tv.detach().contiguous(), tc, ops.expand_faces(faces, n), H, 0.0)
        assert hit is not None
        shared = ops.tex_render(tv.detach(), faces, tc, atlas, H)
        for a, b in zip(alone, shared):
            assert torch.equal(a, b)
        ref = O.tex_render(verts, f, cams, atlas.cpu().numpy(), H)
        np.testing.assert_array_equal(shared[2].cpu().numpy(), ref[2])
        np.testing.assert_allclose(shared[0].cpu().numpy(), ref[0], rtol=0, atol=1e-6)
        mask.sum().backward()                                          # the silhouette backward still finds its workspace intact
        assert torch.isfinite(tv.grad).all() and tv.grad.abs().sum() > 0
        with torch.no_grad():
            tc[:, 1] += 0.1                                            # version bump: no sharing any more
        assert ops._shared_setup(tv.detach().contiguous(), tc, ops.expand_faces(faces, n), H, 0.0) is None
        moved = ops.tex_render(tv.detach(), faces, tc, atlas, H)
        ref2 = O.tex_render(verts, f, tc.cpu().numpy(), atlas.cpu().numpy(), H)
        np.testing.assert_array_equal(moved[2].cpu().numpy(), ref2[2])


def test_texture_render_from_the_cover_plane(meshes):
    """ACFM_RECORD_COVER: the silhouette render notes the nearest COVERING face of every pixel and the texture render
    that takes its workspace over shades from that plane instead of walking the faces (acfm_tex_forward, ws_ready = 2).
    Bit-identical to the stand-alone texture render and to the oracle -- images, silhouettes, face ids, the atlas
    gradient, the fused render + MSE -- with split and unsplit blocks, odd image sizes, batch sizes off the XCD
    grouping, coincident faces (depth ties broken by face id) and K so small that the covering face is not among the
    kept ones.  The flag switches itself on once texture renders are seen to follow silhouette renders."""
    from acfm_video_3d_reconstruction_amd import ops, _lib
    d = _dev()
    cases = (("bird", 3, 128, 20, -3), ("cow", 2, 100, 20, 1), ("horse", 8, 64, 20, 0), ("bird", 5, 72, 2, -3),
             ("bird+dup", 2, 96, 4, -3), ("horse+fine", 2, 256, 20, -3))
    for name, n, H, K, split in cases:
        if name.endswith("+fine"):      # small faces, half of them back-facing: clipping rescales their depth by area / (area + kEps)
            v5, f5 = O.subdivide(meshes["horse_v"], meshes["horse_f"])
            meshes = dict(meshes, **{"horse+fine_v": v5.astype(np.float32), "horse+fine_f": f5.astype(np.int64)})
            verts, f, cams = _setup(meshes, name, n, 91, noise=0.004)
        else:
            verts, f, cams = _setup(meshes, name.split("+")[0], n, 91)
        if name.endswith("+dup"):       # every face twice, the copies under other ids: exact depth ties at every covered pixel
            f = np.concatenate([f, f[::-1]], 0)
        rng = np.random.default_rng(92)
        atlas = torch.tensor(rng.uniform(0, 1, (n, f.shape[0], 3, 3, 3)).astype(np.float32), device=d, requires_grad=True)
        tv, tc = torch.tensor(verts, device=d), torch.tensor(cams, device=d)
        faces = torch.from_numpy(f)[None].to(d).expand(n, -1, -1)
        gimg = torch.tensor(rng.standard_normal((n, 3, H, H)).astype(np.float32), device=d)
        ops._SETUP.clear()
        alone = ops.tex_render(tv, faces, tc, atlas, H)
        g_alone, = torch.autograd.grad((alone[0] * gimg).sum(), atlas)
        outs = {}
        for on in (False, True):
            with _lib.raster_tuning(split=split, record_cover=on):
                m, p = ops.sil_render(tv, faces, tc, H, K=K)
                hit = ops._shared_setup(tv, tc, ops.expand_faces(faces, n), H, 0.0)
                assert hit is not None and bool(hit[3].flags & 4) == on
                t = ops.tex_render(tv, faces, tc, atlas, H)
                g, = torch.autograd.grad((t[0] * gimg).sum(), atlas)
            outs[on] = (m, p[..., 0].clone(), t, g)
        assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])   # the silhouette itself
        for on in (False, True):
            for a, b in zip(alone, outs[on][2]):
                assert torch.equal(a, b), (name, on)
            assert torch.equal(g_alone, outs[on][3]), (name, on)
        ref = O.tex_render(verts, f, cams, atlas.detach().cpu().numpy(), H)
        np.testing.assert_array_equal(outs[True][2][2].cpu().numpy(), ref[2])
        np.testing.assert_allclose(outs[True][2][0].detach().cpu().numpy(), ref[0], rtol=0, atol=1e-6)
        assert (outs[True][2][2] >= 0).any()
    # the fused render + MSE takes the plane too
    verts, f, cams = _setup(meshes, "bird", 4, 93)
    rng = np.random.default_rng(94)
    H = 96
    atlas = torch.tensor(rng.uniform(0, 1, (4, f.shape[0], 4, 4, 3)).astype(np.float32), device=d)
    img = torch.tensor(rng.uniform(0, 1, (4, 3, H, H)).astype(np.float32), device=d)
    gt = torch.tensor((rng.uniform(0, 1, (4, H, H)) > 0.4).astype(np.float32), device=d)
    tv, tc = torch.tensor(verts, device=d), torch.tensor(cams, device=d)
    faces = torch.from_numpy(f)[None].to(d).expand(4, -1, -1)
    res = []
    for on in (False, True):
        with _lib.raster_tuning(record_cover=on):
            ops.sil_render(tv, faces, tc, H)
            res.append(ops.tex_render_mse(tv, faces, tc, atlas, img, gt, H))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    # the policy: off until a texture render has followed a silhouette render, off again once a plane goes unread
    ops._COVER.clear()
    ops._SETUP.clear()
    flag = lambda: bool(ops._shared_setup(tv, tc, ops.expand_faces(faces, 4), H, 0.0)[3] is not None and
                        ops._shared_setup(tv, tc, ops.expand_faces(faces, 4), H, 0.0)[3].flags & 4)
    ops.sil_render(tv, faces, tc, H)
    assert not flag()
    ops.tex_render(tv, faces, tc, atlas, H)
    ops.sil_render(tv, faces, tc, H)
    assert flag()
    ops.tex_render(tv, faces, tc, atlas, H)
    ops.sil_render(tv, faces, tc, H)
    assert flag()
    ops.sil_render(tv, faces, tc, H)          # the plane above was never read
    assert not flag()


def test_texture_atlas_shared_by_hypotheses(meshes):
    """atlas [N/G,...] for G*N/G meshes == atlas.repeat(G,...): same images, gradient = sum over the
    G copies (acfm_tex_forward / _backward, atlas_batch)."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    G, n0, H, R = 3, 2, 64, 4
    verts, f, cams = _setup(meshes, "bird", G * n0, 47)
    rng = np.random.default_rng(48)
    atlas = torch.tensor(rng.uniform(0, 1, (n0, f.shape[0], R, R, 3)).astype(np.float32), device=d)
    tv, tc = torch.tensor(verts, device=d), torch.tensor(cams, device=d)
    faces = torch.from_numpy(f)[None].to(d).expand(G * n0, -1, -1)
    w = torch.tensor(rng.standard_normal((G * n0, 3, H, H)).astype(np.float32), device=d)
    a1 = atlas.clone().requires_grad_(True)
    out1 = ops.tex_render(tv, faces, tc, a1, H)
    (out1[0] * w).sum().backward()
    a2 = atlas.clone().requires_grad_(True)
    out2 = ops.tex_render(tv, faces, tc, a2.repeat(G, 1, 1, 1, 1), H)
    (out2[0] * w).sum().backward()
    for x, y in zip(out1, out2):
        assert torch.equal(x, y)
    np.testing.assert_allclose(a1.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.tex_render(tv, faces, tc, atlas[:1].repeat(4, 1, 1, 1, 1), H)      # 6 meshes, 4 atlases


def test_texture_backward_gather_and_scatter_forms(meshes):
    """The atlas gradient as a per-face gather over the pixels of the face's box
    (acfm_tex_backward_faces) and as per-pixel global atomics (acfm_tex_backward): both equal the
    oracle -- stand-alone workspace, a workspace taken over from a silhouette render (boxes carry its
    blur margin), an atlas shared by G hypotheses, an image size that is no multiple of 8 or 32."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    try:
        for name, n, G, H, R, share_ws in (("bird", 4, 1, 128, 6, False), ("horse", 4, 2, 100, 4, True),
                                           ("cow", 3, 1, 67, 8, True), ("bird", 2, 1, 256, 2, False)):
            verts, f, cams = _setup(meshes, name, n, 61)
            if R == 2:                      # a face count that is no multiple of the kernel's four faces per wave
                f = np.ascontiguousarray(f[:1277])
            rng = np.random.default_rng(62)
            na = n // G
            atlas = rng.uniform(0, 1, (na, f.shape[0], R, R, 3)).astype(np.float32)
            g = rng.standard_normal((n, 3, H, H)).astype(np.float32)
            _, _, _, tidx_ref = O.tex_render(verts, f, cams, np.tile(atlas, (G, 1, 1, 1, 1)), H)
            ref = O.tex_render_backward_atlas(tidx_ref, g, (n,) + atlas.shape[1:])
            ref = ref.reshape((G, na) + atlas.shape[1:]).sum(0)
            tv, tc = torch.tensor(verts, device=d), torch.tensor(cams, device=d)
            faces = torch.from_numpy(f)[None].to(d).expand(n, -1, -1)
            got = {}
            for gather in (True, False):
                ops.TEX_BWD_GATHER = gather
                ops._SETUP.clear()
                if share_ws:
                    ops.sil_render(tv, faces, tc, H)
                    assert ops._shared_setup(tv, tc, ops.expand_faces(faces, n), H, 0.0) is not None
                ta = torch.tensor(atlas, device=d, requires_grad=True)
                imgs, _, _ = ops.tex_render(tv, faces, tc, ta, H)
                (imgs * torch.tensor(g, device=d)).sum().backward()
                got[gather] = ta.grad.cpu().numpy()
                np.testing.assert_allclose(got[gather], ref, rtol=1e-5, atol=1e-5)
            assert (got[True] != 0).sum() == (got[False] != 0).sum()
    finally:
        ops.TEX_BWD_GATHER = True


def test_silhouette_nearest_plane_only(meshes):
    """pix_to_face_slots=1: same mask, same nearest face, same gradients, 1/20 of the id traffic."""
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 3, 51)
    H = 128
    faces = torch.from_numpy(f)[None].repeat(3, 1, 1).to(d)
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    m20, p20 = NeuralRenderer(H)(tv, faces, tc)
    m1, p1 = NeuralRenderer(H, pix_to_face_slots=1)(tv, faces, tc)
    assert p1.shape == (3, H, H, 1)
    np.testing.assert_array_equal(p1.cpu().numpy(), p20[..., :1].cpu().numpy())
    np.testing.assert_array_equal(m1.detach().cpu().numpy(), m20.detach().cpu().numpy())
    np.testing.assert_array_equal(p1._acfm_vis.cpu().numpy(), p20._acfm_vis.cpu().numpy())
    g = torch.randn(3, H, H, device=d)
    a = torch.autograd.grad((m20 * g).sum(), [tv, tc])
    b = torch.autograd.grad((m1 * g).sum(), [tv, tc])
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())


def test_lazy_pix_to_face(meshes):
    """The default NeuralRenderer returns pix_to_face [N,H,W,K] as an ops.LazyPixToFace: `[..., 0]` / `[..., :1]` (what
    the reference's callers read) and bds_loss never produce the other planes; anything else does, and then the tensor
    equals the one rendered with every slot stored (and the oracle's); stale inputs make it raise."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    d = _dev()
    n, H = 3, 64
    verts, f, cams = _setup(meshes, "bird", n, 71)
    faces = torch.from_numpy(f)[None].repeat(n, 1, 1).to(d)
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    ren = NeuralRenderer(H)
    m, p = ren(tv, faces, tc)
    m_e, p_e = NeuralRenderer(H, pix_to_face_slots=20)(tv, faces, tc)
    assert isinstance(p, ops.LazyPixToFace) and not isinstance(p_e, ops.LazyPixToFace)
    assert p.shape == p_e.shape == (n, H, H, 20) and p.dtype == torch.int64 and p.device == p_e.device
    assert torch.equal(p[..., 0], p_e[..., 0]) and torch.equal(p[..., :1], p_e[..., :1]) and not p.is_materialized
    bds = torch.cat([torch.rand(n, 40, 2, device=d) * 2 - 1, torch.ones(n, 40, 1, device=d)], -1)
    la = L.bds_loss(ren.project_points(tv, tc), bds, faces, p, reduce=False)
    lb = L.bds_loss(ren.project_points(tv, tc), bds, faces, p_e, reduce=False)
    assert torch.equal(la, lb) and not p.is_materialized
    assert torch.equal(ops.visible_vertices(p, faces, verts.shape[1]), ops.visible_vertices(p_e, faces, verts.shape[1]))
    assert torch.equal(m, m_e)
    # anything else renders the other planes, once
    assert torch.equal(p[..., 5], p_e[..., 5]) and p.is_materialized
    assert torch.equal(p, p_e) and int((p >= 0).sum()) == int((p_e >= 0).sum())
    _, ref = O.sil_render(verts, f, cams, H)
    np.testing.assert_array_equal(p.cpu().numpy(), ref)
    # gradients are those of the eager render
    g = torch.randn(n, H, H, device=d)
    for x, y in zip(torch.autograd.grad((m * g).sum(), [tv, tc]), torch.autograd.grad((m_e * g).sum(), [tv, tc])):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())
    # stale inputs: loud
    _, p2 = ren(tv, faces, tc)
    with torch.no_grad():
        tv.add_(0.0)
    assert torch.equal(p2[..., 0], p_e[..., 0])
    with pytest.raises(RuntimeError):
        p2.cpu()
    # the fused operator hands out the same kind of tensor
    gt = (torch.rand(n, H, H, device=d) > 0.5).float()
    edt = torch.rand(n, 1, H, H, device=d)
    _, _, p3 = ren.forward_silhouette_losses(tv, faces, tc, gt, edt, raw=True)
    assert isinstance(p3, ops.LazyPixToFace) and torch.equal(p3[..., 0], p_e[..., 0])
    assert torch.equal(p3[..., 19], p_e[..., 19])


def test_vertex_color_render_atlas_false(meshes):
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    d = _dev()
    n, H = 2, 96
    verts, f, cams = _setup(meshes, "cow", n, 61)
    col = np.random.default_rng(62).uniform(0, 1, (n, verts.shape[1], 3)).astype(np.float32)
    ref_img, ref_p2f = O.vertex_color_render(verts, f, cams, col, H)
    imgs, sil, p2f = NeuralRenderer(H)(torch.from_numpy(verts).to(d), torch.from_numpy(f)[None].repeat(n, 1, 1).to(d),
                                       torch.from_numpy(cams).to(d), textures=torch.from_numpy(col).to(d), atlas=False)
    np.testing.assert_array_equal(p2f.cpu().numpy(), ref_p2f)
    np.testing.assert_allclose(imgs.cpu().numpy(), ref_img, rtol=0, atol=2e-6)


def test_refinement_loop_reduces_the_loss(meshes):
    """BASELINE config 3: test-time refinement of handle offsets (+ cameras) on a horse clip
    (predictor.py:287-349) drives the silhouette losses down."""
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    from acfm_video_3d_reconstruction_amd.refine import refine_clip
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    from scipy.ndimage import distance_transform_edt
    d = _dev()
    rng = np.random.default_rng(71)
    v, f = meshes["horse_v"], meshes["horse_f"]
    N, H, Kh = 8, 128, 16
    cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
    faces = torch.tensor(f, device=d)[None].repeat(N, 1, 1)
    solver = DeformSolver(torch.tensor(v, device=d), faces[0], torch.tensor(fps_lbs_logits(v, Kh), device=d))
    r = NeuralRenderer(H)
    with torch.no_grad():
        gt_delta = torch.tensor(rng.normal(0, 0.05, (N, Kh, 3)).astype(np.float32), device=d)
        gt, _ = r(solver(gt_delta), faces, cams)
        gt = (gt > 0.5).float()
    gm = gt.cpu().numpy()
    edt = torch.tensor(np.stack([distance_transform_edt(1 - m) for m in gm]).astype(np.float32)[:, None], device=d)
    ys, xs = np.nonzero(gm[0] > 0.5)
    bds = torch.zeros(N, 64, 3, device=d)                       # a few (invalid-flagged) boundary points
    pred_v, cam, delta, hist = refine_clip(r, solver, torch.zeros(N, Kh, 3, device=d), cams, faces, gt, edt, bds,
                                           num_optim_iter=25, optimize_camera=True)
    assert len(hist) == 25 and hist[-1] < 0.8 * hist[0]
    assert pred_v.shape == (N, v.shape[0], 3) and torch.isfinite(pred_v).all()
    # the same loop with iterations 4..25 replayed from a hipGraph performs the same updates
    pv2, cam2, delta2, hist2 = refine_clip(r, solver, torch.zeros(N, Kh, 3, device=d), cams, faces, gt, edt, bds,
                                           num_optim_iter=25, optimize_camera=True, use_graph=True)
    # (float-atomic summation order makes the two Adam trajectories drift apart slowly)
    assert len(hist2) == 25 and min(hist2) > 0
    np.testing.assert_allclose(hist2[:6], hist[:6], rtol=1e-3)
    np.testing.assert_allclose(hist2, hist, rtol=5e-2)
    assert float((pv2 - pred_v).abs().max()) < 2e-2


def test_refinement_loop_full_size_config3(meshes):
    """BASELINE config 3 at its FULL size (the oracle comparison of one iteration runs at 4 frames @64^2 in
    test_gpu_composed): a 32-frame horse clip @256^2, handle offsets + cameras, 12 Adam iterations with real EDTs and
    boundary points.  Properties: the loss falls; with the deterministic silhouette backward the loop replayed from a
    hipGraph (refine.ClipRefiner.capture) follows the eager loop to the summation order of the two small kernels that
    still use float atomics (the boundary loss's per-vertex LDS sums, the per-mesh loss sums): 1e-5 after 12 steps."""
    from acfm_video_3d_reconstruction_amd import _lib
    from acfm_video_3d_reconstruction_amd import image_utils as IU
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.nnutils.nmr import NeuralRenderer
    from acfm_video_3d_reconstruction_amd.refine import refine_clip
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    d = _dev()
    rng = np.random.default_rng(72)
    v, f = meshes["horse_v"], meshes["horse_f"]
    N, H, Kh, iters = 32, 256, 16, 12
    cams = torch.tensor(make_cams(N, rng, extent=float(np.abs(v).max())), device=d)
    faces = torch.tensor(f, device=d)[None].repeat(N, 1, 1)
    solver = DeformSolver(torch.tensor(v, device=d), faces[0], torch.tensor(fps_lbs_logits(v, Kh), device=d))
    r = NeuralRenderer(H)
    with torch.no_grad():
        gt, _ = r(solver(torch.tensor(rng.normal(0, 0.05, (N, Kh, 3)).astype(np.float32), device=d)), faces, cams)
        gt = (gt > 0.5).float()
    edt = IU.compute_dt(gt, norm=False)[:, None].contiguous()
    bds = IU.compute_boundaries(gt)[:, :1000].contiguous()
    assert bds.shape[1] >= 200 and float(bds[..., 2].sum()) > 0
    z = torch.zeros(N, Kh, 3, device=d)
    with _lib.raster_tuning(deterministic=True):
        pv, cam, delta, hist = refine_clip(r, solver, z, cams, faces, gt, edt, bds, num_optim_iter=iters, optimize_camera=True)
        pv2, cam2, delta2, hist2 = refine_clip(r, solver, z, cams, faces, gt, edt, bds, num_optim_iter=iters,
                                               optimize_camera=True, use_graph=True)
    assert len(hist) == iters and hist[-1] < 0.8 * hist[0] and all(np.isfinite(hist))
    assert pv.shape == (N, v.shape[0], 3) and torch.isfinite(pv).all()
    np.testing.assert_allclose(hist2, hist, rtol=1e-5)
    for a_, b_ in ((delta2, delta), (cam2, cam), (pv2, pv)):
        assert float((a_ - b_).abs().max()) <= 1e-5 * max(1.0, float(b_.abs().max()))


def test_hip_graph_capture_and_replay(meshes):
    """Every entry point is stream-ordered (no allocation or host sync inside): a render +
    loss + backward step captured into a hipGraph replays with identical results."""
    from acfm_video_3d_reconstruction_amd import ops
    from acfm_video_3d_reconstruction_amd.nnutils import loss_utils as L
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 4, 81)
    H = 64
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    faces = torch.from_numpy(f)[None].repeat(4, 1, 1).to(d).contiguous()
    gt = (torch.rand(4, H, H, device=d) > 0.5).float()
    edt = torch.rand(4, 1, H, H, device=d)

    def step():
        mask, p2f = ops.sil_render(tv, faces, tc, H)
        l1, iou, e = L.fused_silhouette_losses(mask, gt, edt)
        gv, gc = torch.autograd.grad((l1 + 0.1 * e).sum(), [tv, tc])
        return mask, p2f, gv, gc

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    got = [t.detach().clone() for t in outs]
    ref = [t.detach() for t in step()]           # eager, after the capture (no autograd graph kept alive across it)
    np.testing.assert_array_equal(got[1].cpu().numpy(), ref[1].cpu().numpy())
    np.testing.assert_array_equal(got[0].cpu().numpy(), ref[0].cpu().numpy())
    for a, b in zip(got[2:], ref[2:]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    del ref
    # new inputs through the same graph: static input buffers are updated in place
    with torch.no_grad():
        tc[:, 1] += 0.05
    g.replay()
    torch.cuda.synchronize()
    m2, p2 = outs[0].detach().clone(), outs[1].clone()
    m3, p3 = [t.detach() for t in step()[:2]]
    np.testing.assert_array_equal(p2.cpu().numpy(), p3.cpu().numpy())
    np.testing.assert_array_equal(m2.cpu().numpy(), m3.cpu().numpy())


def test_cover_plane_inside_a_hip_graph(meshes):
    """The silhouette render + the texture render that shades from its cover plane, captured into one hipGraph: the
    plane lives in the captured workspace, so a replay after an in-place update of the static inputs renders the new
    geometry -- same ids and images as eager renders of the new inputs, with or without the plane."""
    from acfm_video_3d_reconstruction_amd import ops, _lib
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 8, 95)
    H = 96
    tv, tc = torch.tensor(verts, device=d), torch.tensor(cams, device=d)
    faces = torch.from_numpy(f)[None].to(d).expand(8, -1, -1)
    atlas = torch.rand((8, f.shape[0], 3, 3, 3), device=d)

    def step():
        with _lib.raster_tuning(record_cover=True):
            m, p = ops.sil_render(tv, faces, tc, H, k_out=1)
            hit = ops._shared_setup(tv, tc, ops.expand_faces(faces, 8), H, 0.0)
            assert hit is not None and hit[3].flags & 4
            return (m, p) + tuple(ops.tex_render(tv, faces, tc, atlas, H))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    for moved in (False, True):
        if moved:
            with torch.no_grad():
                tc[:, 1] += 0.07
                tv += 0.003
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in outs]
        ops._SETUP.clear()
        with _lib.raster_tuning(record_cover=False):
            ref = ops.sil_render(tv, faces, tc, H, k_out=1) + tuple(ops.tex_render(tv, faces, tc, atlas, H))
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        assert (got[4] >= 0).any()


def test_silhouette_backward_twice_on_one_workspace(meshes):
    """The raster workspace's NDC-gradient scratch is cleared by the face setup and again by every
    backward that reads it (no zero-fill launch of its own): a second backward through the same
    render (retain_graph) gives the same gradients, and so does one with only the cameras asked for."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    verts, f, cams = _setup(meshes, "bird", 3, 71)
    H = 96
    tv = torch.tensor(verts, device=d, requires_grad=True)
    tc = torch.tensor(cams, device=d, requires_grad=True)
    mask, _ = ops.sil_render(tv, torch.from_numpy(f).to(d), tc, H)
    w = torch.randn(3, H, H, device=d)
    loss = (mask * w).sum()
    g1 = torch.autograd.grad(loss, [tv, tc], retain_graph=True)
    g2 = torch.autograd.grad(loss, [tv, tc], retain_graph=True)
    (g3,) = torch.autograd.grad(loss, [tc])
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    assert float((g1[1] - g3).abs().max()) <= 1e-5 * float(g3.abs().max())
    assert float(g1[0].abs().max()) > 0


def test_setup_take_over_never_crosses_a_graph_replay_or_a_stream(meshes):
    """Eager silhouette render -> a hipGraph replay rewrites the vertex buffer in place (no version bump) -> eager
    texture render of the same tensor: the cached face setup must NOT be taken over (the replay is announced with
    ops.graph_replay = invalidate_setups() + replay; torch itself is not patched); the result is the oracle's render of the NEW geometry.  Likewise a texture render on another
    stream, and after ops.invalidate_setups() / with share_setup(False), sets up by itself."""
    from acfm_video_3d_reconstruction_amd import ops
    d = _dev()
    n, H = 2, 64
    verts_a, f, cams = _setup(meshes, "bird", n, 61)
    verts_b = (verts_a + np.array([0.08, -0.05, 0.0], np.float32)).astype(np.float32)
    rng = np.random.default_rng(62)
    atlas = torch.tensor(rng.uniform(0, 1, (n, f.shape[0], 2, 2, 3)).astype(np.float32), device=d)
    tf = torch.from_numpy(f).to(d)[None].repeat(n, 1, 1).contiguous()
    tc = torch.tensor(cams, device=d)
    tv = torch.zeros(n, verts_a.shape[1], 3, device=d)
    src = torch.tensor(verts_a, device=d)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tv.copy_(src)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tv.copy_(src)
    g.replay()
    torch.cuda.synchronize()
    ops.sil_render(tv, tf, tc, H)
    assert ops._shared_setup(tv, tc, tf, H, 0.0) is not None             # plain eager pair: shared
    ver = tv._version
    src.copy_(torch.tensor(verts_b, device=d))
    assert not hasattr(torch.cuda.CUDAGraph.replay, "_acfm_wrapped")     # round 2 edited torch here; no more
    ops.graph_replay(g)                                                  # tv now holds verts_b, same address, same version
    torch.cuda.synchronize()
    assert tv._version == ver and np.array_equal(tv.cpu().numpy(), verts_b)
    assert ops._shared_setup(tv, tc, tf, H, 0.0) is None
    imgs, sil, p2f = ops.tex_render(tv, tf, tc, atlas, H)
    ri, rs, rp, _ = O.tex_render(verts_b, f, cams, atlas.cpu().numpy(), H)
    np.testing.assert_array_equal(p2f.cpu().numpy(), rp)
    np.testing.assert_allclose(imgs.cpu().numpy(), ri, atol=1e-6)
    assert (rp != O.tex_render(verts_a, f, cams, atlas.cpu().numpy(), H)[2]).any()   # the stale setup would have shown
    # another stream: the texture render is not ordered behind the silhouette render -> own setup
    ops.sil_render(tv, tf, tc, H)
    assert ops._shared_setup(tv, tc, tf, H, 0.0) is not None
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        assert ops._shared_setup(tv, tc, tf, H, 0.0) is None
        out2 = ops.tex_render(tv, tf, tc, atlas, H)
    torch.cuda.current_stream().wait_stream(s2)
    np.testing.assert_array_equal(out2[2].cpu().numpy(), rp)
    # explicit controls
    ops.sil_render(tv, tf, tc, H)
    ops.invalidate_setups()
    assert ops._shared_setup(tv, tc, tf, H, 0.0) is None
    old = ops.share_setup(False)
    try:
        ops.sil_render(tv, tf, tc, H)
        assert ops._shared_setup(tv, tc, tf, H, 0.0) is None
    finally:
        ops.share_setup(old)
    # inside ONE capture the pair shares (replayed together, in order); the result replays correctly on new geometry
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            ops.sil_render(tv, tf, tc, H)
            ops.tex_render(tv, tf, tc, atlas, H)
    torch.cuda.current_stream().wait_stream(s)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        m_c, _ = ops.sil_render(tv, tf, tc, H)
        shared_in_capture = ops._shared_setup(tv, tc, tf, H, 0.0) is not None
        i_c, _, p_c = ops.tex_render(tv, tf, tc, atlas, H)
    assert shared_in_capture
    assert ops._shared_setup(tv, tc, tf, H, 0.0) is None                 # ... but not from outside the capture
    src.copy_(torch.tensor(verts_a, device=d))
    g.replay()
    g2.replay()
    torch.cuda.synchronize()
    ra = O.tex_render(verts_a, f, cams, atlas.cpu().numpy(), H)
    np.testing.assert_array_equal(p_c.cpu().numpy(), ra[2])
    np.testing.assert_allclose(i_c.cpu().numpy(), ra[0], atol=1e-6)
