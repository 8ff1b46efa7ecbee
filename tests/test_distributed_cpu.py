"""CPU, world_size 2 over gloo: frame sharding + the single shared-gradient all-reduce give the
same gradients as one process running the full batch.  The per-frame
render is replaced by a smooth stand-in: the sharding/collective logic does not depend on it."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from acfm_video_3d_reconstruction_amd.sharding import SharedGradReducer, clip_shard, frame_shard


def test_shard_ranges_cover_everything():
    for clips, world in ((32, 8), (5, 2), (3, 4), (128, 8)):
        seen = []
        for r in range(world):
            s, e = clip_shard(clips, r, world)
            seen += list(range(s, e))
        assert seen == list(range(clips))
    assert frame_shard(32, 2, 3, 8) == (24, 32)


def _per_frame_loss(mean_v, P, delta, cams):
    v = mean_v[None] + P[None] @ delta                     # deformation apply (closed form)
    proj = cams[:, None, :1] * v[..., :2] + cams[:, None, 1:3]
    return (torch.tanh(proj).pow(2).sum((1, 2)) + 0.1 * v.pow(2).sum((1, 2)))


def _make_problem():
    g = torch.Generator().manual_seed(0)
    V, Kh, clips, T = 40, 6, 6, 2
    mean_v = torch.randn(V, 3, generator=g)
    lbs = torch.randn(V, Kh, generator=g)
    delta = 0.1 * torch.randn(clips * T, Kh, 3, generator=g)
    cams = torch.rand(clips * T, 7, generator=g) + 0.5
    return mean_v, lbs, delta, cams, clips, T


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mean_v, lbs, delta, cams, clips, T = _make_problem()
    mean_v.requires_grad_(True)
    lbs.requires_grad_(True)
    s, e = frame_shard(clips, T, rank, world)
    d_loc = delta[s:e].clone().requires_grad_(True)
    P = torch.softmax(lbs, 0)
    loss = _per_frame_loss(mean_v, P, d_loc, cams[s:e]).sum()
    loss.backward()
    local = (mean_v.grad.clone(), lbs.grad.clone())
    red = SharedGradReducer([mean_v, lbs], deterministic=(rank >= 0))
    tot = red.reduce(extra_scalars=loss.detach().reshape(1))
    # the packed form (the step writes into the exchange buffer, the exchange is one collective)
    red2 = SharedGradReducer([mean_v, lbs])
    flat, views, extra = red2.packed(n_extra=1)
    views[0].copy_(local[0]); views[1].copy_(local[1]); extra.copy_(loss.detach().reshape(1))
    red2.reduce_packed()
    if rank == 0:
        torch.save(dict(mean=mean_v.grad, lbs=lbs.grad, delta=d_loc.grad, loss=tot,
                        mean_packed=views[0].clone(), lbs_packed=views[1].clone(), loss_packed=extra.clone()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_grads_match_full_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    mean_v, lbs, delta, cams, clips, T = _make_problem()
    mean_v.requires_grad_(True)
    lbs.requires_grad_(True)
    delta.requires_grad_(True)
    loss = _per_frame_loss(mean_v, torch.softmax(lbs, 0), delta, cams).sum()
    loss.backward()
    np.testing.assert_allclose(got["mean"].numpy(), mean_v.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["lbs"].numpy(), lbs.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["loss"].item(), loss.item(), rtol=1e-6)
    np.testing.assert_allclose(got["mean_packed"].numpy(), mean_v.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["lbs_packed"].numpy(), lbs.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["loss_packed"].item(), loss.item(), rtol=1e-6)
    s0, e0 = frame_shard(clips, T, 0, 2)
    np.testing.assert_allclose(got["delta"].numpy(), delta.grad[s0:e0].numpy(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# The exchange the north star names: handle weights (lbs) and mean shape are the learned shared parameters; the ranks
# exchange the PRE-SOLVE sums G = sum g delta^T, sum g (+ loss scalars) and finish d lbs through the solve's backward.
def _lbs_problem():
    m = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes.npz"))
    v, f = torch.from_numpy(m["bird_v"])[:], torch.from_numpy(m["bird_f"]).long()
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    g = torch.Generator().manual_seed(1)
    Kh, clips, T = 6, 4, 2
    lbs = torch.from_numpy(fps_lbs_logits(m["bird_v"], Kh))
    delta = 0.05 * torch.randn(clips * T, Kh, 3, generator=g)
    cams = torch.rand(clips * T, 7, generator=g) + 0.5
    extra = torch.randn(5, generator=g)
    return v, f, lbs, delta, cams, extra, clips, T


def _shape_loss(pred_v, cams, mean_p, extra):
    proj = cams[:, None, :1] * pred_v[..., :2] + cams[:, None, 1:3]
    per_frame = torch.tanh(proj).pow(2).sum((1, 2)) + 0.1 * pred_v.pow(2).sum((1, 2)) * extra.pow(2).sum()
    return per_frame.sum() + 0.01 * pred_v.shape[0] * mean_p.pow(2).sum()     # (a prior on the template itself)


def _lbs_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.sharding import SharedShapeExchange
    v, f, lbs, delta, cams, extra, clips, T = _lbs_problem()
    lbs_p, mean_p, extra_p = torch.nn.Parameter(lbs), torch.nn.Parameter(v), torch.nn.Parameter(extra)
    solver = DeformSolver(mean_p, f, lbs_p)
    ex = SharedShapeExchange(solver, extra_params=[extra_p], deterministic=True)
    s, e = frame_shard(clips, T, rank, world)
    d_loc = delta[s:e].clone().requires_grad_(True)
    loss = _shape_loss(ex.apply(d_loc), cams[s:e], mean_p, extra_p)
    loss.backward()
    assert lbs_p.grad is None                                   # the local backward stops at the (P, mean) leaves
    tot = ex.finish(extra_scalars=loss.detach().reshape(1))
    if rank == 0:
        torch.save(dict(lbs=lbs_p.grad, mean=mean_p.grad, extra=extra_p.grad, delta=d_loc.grad, loss=tot,
                        bytes=ex.bytes), out)
    dist.barrier()
    dist.destroy_process_group()


def test_lbs_gradient_through_the_presolve_exchange(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "lbs0.pt")
    mp.spawn(_lbs_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    v, f, lbs, delta, cams, extra, clips, T = _lbs_problem()
    lbs_p, mean_p, extra_p = torch.nn.Parameter(lbs), torch.nn.Parameter(v), torch.nn.Parameter(extra)
    solver = DeformSolver(mean_p, f, lbs_p)
    d_all = delta.clone().requires_grad_(True)
    loss = _shape_loss(solver(d_all), cams, mean_p, extra_p)     # single process, full batch, lbs through the solve
    loss.backward()
    sc = float(lbs_p.grad.abs().max())
    np.testing.assert_allclose(got["lbs"].numpy(), lbs_p.grad.numpy(), rtol=1e-4, atol=1e-5 * sc)
    assert np.linalg.norm(got["lbs"].numpy() - lbs_p.grad.numpy()) <= 1e-5 * np.linalg.norm(lbs_p.grad.numpy())
    np.testing.assert_allclose(got["mean"].numpy(), mean_p.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["extra"].numpy(), extra_p.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["loss"].item(), loss.item(), rtol=1e-6)
    s0, e0 = frame_shard(clips, T, 0, 2)
    np.testing.assert_allclose(got["delta"].numpy(), d_all.grad[s0:e0].numpy(), rtol=1e-5, atol=1e-6)
    V, Kh = lbs.shape
    assert got["bytes"] == 8 * (V * Kh + 3 * V + 5 + 1)           # [G | sum g | extra | loss] in double: ~46 KB here, ~98 KB at K_h = 16


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4: a mixed batch -- every template has its own mean shape and handle weights, the frames of a rank's
# shard interleave the templates -- still costs ONE collective per step (SharedShapeExchange.finish_many: the packed
# buffers of the per-template exchanges laid end to end).
def _mixed_problem():
    m = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes.npz"))
    from acfm_video_3d_reconstruction_amd.synthetic import fps_lbs_logits
    g = torch.Generator().manual_seed(2)
    Kh, clips, T = 5, 6, 2
    tm = []
    for name in ("bird", "cow"):
        tm.append((torch.from_numpy(m[name + "_v"]), torch.from_numpy(m[name + "_f"]).long(),
                   torch.from_numpy(fps_lbs_logits(m[name + "_v"], Kh))))
    delta = 0.05 * torch.randn(clips * T, Kh, 3, generator=g)
    cams = torch.rand(clips * T, 7, generator=g) + 0.5
    which = torch.arange(clips * T) % 2                      # frame n renders template n % 2
    return tm, delta, cams, which, clips, T


def _mixed_loss(preds, cams):
    proj = cams[:, None, :1] * preds[..., :2] + cams[:, None, 1:3]
    return (torch.tanh(proj).pow(2).sum((1, 2)) + 0.1 * preds.pow(2).sum((1, 2))).sum()


def _mixed_forward(solvers_or_ex, delta, which, apply):
    parts, order = [], []
    for t, s in enumerate(solvers_or_ex):
        idx = torch.nonzero(which == t).reshape(-1)
        parts.append(apply(s, delta[idx]))
        order.append(idx)
    inv = torch.argsort(torch.cat(order))
    return torch.cat(parts)[inv]


def _mixed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    from acfm_video_3d_reconstruction_amd.sharding import SharedShapeExchange
    tm, delta, cams, which, clips, T = _mixed_problem()
    params = [(torch.nn.Parameter(v), f, torch.nn.Parameter(l)) for v, f, l in tm]
    exs = [SharedShapeExchange(DeformSolver(v, f, l), deterministic=True) for v, f, l in params]
    s, e = frame_shard(clips, T, rank, world)
    d_loc = delta[s:e].clone().requires_grad_(True)
    loss = _mixed_loss(_mixed_forward(exs, d_loc, which[s:e], lambda ex, d: ex.apply(d)), cams[s:e])
    loss.backward()
    tot = SharedShapeExchange.finish_many(exs, extra_scalars=loss.detach().reshape(1))
    if rank == 0:
        torch.save(dict(lbs=[l.grad for _, _, l in params], mean=[v.grad for v, _, _ in params], delta=d_loc.grad, loss=tot,
                        bytes=exs[0].bytes), out)
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_templates_share_one_collective(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "mixed0.pt")
    mp.spawn(_mixed_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    from acfm_video_3d_reconstruction_amd.deform import DeformSolver
    tm, delta, cams, which, clips, T = _mixed_problem()
    params = [(torch.nn.Parameter(v), f, torch.nn.Parameter(l)) for v, f, l in tm]
    solvers = [DeformSolver(v, f, l) for v, f, l in params]
    d_all = delta.clone().requires_grad_(True)
    loss = _mixed_loss(_mixed_forward(solvers, d_all, which, lambda sv, d: sv(d)), cams)
    loss.backward()
    for t, (v, f, l) in enumerate(params):
        sc = float(l.grad.abs().max())
        np.testing.assert_allclose(got["lbs"][t].numpy(), l.grad.numpy(), rtol=1e-4, atol=1e-5 * sc)
        np.testing.assert_allclose(got["mean"][t].numpy(), v.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got["loss"].item(), loss.item(), rtol=1e-6)
    s0, e0 = frame_shard(clips, T, 0, 2)
    np.testing.assert_allclose(got["delta"].numpy(), d_all.grad[s0:e0].numpy(), rtol=1e-5, atol=1e-6)
    V, Kh = tm[0][2].shape
    assert got["bytes"] == 8 * (2 * (V * Kh + 3 * V) + 1)           # both templates' [G | sum g] + the loss scalar, one buffer
