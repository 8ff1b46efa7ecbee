"""Frame sharding over the GPUs of one node.

The reference only has single-process nn.DataParallel (multiframe/main.py:172-193).  Here:
one process per GPU, clips (pairs of frames + all their camera hypotheses) are dealt in
contiguous blocks, per-frame parameters stay on the owning rank, and the ONLY exchange per
step is one all-reduce (RCCL over xGMI; `nccl` backend) of a flat fp32 buffer holding the
gradients of the shared parameters (mean shape, handle weights, loss scalars).  The buffer is
<= ~200 KB, i.e. latency-bound: one collective, no bucketing."""
import torch
import torch.distributed as dist


def clip_shard(num_clips, rank, world):
    """Contiguous block of clips for `rank`: sizes differ by at most one."""
    base, rem = divmod(num_clips, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def frame_shard(num_clips, frames_per_clip, rank, world):
    """Frame index range owned by `rank`; both frames of a clip (and hence the optical-flow
    pair, loss_utils.py:457, and the hypothesis softmax, main.py:736) stay on one rank."""
    s, e = clip_shard(num_clips, rank, world)
    return s * frames_per_clip, e * frames_per_clip


class SharedGradReducer:
    """Sums the gradients of the shared parameters over ranks with ONE collective.

    deterministic=True uses all_gather + a fixed-order local sum, so every rank gets a
    bit-identical result independent of the collective's internal reduction order."""

    def __init__(self, params, group=None, average=False, deterministic=False):
        self.params = list(params)
        self.group = group
        self.average = average
        self.deterministic = deterministic
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def _buffer(self, ref):
        if self._flat is None or self._flat.device != ref.device:
            self._flat = torch.zeros(self.numel, dtype=torch.float32, device=ref.device)
        return self._flat

    def packed(self, n_extra=0, device=None):
        """The flat exchange buffer and its views: -> (flat, [view per shared parameter], extra view).
        A step that writes its shared gradients (and loss scalars) straight into these views -- e.g.
        as the last nodes of a captured hipGraph -- needs no packing or unpacking launches around the
        collective: `reduce_packed()` is then the ONE launch of the exchange."""
        ref = next(p for p in self.params)
        dev = ref.device if device is None else device
        if self._flat is None or self._flat.device != dev or self._flat.numel() != self.numel + n_extra:
            self._flat = torch.zeros(self.numel + n_extra, dtype=torch.float32, device=dev)
        views, o = [], 0
        for p in self.params:
            views.append(self._flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        return self._flat, views, self._flat[o:o + n_extra]

    def reduce_packed(self):
        """All-reduce the buffer of `packed()` in place (sum, or mean with average=True)."""
        flat = self._flat
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            staged = flat.is_cuda and dist.get_backend(self.group) == "gloo"   # CPU rehearsal of the multi-rank path
            buf = flat.cpu() if staged else flat
            if self.deterministic:
                parts = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(parts, buf, group=self.group)
                buf.zero_()
                for part in parts:
                    buf.add_(part)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if staged:
                flat.copy_(buf)
        if self.average:
            flat[:self.numel].div_(world)
        return flat

    def reduce(self, extra_scalars=None):
        """All-reduce p.grad of every shared parameter in place.  `extra_scalars`: optional 1-D
        fp32 tensor appended to the same buffer (loss values for logging) and returned summed."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        ref = next(p for p in self.params)
        n_extra = 0 if extra_scalars is None else extra_scalars.numel()
        flat = self._buffer(ref)
        if n_extra and flat.numel() != self.numel + n_extra:
            self._flat = flat = torch.zeros(self.numel + n_extra, dtype=torch.float32, device=ref.device)
        o = 0
        for p in self.params:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            flat[o:o + p.numel()].copy_(g.reshape(-1))
            o += p.numel()
        if n_extra:
            flat[o:o + n_extra].copy_(extra_scalars.reshape(-1))
        if world > 1:
            # gloo (CPU rehearsal of the multi-rank path) moves device buffers through the host
            staged = flat.is_cuda and dist.get_backend(self.group) == "gloo"
            buf = flat.cpu() if staged else flat
            if self.deterministic:
                parts = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(parts, buf, group=self.group)
                buf.zero_()
                for part in parts:
                    buf.add_(part)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if staged:
                flat.copy_(buf)
        if self.average:
            flat[:self.numel].div_(world)
        o = 0
        for p in self.params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        return flat[o:o + n_extra].clone() if n_extra else None


class SharedShapeExchange:
    """The one exchange of a frame-sharded step whose shared parameters are the handle weights (lbs,
    multiframe/nnutils/mesh_net.py:543-544) and the mean shape: SURVEY section 8e's packed buffer

        [ G = sum_n g_n delta_n^T  (V x K_h)  |  sum_n g_n  (V x 3)  |  extra shared grads  |  loss scalars ]

    i.e. the PRE-SOLVE sums.  pred_v_n = v + P delta_n with P = (L^T L + A^T A)^-1 A^T (deform.DeformSolver):
    each rank back-propagates its own frames down to (dL/dP, dL/dv) -- exactly deform_apply's backward outputs
    `grad_P = sum_n g_n delta_n^T` and `grad_mean = sum_n g_n` --, ONE all-reduce sums them over the ranks, and every
    rank then finishes dL/dlbs = solve_backward(G) locally and identically (the factorisation is replicated, it is
    cheap; reducing dlbs itself would be equivalent).  Per-frame parameters (handle offsets, cameras) never leave
    their rank.

        ex = SharedShapeExchange(solver, extra_params=[...])      # once
        pred_v = ex.apply(delta_local)                             # forward on this rank's frames
        loss_local.backward()                                      # local backward: stops at (P, mean) leaves
        scalars = ex.finish(extra_scalars=loss_local.detach()[None])   # the collective + the solve backward
        # now solver.lbs.grad / solver.mean_v.grad / extra_params' grads hold the full-batch gradients

    average=True divides the reduced gradients by the world size (per-rank losses are means over equal shards)."""

    def __init__(self, solver, extra_params=(), group=None, average=False, deterministic=False):
        self.solver = solver
        self.extra = [p for p in extra_params if p is not solver.lbs and p is not solver.mean_v]
        self.group, self.average, self.deterministic = group, average, deterministic
        self._flat = None
        self._P = self._P_leaf = self._mean_leaf = None
        self.bytes, self._n_sc, self._shapes = 0, 0, None
        # the exchange buffer holds DOUBLES: G = sum g delta^T is summed in double on every rank (the deformation apply's
        # backward writes it here, unrounded), across the ranks, and rounded to float once, behind the collective --
        # d lbs = solve_backward(G) amplifies G's rounding by the conditioning of the system (~1e5), and this way the
        # split of the frames over the ranks no longer shows in it
        self._store, self._filled, self._sink_key = None, False, None

    def _room(self, n, device):
        """A persistent float64 allocation of at least n elements (the packed buffer is its head)."""
        if self._store is None or self._store.numel() < n or self._store.device != device:
            old = self._store
            self._store = torch.zeros(max(n, 64), dtype=torch.float64, device=device)
            if old is not None and old.device == device:
                self._store[:old.numel()].copy_(old)
        return self._store

    def presolve_buffer(self, V, Kh, device):
        """(called by ops._DeformApply.backward) where this step's local sums go: float64 views [V,K_h] and [V,3] at the
        head of the exchange buffer -- the backward's kernel writes the packed layout itself."""
        tail = sum(p.numel() for p in self.extra) + max(self._n_sc, 8)
        st = self._room(V * Kh + 3 * V + tail, device)
        self._filled = True
        return st[:V * Kh].view(V, Kh), st[V * Kh:V * Kh + 3 * V].view(V, 3)

    def __del__(self):
        try:
            if self._sink_key is not None:
                from . import ops
                ops.drop_presolve_sink(self._sink_key)
        except Exception:
            pass

    def apply(self, delta):
        """delta [n_local,K_h,3] -> pred_v [n_local,V,3]; one factorisation per call (lbs / mean shape may have moved)."""
        s = self.solver
        s.refresh()
        self._P = s.solve_matrix()                                   # carries the autograd path to lbs when it is learned
        self._P_leaf = self._P.detach().requires_grad_(True)
        self._mean_leaf = s.mean_v.detach().requires_grad_(True)
        self._filled = False
        if delta.is_cuda:
            from . import ops
            self._sink_key = ops.register_presolve_sink(self._P_leaf, self, self._sink_key)
            return ops.deform_apply(self._mean_leaf, self._P_leaf, delta)
        return self._mean_leaf[None] + torch.matmul(self._P_leaf[None], delta)

    # finish() = pack() -> reduce() -> unpack(); the three stages are public so that a step replayed from a hipGraph can
    # capture pack() (the last launches of the local backward) and run the collective + the solve's backward after the
    # replay (bench.py --gpus N), and so that several exchanges -- one per template of a mixed batch, BASELINE config 4
    # -- can share ONE collective (finish_many).
    def pack(self, gP=None, gmean=None, extra_scalars=None):
        """Complete [G | sum g | extra shared grads | scalars] in the persistent float64 buffer.  On the GPU the first two
        parts are there already (the deformation apply's backward writes its double sums in place: presolve_buffer);
        otherwise gP / gmean -- by default the .grad of the (P, mean) leaves of the last apply() -- are copied in."""
        s = self.solver
        if gP is None:
            gP = self._P_leaf.grad if self._P_leaf.grad is not None else torch.zeros_like(self._P_leaf)
        if gmean is None:
            gmean = self._mean_leaf.grad if self._mean_leaf.grad is not None else torch.zeros_like(self._mean_leaf)
        direct = s.mean_v.grad if (s.mean_v.requires_grad and s.mean_v.grad is not None) else None   # priors on the template
        self._n_sc = 0 if extra_scalars is None else extra_scalars.numel()
        nP, nm = gP.numel(), gmean.numel()
        n = nP + nm + sum(p.numel() for p in self.extra) + self._n_sc
        filled = self._filled and self._store is not None and self._store.device == gP.device
        flat = self._room(n, gP.device)[:n]
        if filled:
            # the deformation apply's backward wrote G and sum g in double at the head of the buffer already (GPU); only a
            # direct term on the mean shape (a prior on the template) is still to be added
            if direct is not None:
                flat[nP:nP + nm].add_(direct.reshape(-1))
        else:
            flat[:nP].copy_(gP.reshape(-1))
            flat[nP:nP + nm].copy_((gmean if direct is None else gmean + direct).reshape(-1))
        o = nP + nm
        for p in self.extra:
            flat[o:o + p.numel()].copy_((p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1))
            o += p.numel()
        if self._n_sc:
            flat[o:o + self._n_sc].copy_(extra_scalars.detach().reshape(-1))
        self._flat = flat
        self._shapes = (tuple(gP.shape), tuple(gmean.shape))
        self.bytes = 8 * n
        return flat

    def reduce(self, flat=None):
        """The ONE collective: all-reduce (sum; mean of the gradient part with average=True) of the packed buffer."""
        flat = self._flat if flat is None else flat
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            staged = flat.is_cuda and dist.get_backend(self.group) == "gloo"   # CPU rehearsal of the multi-rank path
            buf = flat.cpu() if staged else flat
            if self.deterministic:
                gathered = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(gathered, buf, group=self.group)
                buf.zero_()
                for g in gathered:
                    buf.add_(g)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if staged:
                flat.copy_(buf)
        if self.average and world > 1:
            flat[:flat.numel() - self._n_sc].div_(world)
        return flat

    def unpack(self, flat=None):
        """Reduced buffer -> solver.lbs.grad (through the solve's backward, identical on every rank), solver.mean_v.grad,
        the extra parameters' grads; returns the reduced scalars (or None)."""
        s = self.solver
        flat = self._flat if flat is None else flat
        pshape, mshape = self._shapes
        o = 1
        for d in pshape:
            o *= d
        G = flat[:o].view(pshape).to(s.lbs.dtype if s.lbs.is_floating_point() else torch.float32, copy=True)   # the ONE rounding of G
        if s.lbs.requires_grad:
            direct, s.lbs.grad = s.lbs.grad, None                     # a direct term on lbs (a regulariser: replicated,
            # identical on every rank) is kept, like mean_v's; d lbs through P = solve_backward(G), on every rank.
            # (autograd.grad with retain_graph: the factorisation's node may belong to a captured step that is replayed)
            (g_lbs,) = torch.autograd.grad([self._P], [s.lbs], [G], retain_graph=True)
            s.lbs.grad = g_lbs if direct is None else direct + g_lbs
        nm = 1
        for d in mshape:
            nm *= d
        if s.mean_v.requires_grad:
            s.mean_v.grad = flat[o:o + nm].view(mshape).to(s.mean_v.dtype, copy=True)
        o += nm
        for p in self.extra:
            p.grad = flat[o:o + p.numel()].view_as(p).to(p.dtype, copy=True)
            o += p.numel()
        return flat[flat.numel() - self._n_sc:].to(torch.float32, copy=True) if self._n_sc else None

    def finish(self, extra_scalars=None):
        self.pack(extra_scalars=extra_scalars)
        self.reduce()
        return self.unpack()

    @staticmethod
    def reduce_many(exchanges):
        """ONE collective for the packed buffers of several exchanges (one per template of a mixed batch): laid end to
        end in a joint buffer, reduced together, handed back.  Nothing to do in a single-process run."""
        head = exchanges[0]
        world = dist.get_world_size(head.group) if dist.is_initialized() else 1
        if world == 1:
            return
        flats = [ex._flat for ex in exchanges]
        n = sum(f.numel() for f in flats)
        if getattr(head, "_joint", None) is None or head._joint.numel() != n or head._joint.device != flats[0].device:
            head._joint = torch.zeros(n, dtype=flats[0].dtype, device=flats[0].device)
        joint = head._joint
        torch.cat(flats, out=joint)
        # (scalars sit inside the joint buffer, at the end of each exchange's part: reduce everything as sums here and
        # average the gradient parts per exchange below)
        avg, n_sc, head.average, head._n_sc = head.average, head._n_sc, False, 0
        try:
            head.reduce(joint)
        finally:
            head.average, head._n_sc = avg, n_sc
        o = 0
        for ex, f in zip(exchanges, flats):
            part = joint[o:o + f.numel()]
            if ex.average:
                part[:f.numel() - ex._n_sc].div_(world)
            f.copy_(part)
            o += f.numel()
        head.bytes = joint.element_size() * n

    @staticmethod
    def finish_many(exchanges, extra_scalars=None):
        """pack() of every exchange (the scalars ride with the first), reduce_many(), unpack() of every exchange.
        -> the reduced scalars."""
        for i, ex in enumerate(exchanges):
            ex.pack(extra_scalars=extra_scalars if i == 0 else None)
        SharedShapeExchange.reduce_many(exchanges)
        out = None
        for ex in exchanges:
            r = ex.unpack()
            out = r if out is None else out
        return out
