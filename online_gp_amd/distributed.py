"""Data-parallel accumulation of the WISKI sufficient statistics.

Every cache of the hot path is a sum over data points (reference
online_gp/models/batched_fixed_noise_online_gp.py:44-55,160), so the stream
shards naturally over GPUs: each rank scatters its shard of a batch into a
zeroed *delta* copy of (b, W^T D^-1 W stencil, [y^T D^-1 y, logdet D]), one
all-reduce(SUM) (RCCL over xGMI; gloo in the CPU tests) makes the delta global,
and every rank adds it into its replica of the statistics.  Roots / solves are
then computed redundantly per rank (they do not add).

When a step's shards are small the same sum is cheaper to form redundantly: the ranks all-gather the
raw points (q * (d + 2) reals each) and every rank scatters all of them -- (N - 1) extra scatter passes
against an all-reduce of the whole half stencil.  ``exchange="auto"`` picks per call.

The reference has no distributed code; this is the one collective of the path.
"""
import torch
import torch.distributed as dist


def pick_backend(world, n_devices, force=None):
    """(backend, device index of rank r) for a `world`-rank job on a box that shows `n_devices` GPUs: ``"nccl"`` (= RCCL over
    xGMI) with one device per rank whenever the box has a device for every rank -- RCCL refuses two ranks on one device --,
    otherwise ``"gloo"`` with every rank on device 0: the self-test arrangement of the 1-GPU boxes, never used for reported
    numbers.  ``force`` ("nccl" / "gloo") overrides; forcing nccl without enough devices raises."""
    if force not in (None, "", "nccl", "gloo"):
        raise ValueError("force must be 'nccl' or 'gloo'")
    if force == "nccl" and n_devices < world:
        raise RuntimeError(f"nccl (RCCL) needs one device per rank: {world} ranks, {n_devices} devices visible")
    if force == "gloo" or (not force and n_devices < world):
        return "gloo", (lambda rank: 0)
    return "nccl", (lambda rank: rank)


def rccl_info(group=None, device=None):
    """What carried the collectives of a run, for the record (bench.py's JSON line): torch.distributed's backend and world size,
    the number of ranks an all-reduce of ones really summed over (`ranks_seen`), the RCCL version of torch's communicator, and the
    version code of the librccl that libwiski's own transport resolves (wiski_comm_info; no second communicator is made).
    Collective: every rank must call it."""
    info = {"backend": None, "world_size": 1, "ranks_seen": 1}
    if not (dist.is_available() and dist.is_initialized()):
        return info
    info["backend"], info["world_size"] = dist.get_backend(group), dist.get_world_size(group)
    if info["backend"] == "nccl" and device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    one = torch.ones(1, dtype=torch.float32, device=device if info["backend"] == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    info["ranks_seen"] = int(round(float(one.item())))
    if info["backend"] == "nccl":
        try:
            info["torch_rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # noqa: BLE001
            info["torch_rccl_version"] = "unavailable: " + repr(exc)[:80]
        import ctypes

        from . import _hip

        ver = ctypes.c_int32(0)
        rc = _hip.lib().wiski_comm_info(None, ctypes.byref(ver), None, None)
        info["wiski_rccl_version_code"] = ver.value if rc == 0 else f"unavailable (rc {rc})"
    return info


def allreduce_sum_(tensors, group=None):
    """In-place SUM all-reduce of a list of tensors (no-op without a process group).
    Tensors of one dtype are coalesced by the backend where it pays; the dominant
    message is the stencil (7^d * m reals)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    handles = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
    for h in handles:
        h.wait()
    return tensors


def sharded_posterior_moments(model, X, group=None, want_variance=True):
    """Predictive mean and variance at X [n*, d] with the *solves* divided over the ranks (last sentence:
    predictions shard over query points -- embarrassingly parallel).  The statistics are replicated, so every rank holds the
    same posterior; rank r computes the variance right-hand sides of its contiguous slice of the queries (the expensive
    part: one PCG column per query on large grids) and one all-gather assembles the full vectors on every rank.
    Returns (mean [n*], variance [n*] or None).  Single-output models."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    n = X.shape[0]
    per = (n + world - 1) // world
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    was_training = model.training
    model.eval()
    with torch.no_grad():
        mean = model(X).mean if not want_variance else None                  # the mean is one gather: cheaper than any exchange
        if want_variance:
            mvn = model(X[lo:hi]) if hi > lo else None
            part = torch.zeros(2, per, dtype=model._dtype, device=model._device)
            if mvn is not None:
                part[0, :hi - lo] = mvn.mean.reshape(-1)
                part[1, :hi - lo] = mvn.variance.reshape(-1)
    if was_training:
        model.train()
    if not want_variance:
        return mean.reshape(-1), None
    if world == 1:
        return part[0, :n], part[1, :n]
    parts = [torch.empty_like(part) for _ in range(world)]
    if dist.get_backend(group) == "gloo" and part.is_cuda:
        cpu_parts = [p_.cpu() for p_ in parts]
        dist.all_gather(cpu_parts, part.cpu(), group=group)
        parts = [p_.to(part.device) for p_ in cpu_parts]
    else:
        dist.all_gather(parts, part, group=group)
    full = torch.cat(parts, dim=1)[:, :n]
    return full[0], full[1]


class RcclCommunicator:
    """An ncclComm_t created through the C ABI (``wiski_comm_*``, include/wiski.h) for the ``wiski_allreduce_stats``
    collective: rank 0 draws the unique id, torch.distributed (any backend) ships its bytes to the other ranks, every rank
    joins.  One rank per GPU (RCCL refuses two ranks on one device).  Without a process group: a 1-rank communicator."""

    def __init__(self, group=None):
        import ctypes

        from . import _hip

        lib = _hip.lib()
        nb = int(lib.wiski_comm_unique_id_bytes())
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if world > 1 else 0
        buf = (ctypes.c_ubyte * nb)()
        rc0 = lib.wiski_comm_unique_id(buf) if rank == 0 else 0
        if world > 1:
            # (rank 0's status travels with the id: a rank 0 that raised BEFORE the broadcast would leave the others waiting in it)
            t = torch.tensor(list(buf) + [1 if rc0 else 0], dtype=torch.uint8)
            backend = dist.get_backend(group)
            if backend == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            got = t.cpu().tolist()
            buf = (ctypes.c_ubyte * nb)(*got[:nb])
            if got[nb]:
                raise _hip.WiskiError("wiski_comm_unique_id failed on rank 0 (librccl not resolvable?)")
        else:
            _hip.check(rc0, "wiski_comm_unique_id")
        comm = ctypes.c_void_p()
        _hip.check(lib.wiski_comm_init_rank(buf, ctypes.c_int32(world), ctypes.c_int32(rank), ctypes.byref(comm)), "wiski_comm_init_rank")
        self.handle, self.world, self.rank, self._lib = comm, world, rank, lib

    def allreduce_stats_(self, halves, b, cnt, scal):
        """In-place SUM of (list of half-stencil deltas, W^T D^-1 y, row sums, fp64 scalars) -- one grouped launch per output."""
        import ctypes

        from . import _hip

        dev = b.device
        for o, half in enumerate(halves):
            first = o == 0
            fn = _hip.fn("wiski_allreduce_stats", half.dtype)
            rc = fn(self.handle, _hip.dptr(half), ctypes.c_int64(half.numel()),
                    _hip.dptr(b) if first else None, ctypes.c_int64(b.numel() if first else 0),
                    _hip.dptr(cnt) if (first and cnt is not None) else None, ctypes.c_int64(cnt.numel() if (first and cnt is not None) else 0),
                    _hip.dptr(scal) if first else None, ctypes.c_int64(scal.numel() if first else 0), _hip.stream_ptr(dev))
            _hip.check(rc, "wiski_allreduce_stats")

    def close(self):
        if self.handle:
            self._lib.wiski_comm_destroy(self.handle)
            self.handle = None


class ShardedStatsUpdater:
    """Streams rank-local shards into a model whose statistics stay replicated.

    ``update(X, Y, noise)`` == ``model.condition_on_observations(X_all, Y_all,
    noise_all, inplace=True)`` on every rank, where *_all is the concatenation of
    all ranks' shards."""

    def __init__(self, model, group=None, exchange="auto", equal_shards=False, comm=None):
        """exchange: "stats" (all-reduce the statistics deltas), "points" (all-gather the shards, scatter them all
        on every rank), "stencil" (``stream_step`` only: all-gather the shards, then every rank scatters and multiplies
        only ITS groups of the half stencil and one m-vector all-reduce per CG iteration completes the product -- the one
        exchange that divides the step's work; one output, grids beyond the dense regime with m % 4 == 0, else it behaves like "points") or "auto" (the
        cheaper of the first two by a simple cost model).  The point exchange needs the same
        shard length on every rank; ``equal_shards=True`` promises that (no size check), otherwise the sizes are
        compared first (one tiny all-reduce + host read) and unequal shards fall back to the statistics exchange."""
        if exchange not in ("auto", "stats", "points", "stencil"):
            raise ValueError("exchange must be 'auto', 'stats', 'points' or 'stencil'")
        self.model = model
        self.group = group
        self.exchange = exchange
        self.equal_shards = equal_shards
        self.comm = comm            # RcclCommunicator: the statistics exchange goes through the C ABI's wiski_allreduce_stats
        self._delta = None
        self._res_delta = None
        self._lens = None
        self.last_exchange = None

    def _use_points(self, q, world, dev):
        """Same decision on every rank: exchange the points iff the shards have equal length and the (world - 1)
        redundant scatter passes (~1e11 lane-atomics/s) are cheaper than the stencil all-reduce (~1e11 B/s)."""
        if self.exchange == "stats":
            return False
        m = self.model
        hi, lo = self._shard_lengths(q, dev)
        if hi != lo:
            return False
        if self.exchange == "points":
            return True
        grid = m._grid
        T = 4 ** grid.d
        atomics = world * q * (T * (T + 1) // 2) * m.num_outputs
        stencil_bytes = 2 * ((grid.R + 1) // 2) * grid.m * m.num_outputs * (4 if m._dtype == torch.float32 else 8)
        return atomics < stencil_bytes

    def _shard_lengths(self, q, dev):
        """(longest, shortest) shard length of this call over the ranks: the same pair on every rank, so decisions that gate a
        collective can be taken from it.  ``equal_shards=True`` promises (q, q); otherwise one tiny MAX all-reduce + host read,
        made once per call (``_lens`` is dropped at the top of update() / stream_step())."""
        if self.equal_shards:
            return q, q
        if self._lens is None:
            t = torch.tensor([float(q), -float(q)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            hi, neg_lo = t.tolist()
            self._lens = (int(hi), int(-neg_lo))
        return self._lens

    def _gather_rows(self, packed, world):
        """All-gather the rows of ``packed`` [q_r, c] over the ranks -> (all rows in rank order, offset of this rank's rows).
        With ``equal_shards=True`` one collective; otherwise the shard lengths travel first (one tiny all-gather + host read)
        and shorter shards are padded to the longest for the collective, the padding dropped afterwards."""
        rank = dist.get_rank(self.group)
        q = packed.shape[0]
        if self.equal_shards:
            parts = [torch.empty_like(packed) for _ in range(world)]
            dist.all_gather(parts, packed, group=self.group)
            return torch.cat(parts, dim=0), rank * q
        mine = torch.tensor([q], dtype=torch.int64, device=packed.device)
        lens = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(lens, mine, group=self.group)
        lens = [int(t.item()) for t in lens]
        qmax = max(lens)
        if qmax != q:
            packed = torch.cat([packed, packed.new_zeros(qmax - q, packed.shape[1])], dim=0)
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=self.group)
        return torch.cat([p_[:n_] for p_, n_ in zip(parts, lens)], dim=0), sum(lens[:rank])

    def _update_points(self, X, Y, noise, world):
        m = self.model
        dev = m._device
        d = m._grid.d
        X2 = X.reshape(-1, d).to(dev, m._dtype)
        cols = [X2, Y.to(dev, m._dtype)] + ([noise.to(dev, m._dtype)] if noise is not None else [])
        packed = torch.cat(cols, dim=1).contiguous()
        allp, _ = self._gather_rows(packed, world)
        out = Y.shape[1]
        Xa, Ya = allp[:, :d].contiguous(), allp[:, d:d + out].contiguous()
        Na = allp[:, d + out:].contiguous() if noise is not None else None
        m.condition_on_observations(Xa, Ya, Na, inplace=True)

    def stream_step(self, X, Y, want_mean=True):
        """evaluate -> exchange -> absorb -> refresh for this rank's shard of a streamed batch (unit noise), the data-parallel face
        of ``FixedNoiseOnlineSKIGP.stream_step``.  With the point exchange the all-gathered batch goes through the model's
        one-call step (``wiski_stream_step``, deferred poll) and the rank keeps its slice of the predictive means; with the
        statistics all-reduce the three generic calls run (the exchange sits between the absorb and the refresh).
        Returns the predictive mean of X under the posterior before the update ([q]) or None."""
        m = self.model
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1 and self.comm is None:
            return m.stream_step(X, Y, want_mean)
        if Y.dim() == 1:
            Y = Y[:, None]
        d = m._grid.d
        q = X.reshape(-1, d).shape[0]
        self._lens = None
        # the gathered routes ("points" / "stencil") take shards of any length: unless equal shards were promised the lengths are
        # exchanged first and the gather is padded (a hard-wired torch.empty_like(packed) receive buffer hangs or corrupts the
        # collective when the ranks disagree); the cost-model route keeps its own decision
        gathered = self.comm is None and m.num_outputs == 1 and (
            self.exchange in ("stencil", "points") or self._use_points(q, world, m._device))
        stencil = gathered and self.exchange == "stencil" and self._enter_stencil_shard(world)
        if gathered:
            self.last_exchange = "stencil" if stencil else "points"
            packed = torch.cat([X.reshape(-1, d).to(m._device, m._dtype), Y.to(m._device, m._dtype)], dim=1).contiguous()
            allp, lo = self._gather_rows(packed, world)
            mean = m.stream_step(allp[:, :d].contiguous(), allp[:, d:d + 1].contiguous(), want_mean)
            return mean.reshape(-1)[lo:lo + q] if mean is not None else None
        from . import settings

        m._finish_pending()
        mean = None
        if want_mean:
            with settings.skip_posterior_variances(True):
                mean = m(X).mean
        self.update(X, Y)
        m.prediction_cache
        return mean

    def _enter_stencil_shard(self, world):
        """Put the model into stencil-sharded mode (idempotent); False where it does not apply."""
        m = self.model
        if m.__dict__.get("_stencil_shard") is not None:
            return True
        group = self.group
        gloo_cuda = dist.get_backend(group) == "gloo"

        def allreduce(vec, dots):
            if gloo_cuda and vec.is_cuda:
                # test transport (several ranks on one GPU): through the host, which also orders it against the stream
                for t in (vec, dots):
                    if t is not None:
                        h = t.cpu()
                        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                        t.copy_(h)
                return
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)          # RCCL: ordered against the current stream by torch
            if dots is not None:
                dist.all_reduce(dots, op=dist.ReduceOp.SUM, group=group)

        # Transport of the per-product all-reduce.  Default: the callback above, through torch.distributed's communicator -- the route
        # every multi-process test exercises.  WISKI_SHARD_TRANSPORT=rccl takes the C route instead: wiski_allreduce_stats on an
        # ncclComm_t of our own -- vector + p.Ap slots in ONE grouped launch on the solve's stream, no re-entry into Python per CG
        # iteration.  It is opt-in until a run on >= 2 GPUs has passed the sharded tests on it (no such node was available in rounds
        # 4-5: only its 1-rank form has run on hardware), and it checks itself once against torch's all-reduce before it is used.
        comm = None
        import os

        want = os.environ.get("WISKI_SHARD_TRANSPORT", "torch")
        if want == "rccl" and dist.get_backend(group) == "nccl" and not gloo_cuda:
            if getattr(self, "_shard_comm", None) is None:
                sc = RcclCommunicator(group)
                # (sums of small integers: exact in fp32 / fp64 whatever the reduction order)
                probe = torch.arange(1, 1025, dtype=m._dtype, device=m._device) * (1 + dist.get_rank(group))
                pb, ps = probe[:16].clone(), probe[:4].double()
                refs = [t.clone() for t in (probe, pb, ps)]
                for t in refs:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                try:
                    sc.allreduce_stats_([probe], pb, None, ps)
                    ok = all(bool(torch.equal(a, b)) for a, b in zip((probe, pb, ps), refs))
                except Exception:
                    ok = False
                flag = torch.tensor([0 if ok else 1], device=m._device)
                dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
                self._shard_comm = sc if int(flag.item()) == 0 else False
            if self._shard_comm:
                comm = self._shard_comm.handle
        self.shard_transport = "rccl" if comm is not None else "torch"
        return m.enter_stencil_shard(dist.get_rank(group), world, allreduce, comm=comm)

    def _delta_cache(self):
        """Zeroed delta copies of (b, stats); the W^T W delta lives in the model's symmetric
        half-stencil delta buffers (half the bytes of a full stencil on the wire)."""
        m = self.model
        if self._delta is None:
            b = torch.zeros_like(m._kernel_cache["interpolation_cache"])
            stats = torch.zeros_like(m._kernel_cache["_stats"])
            self._delta = {"interpolation_cache": b, "_stats": stats, "WtW": m._kernel_cache["WtW"]}
            if "_cnt" in m._kernel_cache:
                self._delta["_cnt"] = torch.zeros_like(m._kernel_cache["_cnt"])
        else:
            self._delta["interpolation_cache"].zero_()
            self._delta["_stats"].zero_()
            if "_cnt" in self._delta:
                self._delta["_cnt"].zero_()
        return self._delta

    def update(self, X, Y, noise=None):
        from . import grid_ops
        from .models.batched_fixed_noise_online_gp import _wtw_ops

        m = self.model
        getattr(m, "leave_stencil_shard", lambda: None)()      # (collective; a no-op unless the model is stencil-sharded)
        if Y.dim() == 1:
            Y = Y[:, None]
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.comm is not None:
            world = self.comm.world
        if world == 1 and self.comm is None:
            m.condition_on_observations(X, Y, noise, inplace=True)
            return
        if noise is not None:
            noise = m._canon_noise(noise, Y)
        q = X.reshape(-1, m._grid.d).shape[0]
        self._lens = None
        if self.comm is None and self._use_points(q, world, m._kernel_cache["_stats"].device):
            self.last_exchange = "points"
            self._update_points(X, Y, noise, world)
            return
        self.last_exchange = "stats"
        # the two-level preconditioner's block follows the stream point by point (lazy/two_level.py); an all-reduced increment carries no
        # points, so the shards' coordinates (+ weights) are all-gathered beside it -- q (d + 1) reals per rank, against the 86 MB of the
        # statistics -- and noted by the tracker in _absorb.  Without a process group able to gather (comm-only callers) the block is
        # given up (every replica must keep the SAME preconditioner: the path's collective decisions assume replicated state).
        m.__dict__.pop("_stats_points", None)
        applies = getattr(m, "_two_level_applies", None)
        want_tl = applies is not None and applies() and m.__dict__.get("_two_level") is not None
        in_use = getattr(m, "_spectral_in_use", None)
        # want_fac gates a collective (the gather below): it is decided from replicated state and from shard lengths every rank agrees on,
        # never from this rank's own q -- and the limit (the factor re-projects past 2048 absorbed rows) applies to the gathered total
        want_fac = in_use is not None and in_use() and self.comm is None and m.num_outputs == 1
        if want_fac:
            want_fac = world * self._shard_lengths(q, m._kernel_cache["_stats"].device)[0] <= 2048
        gathered_pts = None
        if self.comm is None and m.num_outputs == 1 and (want_tl or want_fac):
            X2 = X.reshape(-1, m._grid.d).to(m._device, m._dtype)
            n2 = None if noise is None else noise[:, :1].to(m._device, m._dtype)
            wa_loc = torch.ones((X2.shape[0], 1), dtype=m._dtype, device=m._device) if n2 is None else 1.0 / n2.clamp_min(1e-7)
            wby_loc = Y[:, :1].to(m._device, m._dtype) if n2 is None else Y[:, :1].to(m._device, m._dtype) / n2
            allp, _ = self._gather_rows(torch.cat([X2, wa_loc, wby_loc], dim=1).contiguous(), world)
            dd = m._grid.d
            gathered_pts = (allp[:, :dd].contiguous(), None if noise is None else allp[:, dd].contiguous(), allp[:, dd + 1].contiguous())
            if want_tl:
                m.__dict__["_stats_points"] = gathered_pts[:2]
        delta = self._delta_cache()
        halves = m._half_buffers()
        # the carried residual R = b - Z - A U stays valid across the exchange: every rank adds its shard's innovation
        # W^T (wb y - wa (W U)) to a zeroed buffer that is all-reduced with the other deltas and added to R
        ms = getattr(m, "_mean_state", None)
        res_delta = None
        if ms is not None and ms.get("R_ok", False):
            if self._res_delta is None or self._res_delta.shape != ms["R"].shape:
                self._res_delta = torch.zeros_like(ms["R"])
            else:
                self._res_delta.zero_()
            res_delta = self._res_delta
        carried = m._absorb(delta, X, Y, noise, init=False, half_delta=halves, res_delta=res_delta)
        # every rank must make the same choice: carried is a function of replicated state (R_ok, settings) only
        if ms is not None and not carried:
            ms["R_ok"] = False
        dev = delta["_stats"].device
        nloc = float(X.reshape(-1, m._grid.d).shape[0])
        if noise is None:
            wsum = torch.full((Y.shape[1],), nloc, dtype=torch.float64, device=dev)
        else:
            wsum = (1.0 / noise.to(dev, torch.float64).clamp_min(1e-7)).sum(0)        # [out]
        count = torch.cat([torch.tensor([nloc], dtype=torch.float64, device=dev), wsum])
        if self.comm is not None:
            scal = torch.cat([delta["_stats"].reshape(-1), count])
            self.comm.allreduce_stats_(list(halves) + ([res_delta] if carried else []), delta["interpolation_cache"], delta.get("_cnt"), scal)
            ns = delta["_stats"].numel()
            delta["_stats"].copy_(scal[:ns].reshape(delta["_stats"].shape))
            count = scal[ns:]
        else:
            small = [delta["interpolation_cache"], delta["_stats"], count] + ([delta["_cnt"]] if "_cnt" in delta else [])
            allreduce_sum_(small + ([res_delta] if carried else []) + list(halves), self.group)
        c = m._kernel_cache
        if carried:
            ms["R"].add_(res_delta)
        c["interpolation_cache"].add_(delta["interpolation_cache"])
        c["_stats"].add_(delta["_stats"])
        if "_cnt" in delta and "_cnt" in c:
            c["_cnt"].add_(delta["_cnt"])
        for dst, half in zip(_wtw_ops(c["WtW"]), halves):
            dst.root = dst.inv_root = None           # the all-reduced increment bypasses the rank-update path: re-derive on demand
            if grid_ops.is_half_stencil(m._grid, dst.stencil):
                dst.stencil.add_(half)                       # native half storage: plain add
                half.zero_()
            else:
                grid_ops.stencil_expand_add(m._grid, half, dst.stencil)
        tot = count.tolist()
        for o in range(len(tot) - 1):
            m._wsum_host[o] += tot[1 + o]
        m.num_data = m.num_data + int(tot[0])
        m._dump_caches()
        for fac in m.__dict__.get("_spectral", {}).values():
            if gathered_pts is not None and want_fac and fac.ref is not None and not fac.stale:
                fac.absorb(*gathered_pts)                                    # the same points, gathered beside the all-reduce: the factor stays exact
            else:
                fac.stale = True                                             # the all-reduced increment bypassed the factor
