"""Exact block on the dominant spectral modes of wiski_pcg's preconditioner ("two-level" form; include/wiski.h: wiski_twolevel).

The separable model P = (Kt^-1 + a kron_q diag(t_q))^-1 is applied through the generalized eigenbasis X of its factors
(X^T (kron diag t) X = I, Kt = X^-T D X^-1: `grid_ops.kron_eigen`).  In those coordinates the system matrix of the posterior
solve is D^-1 + X^T A X with A = W^T D_noise^-1 W, and P replaces X^T A X by a I.  That is a good model of a stream whose
density is a product of per-dim profiles and a poor one of road-like data (points along lines): a warm-started step of the
50^3 bench needs 6 CG iterations on the clustered stream against 2.5 on the uniform one.  The modes the data inform are few --
the r of largest prior eigenvalue D_j (for a smooth kernel D falls off super-exponentially) -- so this module keeps, for them,
the EXACT block

    G_S = X_S^T A X_S,      N = (D_S^-1 + G_S)^-1 = D_S^1/2 (I + D_S^1/2 G_S D_S^1/2)^-1 D_S^1/2       (r x r)

and leaves the diagonal model on all others: block Jacobi in spectral coordinates, symmetric positive definite whatever N's
age.  Like the reference's rank-limited root space (URLT:69-119 keeps an m x r root of A current by an SVD per batch) the block
follows the stream by a low-rank-structured update -- G_S += F^T diag(wa) F with F = W(x_new) X_S (one projection kernel + one
fp64 MFMA GEMM) -- but nothing downstream depends on its accuracy: a stale N costs CG iterations, never the converged mean.

Refreshes run on a side stream (projection of the points absorbed since the last one, GEMM, wiski_woodbury_c, the one-workgroup
Cholesky + inverse, one more GEMM, cast to fp32 into the spare buffer) while the streaming steps continue with the previous
block; the new block is switched in a fixed number of steps later (deterministic, so that stencil-sharded replicas stay in
lock-step).  A refresh is started when the absorbed weight has grown by `settings.two_level_growth` since the last one.
"""
import ctypes

import numpy as np
import torch

from .. import _hip, grid_ops, settings

_SIDE = {}         # device -> side stream of the refreshes (streams are never destroyed by torch: one per device, not one per model;
                   # HIP offers no priority below the default -- priority_range() = (0, -1) -- so the refresh cannot be made to yield)
_WORK = {}         # (device, r) -> refresh workspace, shared by the blocks of successive models (refreshes of one device serialise on _SIDE)
KMAX = 32          # per-dim eigenvectors the projection kernel can take (wiski_basis_project)
MC_COLS = 64       # columns of a multi-column solve the block can serve (the variance / probe chunks of the model)
MAXR = 480         # one-workgroup Cholesky + inverse (wiski_potrf_inverse) and the slab kernel's LDS staging (512)


def select_modes(D_host, kscale, rank):
    """The `rank` tensor-product modes of largest eigenvalue kscale * prod_q D_q[i_q], each per-dim index among the KMAX largest of
    its dim.  D_host: per-dim eigenvalues in TABLE order (ascending, as kron_eigen stores them).  Returns table indices
    [d, r] sorted by (i0, i1, i2) -- the block order of wiski_twolevel -- and the eigenvalues [r] (fp64)."""
    d = len(D_host)
    tops = [np.arange(len(D))[::-1][:KMAX] for D in D_host]                 # table indices of the largest eigenvalues, descending
    lam = D_host[0][tops[0]]
    for q in range(1, d):
        lam = np.multiply.outer(lam, D_host[q][tops[q]])
    flat = lam.reshape(-1)
    r = int(min(rank, flat.size, MAXR))
    sel = np.argsort(-flat, kind="stable")[:r]
    sub = np.stack(np.unravel_index(sel, lam.shape))                        # [d, r] positions within the tops
    idx = np.stack([tops[q][sub[q]] for q in range(d)])                     # table indices
    order = np.lexsort(tuple(idx[q] for q in range(d - 1, -1, -1)))         # by i0, then i1, then i2
    idx = idx[:, order]
    lam_sel = flat[sel][order] * kscale
    return idx.astype(np.int64), lam_sel


class TwoLevelBlock:
    """The block for ONE eigenbasis (hyper-parameters + density profile).  d = 3, fp32 solves."""

    def __init__(self, grid, device, eig_host, kscale, rank, err):
        assert grid.d == 3 and max(grid.g) <= 64
        self.grid, self.device, self.err = grid, device, err
        X, D = eig_host["X"], eig_host["D"]
        idx, lam = select_modes(D, kscale, rank)
        self.r = r = idx.shape[1]
        self.idx_host = idx
        g0 = grid.g[0]
        # --- what the slab kernel reads
        off = np.zeros(g0 + 1, dtype=np.int32)
        np.add.at(off[1:], idx[0], 1)
        off = np.cumsum(off).astype(np.int32)
        mask = np.zeros((g0, 64), dtype=np.uint64)
        for j in range(r):
            mask[idx[0, j], idx[1, j]] |= np.uint64(1) << np.uint64(idx[2, j])
        pos = (idx[1] * 256 + idx[2]).astype(np.uint16)
        self.nslab = int((np.diff(off) > 0).sum())
        self.d_off = torch.as_tensor(off).to(device)
        self.d_mask = torch.as_tensor(mask.view(np.int64).reshape(-1)).to(device)
        self.d_pos = torch.as_tensor(pos.view(np.int16)).to(device)
        self.d_cs = torch.zeros(max(r, 4) + 1, dtype=torch.int64, device=device)      # exchange words, zeroed once (epoch 0 is never used); [r]: sticky time-out count
        self.N = [torch.zeros((r, r), dtype=torch.float32, device=device) for _ in range(2)]
        self.active = -1                       # index of the N buffer the solver reads, -1: none yet
        self.mc = torch.empty((2, MC_COLS, r), dtype=torch.float32, device=device)   # c_S and d = N c_S per column of a multi-column solve
        self.struct = grid_ops.TwoLevelStruct(r, self.nslab, self.d_mask.data_ptr(), self.d_off.data_ptr(), self.d_pos.data_ptr(),
                                              self.N[0].data_ptr(), self.d_cs.data_ptr(), self.mc.data_ptr(), MC_COLS)
        # --- what the refresh reads: per-dim tables [g_q, kw] of the eigenvectors the selection uses (descending), fp64
        kw = int(max((len(D[q]) - 1 - idx[q]).max() for q in range(3))) + 1
        self.kw = kw
        tabs = [np.ascontiguousarray(X[q][:, ::-1][:, :kw]) for q in range(3)]
        self.Vtab = torch.as_tensor(np.concatenate([t.reshape(-1) for t in tabs])).to(device)
        S = np.stack([len(D[q]) - 1 - idx[q] for q in range(3)]).astype(np.int32)
        self.S = torch.as_tensor(S).to(device).contiguous()
        self._S_long = self.S.long()
        self.lam_unit = torch.as_tensor(lam / kscale).to(device)         # eigenvalues of kron D (kscale applied by wiski_woodbury_c)
        self.kscale = float(kscale)
        self.G = torch.zeros((r, r), dtype=torch.float64, device=device)
        wkey = (str(device), r)
        if wkey not in _WORK:
            nb = int(_hip.lib().wiski_twolevel_refresh_workspace_bytes(ctypes.c_int32(r)))
            _WORK[wkey] = torch.empty(nb, dtype=torch.uint8, device=device)
        self.work = _WORK[wkey]
        if str(device) not in _SIDE:
            # lowest device priority (torch's own range stops at "normal"): the refresh kernels then yield wave slots to the SpMV of the
            # main stream where they meet (measured against a plain torch stream: profiles/r05_side_priority.txt)
            side = None
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                if _hip.lib().wiski_side_stream_create(ctypes.c_int32(1), ctypes.byref(h)) == 0 and h.value:
                    side = torch.cuda.ExternalStream(h.value, device=device)
            _SIDE[str(device)] = side if side is not None else torch.cuda.Stream(device=device)
        self.side = _SIDE[str(device)]
        self.in_flight = None                  # (event, step it was launched at, buffer index)
        self.failed = False
        self.weight_at_launch = 0.0
        self.refreshes = 0

    def ensure_cols(self, k):
        """Room in the multi-column scratch for a solve of k columns (grown on demand: settings.variance_chunk above 64)."""
        if k > self.struct.mc_cols:
            self.mc = torch.empty((2, int(k), self.r), dtype=torch.float32, device=self.device)
            self.struct.d_mc, self.struct.mc_cols = self.mc.data_ptr(), int(k)

    # ------------------------------------------------------------------------------------------------- refresh --
    def launch_refresh(self, points, step, weight, gscale=1.0, subsample=None):
        """Queue, on the side stream: G += sum F^T diag(wa) F over `points` [(X, wa or None), ...], then N -> the spare buffer."""
        tgt = 1 - self.active if self.active >= 0 else 0
        # (the concatenation / weights / sub-sampling run on the MAIN stream, where their sources were produced and will be recycled: a
        #  side-stream read of a caller's temporary could see it overwritten by the main stream's allocator)
        if len(points) == 1:
            X, wa = points[0]
        else:
            X = torch.cat([p_[0] for p_ in points], dim=0)
            wa = None if all(p_[1] is None for p_ in points) else torch.cat(
                [p_[1] if p_[1] is not None else torch.ones(p_[0].shape[0], dtype=p_[0].dtype, device=self.device) for p_ in points])
        sc = None if wa is None else wa.to(X.dtype).sqrt().contiguous()
        sub = subsample if subsample else settings.two_level_subsample.value()
        if sub > 1 and X.shape[0] >= 64 * sub:
            # G is a sum of one rank-one term per point: every sub-th point with weight `sub` estimates it without bias, and a
            # preconditioner needs no more (the projection + Gram product are the refresh's only O(points) work)
            X = X[::sub]
            sc = (sc[::sub] * (sub ** 0.5)).contiguous() if sc is not None else torch.full((X.shape[0],), float(sub) ** 0.5, dtype=X.dtype, device=self.device)
        X = X.contiguous()
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)                       # the batch and everything derived from it above
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            # ONE C call queues projection + Gram product per 16384 points, wiski_woodbury_c, the one-workgroup Cholesky + inverse,
            # N = Lam^1/2 C^-1 Lam^1/2 in fp32 (wiski_twolevel_refresh)
            X.record_stream(self.side)
            if sc is not None:
                sc.record_stream(self.side)
            X = X.contiguous()
            # verdict of the refresh (a failed factorisation poisons N; an exchange word that timed out in the slab kernel since the block was
            # built: d_cs[r], sticky): written by the refresh's last kernel straight into pinned memory, read at tick() without a sync
            if getattr(self, "_bad_host", None) is None:
                self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._bad_host[0] = 0                    # (the previous verdict was consumed when its refresh was switched in: nothing is in flight)
            rc = _hip.lib().wiski_twolevel_refresh_f32(self.grid.ref, _hip.dptr(X), ctypes.c_int64(X.shape[0]), _hip.dptr(sc), _hip.dptr(self.Vtab),
                                                        ctypes.c_int32(self.kw), _hip.dptr(self.S), ctypes.c_int32(self.r), _hip.dptr(self.lam_unit),
                                                        ctypes.c_double(self.kscale), ctypes.c_double(gscale), _hip.dptr(self.G), _hip.dptr(self.work),
                                                        ctypes.c_int64(self.work.numel()), _hip.dptr(self.N[tgt]),
                                                        ctypes.c_void_p(self.d_cs.data_ptr() + 8 * self.r), ctypes.c_void_p(self._bad_host.data_ptr()),
                                                        _hip.stream_ptr(self.device))
            _hip.check(rc, "wiski_twolevel_refresh")
            self._keep = (X, sc)                     # alive until the side stream has read them (replaced by the next refresh)
            done = torch.cuda.Event()
            done.record(self.side)
        self.in_flight = (done, step, tgt)
        self.weight_at_launch = weight
        self.refreshes += 1

    def rebase(self, eig_host, kscale):
        """The same index set S on a NEW eigenbasis (a hyper-parameter step moved the tables a little: the r modes of largest
        eigenvalue are nearly the same ones, and any S gives a symmetric positive definite preconditioner): new eigenvector tables
        and eigenvalues, G and N to be rebuilt by the caller -- two host-to-device copies instead of a new block."""
        X, D = eig_host["X"], eig_host["D"]
        idx = self.idx_host
        lam = np.ones(self.r)
        for q in range(3):
            lam = lam * D[q][idx[q]]
        tabs = [np.ascontiguousarray(X[q][:, ::-1][:, :self.kw]) for q in range(3)]
        self.Vtab.copy_(torch.as_tensor(np.concatenate([t.reshape(-1) for t in tabs])))
        self.lam_unit.copy_(torch.as_tensor(lam))
        self.kscale = float(kscale)
        self.active, self.in_flight, self.failed = -1, None, False
        self.rebases = getattr(self, "rebases", 0) + 1

    def rebuild_from_stencil(self, stencil, weight):
        """G = X_S^T A X_S from the statistics themselves (r stencil-product columns in chunks of 64 + d mode products each), N from it,
        on the CURRENT stream, switched in at once: what a block costs where no history of points in this eigenbasis exists -- after a
        hyper-parameter step, a handed-over kernel cache, an all-reduced statistics delta (~1.5 ms at 50^3, r = 192; the streaming
        refresh pipeline then continues from it)."""
        g, dev, r = self.grid.g, self.device, self.r
        torch.cuda.current_stream(dev).wait_stream(self.side)     # (a refresh of an earlier block may still be using the shared workspace)
        Vq, o = [], 0
        for q in range(3):
            Vq.append(self.Vtab[o:o + g[q] * self.kw].reshape(g[q], self.kw).float())
            o += g[q] * self.kw
        S = self._S_long
        # (fp32 throughout: G only preconditions; all r columns in one pass where they fit in 256 MB, else in chunks of 64)
        step = r if r * self.grid.m * 4 <= (1 << 28) else 64
        G = torch.empty((r, r), dtype=torch.float32, device=dev)
        for lo in range(0, r, step):
            hi = min(lo + step, r)
            c = hi - lo
            Bc = Vq[0][:, S[0, lo:hi]].t()
            for q in (1, 2):
                Bc = Bc.reshape(c, -1, 1) * Vq[q][:, S[q, lo:hi]].t().reshape(c, 1, -1)
            AB = grid_ops.stencil_spmv(self.grid, stencil, Bc.reshape(c, self.grid.m).contiguous())
            P = AB.reshape((c,) + tuple(g))
            for q in range(3):
                P = torch.tensordot(P, Vq[q], dims=([1], [0]))
            G[lo:hi] = P[(slice(None), S[0], S[1], S[2])]
        G = G.double()
        self.G = (0.5 * (G + G.t())).contiguous()
        if getattr(self, "_bad_host", None) is None:
            self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._bad_host[0] = 0
        rc = _hip.lib().wiski_twolevel_refresh_f32(self.grid.ref, None, ctypes.c_int64(0), None, _hip.dptr(self.Vtab), ctypes.c_int32(self.kw),
                                                    _hip.dptr(self.S), ctypes.c_int32(r), _hip.dptr(self.lam_unit), ctypes.c_double(self.kscale),
                                                    ctypes.c_double(1.0), _hip.dptr(self.G), _hip.dptr(self.work), ctypes.c_int64(self.work.numel()),
                                                    _hip.dptr(self.N[0]), ctypes.c_void_p(self.d_cs.data_ptr() + 8 * self.r),
                                                    ctypes.c_void_p(self._bad_host.data_ptr()), _hip.stream_ptr(dev))
        _hip.check(rc, "wiski_twolevel_refresh")
        # the verdict (pinned word) must be read before N preconditions anything: G was accumulated in fp32 here, and a slightly
        # indefinite G (long or heavy streams) or a non-finite stencil fails the one-workgroup Cholesky, which poisons N with NaN by
        # design.  The solve this block was rebuilt for polls the host every iteration anyway: waiting for the ~1.5 ms refresh
        # shifts nothing
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        done.synchronize()
        if int(self._bad_host[0]):
            self.failed = True
            self.active = -1
            return False
        self.active = 0
        self.struct.d_N = self.N[0].data_ptr()
        self.in_flight = None
        self.weight_at_launch = weight
        self.refreshes += 1
        self.rebuilt = True
        return True

    def tick(self, step, lag, lockstep=False):
        """Switch a finished refresh in.  lockstep (replicas that must take identical iterations): exactly `lag` steps after it was
        launched, waiting for it if need be.  Otherwise: as soon as it is found complete (no wait), at the latest 4 `lag` steps on.
        True if N changed."""
        fl = self.in_flight
        if fl is None:
            return False
        done, at, tgt = fl
        if lockstep:
            if step < at + lag:
                return False
            done.synchronize()
        elif not done.query():
            if step < at + 4 * max(lag, 1):
                return False
            done.synchronize()
        self.in_flight = None
        if int(self._bad_host[0]):
            self.failed = True                  # G is corrupt (non-finite points reached it): the tracker drops the block
            return False
        self.active = tgt
        self.struct.d_N = self.N[tgt].data_ptr()
        return True

    def finish(self):
        """Wait for a refresh in flight and switch it in (tests, teardown)."""
        if self.in_flight is not None:
            self.tick(self.in_flight[1] + 10 ** 9, 0, lockstep=True)


class TwoLevelTracker:
    """Model-side bookkeeping: every point the statistics have absorbed either sits in `pending` or is part of the block's G."""

    def __init__(self):
        self.pending = []          # [(X [n, d], wa [n] or None)]
        self.pending_n = 0
        self.covered = True        # False once points were absorbed behind our back: no block until the statistics are rebuilt
        self.block = None
        self.block_key = None
        self.step = 0
        self.switched = False      # the last for_step() switched a new block in
        self.seen_iters = []       # CG iterations of the first warm steps under the separable model alone (the gate below)
        self.last_q = 0
        self.wanted = False        # the gate has opened once for this stream (survives lose(): see rebuild())
        self.rebuilds = 0

    def reset(self):
        self.__init__()

    def note(self, X, wa):
        if not self.covered:
            return
        self.pending.append((X, wa))
        self.pending_n += X.shape[0]
        if self.block is None and self.pending_n > 4_000_000:       # nobody is streaming through the fast path: stop hoarding
            self.lose()

    def lose(self):
        if self.block is not None and not self.block.failed:
            self._spare = self.block                   # rebuild() may put it on the next eigenbasis (TwoLevelBlock.rebase)
        self.pending, self.pending_n, self.covered, self.block, self.block_key = [], 0, False, None, None

    def rebuild(self, grid, device, pst, kscale, stencil, weight, err):
        """A block for the eigenbasis of `pst` straight from the statistics (TwoLevelBlock.rebuild_from_stencil): for streams the gate
        has opened for, when the block was lost to a hyper-parameter step / a density-profile re-solve / points that bypassed the
        tracker.  Every absorbed point is in the stencil, so the tracker is whole again afterwards.  Returns the struct, or None."""
        if not self.wanted or pst is None or "eig_host" not in pst or settings.two_level_rebuild.off():
            return None
        blk = getattr(self, "_spare", None)
        self._spare = None
        if (blk is not None and blk.grid is grid and blk.r == min(settings.two_level_rank.value(), MAXR) and getattr(blk, "rebases", 0) % 32 != 31
                and (blk.in_flight is None or blk.in_flight[0].query())):
            blk.rebase(pst["eig_host"], kscale)        # (every 32nd time the modes are selected afresh)
        else:
            blk = TwoLevelBlock(grid, device, pst["eig_host"], kscale, settings.two_level_rank.value(), err)
        if not blk.rebuild_from_stencil(stencil, weight):
            self.rebuild_failures = getattr(self, "rebuild_failures", 0) + 1
            return None                                # the solve runs without the block (two_level = None)
        self.block, self.block_key, self._eig_ref = blk, (id(pst["eig"][0]), float(kscale), settings.two_level_rank.value()), pst["eig"]
        self.pending, self.pending_n, self.covered = [], 0, True
        self.rebuilds += 1
        return blk.struct

    def current(self, pst, kscale, cols=1):
        """The block as it stands, for a solve that is not a streaming step (variance / probe / fantasy columns; any number of
        columns up to the block's scratch): its struct if one is active for exactly this eigenbasis, else None.  No side effects --
        the refresh pipeline moves with for_step() only; points still pending cost such a solve iterations, nothing else."""
        blk = self.block
        if (blk is None or not self.covered or blk.failed or blk.active < 0 or pst is None or "eig" not in pst or
                self.block_key != (id(pst["eig"][0]), float(kscale), settings.two_level_rank.value())):
            return None
        blk.ensure_cols(cols)
        return blk.struct

    def for_step(self, grid, device, pst, kscale, weight, err, lockstep=False, last_iters=0, subsample=None):
        """Called once per fast streaming step, after its batch was noted: returns the TwoLevelStruct to solve with (or None)
        and keeps the refresh pipeline going.  `weight`: absorbed weight (sum of wa) including this step's batch; `last_iters`:
        CG iterations of the most recent finished solve."""
        if not self.covered or "eig_host" not in pst:
            return None
        key = (id(pst["eig"][0]), float(kscale), settings.two_level_rank.value())
        if self.block is None or self.block_key != key:
            if self.block is not None:
                # the eigenbasis moved (hyper-parameters or density profile): G lives in the old one and the points are gone
                self.lose()
                return None
            # the gate: the block pays (a projection + Gram product per batch, an r x r factorisation per refresh, a coefficient
            # exchange per CG iteration) only where the separable density model is poor.  The first warm steps run without it;
            # if they converge in fewer than `two_level_min_iters` iterations (uniform streams: 2-3) it is never built.
            if last_iters > 0:
                self.seen_iters.append(last_iters)
            if len(self.seen_iters) < 3:                # [0] is the cold solve on the init data; two warm steps decide
                return None
            if min(self.seen_iters[1:3]) < settings.two_level_min_iters.value():
                if len(self.seen_iters) >= 6:
                    self.lose()                         # decided: not needed for this stream (stops the hoarding of points)
                return None
            self.block = TwoLevelBlock(grid, device, pst["eig_host"], kscale, settings.two_level_rank.value(), err)
            self.wanted = True
            self.block_key = key
            self._eig_ref = pst["eig"]          # keeps the id in the key from being recycled
        blk = self.block
        self.step += 1
        self.switched = blk.tick(self.step, settings.two_level_lag.value(), lockstep)
        if blk.failed:
            self.lose()
            return None
        growth = settings.two_level_growth.value()
        if blk.in_flight is None and self.pending and weight >= growth * blk.weight_at_launch:
            # the block will serve from ~`lag` steps on until the next one arrives (~`lag` steps after the weight has grown by
            # `growth`): aim it at the middle of that span (for a stationary stream G grows with the absorbed weight)
            dq = float(self.pending[-1][0].shape[0])
            lag = float(max(settings.two_level_lag.value(), 1))
            lo, hi = weight + lag * dq, max(growth * weight, weight + dq) + lag * dq
            gscale = 0.5 * (lo + hi) / max(weight, 1.0)
            pts, self.pending, self.pending_n = self.pending, [], 0
            blk.launch_refresh(pts, self.step, weight, gscale, subsample)
        return blk.struct if blk.active >= 0 else None
