"""FixedNoiseOnlineSKIGP -- the WISKI model core, MI355X-native.

Host-side mirror of the reference's
online_gp/models/batched_fixed_noise_online_gp.py (same constructor, methods,
cache keys and output shapes) with the arithmetic moved to the HIP kernels of
libwiski_hip.so:

  reference (dense torch/gpytorch)                    here
  ------------------------------------------------    ---------------------------------------------
  W^T densified m x n            (:22-28)             never formed; taps recomputed from x in-kernel
  W^T D^-1 W dense m x m         (:50-53, URLT:58)    block stencil [7^d, m], atomically scattered
  root L, Q = I + L^T Kuu L, chol(Q)   (:343-383)     preconditioned CG in inducing space (wiski_pcg)
  pred_cov dense m x m           (:385-404)           lazy operator M = (Kt^-1 + A)^-1
  left_interp / W* pred_cov W*^T (:206-228)           fused gather kernels

The posterior it returns is the exact one, mean = w*^T (Kt^-1+A)^-1 b and
cov = sigma2 W* (Kt^-1+A)^-1 W*^T, to the CG tolerance.
"""
import math

import torch

from .. import grid_ops, settings
from ..distributions import MultivariateNormal, ZeroLazyTensor, DenseLazyTensor, LazyCovariance
from ..kernels import GridInterpolationKernel, RBFKernel, ScaleKernel
from ..lazy.operators import (InducingPosterior, InterpolatedKernel, KroneckerToeplitz, PredictiveCovariance, StencilWtW)
from ..lazy.dense_woodbury import DenseInducingPosterior
from ..likelihoods import FNMGLikelihood


class BatchOperator(LazyCovariance):
    """Stack of per-output operators: shape [out, n, n]."""

    def __init__(self, ops):
        self.ops = list(ops)
        self.shape = torch.Size([len(self.ops)]) + self.ops[0].shape
        self.dtype = self.ops[0].dtype
        self.device = self.ops[0].device

    def __getitem__(self, i):
        return self.ops[i]

    def __len__(self):
        return len(self.ops)

    def matmul(self, rhs):
        if rhs.dim() == 2:
            return torch.stack([op.matmul(rhs) for op in self.ops])
        return torch.stack([op.matmul(rhs[i]) for i, op in enumerate(self.ops)])

    __matmul__ = matmul

    def evaluate(self):
        return torch.stack([op.evaluate() for op in self.ops])

    def diag(self):
        return torch.stack([op.diag() for op in self.ops])

    def clone(self):
        return BatchOperator([op.clone() for op in self.ops])

    def to(self, device):
        return BatchOperator([op.to(device) for op in self.ops])


def _wtw_ops(wtw):
    return wtw.ops if isinstance(wtw, BatchOperator) else [wtw]


def _default_tol(dtype):
    v = settings.cg_tolerance.value()
    if v is not None:
        return v
    return 1e-6 if dtype == torch.float32 else 1e-10


class FixedNoiseOnlineSKIGP(torch.nn.Module):
    def __init__(
        self,
        train_inputs=None,
        train_targets=None,
        train_noise_term=None,
        covar_module=None,
        kernel_cache=None,
        grid_bounds=None,
        grid_size=30,
        likelihood=None,
        learn_additional_noise=False,
        num_data=None,
    ):
        super().__init__()
        assert train_inputs is not None or kernel_cache is not None

        if train_targets is not None:
            if train_targets.dim() == 1:
                train_targets = train_targets[:, None]
            num_outputs = train_targets.shape[-1]
            self.num_data = train_inputs.shape[-2]
            device, dtype = train_inputs.device, train_inputs.dtype
            num_dims = train_inputs.shape[-1]
        else:
            ic = kernel_cache["interpolation_cache"]
            num_outputs = ic.shape[0]
            self.num_data = num_data
            device, dtype = ic.device, ic.dtype
            num_dims = None
        self.num_outputs = num_outputs
        _batch_shape = torch.Size([num_outputs]) if num_outputs > 1 else torch.Size()

        if covar_module is None:
            if grid_bounds is None:
                grid_bounds = torch.stack((train_inputs.min(dim=-2)[0] - 0.1, train_inputs.max(dim=-2)[0] + 0.1)).transpose(-1, -2)
            covar_module = ScaleKernel(RBFKernel(batch_shape=_batch_shape, ard_num_dims=train_inputs.size(-1)), batch_shape=_batch_shape)
        if not isinstance(covar_module, GridInterpolationKernel):
            covar_module = GridInterpolationKernel(base_kernel=covar_module, grid_size=grid_size, num_dims=train_inputs.shape[-1],
                                                   grid_bounds=grid_bounds)
        self._batch_shape = _batch_shape
        self.train_inputs = [None]
        self.train_targets = None
        self.covar_module = covar_module.to(device)
        self._grid = self.covar_module.grid_spec
        self._dtype = dtype
        self._device = device
        if num_dims is not None and num_dims != self._grid.d:
            raise RuntimeError(f"inputs have {num_dims} dims but the grid has {self._grid.d}")

        if likelihood is None:
            if train_noise_term is None:
                train_noise_term = torch.ones_like(train_targets)
            train_noise_term = self._canon_noise(train_noise_term, train_targets)
            self.likelihood = FNMGLikelihood(noise=train_noise_term.transpose(-1, -2), learn_additional_noise=learn_additional_noise,
                                             batch_shape=_batch_shape).to(device)
        else:
            self.likelihood = likelihood
            if train_noise_term is not None and train_targets is not None:
                train_noise_term = self._canon_noise(train_noise_term, train_targets)
        self.has_learnable_noise = learn_additional_noise

        self._err = grid_ops.new_err_flag(device)
        # sum_p 1/noise_p per output: host part (unit-noise updates, exact) + device part (explicit noise tensors)
        self._wsum_host = [0.0] * num_outputs
        self._wsum_dev = torch.zeros(num_outputs, dtype=torch.float64, device=device)
        self._wsum_dev_host = [0.0] * num_outputs
        self._wsum_dirty = False
        self._pcg_ws = grid_ops.PCGWorkspace()
        self._memo = {}
        self._mean_state = None  # warm-start state of the posterior-mean solve

        if kernel_cache is None:
            self._kernel_cache = self._fresh_cache()
            self._absorb(self._kernel_cache, train_inputs, train_targets, train_noise_term, init=True)
        else:
            self._kernel_cache = kernel_cache
            facs = kernel_cache.get("_spectral")
            if facs:
                # the spectral factor(s) follow the statistics they describe (bayesopt.py:86-96 re-initialises the model from the
                # previous model's cache at every step: a factor kept on the model object would be rebuilt from the stencil each time)
                for fac in facs.values():
                    fac.err = self._err
                self.__dict__["_spectral"] = facs
            if "_cnt" in kernel_cache:
                # hand-over path (bayesopt.py:86-96): recover sum_p 1/noise_p from the row sums (rows of W sum to one)
                self._wsum_dev = kernel_cache["_cnt"].sum(dim=1, dtype=torch.float64)
                self._wsum_dirty = True

    # ------------------------------------------------------------ helpers --
    @staticmethod
    def _canon_noise(noise, targets):
        """-> [n, out] like the targets."""
        if noise.dim() == 1:
            noise = noise[:, None]
        if noise.shape != targets.shape and noise.transpose(-1, -2).shape == targets.shape:
            noise = noise.transpose(-1, -2)
        return noise.expand_as(targets) if noise.shape != targets.shape else noise

    def _fresh_cache(self):
        out, m = self.num_outputs, self._grid.m
        b = torch.zeros((out, m, 1), dtype=self._dtype, device=self._device)
        stats = torch.zeros((out, 2), dtype=torch.float64, device=self._device)
        # the half stencils of all outputs live in ONE [out, H, m] tensor (each operator holds a view), so that an absorb of
        # several outputs is a single launch (wiski_scatter_stats_multi)
        pack = torch.zeros((out, (self._grid.R + 1) // 2, m), dtype=self._dtype, device=self._device)
        ops = [StencilWtW(self._grid, pack[o]) for o in range(out)]
        cnt = torch.zeros((out, m), dtype=self._dtype, device=self._device)
        return self._pack_cache(b, stats, ops, cnt)

    def _pack_cache(self, b, stats, ops, cnt=None):
        out = b.shape[0]
        if cnt is None:
            cnt = torch.zeros((out, b.shape[1]), dtype=b.dtype, device=b.device)
        return {
            "_cnt": cnt,                                     # W^T D^-1 1 = row sums of W^T D^-1 W (preconditioner density model)
            "response_cache": stats[:, 0].view(out, 1, 1),   # y^T D^-1 y     (:45)   [float64 view]
            "interpolation_cache": b,                        # W^T D^-1 y     (:46)
            "WtW": ops[0] if out == 1 else BatchOperator(ops),  # W^T D^-1 W  (:50-53)
            "D_logdet": stats[:, 1],                         # logdet D       (:55)   [float64 view]
            "_stats": stats,
        }

    def _clone_cache(self, cache):
        stats = cache["_stats"].clone()
        cnt = cache["_cnt"].clone() if "_cnt" in cache else None
        ops = _wtw_ops(cache["WtW"])
        pack = self._stencil_pack(ops)
        if pack is not None and len(ops) > 1:           # keep the outputs' stencils in one tensor (see _fresh_cache)
            cp = pack.clone()
            cl = lambda t: None if t is None else t.clone()
            new_ops = [StencilWtW(self._grid, cp[o], cl(op.root), cl(op.inv_root)) for o, op in enumerate(ops)]
        else:
            new_ops = [op.clone() for op in ops]
        new = self._pack_cache(cache["interpolation_cache"].clone(), stats, new_ops, cnt)
        facs = cache.get("_spectral")
        if facs:
            new["_spectral"] = {o: fac.clone() for o, fac in facs.items() if fac.ref is not None and not fac.stale}
        return new

    def _stencil_pack(self, ops):
        """The [out, H, m] tensor the outputs' half stencils are views of, if they (still) are: consecutive, same shape."""
        st = [op.stencil for op in ops]
        if not all(grid_ops.is_half_stencil(self._grid, t) and t.is_contiguous() for t in st):
            return None
        H, m = st[0].shape
        es = st[0].element_size()
        base = st[0].data_ptr()
        if any(t.shape != (H, m) or t.data_ptr() != base + o * H * m * es for o, t in enumerate(st)):
            return None
        if len(st) == 1:
            return st[0][None]
        root = st[0]._base if st[0]._base is not None else None
        if root is None or root.dim() != 3 or root.shape != (len(st), H, m) or root.data_ptr() != base:
            return None
        return root

    def _half_buffers(self):
        """Per-output symmetric half-stencil delta buffers [(R+1)/2, m] for the data-parallel
        path (rank-local increment -> all-reduce -> add; zero between uses)."""
        if getattr(self, "_half_delta", None) is None:
            H = (self._grid.R + 1) // 2
            self._half_delta = [torch.zeros((H, self._grid.m), dtype=self._dtype, device=self._device) for _ in range(self.num_outputs)]
        return self._half_delta

    def _absorb(self, cache, X, Y, noise, init, half_delta=None, res_delta=None):
        """_initialize_caches (:31-60) / _update_cache_dicts (:155-171) fused into
        one scatter launch per output; mutates `cache` in place.

        W^T D^-1 W accumulates straight into the symmetric half stencil (T(T+1)/2 atomics
        per point).  With `half_delta` given (data-parallel path) the increments go to those
        buffers instead, for the caller to all-reduce and add; `res_delta` ([out, m], zeroed) then receives this shard's
        innovation W^T (wb y - wa (W U)) of the carried residual, to be all-reduced and added to R alongside."""
        self._finish_pending()
        if cache is self._kernel_cache:
            self.leave_stencil_shard()               # the generic absorb writes every group
        X = X.reshape(-1, self._grid.d).to(self._device, self._dtype).contiguous()
        Y = Y.to(self._device, self._dtype)
        if Y.dim() == 1:
            Y = Y[:, None]
        n = X.shape[0]
        unit = noise is None            # unit noise (OnlineSKIRegression, OSR:25,122): no per-point weight tensors at all
        if unit:
            if getattr(self, "_ones_cache", None) is None or self._ones_cache.shape[0] < n:
                self._ones_cache = torch.ones(max(n, 4096), dtype=self._dtype, device=self._device)
            ones = self._ones_cache[:n]
        else:
            noise = noise.to(self._device, self._dtype)
        b = cache["interpolation_cache"]
        stats = cache["_stats"]
        ops = _wtw_ops(cache["WtW"])
        dst = half_delta if half_delta is not None else [op.stencil for op in ops]
        # residual carry-over: while the posterior-mean state (U, Z, R = b - Z - A U) is current, the scatter
        # keeps R exact under the increment, and the next refresh starts without an A U product
        ms = self._mean_state
        mine = cache is self._kernel_cache
        if (mine or half_delta is not None) and self.num_outputs == 1:
            if half_delta is not None:
                # the increment arrives by all-reduce: other ranks' points never pass here.  The updater hands the gathered coordinates over
                # (_stats_points).  Failing that (callers with only an ncclComm_t, nothing to gather through) the block is given up: noting
                # this rank's own shard instead would leave every replica with a DIFFERENT preconditioner, hence different CG iterates and
                # iteration counts -- and the collective decisions of the path (carried residual, poll hints) assume replicated state
                pts = self.__dict__.pop("_stats_points", None)
                if pts is not None and not init:
                    self._two_level_note(pts[0], pts[1])            # every rank's points, gathered beside the all-reduce (distributed.py)
                else:
                    self._two_level_lose()
            else:
                self._two_level_note(X, None if unit else (1.0 / noise[:, 0] if init else 1.0 / noise[:, 0].clamp_min(1e-7)), init=init)
        carry = (mine and half_delta is None and not init and ms is not None and ms.get("R_ok", False)
                 and settings.residual_carry_over.on())
        carry_delta = (half_delta is not None and res_delta is not None and not init and ms is not None and ms.get("R_ok", False)
                       and settings.residual_carry_over.on())
        if mine and ms is not None and not carry:
            ms["R_ok"] = False
        if getattr(self, "_scratch_stats", None) is None:
            self._scratch_stats = torch.zeros(2, dtype=torch.float64, device=self._device)
        if self._absorb_all_outputs(cache, X, Y, noise, unit, init, half_delta, ops, dst, carry, ms, mine, n):
            return carry_delta
        for o in range(self.num_outputs):
            yo = Y[:, o].contiguous()
            if unit:
                no = wa = wb = ones
            else:
                no = noise[:, o].contiguous()
                wb = 1.0 / no
                wa = wb if init else 1.0 / no.clamp_min(1e-7)   # clamp_min(1e-7)**0.5 of :163, squared
            cnt_o = cache["_cnt"][o] if "_cnt" in cache else None     # row sums W^T wa ride on the same launch
            half = grid_ops.is_half_stencil(self._grid, dst[o])      # a handed-over cache may carry a full stencil
            if carry and not half:
                carry = ms["R_ok"] = False
            if carry_delta and not half:
                carry_delta = False
            grid_ops.scatter_stats_cnt(self._grid, X, yo, wa, wb, no, b[o, :, 0], dst[o], half, cnt_o, stats[o], self._err,
                                       u=ms["U"][o] if (carry or carry_delta) else None,
                                       res=ms["R"][o] if carry else (res_delta[o] if carry_delta else None))
            if (init or half_delta is not None) and getattr(ops[o], "root", None) is not None:
                # the stencil is being rebuilt (set_train_data) or receives its increment later, after an all-reduce
                # (data-parallel path): a carried root pair would describe the OLD matrix -- drop it, it is re-derived on demand
                ops[o].root = ops[o].inv_root = None
            if half_delta is None and getattr(ops[o], "root", None) is not None and n > 0:
                # the reference's root pair, once somebody asked for it: L L^T follows A by a rank-n root update (URLT:62-119)
                Wd = grid_ops.wt_columns(self._grid, X, self._err)                 # [n, m]
                ops[o].update_roots_((Wd * wa.sqrt()[:, None]).t().contiguous())   # V = W^T diag(wa)^(1/2), BFN:163-168
            if mine and n > 0:
                self._spectral_absorb(o, X, None if unit else wa, yo if unit else yo * wb, init=init, bypass=half_delta is not None)
            if cache is self._kernel_cache or init:
                if unit:
                    self._wsum_host[o] += float(n)
                else:
                    self._wsum_dev[o] += wa.sum(dtype=torch.float64)
                    self._wsum_dirty = True
        return carry_delta

    def _absorb_all_outputs(self, cache, X, Y, noise, unit, init, half_delta, ops, dst, carry, ms, mine, n):
        """Several outputs, native packed half stencils, no root pairs to carry: ONE scatter launch for all of them
        (wiski_scatter_stats_multi) instead of one per output.  False: not applicable, the caller loops."""
        out = self.num_outputs
        if out == 1 or half_delta is not None or n == 0 or "_cnt" not in cache or any(getattr(op, "root", None) is not None for op in ops):
            return False
        pack = self._stencil_pack(ops)
        if pack is None:
            return False
        Yt = Y.t().contiguous()                                   # [out, n]
        if unit:
            wa = wb = no = self._ones_cache[:n]
        else:
            no = noise.t().contiguous()
            wb = 1.0 / no
            wa = wb if init else 1.0 / no.clamp_min(1e-7)        # clamp_min(1e-7)**0.5 of :163, squared
        b = cache["interpolation_cache"][:, :, 0]
        if not b.is_contiguous():
            return False
        grid_ops.scatter_stats_multi(self._grid, X, Yt, wa, wb, no, b, pack, cache["_cnt"], cache["_stats"], self._err,
                                     u=ms["U"] if carry else None, res=ms["R"] if carry else None)
        for o in range(out):
            if mine:
                self._spectral_absorb(o, X, None if unit else wa[o], Yt[o] if unit else Yt[o] * wb[o], init=init)
            if mine or init:
                if unit:
                    self._wsum_host[o] += float(n)
                else:
                    self._wsum_dev[o] += wa[o].sum(dtype=torch.float64)
                    self._wsum_dirty = True
        return True

    # (the return value of _absorb tells the data-parallel caller whether res_delta was filled)
    @property
    def _wsum(self):
        if self._wsum_dirty:
            self._wsum_dev_host = self._wsum_dev.tolist()
            self._wsum_dirty = False
        return [h + d for h, d in zip(self._wsum_host, self._wsum_dev_host)]

    def check_bounds(self):
        """Raise like gpytorch's grid check if any point seen so far was outside the
        grid (the kernels only set a device flag; this is the one host sync; the
        device part of the noise-weight sum rides on the same transfer)."""
        self._finish_pending()
        if self._wsum_dirty:
            vals = torch.cat([self._err.double(), self._wsum_dev]).tolist()
            self._wsum_dev_host, self._wsum_dirty = vals[1:], False
            flag = int(vals[0])
        else:
            flag = int(self._err.item())
        if flag:
            self._raise_out_of_bounds(flag)

    def _raise_out_of_bounds(self, flag):
        """`flag` is the raw device word: bit 0 = some point (query or training) was outside the grid, bits 1.. = number
        of *training* points the scatter dropped.  Dropped points contributed nothing to A, b, y^T D^-1 y or log|D|
        (scatter_stats.hip), so taking them out of `num_data` and of the noise-weight sum leaves statistics that
        describe exactly the points that were absorbed -- a caller may catch the error and carry on."""
        dropped = flag >> 1
        self._err.zero_()
        self.__dict__.pop("_stream_step_cache", None)
        self._drop_spectral()          # rows of out-of-grid points were zero for the factor too, but a prepared state may be half-updated
        if dropped:
            self.num_data = self.num_data - dropped
            cnt = self._kernel_cache.get("_cnt")
            if cnt is not None:                       # row sums of W^T D^-1 W: sum_i cnt_i = sum over absorbed points of 1/noise
                self._wsum_dev = cnt.sum(dim=1, dtype=torch.float64)
                self._wsum_host = [0.0] * self.num_outputs
                self._wsum_dirty = True
            self._dump_caches()
        raise RuntimeError("Received data that was out of bounds for the specified grid. "
                           f"Grid bounds were {self.covar_module.grid_bounds}.")

    def _sigma2(self, o=0):
        if not self.has_learnable_noise:
            return 1.0
        n = self.likelihood.second_noise_covar.noise.detach().reshape(-1)
        return float(n[o] if n.numel() > 1 else n[0])

    def _hyper_version(self):
        ps = self.__dict__.get("_hyper_params")
        if ps is None:
            ps = list(self.covar_module.parameters()) + (list(self.likelihood.second_noise_covar.parameters()) if self.has_learnable_noise else [])
            self.__dict__["_hyper_params"] = ps
        return (self.__dict__.get("_hyper_epoch", 0),) + tuple(p._version for p in ps)

    def _hyper(self):
        """Per-output (tcol on device in the data dtype, sigma2 float); memoised on
        the parameters' version counters."""
        ver = self._hyper_version()
        h = self._memo.get("hyper")
        if h is None or h[0] != ver:
            vals = []
            with torch.no_grad():
                for o in range(self.num_outputs):
                    bi = o if self.num_outputs > 1 else None
                    tcol64 = self.covar_module.toeplitz_columns(batch_index=bi, device=self._device).contiguous()
                    vals.append((tcol64.to(self._dtype).contiguous(), self._sigma2(o), tcol64))
            h = (ver, vals)
            self._memo["hyper"] = h
        return h[1]

    def _use_dense(self):
        return settings.dense_small_grids.on() and self._grid.m <= settings.max_cholesky_size.value()

    def _spectral_allowed(self):
        """May a request go to the spectral factor (lazy/spectral_woodbury.py)?  Beyond the dense regime always; inside it only for
        the reference's per-batch loop, whose owner (a streaming wrapper) has said so -- direct users of the model, BO posteriors
        and fantasies keep the nodal dense factor with its cached M and rank-q updates."""
        if settings.spectral_factor.off():
            return False
        if not self._use_dense():
            return True
        return settings.spectral_dense_regime.on() and bool(self.__dict__.get("_stream_owner"))

    def _precond(self, o, tcol):
        """(eigen tuple, shift) of wiski_pcg's preconditioner (Kt^-1 + a kron_q diag(t_q))^-1.
        t_q = per-dim marginal of the row sums of W^T D^-1 W (the data-density profile: the
        grid nodes outside the data box carry no data), a = total mass / prod_q sum(t_q).
        The d small generalized eigenproblems are re-solved when the hyper-parameters change,
        or when the normalised density profile has moved by more than
        settings.precond_profile_drift (a stationary stream keeps its eigenbasis).  The profile is
        looked at (3 small reductions + one host read, ~0.15 ms) each time the data volume has
        doubled, or as soon as a warm refresh needs 2 more CG iterations than the first one after
        the last re-solve did -- the symptom of a stale basis; `a` follows the stream exactly."""
        if settings.spectral_preconditioner.off():
            return None, 0.0
        # (the eigenbasis must belong to the CURRENT hyper-parameters exactly: the fused solver kernels take the u = Kt z image of every
        # search direction from it, so a basis kept across even a 1 % lengthscale step changes the converged mean at the 1e-4 level --
        # tried and reverted in round 3, tests/test_model_gpu.py::test_preconditioner_eigenbasis_is_resolved_for_every_hyperparameter_change)
        ver = self._hyper_version()
        st = self._memo.setdefault("precond", {}).get(o)
        wsum = float(self._wsum[o])
        stale = st is None or st["ver"] != ver
        its = (getattr(self, "_last_iters", None) or [0] * (o + 1))[o]
        if not stale:
            if st.get("it0") is None and its > 0:
                st["it0"] = its                          # iteration level of this basis when it was fresh
            slow = st.get("it0") is not None and its >= st["it0"] + 2 and wsum > 1.1 * st["wsum"]
        if stale or wsum > 2.0 * st["wsum"] or wsum < 0.5 * st["wsum"] or slow:
            profiles, norm = None, float(self._grid.m)
            cnt = self._kernel_cache.get("_cnt") if settings.density_profile_preconditioner.on() else None
            if cnt is not None and wsum > 0:
                c3 = cnt[o].reshape(self._grid.g).double()
                margs = [c3.sum(dim=[r for r in range(self._grid.d) if r != q]) if self._grid.d > 1 else c3 for q in range(self._grid.d)]
                margs = torch.stack([torch.nn.functional.pad(mg, (0, max(self._grid.g) - mg.numel())) for mg in margs]).cpu().numpy()
                if margs.max() > 0:
                    profiles, norm = [], 1.0
                    for q, gq in enumerate(self._grid.g):
                        t = margs[q, :gq] / margs[q, :gq].max()
                        t = t.clip(1e-2, None)
                        profiles.append(t)
                        norm *= float(t.sum())
            old = None if stale else st.get("profiles")
            if (old is not None and profiles is not None and
                    max(float(abs(a - b).max()) for a, b in zip(profiles, old)) <= settings.precond_profile_drift.value()):
                st["wsum"] = wsum                      # same density shape: keep the eigenbasis, only the scale moves
                st["it0"] = None
            else:
                host = {}
                eig = grid_ops.kron_eigen(self._grid, tcol, profiles=profiles, host_out=host)
                st = {"ver": ver, "wsum": wsum, "eig": eig, "norm": norm, "profiles": profiles, "it0": None, "eig_host": host}
                self._memo["precond"][o] = st
        return st["eig"], wsum / st["norm"]

    def _posterior_op(self, o):
        self.leave_stencil_shard()                   # solves outside the sharded streaming step need the whole stencil
        tcol, s2, _ = self._hyper()[o]
        if self._use_dense():
            return DenseInducingPosterior(self._grid, _wtw_ops(self._kernel_cache["WtW"])[o], tcol, 1.0 / s2, grid_ops.kron_eigen(self._grid, tcol))
        eig, shift = self._precond(o, tcol)
        post = InducingPosterior(self._grid, _wtw_ops(self._kernel_cache["WtW"])[o], tcol, 1.0 / s2, _default_tol(self._dtype),
                                 settings.max_cg_iterations.value(), workspace=self._pcg_ws, check_every=settings.cg_check_every.value(),
                                 eigen=eig, shift=shift, err=self._err)
        # the exact block of the two-level preconditioner, where the stream keeps one for this eigenbasis: every solve through this
        # operator that names no block of its own asks for it when it runs (the operator outlives many streaming steps)
        if o == 0:
            import weakref

            ref = weakref.ref(self)
            post.two_level_provider = lambda op, k: (ref()._two_level_for_solve(op, k) if ref() is not None else None)
        return post

    def _two_level_for_solve(self, post, k):
        """The stream's two-level block for a solve of k columns through `post` (variances, probes, fantasies: 15 -> 4-5 iterations
        per 64-column solve on the road-like stream): the tracker's block if it belongs to the operator's
        eigenbasis; where it was lost (hyper-parameter step, profile re-solve, points behind the tracker's back) and the solve is
        wide, a block rebuilt from the statistics (settings.two_level_rebuild)."""
        tr = self.__dict__.get("_two_level") if self._two_level_applies() else None
        if tr is None:
            return None
        pst = self._memo.get("precond", {}).get(0)
        if pst is None or pst.get("eig") is not post.eigen:
            return None
        tl = tr.current(pst, post.kscale, cols=k)
        warming = tr.block is not None and tr.covered and not tr.block.failed and tr.block.active < 0 and tr.block.in_flight is not None
        if tl is None and k >= 16 and tr.wanted and not warming:   # (a block whose first refresh is in flight is not thrown away for a ~1.5 ms rebuild; 8 or 1 columns: the q = 1 reference step on the PCG path 5.0 -> 5.2 / 6.4 ms: the rebuild costs more than narrow solves save)
            tl = tr.rebuild(self._grid, self._device, pst, post.kscale, post.wtw.stencil, float(self._wsum[0]), self._err)
            if tl is not None:
                tr.block.ensure_cols(k)
                self._poll_hint_sticky = 2                   # a new block: the next warm steps poll after 2 iterations (as _two_level_step)
        return tl

    # ------------------------------------------------------- stencil shard --
    def enter_stencil_shard(self, rank, world, allreduce, allreduce_full=None, comm=None):
        """Multi-GPU step that divides the work (include/wiski.h: wiski_shard): from now on this replica keeps
        only ITS groups of the half stencil current -- the streaming step scatters 1 / world of the tap pairs per point and
        computes 1 / world of A p, one all-reduce of an m-vector per CG iteration makes the product whole.  Every rank must
        see every point (the caller all-gathers the shards) and must make the same calls.  `allreduce(vec, dots)`: in-place
        SUM over the ranks of the two tensors (dots may be None); `allreduce_full(t)`: the same for one large tensor, used
        when a consumer needs the whole stencil again (leave_stencil_shard).  `comm`: an ncclComm_t (wiski_comm_*) -- the per-product
        all-reduce is then issued from C on the solve's stream (one grouped RCCL launch, no re-entry into Python); a single rank
        with a communicator owns every group and still takes that path (both precisions).  Returns False
        where the sharded step does not apply (then nothing changes): one output, native half stencil, m % 4 == 0, a grid beyond the
        dense regime.  Any d and both precisions: the d = 3 fp32 products run on the LDS-DMA kernel's part table, all others on
        the LDS-window kernel restricted to the replica's group range."""
        op = _wtw_ops(self._kernel_cache["WtW"])[0]
        if (world <= 1 and not comm) or self.num_outputs != 1 or self._use_dense() or not op.is_half or self._grid.m % 4 or op.root is not None:
            return False
        if self.__dict__.get("_stencil_shard") is not None:
            return True
        self._finish_pending()
        lo, hi = grid_ops.shard_groups(self._grid.d, rank, world)
        ng = (self._grid.R // 7 + 1) // 2
        flat = op.stencil.reshape(-1)
        for a, b in grid_ops.half_stencil_group_slices(self._grid, 0, lo) + grid_ops.half_stencil_group_slices(self._grid, hi, ng):
            flat[a:b].zero_()
        self.__dict__["_stencil_shard"] = {"rank": rank, "world": world, "allreduce": allreduce, "allreduce_full": allreduce_full or (lambda t: allreduce(t, None)),
                                           "comm": comm}
        self.__dict__.pop("_stream_step_cache", None)
        self._drop_spectral()
        return True

    def leave_stencil_shard(self):
        """Collective: sum the disjoint stencil shards back into a full replica on every rank (one all-reduce of the half
        stencil).  Called automatically by every consumer that reads the stencil outside the sharded streaming step."""
        sh = self.__dict__.get("_stencil_shard")
        if sh is None:
            return
        self._finish_pending()
        self.__dict__["_stencil_shard"] = None
        self.__dict__.pop("_stream_step_cache", None)
        sh["allreduce_full"](_wtw_ops(self._kernel_cache["WtW"])[0].stencil)

    # ------------------------------------------------------ spectral factor --
    def _spectral_state(self, o=0):
        """(factor, state, fp64 Toeplitz columns on the device) of the reduced-eigenbasis Woodbury factor
        (lazy/spectral_woodbury.py) for the current hyper-parameters and statistics, or None where it does not apply:
        dense regime, switched off, or a prior whose numerical rank exceeds settings.spectral_max_rank."""
        if not self._spectral_allowed():
            return None
        from ..lazy import spectral_woodbury as sw

        self._finish_pending()
        self.leave_stencil_shard()
        ver = self._hyper_version()
        key = (ver, sw.default_tail(self._dtype), settings.spectral_max_rank.value(), settings.fast_pred_var.on(),
               settings.max_root_decomposition_size.value())
        memo = self._memo.setdefault("spectral", {})
        ent = memo.get(o)
        if ent is None or ent[0] != key:
            tcol64 = self._hyper()[o][2]                 # fp64 Toeplitz columns, computed once per hyper-parameter version
            ent = (key, tcol64)
            memo[o] = ent
        if ent[1] is None:
            return None                                  # not applicable at these hyper-parameters (remembered)
        _, tcol64 = ent
        facs = self.__dict__.setdefault("_spectral", {})
        self._kernel_cache["_spectral"] = facs           # (travels with the statistics: kernel_cache hand-over, functional conditioning)
        fac = facs.get(o)
        if fac is None:
            fac = facs[o] = sw.SpectralWoodburyFactor(self._grid, self._dtype, self._device, self._err)
        kscale = 1.0 / self._hyper()[o][1]
        if fac.stale:
            fac.stale = False
            # statistics changed behind the factor's back: rebuild from the stencil (and forget the state derived from the old ones)
            fac.ref = fac.cur = None
            fac.data_version += 1
        # the columns go in as a device tensor: after a hyper-parameter step the factor refreshes its eigenvectors on the
        # device (no host copy, no synchronisation) unless it has to re-select its index set
        st = fac.state(key, tcol64, kscale)
        if st is None:
            memo[o] = (key, None)
            return None
        if st.get("need_reference"):
            tc_host = tcol64.detach().cpu().numpy()
            # reference basis with a margin, so that the eigenbasis may drift with the hyper-parameters before the
            # reference has to be rebuilt; if the margin does not fit the rank cap, the basis itself
            refb = None if settings.fast_pred_var.on() else sw.select_basis(self._grid, tc_host, st["tail"] * 1e-2, 2 * settings.spectral_max_rank.value(),
                                                                            self._device)
            op = _wtw_ops(self._kernel_cache["WtW"])[o]
            fac.build_reference(refb if refb is not None else st["basis"], op.stencil, self._kernel_cache["interpolation_cache"][o, :, 0])
            st = fac.state(key, tcol64, kscale)
            if st is None or st.get("need_reference"):
                memo[o] = (key, None)
                return None
        return fac, st, tcol64

    def _measure_factor_means(self, sps):
        """Factors whose mean monitor can no longer decide by its bound (SpectralWoodburyFactor.measure_due): measure the factor's mean
        against the PCG mean at the current hyper-parameters on the factor's probe set (one PCG solve, rare: every few hundred steps at
        most), which either keeps the factor serving the mean or switches it off."""
        due = [o for o, sp in enumerate(sps) if sp[0].measure_due]
        if not due:
            return
        pc = self.prediction_cache
        for o in due:
            fac, st, tc = sps[o]
            fac.measure_mean(st, tc, lambda pts, o=o: grid_ops.gather(self._grid, pts, pc["pred_mean"][..., 0], self._err)[:, o])

    def _spectral_in_use(self):
        """Every output has a spectral factor that follows the stream and has been asked for a state recently."""
        if not self._spectral_allowed():
            return False
        facs = self.__dict__.get("_spectral", {})
        return all((f := facs.get(o)) is not None and f.ref is not None and f.cur is not None and not f.stale and f.idle_absorbs < 8
                   for o in range(self.num_outputs))

    def _spectral_absorb(self, o, X, wa, wby, init=False, bypass=False):
        """Keep the spectral factor of output o (if one exists) in step with the statistics."""
        fac = self.__dict__.get("_spectral", {}).get(o)
        if fac is None or fac.ref is None:
            return
        if init or bypass or X.shape[0] > 2048 or fac.idle_absorbs >= 8:
            # rebuilt from scratch / changed by an all-reduce / a batch large enough that re-projecting the stencil on
            # demand (r SpMV columns) is cheaper than following it / nobody has asked the factor anything for 8 batches (a
            # streaming loop that only wants means must not pay a projection + GEMM per step): mark, rebuild when next asked
            fac.stale = True                             # (on the factor, not the model: it travels with the kernel cache)
            return
        fac.absorb(X, wa, wby)

    def _drop_spectral(self):
        self.__dict__.pop("_spectral", None)
        if self._kernel_cache is not None:
            self._kernel_cache.pop("_spectral", None)
        self._memo.pop("spectral", None)

    # --------------------------------------------------------------- caches --
    @property
    def Kuu(self):
        """Lazy Kuu (/ sigma2 when the second noise is learnable), :334-341."""
        ops = [KroneckerToeplitz(self._grid, tcol, 1.0 / s2) for tcol, s2, _ in self._hyper()]
        return ops[0] if self.num_outputs == 1 else BatchOperator(ops)

    @property
    def Kuu_response(self):
        """Kuu @ W^T D^-1 y, :363-366; [out, m, 1]."""
        b = self._kernel_cache["interpolation_cache"]
        outs = [grid_ops.kron_toeplitz_mm(self._grid, tcol, b[o, :, 0], 1.0 / s2) for o, (tcol, s2, _) in enumerate(self._hyper())]
        return torch.stack(outs)[..., None]

    @property
    def prediction_cache(self):
        """pred_mean = (Kt^-1 + A)^-1 W^T D^-1 y  [out, m, 1]   (:368-383), and the
        lazy pred_cov operator(s).  The mean solve is warm-started from the
        previous solution (U, Z) after every streaming update."""
        self._finish_pending()
        self.leave_stencil_shard()                   # whoever asks for the cache may go on to solve with the whole stencil
        self._apply_pending_rank_update()
        pc = self._memo.get("prediction_cache")
        if pc is not None:
            return pc
        if self._use_dense() or self._wsum_dirty:
            self.check_bounds()          # dense path has no solver poll to ride on; a dirty weight sum needs the read anyway
        out, m = self.num_outputs, self._grid.m
        b = self._kernel_cache["interpolation_cache"]
        hyper = self._hyper()
        ver = self._hyper_version()
        ms = self._mean_state
        if ms is not None and ms["U"].shape == (out, m):
            U, Z, R = ms["U"], ms["Z"], ms["R"]       # refreshed in place by the warm-started solves
        else:
            U = torch.empty((out, m), dtype=self._dtype, device=self._device)
            Z = torch.empty_like(U)
            R = torch.empty_like(U)
            ms = None
        iters = []
        posts = []
        converged = True
        for o in range(out):
            post = self._posterior_op(o)
            if isinstance(post, DenseInducingPosterior):
                Uo, _ = post.solve_columns(b[o, :, 0][None])
                U[o] = Uo[0]
                Z[o].zero_()
                iters.append(0)
                posts.append(post)
                continue
            warm = ms is not None
            Uo, Zo, Ro = U[o:o + 1], Z[o:o + 1], R[o:o + 1]   # contiguous row views: wiski_pcg updates them in place
            carried = warm and ms.get("R_ok", False)
            if warm and ms["ver"] != ver:
                tcol, s2, _ = hyper[o]
                Uo.copy_(grid_ops.kron_toeplitz_mm(self._grid, tcol, Zo, 1.0 / s2))   # keep U = Kt Z under the new hypers
                carried = False
            # the exact block of the two-level preconditioner, where one is being tracked (the streaming fast path takes it through
            # wiski_stream_step; this is the generic refresh -- e.g. after a statistics all-reduce, the north-star exchange)
            tl = None
            tr = self.__dict__.get("_two_level") if (out == 1 and warm and self._two_level_applies()) else None
            pst = self._memo.get("precond", {}).get(o) if tr is not None else None
            if pst is not None and "eig_host" in pst and pst.get("eig") is post.eigen:
                tl = tr.for_step(self._grid, self._device, pst, post.kscale, float(self._wsum[0]), self._err,
                                 lockstep=settings.two_level_lockstep.on(), last_iters=(getattr(self, "_last_iters", None) or [0])[0])
                if tr.switched:
                    self._poll_hint_sticky = 2               # a new block: poll after 2 iterations, then after every one (as _two_level_step)
            # warm refreshes poll convergence first where the previous one converged (streaming steps are
            # alike), every 8th one an iteration earlier, and then after every iteration
            fc, probe = 0, False
            if warm and getattr(self, "_last_iters", None):
                self._refresh_count = getattr(self, "_refresh_count", 0) + 1
                fc, probe = self._first_poll(self._last_iters[o])
                post.check_every = 1
            # warm = 2: R was kept equal to b - Z - A U by the scatter launches since the last solve (recomputed
            # from scratch every 16th refresh so that fp rounding of the recursion cannot accumulate)
            if carried and getattr(self, "_refresh_count", 0) % 16 == 0:
                carried = False
            post.solve_columns(b[o, :, 0][None], U=Uo, Z=Zo, warm=2 if carried else warm, first_check=fc, inplace=True, R=Ro, two_level=tl)
            if fc:
                self._note_poll(post.last_iters, fc, probe)
            self._last_rel = None                    # this path does not keep the converged residual: timer-paced probes only
            iters.append(post.last_iters)
            posts.append(post)
            if post.last_err:            # out-of-grid flag delivered with the convergence poll (no extra sync)
                self._mean_state = None if ms is None else dict(ms, R_ok=False)
                self._raise_out_of_bounds(post.last_err)
            converged = converged and getattr(post, "last_converged", True)
        # a solve that stopped at max_cg_iterations (warned about in grid_ops.pcg) leaves a residual that is not small:
        # do not carry it into the next refresh as if it were
        self._mean_state = None if self._use_dense() else {"U": U, "Z": Z, "R": R, "R_ok": converged, "ver": ver}
        self._last_iters = list(iters)
        self._poll_hint = 0 if ms is not None else 2      # after a cold solve: see _first_poll
        pc = {"pred_mean": U[..., None], "pred_cov": posts[0] if out == 1 else BatchOperator(posts), "cg_iters": iters, "ver": ver}
        self._memo["prediction_cache"] = pc
        return pc

    def _make_predictive_covar(self, *args, **kwargs):
        return self.prediction_cache["pred_cov"]

    def _root_space_unavailable(self, name):
        raise NotImplementedError(
            f"{name} is a root-space quantity of the reference's dense formulation (L, Q = I + L^T Kuu L); the matrix-free "
            "formulation has no root.  Use prediction_cache / Kuu / Kuu_response, or the dense path for small grids.")

    def _root_space(self):
        """Reference root-space objects (BFN:343-366) for small grids: L = chol(A + jitter),
        Kt L, Q = I + L^T Kt L, L^T Kt b -- dense, on the MFMA GEMM / Cholesky kernels."""
        if not self._use_dense():
            self._root_space_unavailable("root-space quantities")
        rs = self._memo.get("root_space")
        if rs is None:
            Ls, KLs, Qs, projs = [], [], [], []
            b = self._kernel_cache["interpolation_cache"]
            for o, (tcol, s2, _) in enumerate(self._hyper()):
                # L of the reference's UpdatedRootLazyTensor: chol(A + jitter) when first asked for, afterwards carried
                # through every streaming update by the rank-q root update (so it stops being triangular: BFN:343-366 only
                # ever use L L^T = A, which holds)
                L = _wtw_ops(self._kernel_cache["WtW"])[o].root_decomposition().root.evaluate()
                KL = grid_ops.kron_toeplitz_mm(self._grid, tcol, L.t().contiguous(), 1.0 / s2).t().contiguous()     # Kt L
                Q = grid_ops.gemm(L, KL, ta=True)
                Q.diagonal().add_(1.0)                                                                                # add_jitter(1.0), :355
                Kb = grid_ops.kron_toeplitz_mm(self._grid, tcol, b[o, :, 0], 1.0 / s2)
                Ls.append(L); KLs.append(KL); Qs.append(Q); projs.append(grid_ops.gemm(L, Kb[:, None].contiguous(), ta=True))
            st = (lambda xs: xs[0]) if self.num_outputs == 1 else torch.stack
            rs = {"L": st(Ls), "KL": st(KLs), "Q": st(Qs), "proj": st(projs)}
            self._memo["root_space"] = rs
        return rs

    @property
    def current_inducing_compression_matrix(self):
        return self._root_space()["KL"]

    @property
    def current_qmatrix(self):
        return self._root_space()["Q"]

    @property
    def root_space_projection(self):
        return self._root_space()["proj"]

    def _dump_caches(self):
        self._memo.pop("prediction_cache", None)
        self._memo.pop("pending_rank_update", None)
        self._memo.pop("root_space", None)
        # "hyper" (Toeplitz columns + Kronecker eigenbasis) is keyed on the parameters' version
        # counters and survives streaming updates: only a hyper-parameter change invalidates it

    def zero_grad(self, *args, **kwargs):
        # the reference calls gp.zero_grad() after every optimiser step (OSR:146, BFN:416-418): treat it as "the
        # hyper-parameters may have moved".  The memoisation below keys on the parameters' version counters, which a
        # *fused* optimiser (torch.optim.Adam(fused=True)) does not advance -- the epoch covers that.
        self.__dict__["_hyper_epoch"] = self.__dict__.get("_hyper_epoch", 0) + 1
        self._dump_caches()
        return super().zero_grad(*args, **kwargs)

    def hyperparameters_changed(self):
        """Tell the model that hyper-parameters were modified in a way autograd's version counters do not see (fused
        optimisers, writes through .data): every hyper-parameter-dependent cache is recomputed on next use."""
        self.__dict__["_hyper_epoch"] = self.__dict__.get("_hyper_epoch", 0) + 1
        self._dump_caches()

    # -------------------------------------------------------------- forward --
    def forward(self, X, **kwargs):
        if self.training:
            return self._train_forward(X)
        return self._eval_forward(X)

    def __call__(self, *args, **kwargs):
        if len(args) == 0:
            args = (None,)
        return super().__call__(*args, **kwargs)

    def _train_forward(self, X):
        # dummy: the real action happens in the MLL (:173-203)
        out = self.num_outputs
        n = X.shape[-2] if X is not None else self.num_data
        mean_shape = (out, n) if out > 1 else (n,)
        mean = torch.zeros(mean_shape, dtype=self._dtype, device=self._device)
        if X is None:
            return MultivariateNormal(mean, ZeroLazyTensor(*mean_shape, n, dtype=self._dtype, device=self._device))
        Xf = X.reshape(-1, self._grid.d).to(self._device, self._dtype)
        ops = [InterpolatedKernel(self._grid, Xf, tcol, 1.0, self._err) for tcol, _, _ in self._hyper()]
        return MultivariateNormal(mean, ops[0] if out == 1 else BatchOperator(ops))

    def _eval_forward(self, X):
        grid, out = self._grid, self.num_outputs
        X = X.to(self._device, self._dtype)
        block = None
        lead = X.shape[:-1]
        if X.dim() > 2:
            block = X.shape[-2]
        Xf = X.reshape(-1, grid.d).contiguous()
        n = Xf.shape[0]
        # smooth kernel on a large grid, variances wanted: the variance of the batch comes from the spectral factor (no solve at
        # all, with a per-query truncation bound).  The MEAN comes from the warm-started PCG state whenever that is current
        # (it carries its own tolerance and costs one gather).  Where it is not -- the hyper-parameters have moved since the
        # last solve, or no solve has run yet: a streaming wrapper that takes an MLL step per batch keeps a factor current for
        # that step anyway, while a PCG mean would first re-solve the preconditioner's eigenproblems and then iterate -- the
        # factor serves the mean too, but only while its mean monitor is green (sqrt(tail(w) b^T (Kt - Kt_B) b), evaluated every
        # few states: the variance bound does not control the mean).
        sq = None
        ms = self._mean_state
        pcg_current = self._memo.get("prediction_cache") is not None or (ms is not None and ms.get("ver") == self._hyper_version())
        hypers_moved = not pcg_current and ms is not None          # (means only, no PCG state at all: a cold solve, then warm ones)
        factor_mean = False
        try_spectral = settings.skip_posterior_variances.off() or (hypers_moved and self._spectral_in_use())
        if not try_spectral and self._use_dense() and self._spectral_in_use():
            try_spectral = True                      # small grid, means only (the classifier's predict): a factor in use serves them too
        if try_spectral and self._use_dense():
            # small grid: the cached nodal factor (M: two gathers per request, rank-q updates after conditioning at fixed hyper-parameters --
            # acquisition loops, fantasies) answers whenever it is current or one rank-q update away; the spectral factor serves the requests
            # that follow a hyper-parameter step (its refresh stays on the device), and only where somebody -- an MLL step, evaluate() --
            # has already built it
            pend = self._memo.get("pending_rank_update")
            dense_current = self._memo.get("prediction_cache") is not None or (pend is not None and pend[3] == self._hyper_version())
            facs = self.__dict__.get("_spectral", {})
            have_factor = all((f := facs.get(o)) is not None and f.ref is not None and not f.stale for o in range(out))
            try_spectral = have_factor and not dense_current
        if try_spectral:
            sps = [self._spectral_state(o) for o in range(out)]
            if all(sp is not None for sp in sps):
                if not pcg_current:
                    self._measure_factor_means(sps)
                    pcg_current = self._memo.get("prediction_cache") is not None      # (a measurement leaves a current PCG state behind)
                sq = [sp[0].query(sp[1], Xf, sp[2]) for sp in sps]
                factor_mean = not pcg_current and all(sp[0].mean_ok for sp in sps)
        pc = None
        if factor_mean:
            mean = torch.stack([s_.mean() for s_ in sq], dim=1)                   # [n, out] fp64
            scale = mean.abs().amax(0)
            for o, sp in enumerate(sps):
                sp[0].mean_monitor(sp[1], sq[o], self._kernel_cache["interpolation_cache"][o, :, 0], sp[2], scale[o])
            factor_mean = all(sp[0].mean_ok for sp in sps)                         # (a verdict read just now may have turned it off)
            mean = mean.to(self._dtype)
        if not factor_mean:
            pc = self.prediction_cache
            mean = grid_ops.gather(grid, Xf, pc["pred_mean"][..., 0], self._err)  # [n, out]   left_interp, :206-210
        if settings.deferred_bounds_check.off():
            flag = grid_ops.read_flag(self._err)       # gpytorch raises inside this call for queries outside the grid
            if flag:
                self._raise_out_of_bounds(flag)
        if settings.skip_posterior_variances.on():
            covs = None
        else:
            chunk = settings.variance_chunk.value()
            if sq is not None:
                posts = [None] * out
            else:
                posts = pc["pred_cov"].ops if out > 1 else [pc["pred_cov"]]
            covs = [PredictiveCovariance(_wtw_post, Xf, self._hyper()[o][1] if self.has_learnable_noise else 1.0, self._err, chunk=chunk,
                                         block=block if X.dim() > 2 else None, spectral=None if sq is None else (lambda o=o: sq[o]))
                    for o, _wtw_post in enumerate(posts)]
        if covs is not None and settings.fast_pred_samples.on():
            # BFN:229-243: hand out a root form of the covariance where a factor provides one (else the exact covariance, as always)
            covs = [(c.root_decomposition() or c) for c in covs]
        # output shapes follow :248-252
        if out == 1:
            mean_o = mean[:, 0].reshape(lead)
            if covs is None:
                cov = ZeroLazyTensor(*lead, lead[-1], dtype=self._dtype, device=self._device)
            else:
                cov = covs[0]
            return MultivariateNormal(mean_o, cov)
        mean_o = mean.t().reshape((out,) + tuple(lead))
        if covs is None:
            cov = ZeroLazyTensor(out, *lead, lead[-1], dtype=self._dtype, device=self._device)
        else:
            cov = BatchOperator(covs)
        return MultivariateNormal(mean_o, cov)

    # -------------------------------------------------------------- updates --
    def condition_on_observations(self, X, Y, noise=None, inplace=False):
        """a7, :258-285.  inplace: the statistics buffers are updated where they
        live (O(4^{2d}) atomics per point); otherwise they are cloned first and a
        sibling model sharing covar_module / likelihood is returned."""
        if X.dim() > 2 or Y.dim() > 2:
            # batch-expanded conditioning (what OSB.fantasize asks for, OSB:51-61): a batch of conditioned copies
            if inplace:
                raise RuntimeError("batched conditioning returns a batch of models and cannot be done in place")
            from .fantasy import BatchedFantasyModel

            return BatchedFantasyModel(self, X, Y, noise)
        if Y.dim() == 1:
            Y = Y[:, None]
        if noise is not None:
            noise = self._canon_noise(noise, Y)
        q = X.reshape(-1, self._grid.d).shape[0]
        old_pc = self._rank_update_source(q)
        if inplace:
            self._absorb(self._kernel_cache, X, Y, noise, init=False)
            self.num_data = self.num_data + q
            self._dump_caches()
            self._seed_rank_updated_cache(self, old_pc, X, noise, Y)
            return None
        new_cache = self._clone_cache(self._kernel_cache)
        new_gp = type(self)(
            covar_module=self.covar_module,
            kernel_cache=new_cache,
            learn_additional_noise=self.has_learnable_noise,
            likelihood=self.likelihood,
            num_data=self.num_data + q,
        )
        new_gp._wsum_dev = self._wsum_dev.clone()
        new_gp._wsum_host = list(self._wsum_host)
        new_gp._wsum_dev_host = list(self._wsum_dev_host)
        new_gp._wsum_dirty = self._wsum_dirty
        new_gp._absorb(new_cache, X, Y, noise, init=False)
        if self._mean_state is not None:
            new_gp._mean_state = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self._mean_state.items()}
            new_gp._mean_state["R_ok"] = False          # the copied residual predates the increment just absorbed
        if not self.training:
            new_gp.eval()
        self._seed_rank_updated_cache(new_gp, old_pc, X, noise, Y)
        return new_gp

    def stream_step(self, X, Y, want_mean=True):
        """evaluate -> absorb -> refresh for one streamed batch (the reference driver's online step at batch granularity,
        experiments/regression.py:48-54 with fixed hyper-parameters): predictive mean of X under the current posterior, then
        ``condition_on_observations(X, Y, inplace=True)``, then the posterior mean refreshed.  Equivalent to
        ``m = self(X).mean; self.condition_on_observations(X, Y, inplace=True); self.prediction_cache`` -- which is also the
        fallback -- but on large single-output grids the three launches go through ONE C-ABI call (``wiski_stream_step``) with
        the per-step host work reduced to bookkeeping.  Unit noise.  Returns the mean [n] (or None)."""
        st = self._stream_fast_state(X, Y)
        if st is None:
            self._finish_pending()
            self.leave_stencil_shard()               # the generic path below reads and writes the whole stencil
            mean = None
            if want_mean:
                with settings.skip_posterior_variances(True):
                    mean = self(X).mean
            self.condition_on_observations(X, Y, None, inplace=True)
            self.prediction_cache
            return mean
        step, ms, pst = st
        q = X.shape[0]
        if getattr(self, "_ones_cache", None) is None or self._ones_cache.shape[0] < q:
            self._ones_cache = torch.ones(max(q, 4096), dtype=self._dtype, device=self._device)
        ones = self._ones_cache[:q]
        mean = torch.empty(q, dtype=self._dtype, device=self._device) if want_mean else None
        # bookkeeping of _absorb / prediction_cache
        self._wsum_host[0] += float(q)
        self.num_data = self.num_data + q
        self._refresh_count = getattr(self, "_refresh_count", 0) + 1
        self._two_level_step(step, pst, X, q)            # (may set the poll hint: before _first_poll)
        last = (getattr(self, "_last_iters", None) or [0])[0]
        fc, probe = self._first_poll(last)
        carry = ms.get("R_ok", False) and self._refresh_count % 16 != 0
        step.args.shift = float(self._wsum[0]) / pst["norm"]
        y1 = Y.reshape(-1)
        y1 = y1 if y1.is_contiguous() else y1.contiguous()
        if "_spectral" in self.__dict__:
            self._spectral_absorb(0, X, None, y1)
        if settings.deferred_refresh.on() or step.pending:
            prev, pending = step(X, y1, ones, ones, ones, mean, carry, fc, defer=settings.deferred_refresh.on())
            self._pending_step = step if pending else None
            if prev is not None:
                if prev[2]:                              # the PREVIOUS batch held out-of-grid points: this one was not queued at all
                    self._wsum_host[0] -= float(q)
                    self.num_data = self.num_data - q
                    self._refresh_count -= 1
                self._note_solve(ms, prev, *getattr(self, "_pending_fc", (0, False)))
            self._pending_fc = (fc, probe)
            if not pending:                              # deferral switched off meanwhile: this call ran to convergence
                self._note_solve(ms, (step.it.value, step.rr.value, step.herr.value, True), fc, probe)
            return mean
        self._note_solve(ms, step(X, y1, ones, ones, ones, mean, carry, fc), fc, probe)
        return mean

    # ------------------------------------------------- two-level preconditioner --
    def _two_level_applies(self):
        if not (settings.two_level_preconditioner.on() and self.num_outputs == 1 and self._grid.d == 3 and self._dtype == torch.float32
                and max(self._grid.g) <= 64 and not self._use_dense()):
            return False
        # the slab kernel's coefficient exchange spins on words written by other blocks of the same launch: all 2 g0 blocks must be
        # co-resident (one per CU) -- not on a partitioned device or a part with fewer CUs (the C side refuses as well)
        cus = self.__dict__.get("_cu_count")
        if cus is None:
            cus = self.__dict__["_cu_count"] = torch.cuda.get_device_properties(self._device).multi_processor_count if self._device.type == "cuda" else 0
        return 2 * self._grid.g[0] <= cus

    def _two_level_note(self, X, wa, init=False):
        """Every point the statistics absorb is either pending for, or part of, the exact block of the two-level preconditioner
        (lazy/two_level.py); points absorbed without passing here make the tracker give up until the statistics are rebuilt."""
        if not self._two_level_applies():
            self.__dict__.pop("_two_level", None)
            return
        from ..lazy.two_level import TwoLevelTracker

        tr = self.__dict__.get("_two_level")
        if tr is None or init:
            if tr is None and not init:
                return                                 # statistics older than the tracker: never covered
            tr = self.__dict__["_two_level"] = TwoLevelTracker()
        tr.note(X.reshape(-1, self._grid.d), wa)

    def _two_level_lose(self):
        tr = self.__dict__.get("_two_level")
        if tr is not None:
            tr.lose()

    def _two_level_step(self, step, pst, X, q):
        """Before a one-call streaming step: note its batch, keep the block's refresh pipeline going, point the solve at the block."""
        tr = self.__dict__.get("_two_level") if self._two_level_applies() else None
        tl = None
        if tr is not None:
            tr.note(X, None)
            _, s2, _ = self._hyper()[0]
            sh = self.__dict__.get("_stencil_shard")
            # stencil-sharded replicas see the gathered batch of ALL ranks and each keeps its own copy of the block: every world-th
            # point (weighted) keeps a replica's refresh work at what one GPU's stream costs (the side stream must keep pace with
            # the steps, or the lock-step switch would stall them)
            sub = max(settings.two_level_subsample.value(), sh["world"]) if sh is not None and sh["world"] > 1 else None
            tl = tr.for_step(self._grid, self._device, pst, 1.0 / s2, float(self._wsum[0]), self._err,
                             lockstep=sh is not None or settings.two_level_lockstep.on(),
                             last_iters=(getattr(self, "_last_iters", None) or [0])[0], subsample=sub)
            if tr.switched:
                # a new block: the iteration count of the previous solves says little about the next one -- poll after 2 iterations,
                # then after every one (the poll placement would otherwise walk down one iteration per probe)
                self._poll_hint_sticky = 2
        if getattr(step, "_two_level", None) is not tl:
            step.set_two_level(tl)

    def _first_poll(self, last):
        """Where a warm refresh polls convergence first: (iteration count, is this a probe?).  Streaming steps are alike (at
        50^3 the uniform bench stream needs 3 iterations for its first ~40 steps and 2 ever after; the clustered one 6, now and
        then 5), so the first poll goes where the previous refresh converged.  A *probe* polls one iteration earlier to notice
        that the stream got easier; one that fails costs a stand-alone vector update + poll and a host round trip (~14 us of a
        ~215 us step), one that is not made when it would have succeeded costs an iteration (41 us) per step.  So: probe
        every 4th refresh while the last converged residual says one iteration less might do (it is below tol / 10; the
        contraction per iteration is ~0.08 here: the residual after 3 iterations falls from 8e-5 to 7e-6 before 2 suffice),
        otherwise after 8, 16, 32 refreshes.  (Probing every other refresh, as before: 50 % failed polls, 0.221 ms per step
        against 0.216 with this placement, tools/policy_probe.py.)"""
        if not last:
            return 0, False
        sticky = getattr(self, "_poll_hint_sticky", 0)
        if sticky:
            # a new block of the two-level preconditioner went in: for this step and the next (whose `last` still predates the
            # switch: deferred refreshes report one step late) poll after 2 iterations, then after every one
            self._poll_hint_sticky = sticky - 1
            return min(last, 2), False
        hint = getattr(self, "_poll_hint", 0)
        if hint:
            # the previous solve was a cold one: its count (8 at 50^3) says nothing about a warm, residual-carrying refresh
            # (3 there).  Poll early and then after every iteration -- a few extra polls instead of walking down from the
            # cold count one wasted iteration per refresh.
            return min(last, hint), False
        pend = getattr(self, "_probe_pending", 0)
        if pend:
            # deferred refreshes: the verdict of the probe queued by the previous step is not in yet (and `last` is older
            # still).  Poll where the probe did: if it fails this costs one more cheap poll, if it succeeds an iteration less.
            self._probe_pending = 0
            return pend, False
        wait = getattr(self, "_probe_wait", 0)
        rel = getattr(self, "_last_rel", None)
        tol = settings.cg_tolerance.value() or (1e-7 if self._dtype == torch.float32 else 1e-11)
        informed = rel is not None and rel < 0.1 * tol
        if informed:
            wait = min(wait, 4)
        probe = wait <= 0 and last > 1
        self._probe_informed = informed
        self._probe_wait = wait - 1
        fc = max(1, last - (1 if probe else 0))
        if probe:
            self._probe_pending = fc
        return fc, probe

    def _note_poll(self, it, fc, probe):
        if probe:
            self._probe_pending = 0
            if it <= fc:
                self._probe_gap, self._probe_wait = 0, 8        # cut to 4 by _first_poll if the new residual invites it
            elif getattr(self, "_probe_informed", False):
                self._probe_wait = 4
            else:
                self._probe_gap = min(32, max(8, 2 * getattr(self, "_probe_gap", 0)))
                self._probe_wait = self._probe_gap

    def _note_solve(self, ms, res, fc, probe=False):
        """Host bookkeeping after a refresh: iteration history for the poll placement, residual validity, out-of-grid error."""
        it, rel, flag, conv = res
        if fc:
            self._note_poll(it, fc, probe)
        self._last_iters = [it]
        self._last_rel = rel
        self._poll_hint = 0
        ms["R_ok"] = bool(conv)
        pc = self._memo.get("prediction_cache")
        if pc is not None:
            pc["cg_iters"] = [it]
        if flag:
            ms["R_ok"] = False
            self._raise_out_of_bounds(flag)

    def _finish_pending(self):
        """A deferred refresh (settings.deferred_refresh) is still in flight: wait for its poll, finish the solve if needed."""
        step = self.__dict__.get("_pending_step")
        if step is None:
            return
        self._pending_step = None
        if step.pending:
            prev, _ = step(None, None, None, None, None, None, 0, 0, defer=False)
            if prev is not None and self._mean_state is not None:
                self._note_solve(self._mean_state, prev, *getattr(self, "_pending_fc", (0, False)))

    def _stream_fast_state(self, X, Y):
        """(prepared StreamStep, mean state, preconditioner state) when the one-call streaming step applies, else None."""
        if (self.num_outputs != 1 or self._use_dense() or settings.spectral_preconditioner.off() or settings.residual_carry_over.off()
                or X.dim() != 2 or not X.is_cuda or X.dtype != self._dtype or not X.is_contiguous() or Y.dtype != self._dtype):
            return None
        ms = self._mean_state
        pc = self._memo.get("prediction_cache")
        ver = self._hyper_version()
        if ms is None or pc is None or ms.get("ver") != ver or "pending_rank_update" in self._memo:
            return None
        op = _wtw_ops(self._kernel_cache["WtW"])[0]
        if not op.is_half or op.root is not None or "_cnt" not in self._kernel_cache:
            return None
        pst = self._memo.get("precond", {}).get(0)
        wsum_new = float(self._wsum[0]) + X.shape[0]
        if pst is None or pst["ver"] != ver:
            return None
        if wsum_new > 2.0 * pst["wsum"] and not self._density_profile_still_fits(pst):
            return None                                   # the density profile has moved: generic path (new eigenbasis) this step
        its = (getattr(self, "_last_iters", None) or [0])[0]
        if pst.get("it0") is None and its > 0:
            pst["it0"] = its
        if pst.get("it0") is not None and its >= pst["it0"] + 2 and wsum_new > 1.1 * pst["wsum"]:
            return None
        tol = _default_tol(self._dtype)
        c = self._kernel_cache
        # raw device pointers go into the prepared call: key it on every one of them (ids of Python wrappers can be reused)
        sh = self.__dict__.get("_stencil_shard")
        key = (None if sh is None else (sh["rank"], sh["world"]), ver, ms["U"].data_ptr(), ms["Z"].data_ptr(), ms["R"].data_ptr(), op.stencil.data_ptr(), c["interpolation_cache"].data_ptr(),
               c["_cnt"].data_ptr(), c["_stats"].data_ptr(), self._err.data_ptr(), pst["eig"][0].data_ptr(), str(self._device), tol,
               settings.cg_check_every.value(), settings.max_cg_iterations.value())
        cached = self.__dict__.get("_stream_step_cache")
        if cached is None or cached[0] != key:
            tcol, s2, _ = self._hyper()[0]
            step = grid_ops.StreamStep(self._grid, self._dtype, self._device, op.stencil, c["interpolation_cache"][0, :, 0], c["_cnt"][0],
                                       c["_stats"][0], self._err, ms["U"][0], ms["Z"][0], ms["R"][0], tcol, self._pcg_ws,
                                       settings.max_cg_iterations.value())
            step.set_solver(1.0 / s2, pst["eig"], 0.0, tol, 1)
            if sh is not None:
                step.set_shard(sh["rank"], sh["world"], comm=sh.get("comm"), allreduce=sh["allreduce"])
            cached = (key, step)
            self.__dict__["_stream_step_cache"] = cached
        return cached[1], ms, pst

    def _density_profile_still_fits(self, pst):
        """The data volume has doubled since the preconditioner's density profile was last looked at.  Look at it from inside
        the one-call streaming path instead of sending the step through the generic three-call one (4 looks per 3droad-sized
        pass used to cost a ~1 ms generic step each), and without draining the pipeline: the 3 small reductions and the copy
        of the marginals to pinned memory are QUEUED now, the verdict is read by whichever later step finds the copy done;
        meanwhile the stream carries on with the basis it has.  On a stationary stream the normalised profile has not moved
        (settings.precond_profile_drift): the eigenbasis stays, only the scale follows.  False: the profile moved (or there is
        none) -- the caller falls back to the generic path and `_precond` re-solves the eigenbasis."""
        old = pst.get("profiles")
        cnt = self._kernel_cache.get("_cnt") if settings.density_profile_preconditioner.on() else None
        if old is None or cnt is None:
            return False
        g = self._grid.g
        look = self.__dict__.get("_profile_look")
        if look is None:
            c3 = cnt[0].reshape(g).double()
            margs = [c3.sum(dim=[r for r in range(self._grid.d) if r != q]) if self._grid.d > 1 else c3 for q in range(self._grid.d)]
            margs = torch.stack([torch.nn.functional.pad(mg, (0, max(g) - mg.numel())) for mg in margs])
            host = self.__dict__.get("_profile_look_host")            # pinned once (a pinned allocation costs ~0.3 ms)
            if host is None or host.shape != margs.shape:
                host = torch.empty(margs.shape, dtype=margs.dtype, pin_memory=True)
                self.__dict__["_profile_look_host"] = host
            host.copy_(margs, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.__dict__["_profile_look"] = (host, ev, float(self._wsum[0]), margs)
            return True
        host, ev, wsum_then, _ = look
        if self.__dict__.get("_stencil_shard") is not None:
            ev.synchronize()                              # sharded replicas must all read the verdict at the SAME step (control flow
        elif not ev.query():                              # that depends on copy timing would let the ranks' collectives diverge)
            return True                                   # still in flight
        self.__dict__["_profile_look"] = None
        margs = host.numpy()
        if not margs.max() > 0:
            return False
        for q, gq in enumerate(g):
            t = (margs[q, :gq] / margs[q, :gq].max()).clip(1e-2, None)
            if float(abs(t - old[q]).max()) > settings.precond_profile_drift.value():
                return False
        pst["wsum"] = wsum_then                           # same density shape: keep the eigenbasis, only the scale moves
        pst["it0"] = None
        return True

    def _rank_update_source(self, q):
        """The cached dense posterior(s), if a rank-q Woodbury update of them is valid and cheaper than a fresh factor:
        dense regime, cache built for the current hyper-parameters, 0 < q <= m / 8, fewer than 64 stacked updates."""
        if q == 0 or not self._use_dense() or q > self._grid.m // 8 or settings.dense_rank_updates.off():
            return None
        self._apply_pending_rank_update()
        pc = self._memo.get("prediction_cache")
        if pc is None or pc.get("ver") != self._hyper_version():
            return None
        posts = pc["pred_cov"].ops if self.num_outputs > 1 else [pc["pred_cov"]]
        if not all(isinstance(p, DenseInducingPosterior) and p.updates < 64 for p in posts):
            return None
        return posts

    @staticmethod
    def _seed_rank_updated_cache(target, old_posts, X, noise, Y):
        """Note on `target` (self or the sibling model) that its prediction cache can be obtained from `old_posts` by a
        rank-q update.  Applied lazily by the next prediction_cache request; dropped if the caches are dumped first
        (e.g. by a hyper-parameter step), so a BO loop that refits after every update pays nothing for it."""
        if old_posts is None:
            return
        X = X.reshape(-1, target._grid.d).to(target._device, target._dtype).contiguous()
        was = []
        for o in range(target.num_outputs):
            if noise is None:
                was.append(torch.ones(X.shape[0], dtype=target._dtype, device=target._device))
            else:
                was.append(1.0 / noise.to(target._device, target._dtype)[:, o].clamp_min(1e-7))
        target._memo["pending_rank_update"] = (old_posts, X, was, target._hyper_version())

    def _apply_pending_rank_update(self):
        pend = self._memo.pop("pending_rank_update", None)
        if pend is None or "prediction_cache" in self._memo:
            return
        old_posts, X, was, ver = pend
        if ver != self._hyper_version():
            return
        self.check_bounds()         # as a fresh dense factor would: raise for out-of-grid inputs before trusting the update
        b = self._kernel_cache["interpolation_cache"]
        out = self.num_outputs
        posts, U = [], torch.empty((out, self._grid.m), dtype=self._dtype, device=self._device)
        ops = _wtw_ops(self._kernel_cache["WtW"])
        for o in range(out):
            post = old_posts[o].rank_update(ops[o], X, was[o], self._err)
            U[o] = post.solve_columns(b[o, :, 0][None])[0][0]
            posts.append(post)
        self._last_iters = [0] * out
        self._memo["prediction_cache"] = {"pred_mean": U[..., None], "pred_cov": posts[0] if out == 1 else BatchOperator(posts),
                                          "cg_iters": [0] * out, "ver": ver}

    def get_fantasy_model(self, inputs, targets, noise_term=None, **kwargs):
        """BFN:287-332.  inputs [*b, q, d], targets [*b, q] or [num_fantasies, *b, q]: a batch of conditioned copies
        (``models/fantasy.py``: specified from the maths, the reference's cache expansion is broken at HEAD);
        unbatched inputs [q, d] with targets [q] / [q, 1]: a plain functional ``condition_on_observations``."""
        plain = inputs.dim() == 2 and (targets.dim() == 1 or (targets.dim() == 2 and tuple(targets.shape) == (inputs.shape[0], self.num_outputs)))
        if not plain:
            from .fantasy import BatchedFantasyModel

            return BatchedFantasyModel(self, inputs, targets, noise_term)
        if targets.dim() == 1:
            targets = targets[:, None]
        if noise_term is None:
            noise_term = torch.ones_like(targets)
        return self.condition_on_observations(inputs, targets, noise_term, inplace=False)

    def set_train_data(self, train_inputs, train_targets, train_noise_term):
        """:420-428 -- rebuild every statistic from scratch."""
        if train_targets.dim() == 1:
            train_targets = train_targets[:, None]
        noise = self._canon_noise(train_noise_term, train_targets)
        self._finish_pending()                      # a deferred solve must not be resumed on zeroed statistics
        self.__dict__.pop("_stream_step_cache", None)
        cache = self._kernel_cache
        cache["interpolation_cache"].zero_()
        cache["_stats"].zero_()
        if "_cnt" in cache:
            cache["_cnt"].zero_()
        for op in _wtw_ops(cache["WtW"]):
            op.stencil.zero_()
            op.root = op.inv_root = None            # L L^T described the old matrix
        self._wsum_dev.zero_()
        self._wsum_host = [0.0] * self.num_outputs
        self._wsum_dev_host = [0.0] * self.num_outputs
        self._wsum_dirty = False
        self._memo.pop("precond", None)
        self._drop_spectral()
        self._absorb(cache, train_inputs, train_targets, noise, init=True)
        self.num_data = train_inputs.reshape(-1, self._grid.d).shape[0]
        self._mean_state = None
        self._dump_caches()

    def to(self, *args, **kwargs):
        res = super().to(*args, **kwargs)
        device = None
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = torch.device(a)
            elif torch.is_tensor(a):
                device = a.device
        device = kwargs.get("device", device)
        if device is not None and self._kernel_cache is not None and torch.device(device) != self._device:
            c = self._kernel_cache
            stats = c["_stats"].to(device)
            ops = _wtw_ops(c["WtW"])
            pack = self._stencil_pack(ops)
            if pack is not None and len(ops) > 1:
                pk = pack.to(device)
                mv = lambda t: None if t is None else t.to(device)
                new_ops = [StencilWtW(self._grid, pk[o], mv(op.root), mv(op.inv_root)) for o, op in enumerate(ops)]
            else:
                new_ops = [op.to(device) for op in ops]
            self._kernel_cache = self._pack_cache(c["interpolation_cache"].to(device), stats, new_ops, c["_cnt"].to(device) if "_cnt" in c else None)
            self._device = torch.device(device)
            self._err = grid_ops.new_err_flag(device)
            self._mean_state = None
            self._memo = {}
            self.__dict__.pop("_stream_step_cache", None)
            self._drop_spectral()
        return res

    # ------------------------------------------------- distributed statistics --
    def stats_buffers(self):
        """Tensors that are additive over data shards (what RCCL all-reduces):
        b, the W^T W stencils and (y^T D^-1 y, logdet D)."""
        c = self._kernel_cache
        return [c["interpolation_cache"], c["_stats"], c["_cnt"]] + [op.stencil for op in _wtw_ops(c["WtW"])]
