"""The hyper-parameter step of the streaming loop as ONE captured HIP graph.

One step of the reference's online loop (experiments/regression.py:48-54) does, between absorbing two batches, a single Adam step
on the marginal log-likelihood (online_ski_regression.py:135-147).  With the spectral Woodbury factor serving the MLL
(lazy/spectral_woodbury.py) that step is ~45 small launches -- kernel columns, the factor's reduced-basis gradient, the scalar
tail, autograd's bookkeeping nodes, the fused Adam update -- whose ~0.25 ms of GPU work costs ~1 ms of host time when issued one
framework op at a time.  Here the whole sequence (forward, backward, optimiser update) is recorded once into a HIP graph
(``torch.cuda.CUDAGraph``: the library's kernels are launched on the capture stream like any other) and replayed with one call.

What a replay needs is that every address the recorded kernels read is still the right one:
  * parameters, optimiser state, the model's statistics buffer: live in place, so they are;
  * the factor state (G, chol^-1, sqrt(lam), zeta, the eigenvector tables, b^T M b, logdet) is rebuilt into fresh tensors by every
    refresh -- it is copied into static staging buffers before each replay, by ONE launch (``wiski_multi_copy_f64``: three r x r
    matrices, seven packed pieces and the data count, which enters the recorded kernels as a device scalar like 1 / sigma2);
  * the index set S of the basis is used in place: the graph is re-captured when the factor re-selects it (or when the learning
    rate, the optimiser, the dtype or the set of priors changes).
The first steps run eagerly (they also serve as the warm-up the allocator wants before a capture); whenever the spectral path does
not apply -- dense regime, rough kernel, a foreign optimiser -- the caller's eager path runs instead.  Several outputs (the Dirichlet
classifier's two) are one graph: one factor, one set of staging buffers per output.

Two refinements (round 4).  (i) For the reference's own parameterisation -- (Scale of)* RBF | Matern, homoskedastic second
noise, plain Adam, no registered priors -- the step is recorded WITHOUT autograd (``_capture_fused``, csrc/hyper_step.hip): 11 graph
nodes instead of 40, and the graph leaves the Toeplitz columns and sigma2 of the UPDATED hyper-parameters behind (``read_loss`` hands
them to the model's memo together with the loss: one host read).  (ii) ``prepare()``: evaluate() of a batch checks and stages the
step of the same batch before its own host read, so update() only has to replay.
"""
import contextlib
import gc

import torch

from .. import _hip, grid_ops, settings
from ..lazy.spectral_woodbury import SpectralBasis

WARMUP_STEPS = 3


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is being captured: a collection that happens to free another model's CUDAGraph (or
    anything else whose destructor calls into the runtime) in the middle of a capture aborts the process.  torch.cuda.graph() collects
    once on entry; this keeps the allocation-triggered collections out until the capture has ended."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class GraphedHyperStep:
    def __init__(self, wrapper):
        self.w = wrapper
        self.key = None
        self.graph = None
        self.loss = None
        self.eager_calls = 0
        self.captures = 0          # diagnostics / tests
        self.replays = 0
        self.disabled = None       # reason, once a capture has failed
        self.churn = self.cooldown = self.replays_at_capture = 0
        self.fused = False         # the current graph was recorded without autograd (csrc/hyper_step.hip)
        self.fused_captures = 0
        self._seed = None
        self._prepared = self._prepared_sps = None

    # ---------------------------------------------------------------------------------------------------------------------
    def _applicable(self, skip_bounds=False):
        """The live factor state if this step can run as a graph, else None."""
        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        if self.disabled is not None or settings.graphed_hyper_step.off() or settings.spectral_factor.off():
            return None
        if torch.device(gp._device).type != "cuda" or not gp._spectral_allowed() or w.mll.clear_caches_every_iteration:
            return None
        if not isinstance(opt, torch.optim.Adam) or not all(g.get("capturable") and g.get("fused") for g in opt.param_groups):
            return None
        if max(gp._grid.g) > 64:                      # (the one-launch lag gradient; larger factors take the op-by-op form)
            return None
        gp._finish_pending()
        # the eager MLL checks the out-of-grid flag first (a host read; it may not happen inside a capture) -- unless evaluate() has
        # just read it clean with the same statistics (models/online_ski_regression.py)
        clean = gp.__dict__.get("_bounds_clean_at")
        fac0 = gp.__dict__.get("_spectral", {}).get(0)
        if not skip_bounds and (gp._wsum_dirty or clean is None or fac0 is None or clean != (gp.num_data, fac0.data_version)):
            gp.check_bounds()
        sps = [gp._spectral_state(o) for o in range(gp.num_outputs)]      # one factor per output (own statistics, own hyper-parameters)
        return sps if all(sp is not None for sp in sps) else None

    def _token(self, sps):
        """What a staged-but-not-yet-replayed step depends on: the factor states themselves, hyper-parameters, data count, optimiser."""
        gp, opt = self.w.gp, self.w.gp_optimizer
        g0 = opt.param_groups[0]
        return (tuple(id(sp[1]) for sp in sps), tuple(sp[0].data_version for sp in sps), gp._hyper_version(), gp.num_data, id(opt), g0.get("lr") if not torch.is_tensor(g0.get("lr")) else id(g0.get("lr")),
                tuple(g0.get("betas", ())), g0.get("eps"), g0.get("weight_decay"), g0.get("amsgrad"), g0.get("maximize"))

    def _prepare(self, skip_bounds=False):
        """Everything of a step up to the replay: applicability, warm-up count, (re-)capture, staging of the factor state.  Returns True when
        the graph is ready to be replayed for the current state."""
        sps = self._applicable(skip_bounds)
        if sps is None:
            return False
        if self.eager_calls < WARMUP_STEPS:
            return None                               # (the caller counts the eager step)
        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        # everything the captured graph bakes in: the index sets, the optimiser's hyper-parameters (lr, betas, eps, weight decay ...),
        # the set of registered priors, the statistics buffer the MLL reads, which parameters are trainable
        from ..priors import REGISTRY_EPOCH

        def _hp(g):
            # (the entries a recorded Adam step bakes in; looked up one by one -- this runs every step)
            return (g.get("lr") if not torch.is_tensor(g.get("lr")) else id(g.get("lr")), tuple(g.get("betas", ())), g.get("eps"), g.get("weight_decay"),
                    g.get("amsgrad"), g.get("maximize"), g.get("capturable"), g.get("fused"), g.get("foreach"), g.get("differentiable"))

        key = (tuple((sp[1]["basis"].S.data_ptr(), sp[1]["basis"].r, sp[1]["basis"].kmax) for sp in sps), str(gp._dtype),
               tuple(_hp(g) for g in opt.param_groups), id(opt),
               settings.fused_hyper_columns.on(), tuple(p.requires_grad for g in opt.param_groups for p in g["params"]),
               REGISTRY_EPOCH[0], gp._kernel_cache["_stats"].data_ptr(), settings.fused_hyper_step.on())
        if key != self.key:
            # a capture costs ~2 ms: worth it only if it is then replayed.  If the factor keeps re-selecting its index set (host-side
            # refresh, a kernel whose spectrum moves fast), stop re-capturing for a while and let the eager path run.
            if self.cooldown > 0:
                self.cooldown -= 1
                return False
            if self.key is not None and self.replays - self.replays_at_capture < 4:
                self.churn += 1
                if self.churn >= 3:
                    self.churn, self.cooldown, self.key, self.graph = 0, 64, None, None
                    return False
            else:
                self.churn = 0
            try:
                self._capture(sps, key)
            except Exception as exc:                  # a capture that cannot be made is not an error of the step: run eagerly from now on
                self.disabled = f"{type(exc).__name__}: {exc}"
                self.graph = self.key = None
                gp.__dict__.pop("_graph_ctx", None)
                return False
        self._stage(sps)
        self._prepared = self._token(sps)
        self._prepared_sps = sps
        return True

    def prepare(self):
        """Called by evaluate() just before its one host read: the step that follows (update() of the same batch) finds its graph
        checked and its factor state staged -- that work is done while the host would otherwise wait for the metrics, instead of in
        the gap between the metrics and the replay, where the device idles."""
        if self.graph is None or self.disabled is not None or self.eager_calls < WARMUP_STEPS or settings.graphed_hyper_step.off():
            return
        try:
            if self._prepare(skip_bounds=True) is not True:       # (the flag is read by the caller, with its metrics)
                self._prepared = None
        except Exception:
            self._prepared = None

    def step(self, lazy=False):
        """One Adam step on -MLL; returns the loss, or None when the caller has to take the eager path.  lazy: return the device
        scalar instead of reading it (the read is a synchronisation: a caller with more launches to queue -- the absorb of the
        batch -- reads it after those; valid until the next replay)."""
        gp = self.w.gp
        ready = False
        if (self._prepared is not None and self.graph is not None and self.disabled is None and settings.graphed_hyper_step.on()
                and settings.spectral_factor.on() and not gp._wsum_dirty and gp.__dict__.get("_pending_step") is None):
            sps = self._prepared_sps
            ready = (all(sp[0].cur is sp[1] for sp in sps) and self._token(sps) == self._prepared
                     and gp.__dict__.get("_bounds_clean_at") == (gp.num_data, sps[0][0].data_version))
        self._prepared = None
        if not ready:
            r = self._prepare()
            self._prepared = None
            if r is None:
                self.eager_calls += 1
                return None
            if not r:
                return None
        self.graph.replay()
        self.replays += 1
        gp.zero_grad()                                # (as the eager step: drops the gradients and moves the hyper-parameter epoch on)
        self._seed = gp._hyper_version() if self.fused else None
        return self.loss if lazy else self.read_loss()

    def read_loss(self):
        """The loss of the last replay (a host read: everything queued so far completes).  The fused graph also left the Toeplitz
        columns and sigma2 of the UPDATED hyper-parameters behind: they are handed to the model's memo, so the next factor refresh
        neither re-evaluates the kernel nor reads sigma2 back (one synchronisation and ~8 launches less per step)."""
        if not self.fused:
            return float(self.loss)
        loss, s2 = self.host_read.tolist()
        gp = self.w.gp
        if self._seed is not None and self._seed == gp._hyper_version():
            gp._memo["hyper"] = (self._seed, [(self.f_tc, s2, self.f_tc64)])
            gp.__dict__["_s2_dev"] = (self._seed, self.f_s2)          # sigma2 on the device, for consumers that take it from there
        self._seed = None
        return loss

    # ---------------------------------------------------------------------------------------------------------------------
    def _stage(self, sps):
        n = float(self.w.gp.num_data)
        for k, ((fac, st, _), buf) in enumerate(zip(sps, self.bufs)):
            basis = st["basis"]
            fac.coefficients(st)
            v_sq, v_zeta, v_lam, v_ev, v_V, v_bMb, v_logdet = buf["views"]
            # ONE launch per output: the three r x r matrices, the seven packed pieces and (with the first output) the data count
            grid_ops.multi_copy([(buf["G"], st["G"]), (buf["Linv"], st["Linv"]), (buf["sqG"], st["sqG"]), (v_sq, st["sq"]), (v_zeta, st["zeta"]),
                                 (v_lam, basis.lam_kuu), (v_ev, basis.ev_tab.reshape(-1)), (v_V, basis.Vtab), (v_bMb, st["bMb"].reshape(1)),
                                 (v_logdet, st["logdet"].reshape(1))], scalar=n if k == 0 else None, scalar_dst=self.n_dev if k == 0 else None)

    # ---------------------------------------------------------------------------------------------------------------------
    def _fused_plan(self):
        """(plan, kind, lr, betas, eps, dtype) when the step can be recorded without autograd (settings.fused_hyper_step), else None."""
        from ..constraints import GreaterThan, Interval
        from ..kernels import GridInterpolationKernel, MaternKernel, RBFKernel, ScaleKernel
        from ..likelihoods.fnmg_likelihood import HomoskedasticNoise
        from ..priors import named_priors

        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        if settings.fused_hyper_step.off() or settings.fused_hyper_columns.off() or gp.num_outputs != 1 or not gp.has_learnable_noise:
            return None
        if next(iter(named_priors(w.mll)), None) is not None:
            return None
        cm = gp.covar_module
        if not isinstance(cm, GridInterpolationKernel):
            return None
        entries, k = [], cm.base_kernel
        while isinstance(k, ScaleKernel):
            entries.append((k.raw_outputscale, 1, k.raw_outputscale_constraint))
            k = k.base_kernel
        if type(k) is RBFKernel:
            kind = 0
        elif type(k) is MaternKernel:
            kind = {0.5: 1, 1.5: 2, 2.5: 3}[k.nu]
        else:
            return None
        entries.append((k.raw_lengthscale, 0, k.raw_lengthscale_constraint))
        noise = gp.likelihood.second_noise_covar
        if type(noise) is not HomoskedasticNoise:
            return None
        entries.append((noise.raw_noise, 2, GreaterThan(noise.LOWER)))
        if len(entries) > _hip.HYPER_MAX_PARAMS or len(opt.param_groups) != 1:
            return None
        grp = opt.param_groups[0]
        ps = grp["params"]
        if len(ps) != len(entries) or {id(p) for p in ps} != {id(e[0]) for e in entries}:
            return None                               # (a mean constant, a foreign parameter ...: the autograd recording handles those)
        if grp.get("weight_decay", 0) != 0 or grp.get("amsgrad") or grp.get("maximize") or grp.get("differentiable") or torch.is_tensor(grp["lr"]):
            return None
        dt = ps[0].dtype
        if dt not in (torch.float32, torch.float64):
            return None
        plan = _hip.wiski_hyper_plan()
        plan.count = len(entries)
        for i, (p, role, con) in enumerate(entries):
            st = opt.state.get(p)
            if (not p.requires_grad or not p.is_cuda or p.dtype != dt or not p.is_contiguous() or not st or "exp_avg" not in st
                    or not torch.is_tensor(st.get("step")) or not st["step"].is_cuda or st["step"].dtype != torch.float32 or st["step"].numel() != 1
                    or st["exp_avg"].dtype != dt or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous()):
                return None
            if role != 0 and p.numel() != 1:
                return None
            if role == 0 and p.numel() not in (1, gp._grid.d):
                return None
            if type(con) is Interval:
                ckind, lo, hi = 1, con.lower_bound, con.upper_bound
            elif isinstance(con, GreaterThan):
                ckind, lo, hi = 0, con.lower_bound, 0.0
            else:
                return None
            e = plan.p[i]
            e.raw, e.exp_avg, e.exp_avg_sq, e.step = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
            e.numel, e.step_numel, e.role, e.kind, e.lower, e.upper = p.numel(), 1, role, ckind, float(lo), float(hi)
        has_scale = any(r == 1 for _, r, _ in entries)
        b1, b2 = grp["betas"]
        return plan, kind, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), dt, has_scale, entries

    def _capture_fused(self, sps, key, fp):
        """The step without autograd: constraint transforms -> MLL tail -> the factor's backward launches -> columns' gradient -> chain
        rule + Adam -> the updated hyper-parameters' columns and sigma2.  11 graph nodes."""
        plan, kind, lr, b1, b2, eps, dt, has_scale, entries = fp
        w = self.w
        gp = w.gp
        fac, st, _ = sps[0]
        dev = st["G"].device
        f64 = dict(dtype=torch.float64, device=dev)
        basis = st["basis"]
        r = basis.r
        sizes = [r, r, r, basis.ev_tab.numel(), basis.Vtab.numel(), 1, 1]
        buf = {"G": torch.empty((r, r), **f64), "Linv": torch.empty((r, r), **f64), "sqG": torch.empty((r, r), **f64), "packed": torch.empty(sum(sizes), **f64)}
        v_sq, v_zeta, v_lam, v_ev, v_V, v_bMb, v_logdet = buf["views"] = torch.split(buf["packed"], sizes)
        sbasis = SpectralBasis.on_device(basis, v_V, v_ev.view(basis.ev_tab.shape), None, lam=v_lam)
        sst = {"basis": sbasis, "G": buf["G"], "sq": v_sq, "zeta": v_zeta, "coef": v_zeta, "Linv": buf["Linv"], "sqG": buf["sqG"], "bMb": v_bMb.reshape(()),
               "logdet": v_logdet, "kscale": None}
        self.bufs = [buf]
        fac._grid_dev()
        self.n_dev = torch.zeros(1, **f64)
        grid = gp._grid
        nell = next(p.numel() for p, role, _ in entries if role == 0)
        mk = lambda n_: torch.empty(n_, dtype=dt, device=dev)
        ell, s2, scale = mk(nell), mk(1), (mk(1) if has_scale else None)
        ell2, s2n, scale2 = mk(nell), mk(1), (mk(1) if has_scale else None)
        mid = torch.empty(9, **f64)
        self.host_read = torch.zeros(2, **f64)
        self.f_tc64 = torch.empty(sum(grid.g), **f64)
        self.f_tc = torch.empty(sum(grid.g), dtype=gp._dtype, device=dev)
        same = dt == gp._dtype                        # (parameters may be fp32 under an fp64 model: the columns in the model dtype by a cast node then)
        stats = gp._kernel_cache["_stats"]
        self._stage(sps)
        self._fused_keep = (plan, entries, ell, s2, scale, ell2, s2n, scale2, mid)      # (addresses recorded into the graph stay alive)
        # sigma2 of the updated hyper-parameters, left on the device in the MODEL's dtype for evaluate() (fp32 parameters under an fp64
        # model -- the default kernel under fp64 data -- : the fp64 copy the graph writes next to the loss)
        self.f_s2 = s2n if same else (self.host_read[1:2] if gp._dtype == torch.float64 else s2n)
        graph = torch.cuda.CUDAGraph()
        with _no_gc(), torch.cuda.graph(graph), torch.no_grad():
            grid_ops.hyper_columns(plan, grid, kind, ell, scale, s2)
            # (the value leaves the log-determinant out, as the eager step does under skip_logdet_forward; its gradient is taken)
            grid_ops.hyper_mid(sst["bMb"], None, s2, stats[0, 0], stats[0, 1], self.n_dev, mid, self.host_read[0:1])
            g_tcol, g_kap = fac.mll_backward(sst, mid[5], mid[6], kap=mid[8:9])
            g_ell, g_scale = grid_ops.stationary_columns_grad(grid, kind, ell, scale, g_tcol)
            grid_ops.hyper_adam(plan, scale, s2, g_ell, g_scale, mid, g_kap, self.n_dev, lr, b1, b2, eps)
            grid_ops.hyper_columns(plan, grid, kind, ell2, scale2, s2n, self.host_read[1:2], self.f_tc64, self.f_tc if same else None)
            if not same:
                self.f_tc.copy_(self.f_tc64)
        self.graph, self.loss, self.key = graph, self.host_read[0], key
        self.fused = True
        self.captures += 1
        self.fused_captures += 1
        self.replays_at_capture = self.replays

    def _capture(self, sps, key):
        fp = self._fused_plan() if len(sps) == 1 else None
        if fp is not None:
            return self._capture_fused(sps, key, fp)
        self.fused = False
        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        dev = sps[0][1]["G"].device
        f64 = dict(dtype=torch.float64, device=dev)
        self.bufs, static = [], []
        for fac, st, _ in sps:
            basis = st["basis"]
            r = basis.r
            sizes = [r, r, r, basis.ev_tab.numel(), basis.Vtab.numel(), 1, 1]
            buf = {"G": torch.empty((r, r), **f64), "Linv": torch.empty((r, r), **f64), "sqG": torch.empty((r, r), **f64),
                   "packed": torch.empty(sum(sizes), **f64)}
            v_sq, v_zeta, v_lam, v_ev, v_V, v_bMb, v_logdet = buf["views"] = torch.split(buf["packed"], sizes)
            sbasis = SpectralBasis.on_device(basis, v_V, v_ev.view(basis.ev_tab.shape), None, lam=v_lam)
            static.append((fac, {"basis": sbasis, "G": buf["G"], "sq": v_sq, "zeta": v_zeta, "coef": v_zeta, "Linv": buf["Linv"], "sqG": buf["sqG"], "bMb": v_bMb.reshape(()),
                                 "logdet": v_logdet, "kscale": None}, None))
            self.bufs.append(buf)
            fac._grid_dev()                           # (an upload: must exist before the capture)
        self.n_dev = torch.zeros(1, **f64)
        self._stage(sps)
        opt.zero_grad(set_to_none=True)               # the capture allocates the gradients in its own pool; replays rewrite them
        graph = torch.cuda.CUDAGraph()
        gp._graph_ctx = {"sp": static, "n": self.n_dev}
        try:
            # (enable_grad: evaluate() may have been called under torch.no_grad() -- the reference driver does -- and prepare() re-captures
            #  from there; a recording without autograd would fail and disable the graphed step for good)
            with _no_gc(), torch.enable_grad(), torch.cuda.graph(graph):
                with settings.skip_logdet_forward(True):
                    loss = -w.mll(None, None).sum()
                loss.backward()
                opt.step()
        finally:
            gp.__dict__.pop("_graph_ctx", None)
        self.graph, self.loss, self.key = graph, loss.detach(), key
        self.captures += 1
        self.replays_at_capture = self.replays
