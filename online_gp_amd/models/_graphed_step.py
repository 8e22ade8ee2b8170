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
    refresh -- it is copied into static staging buffers before each replay (two r x r copies and one packed ``cat``);
  * the data count n and 1 / sigma2 enter the recorded kernels as device scalars (n through a pinned one-element copy);
  * the index set S of the basis is used in place: the graph is re-captured when the factor re-selects it (or when the learning
    rate, the optimiser, the dtype or the set of priors changes).
The first steps run eagerly (they also serve as the warm-up the allocator wants before a capture); whenever the spectral path does
not apply -- dense regime, rough kernel, a foreign optimiser -- the caller's eager path runs instead.  Several outputs (the Dirichlet
classifier's two) are one graph: one factor, one set of staging buffers per output.
"""
import torch

from .. import settings
from ..lazy.spectral_woodbury import SpectralBasis

WARMUP_STEPS = 3


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

    # ---------------------------------------------------------------------------------------------------------------------
    def _applicable(self):
        """The live factor state if this step can run as a graph, else None."""
        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        if self.disabled is not None or settings.graphed_hyper_step.off() or settings.spectral_factor.off():
            return None
        if torch.device(gp._device).type != "cuda" or gp._use_dense() or w.mll.clear_caches_every_iteration:
            return None
        if not isinstance(opt, torch.optim.Adam) or not all(g.get("capturable") and g.get("fused") for g in opt.param_groups):
            return None
        if max(gp._grid.g) > 64:                      # (the one-launch lag gradient; larger factors take the op-by-op form)
            return None
        gp._finish_pending()
        gp.check_bounds()                             # the eager MLL does both first; neither may happen inside a capture
        sps = [gp._spectral_state(o) for o in range(gp.num_outputs)]      # one factor per output (own statistics, own hyper-parameters)
        return sps if all(sp is not None for sp in sps) else None

    def step(self, lazy=False):
        """One Adam step on -MLL; returns the loss, or None when the caller has to take the eager path.  lazy: return the device
        scalar instead of reading it (the read is a synchronisation: a caller with more launches to queue -- the absorb of the
        batch -- reads it after those; valid until the next replay)."""
        sps = self._applicable()
        if sps is None:
            return None
        if self.eager_calls < WARMUP_STEPS:
            self.eager_calls += 1
            return None
        w = self.w
        gp, opt = w.gp, w.gp_optimizer
        # everything the captured graph bakes in: the index sets, the optimiser's hyper-parameters (lr, betas, eps, weight decay ...),
        # the set of registered priors, the statistics buffer the MLL reads, which parameters are trainable
        from ..priors import REGISTRY_EPOCH

        def _hp(g):
            return tuple((k_, (tuple(v) if isinstance(v, (tuple, list)) else v)) for k_, v in sorted(g.items())
                         if k_ != "params" and isinstance(v, (int, float, bool, tuple, list, type(None))))

        key = (tuple((sp[1]["basis"].S.data_ptr(), sp[1]["basis"].r, sp[1]["basis"].kmax) for sp in sps), str(gp._dtype),
               tuple(_hp(g) for g in opt.param_groups), id(opt),
               settings.fused_hyper_columns.on(), tuple(p.requires_grad for g in opt.param_groups for p in g["params"]),
               REGISTRY_EPOCH[0], gp._kernel_cache["_stats"].data_ptr())
        if key != self.key:
            # a capture costs ~2 ms: worth it only if it is then replayed.  If the factor keeps re-selecting its index set (host-side
            # refresh, a kernel whose spectrum moves fast), stop re-capturing for a while and let the eager path run.
            if self.cooldown > 0:
                self.cooldown -= 1
                return None
            if self.key is not None and self.replays - self.replays_at_capture < 4:
                self.churn += 1
                if self.churn >= 3:
                    self.churn, self.cooldown, self.key, self.graph = 0, 64, None, None
                    return None
            else:
                self.churn = 0
            try:
                self._capture(sps, key)
            except Exception as exc:                  # a capture that cannot be made is not an error of the step: run eagerly from now on
                self.disabled = f"{type(exc).__name__}: {exc}"
                self.graph = self.key = None
                gp.__dict__.pop("_graph_ctx", None)
                return None
        self._stage(sps)
        self.graph.replay()
        self.replays += 1
        gp.zero_grad()                                # (as the eager step: drops the gradients and moves the hyper-parameter epoch on)
        return self.loss if lazy else float(self.loss)

    # ---------------------------------------------------------------------------------------------------------------------
    def _stage(self, sps):
        for (fac, st, _), buf in zip(sps, self.bufs):
            basis = st["basis"]
            fac.coefficients(st)
            buf["G"].copy_(st["G"])
            buf["Linv"].copy_(st["Linv"])
            buf["sqG"].copy_(st["sqG"])
            torch.cat([st["sq"], st["zeta"], basis.lam_kuu, basis.ev_tab.reshape(-1), basis.Vtab, st["bMb"].reshape(1), st["logdet"].reshape(1)],
                      out=buf["packed"])
        self.n_pin[0] = float(self.w.gp.num_data)
        self.n_dev.copy_(self.n_pin, non_blocking=True)

    def _capture(self, sps, key):
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
            v_sq, v_zeta, v_lam, v_ev, v_V, v_bMb, v_logdet = torch.split(buf["packed"], sizes)
            sbasis = SpectralBasis.on_device(basis, v_V, v_ev.view(basis.ev_tab.shape), None, lam=v_lam)
            static.append((fac, {"basis": sbasis, "G": buf["G"], "sq": v_sq, "zeta": v_zeta, "coef": v_zeta, "Linv": buf["Linv"], "sqG": buf["sqG"], "bMb": v_bMb.reshape(()),
                                 "logdet": v_logdet, "kscale": None}, None))
            self.bufs.append(buf)
            fac._grid_dev()                           # (an upload: must exist before the capture)
        self.n_dev = torch.zeros(1, **f64)
        self.n_pin = torch.zeros(1, dtype=torch.float64).pin_memory()
        self._stage(sps)
        opt.zero_grad(set_to_none=True)               # the capture allocates the gradients in its own pool; replays rewrite them
        graph = torch.cuda.CUDAGraph()
        gp._graph_ctx = {"sp": static, "n": self.n_dev}
        try:
            with torch.cuda.graph(graph):
                with settings.skip_logdet_forward(True):
                    loss = -w.mll(None, None).sum()
                loss.backward()
                opt.step()
        finally:
            gp.__dict__.pop("_graph_ctx", None)
        self.graph, self.loss, self.key = graph, loss.detach(), key
        self.captures += 1
        self.replays_at_capture = self.replays
