"""BatchedWoodburyMarginalLogLikelihood -- the WISKI marginal log-likelihood from the
caches alone (reference online_gp/mlls/batched_woodbury_marginal_log_likelihood.py:6-51).

Reference (root space, L L^T = W^T D^-1 W, Q = I + L^T Kt L, Kt = Kuu / sigma2):
    inv_quad = (y^T D^-1 y - b^T Kt b) + (L^T Kt b)^T Q^-1 (L^T Kt b)       (BWM:27-34)
    logdet   = logdet(Q) + logdet(D)                                         (BWM:35)
    res      = -0.5 * (inv_quad / sigma2 + logdet + n log sigma2 + n log 2 pi) / n   (BWM:37-51)
Matrix-free equivalents used here (same values):
    inv_quad = y^T D^-1 y - b^T M b,  M = (Kt^-1 + A)^-1     (one wiski_pcg solve, mu = M b, z = Kt^-1 mu)
    logdet(Q) = logdet(I + Kt A) = logdet(I + Kt^1/2 A Kt^1/2)   (stochastic Lanczos quadrature on the symmetric form)
Gradients w.r.t. the hyper-parameters flow through torch autograd into the Toeplitz columns and sigma2:
    d(b^T M b)   = z^T dKt z                                 (wiski_kron_toeplitz_grad on (z, z))
    d logdet(Q)  = tr(S dKt),  S = (I + A Kt)^-1 A           (Hutchinson: s_j = Z-output of wiski_pcg for rhs A e_j)
Like the reference, the `distro` and `targets` arguments are ignored: only `model._kernel_cache` is read.
"""
import math

import torch

from .. import grid_ops, settings
from ..priors import named_priors


class num_trace_samples(settings._value_context):
    """Probe vectors for the logdet gradient / SLQ value (gpytorch default: 10)."""

    _global_value = 10


class fixed_trace_probes(settings._feature_flag):
    """Matrix-free MLL: keep the Rademacher probe vectors of the trace estimator fixed from step to step (a sample-average
    objective: the same probes score every hyper-parameter setting) and warm-start their solves from the previous step's
    solutions -- streaming steps change (A, hyper-parameters) a little at a time, so the probe solves then cost 2-3 CG
    iterations instead of a cold solve each.  A fixed probe set is a sample-average objective whose Hutchinson error does not
    average out over steps, so the set is redrawn every ``trace_probe_refresh_every`` steps (one cold multi-column solve then).
    Off: fresh probes seeded by the data count at every step, cold solves (gpytorch's habit)."""

    _state = True


class trace_probe_refresh_every(settings._value_context):
    """Steps a fixed set of trace probes is kept before it is redrawn (``fixed_trace_probes``)."""

    _global_value = 16


def _current_dense_posterior(model, o, A):
    """The model's cached dense posterior factor of output o if it belongs to the CURRENT hyper-parameters and statistics, else None.
    The model drops its prediction cache whenever data are absorbed or zero_grad() / hyperparameters_changed() is called, and stamps
    it with the parameters' version counters (models/batched_fixed_noise_online_gp.py: prediction_cache, _hyper_version)."""
    from ..lazy.dense_woodbury import DenseInducingPosterior

    memo = getattr(model, "_memo", None)
    if not memo or "pending_rank_update" in memo:
        return None
    pc = memo.get("prediction_cache")
    if pc is None or pc.get("ver") != model._hyper_version():
        return None
    cov = pc.get("pred_cov")
    cand = cov.ops[o] if hasattr(cov, "ops") else (cov if o == 0 else None)
    if isinstance(cand, DenseInducingPosterior) and cand.wtw is A and cand.dense is not None:
        return cand
    return None


class _WoodburyTerms(torch.autograd.Function):
    """(b^T M b, logdet(I + Kt A)) as a differentiable function of (tcol, kappa = 1/sigma2)."""

    @staticmethod
    def forward(ctx, tcol64, kappa, model, o, want_logdet):
        grid = model._grid
        dt, dev = model._dtype, model._device
        A = model._kernel_cache["WtW"]
        A = A.ops[o] if hasattr(A, "ops") else A
        b = model._kernel_cache["interpolation_cache"][o, :, 0]
        m = grid.m
        dense = model._use_dense()
        ctx.spectral = None
        sp = model._spectral_state(o) if model._spectral_allowed() else None
        if sp is not None:
            # smooth kernel on a large grid: both terms, exactly differentiable, from the dense factor in the dominant
            # Kronecker eigenspace (lazy/spectral_woodbury.py) -- no probe vectors, no solves
            fac, st, _ = sp
            ctx.spectral = (fac, st)
            logdet = st["logdet"].clone() if want_logdet else torch.zeros((), dtype=torch.float64, device=dev)
            return st["bMb"].clone(), logdet
        tcol = tcol64.detach().to(dt).contiguous()
        post = _current_dense_posterior(model, o, A) if dense else None
        # (the reference's loop evaluates a batch and then takes its MLL step at the same hyper-parameters and data: the posterior
        # factor evaluate() has just built -- Kronecker eigenbasis on the host, Cholesky + inverse, two m x m GEMMs -- IS this step's)
        kap = post.kscale if post is not None else float(kappa.detach())
        # plain eigenbasis of Kt: the dense factor and the SLQ logdet need it; the streaming hyper step
        # (skip_logdet_forward, large grid) does not -- 3 host eigh + an upload saved per step
        eig = grid_ops.kron_eigen(grid, tcol) if ((dense and post is None) or (want_logdet and not dense)) else None
        if dense:
            # small grid: everything from the dense factor, exact trace (S = A - A M A by Woodbury)
            from ..lazy.dense_woodbury import DenseInducingPosterior

            if post is None:
                post = DenseInducingPosterior(grid, A, tcol, kap, eig)
            U, _ = post.solve_columns(b[None])
            Z = b[None] - grid_ops.stencil_spmv(grid, A.stencil, U)              # z = Kt^-1 mu = b - A mu
            Ad = A.evaluate().contiguous()
            AM = grid_ops.gemm(Ad, post.dense)
            S_cols = Ad - grid_ops.gemm(AM, Ad)
            E = torch.eye(m, dtype=dt, device=dev)
            P = 1
            bMb = (b.double() * U[0].double()).sum()
            logdet = post.logdet.clone() if want_logdet else torch.zeros((), dtype=torch.float64, device=dev)
            ctx.grid, ctx.P, ctx.kap = grid, 1, kap
            ctx.save_for_backward(tcol, Z[0].clone(), U[0].clone(), S_cols, E)
            return bMb, logdet
        # same density-profile preconditioner as the posterior refresh (re-solved because the hypers moved)
        peig, shift = model._precond(o, tcol)
        if peig is None:
            peig, shift = (eig if eig is not None else grid_ops.kron_eigen(grid, tcol)), float(model._wsum[o]) / grid.m
        tol = settings.cg_tolerance.value() or (1e-7 if dt == torch.float32 else 1e-11)
        kw = dict(tol=tol, max_iter=settings.max_cg_iterations.value(), check_every=settings.cg_check_every.value(), workspace=model._pcg_ws,
                  eigen=peig, shift=shift)
        ms = model._mean_state
        P = num_trace_samples.value()
        chunk = settings.variance_chunk.value()
        if fixed_trace_probes.on() and P + 1 <= chunk:
            # ONE multi-column solve per step: column 0 is the mean (rhs b, warm-started from the last posterior mean),
            # columns 1..P the trace probes (rhs A e_j, FIXED e_j, warm-started from the previous step's solutions: keep
            # the pre-images z, re-derive u = Kt_new z).  The k = P + 1 products read A_h once per 4 columns.
            ps = model.__dict__.get("_mll_probes", {}).get(o)
            if (ps is None or ps["E"].shape != (P, m) or ps["E"].dtype != dt or ps["E"].device != dev
                    or ps["age"] >= trace_probe_refresh_every.value()):
                draw = 0 if ps is None else ps["draw"] + 1
                gen = torch.Generator(device=dev).manual_seed(0x5EED + o + 7919 * draw)
                E = torch.randint(0, 2, (P, m), generator=gen, device=dev).to(dt) * 2 - 1
                ps = {"E": E, "Z": None, "age": 0, "draw": draw}
                model.__dict__.setdefault("_mll_probes", {})[o] = ps
            ps["age"] += 1
            E = ps["E"]
            RHS = torch.cat([b[None], grid_ops.stencil_spmv(grid, A.stencil, E)])
            warm = ms is not None and ps["Z"] is not None
            U0 = Z0 = None
            if warm:
                Z0 = torch.cat([ms["Z"][o:o + 1], ps["Z"]])
                U0 = grid_ops.kron_toeplitz_mm(grid, tcol, Z0, kap)
            Uall, Zall, _, _ = grid_ops.pcg(grid, A.stencil, tcol, kap, RHS, U=U0, Z=Z0, warm=warm, **kw)
            U, Z = Uall[:1], Zall[:1]
            S_cols = Zall[1:].clone()
            ps["Z"] = S_cols
        else:
            # warm start from the last posterior mean: keep the pre-image z, re-derive u = Kt_new z
            U0 = Z0 = None
            if ms is not None:
                Z0 = ms["Z"][o:o + 1].clone()
                U0 = grid_ops.kron_toeplitz_mm(grid, tcol, Z0, kap)
            U, Z, _, _ = grid_ops.pcg(grid, A.stencil, tcol, kap, b[None], U=U0, Z=Z0, warm=U0 is not None, **kw)
            # probes for tr(S dKt): Rademacher (Hutchinson), fresh per data count
            gen = torch.Generator(device=dev).manual_seed(0x5EED + model.num_data)
            E = torch.randint(0, 2, (P, m), generator=gen, device=dev).to(dt) * 2 - 1
            S_cols = torch.empty_like(E)
            for s in range(0, P, chunk):
                rhs = grid_ops.stencil_spmv(grid, A.stencil, E[s:s + chunk])
                _, Zs, _, _ = grid_ops.pcg(grid, A.stencil, tcol, kap, rhs, **kw)
                S_cols[s:s + chunk] = Zs
        bMb = (b.double() * U[0].double()).sum()
        P = E.shape[0]
        logdet = torch.zeros((), dtype=torch.float64, device=dev)
        if want_logdet:
            logdet = _logdet_value(grid, A, eig, kap, E, False)
        ctx.grid, ctx.P, ctx.kap = grid, P, kap   # Rademacher probes average to the trace
        ctx.save_for_backward(tcol, Z[0].clone(), U[0].clone(), S_cols, E)
        return bMb, logdet

    @staticmethod
    def backward(ctx, g_bMb, g_logdet):
        if ctx.spectral is not None:
            fac, st = ctx.spectral
            g_tcol, g_kap = fac.mll_backward(st, g_bMb, g_logdet)
            return (g_tcol if ctx.needs_input_grad[0] else None), (g_kap if ctx.needs_input_grad[1] else None), None, None, None
        tcol, z, u, S_cols, E = ctx.saved_tensors
        grid, P, kap = ctx.grid, ctx.P, ctx.kap
        # columns [z | s_1..s_P] against [z | e_1..e_P], weighted by the incoming gradients
        X = torch.cat([z[None] * float(g_bMb), S_cols * (float(g_logdet) / P)])
        Y = torch.cat([z[None], E])
        g_tcol = None
        if ctx.needs_input_grad[0]:
            g_tcol = kap * grid_ops.kron_toeplitz_grad(grid, tcol, X, Y)
        g_kap = None
        if ctx.needs_input_grad[1]:
            KY = grid_ops.kron_toeplitz_mm(grid, tcol, Y, 1.0)
            g_kap = (X.double() * KY.double()).sum()
        return g_tcol, g_kap, None, None, None


def _logdet_value(grid, A, eig, kap, E, exact, lanczos_steps=40):
    """logdet(I + G A G), G = Kt^(1/2): exact via the dense m x m matrix for small grids
    (E is the identity there), otherwise stochastic Lanczos quadrature with probes E."""
    dev = E.device

    def C(V):
        GV = grid_ops.kron_spectral_mm(grid, eig, V, kscale=kap, power=0.5)
        AGV = grid_ops.stencil_spmv(grid, A.stencil, GV)
        return V + grid_ops.kron_spectral_mm(grid, eig, AGV, kscale=kap, power=0.5)

    if exact:
        Cm = C(E)
        Cm = (0.5 * (Cm + Cm.t())).contiguous()
        return grid_ops.chol_logdet(grid_ops.psd_safe_cholesky(Cm)).to(dev)
    P, m = E.shape
    Q_prev = torch.zeros_like(E)
    q = E / E.norm(dim=1, keepdim=True)
    alphas, betas = [], []
    beta = torch.zeros(P, dtype=torch.float64, device=dev)
    for _ in range(lanczos_steps):
        w = C(q) - beta.to(q.dtype)[:, None] * Q_prev
        alpha = (w.double() * q.double()).sum(1)
        w = w - alpha.to(q.dtype)[:, None] * q
        beta = w.double().norm(dim=1)
        alphas.append(alpha)
        betas.append(beta)
        Q_prev, q = q, w / beta.clamp_min(1e-300).to(q.dtype)[:, None]
    al = torch.stack(alphas, 1).cpu()
    be = torch.stack(betas, 1).cpu()
    est = 0.0
    for j in range(P):
        T = torch.diag(al[j]) + torch.diag(be[j, :-1], 1) + torch.diag(be[j, :-1], -1)
        ev, evec = torch.linalg.eigh(T)
        est += float((evec[0] ** 2 * ev.clamp_min(1e-300).log()).sum())
    return torch.tensor(est / P * m, dtype=torch.float64, device=dev)


class _SpectralMll(torch.autograd.Function):
    """One output's MLL term  -1/2 ((c - b^T M b) / s2 + logdet + ld + n log 2 pi + n log s2)  (BWM:34-47) from the spectral
    Woodbury factor, as a function of (Toeplitz columns, sigma2): the factor's two numbers and the scalar arithmetic around them in
    one launch forward (wiski_mll_value), the factor's reduced-basis gradient + one launch backward -- instead of ~15
    zero-dimensional framework ops and their ~20 autograd nodes per step."""

    @staticmethod
    def forward(ctx, tcol64, s2, model, o, want_logdet, sp, n):
        """n: the data count, a Python number -- or a device scalar when the step is recorded into a captured graph
        (models/_streaming_wrapper.py), in which case 1 / sigma2 is taken from the device as well."""
        fac, st, _ = sp
        stats = model._kernel_cache["_stats"]
        s2d = s2.detach()
        val, coef = grid_ops.mll_value(st["bMb"], st["logdet"] if want_logdet else None, s2d, stats[o, 0], stats[o, 1], n)
        ctx.fac, ctx.st, ctx.n = fac, st, n
        ctx.kap = (1.0 / s2d.double()).reshape(1) if torch.is_tensor(n) else None
        ctx.save_for_backward(coef, s2d)
        return val

    @staticmethod
    def backward(ctx, g):
        coef, s2d = ctx.saved_tensors
        gab = coef[:2] * g
        g_tcol, g_kap = ctx.fac.mll_backward(ctx.st, gab[0], gab[1], kap=ctx.kap)
        g_s2 = grid_ops.mll_s2_grad(g.contiguous(), coef, s2d, ctx.n, g_kap) if ctx.needs_input_grad[1] else None
        return (g_tcol if ctx.needs_input_grad[0] else None), g_s2, None, None, None, None, None


class BatchedWoodburyMarginalLogLikelihood(torch.nn.Module):
    def __init__(self, likelihood, model, clear_caches_every_iteration=False):
        super().__init__()
        self.likelihood = likelihood
        self.model = model
        self.has_learnable_noise = self.likelihood.second_noise_covar is not None
        self.clear_caches_every_iteration = clear_caches_every_iteration

    def forward(self, distro=None, targets=None, *args):
        model = self.model
        # graph context (models/_streaming_wrapper.py): the call is being recorded into a captured graph -- the factor state comes
        # from static staging buffers, the data count from a device scalar, and nothing here may touch the host-device boundary
        gctx = model.__dict__.get("_graph_ctx")
        if gctx is None:
            model._finish_pending()  # the solves below share the model's PCG workspace with a deferred refresh still in flight
            if self.clear_caches_every_iteration:
                model.zero_grad()
            model.check_bounds()
        cache = model._kernel_cache
        n = model.num_data if gctx is None else gctx["n"]
        want_logdet = settings.skip_logdet_forward.off()
        out = []
        for o in range(model.num_outputs):
            bi = o if model.num_outputs > 1 else None
            tcol = model.covar_module.toeplitz_columns(batch_index=bi, device=model._device)          # float64, differentiable
            if self.has_learnable_noise:
                s2 = self.likelihood.second_noise_covar.noise.reshape(-1)
                s2 = s2[o] if s2.numel() > 1 else s2[0]
            else:
                s2 = torch.ones((), dtype=torch.float64, device=model._device)
            sp = gctx["sp"][o] if gctx is not None else (model._spectral_state(o) if model._spectral_allowed() else None)
            if sp is not None:
                out.append(_SpectralMll.apply(tcol, s2, model, o, want_logdet, sp, n))
                continue
            s2 = s2.double()
            bMb, logdet_q = _WoodburyTerms.apply(tcol, 1.0 / s2, model, o, want_logdet)
            c = cache["_stats"][o, 0]
            ld = cache["_stats"][o, 1]
            inv_quad = (c - bMb) / s2                                               # BWM:34,43
            final = n * math.log(2 * math.pi) + n * torch.log(s2)                   # BWM:40-47 (log sigma2 = 0 when fixed)
            out.append(-0.5 * (inv_quad + logdet_q + ld + final))                  # BWM:46
        res = torch.stack(out)
        # registered hyper-parameter priors: + sum of log-densities, before the division by n (BWM:48-51)
        for _, prior, closure in named_priors(self):
            res = res + prior.log_prob(closure()).sum().to(res)
        res = res / n
        return res[0] if model.num_outputs == 1 else res
