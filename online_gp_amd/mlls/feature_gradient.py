"""Gradient of the Woodbury MLL w.r.t. the *features* (the inputs of the interpolation), for learned stems.

The reference trains stem and GP jointly in ``OnlineSKIRegression.fit`` by back-propagating the MLL through the
dense interpolation matrix W(features) (online_gp/models/online_ski_regression.py:80-112).  Here the statistics are
accumulated by kernels outside autograd, so the feature gradient is written out.  With M = (Kt^-1 + A)^-1,
mu = M b, and per point p the interpolation row w_p, weights wb_p = wa_p = 1/noise_p (BFN:44-53):

    -2 n MLL = (c - b^T M b) / sigma2 + logdet(I + Kt A) + logdet D + n log(2 pi sigma2)          (BWM:26-51)
    d(-b^T M b)        = -2 mu^T db + mu^T dA mu,   db = wb_p y_p dw_p,   dA = wa_p (dw_p w_p^T + w_p dw_p^T)
    d logdet(I + Kt A) = tr(M dA) = 2 wa_p dw_p^T (M w_p)

so  d(-2 n MLL)/dw_p = -(2 / sigma2) (wb_p y_p - wa_p w_p^T mu) mu + 2 wa_p M w_p, and the chain through
dw_p/dx_p is wiski_gather_grad (``grid_ops.InterpDot``).  ``mll_feature_surrogate`` returns a scalar whose value is
zero and whose gradient w.r.t. ``features`` is d(-MLL)/d features, to be added to the loss.

Dense regime (m <= settings.max_cholesky_size -- the reference's stems map to a 2-D 16 x 16 grid): the rows M w_p of every
training point are one GEMM.  Beyond it that would be n solves, so the trace term is estimated the way the hyper-parameter
gradient of the same logdet is (BWM, Hutchinson): with probes z_j, E[z z^T] = I,

    dw_p^T M w_p  ~=  (1/P) sum_j (dw_p^T s_j) (z_j^T w_p),      s_j = M z_j      (P solves, whatever n)

-- two interpolations per probe and point (``wiski_gather`` of z_j, ``wiski_gather_grad`` of s_j).  The data term needs no
estimate.  ``probes`` / ``probe_weight`` let a caller pass its own z_j (the identity with weight 1 makes the estimate exact)."""
import torch

from .. import grid_ops, settings


def mll_feature_surrogate(model, features, targets, noise=None, chunk=4096, probes=None, probe_weight=None):
    """features [n, d] (requires grad), targets [n, out], noise [n, out] or None (= ones).  The model's statistics must
    have been built from ``features.detach()`` (``set_train_data``).  probes [P, m] / probe_weight: matrix-free regime only."""
    dense = model._use_dense()
    grid = model._grid
    dev, dt = model._device, model._dtype
    x = features.reshape(-1, grid.d).to(dev, dt)
    n = x.shape[0]
    y = targets.to(dev, dt).reshape(n, -1)
    with torch.no_grad():
        pc = model.prediction_cache
    total = x.new_zeros(())
    for o in range(model.num_outputs):
        post = pc["pred_cov"].ops[o] if model.num_outputs > 1 else pc["pred_cov"]
        mu = pc["pred_mean"][o, :, 0].detach()
        s2 = model._sigma2(o)
        w = torch.ones(n, dtype=dt, device=dev) if noise is None else 1.0 / noise.to(dev, dt).reshape(n, -1)[:, o]
        if not dense:
            from .batched_woodbury_marginal_log_likelihood import num_trace_samples

            if probes is None:
                gen = torch.Generator(device=dev).manual_seed(0xFEA7 + int(model.num_data) + o)
                Zp = torch.randint(0, 2, (num_trace_samples.value(), grid.m), generator=gen, device=dev).to(dt) * 2 - 1
            else:
                Zp = probes.to(dev, dt).reshape(-1, grid.m)
            P = Zp.shape[0]
            pw = (1.0 / P) if probe_weight is None else float(probe_weight)
            cc = settings.variance_chunk.value()
            with torch.no_grad():
                Sp = torch.cat([post.solve_columns(Zp[j:j + cc].contiguous())[0] for j in range(0, P, cc)])   # s_j = M z_j
        for s in range(0, n, chunk):
            xs = x[s:s + chunk]
            alpha = grid_ops.InterpDot.apply(grid, xs, mu, False, model._err)             # w_p^T mu      (grad: mu^T dw_p)
            if dense:
                with torch.no_grad():
                    W = grid_ops.wt_columns(grid, xs.detach().contiguous(), model._err)   # [q, m]
                    U, _ = post.solve_columns(W)                                           # rows M w_p
                s_half = grid_ops.InterpDot.apply(grid, xs, U, True, model._err)          # w_p^T (M w_p) (grad: (M w_p)^T dw_p)
            else:
                with torch.no_grad():
                    Bz = grid_ops.gather(grid, xs.detach().contiguous(), Zp, model._err)  # [q, P]: z_j^T w_p
                s_half = 0.0
                for j in range(P):
                    s_half = s_half + pw * Bz[:, j] * grid_ops.InterpDot.apply(grid, xs, Sp[j], False, model._err)
            ws = w[s:s + chunk]
            c1 = (ws * y[s:s + chunk, o] - ws * alpha).detach()
            total = total + (-(2.0 / s2) * c1 * alpha + 2.0 * ws * s_half).sum()
    surr = 0.5 * total / float(model.num_data)
    return surr - surr.detach()
