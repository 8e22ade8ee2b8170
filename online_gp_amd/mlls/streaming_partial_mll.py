"""sm_partial_mll -- Sherman-Morrison one-point MLL increment used as the stem loss
(reference online_gp/mlls/streaming_partial_mll.py:6-62).

With M = (Kt^-1 + A)^-1 (pred_cov, detached), b = W^T D^-1 y (detached), w = w(x') and
new_Wy = b + w y'  (SPM:24-26):
    v = M w,  div = 1 + v^T w,                                  (SPM:31-36)
    quad = [ new_Wy^T M new_Wy - (v^T new_Wy)^2 / div ] / sigma2     (SPM:45-53)
    partial_mll = (quad - log div) / 2 / (num_seen + 1)               (SPM:57-62)
Everything depends on x' only through the scalars alpha = w^T mu (mu = M b) and
s = w^T M w, which are evaluated by the fused gather kernels; their input gradients
(d alpha = mu^T dw, d s = 2 (M w)^T dw) come from wiski_gather_grad, so the loss is
differentiable w.r.t. the features a learned stem produces.  A batch of q points is
treated as q independent one-point increments (identical to the reference for its
batch_size = 1 configuration)."""
import torch

from .. import grid_ops, settings


def sm_partial_mll(ski_gp, new_x, new_y, num_seen):
    grid = ski_gp._grid
    dev, dt = ski_gp._device, ski_gp._dtype
    x = new_x.reshape(-1, grid.d).to(dev, dt)
    q = x.shape[0]
    y = new_y.to(dev, dt)
    y = y.reshape(ski_gp.num_outputs, q) if y.numel() == ski_gp.num_outputs * q and y.dim() >= 2 and y.shape[0] == ski_gp.num_outputs else y.reshape(q, -1).t()
    with settings.skip_posterior_variances(False), torch.no_grad():
        pc = ski_gp.prediction_cache
    outs = []
    for o in range(ski_gp.num_outputs):
        post = pc["pred_cov"].ops[o] if ski_gp.num_outputs > 1 else pc["pred_cov"]
        mu = pc["pred_mean"][o, :, 0].detach()
        b = ski_gp._kernel_cache["interpolation_cache"][o, :, 0].detach()
        s2 = ski_gp._sigma2(o)
        with torch.no_grad():
            W = grid_ops.wt_columns(grid, x.detach().contiguous(), ski_gp._err)          # [q, m]
            U, _ = post.solve_columns(W)                                                  # rows u_p = M w_p
            bMb = (b.double() * mu.double()).sum().to(dt)
        alpha = grid_ops.InterpDot.apply(grid, x, mu, False, ski_gp._err)               # w_p^T mu
        s_half = grid_ops.InterpDot.apply(grid, x, U, True, ski_gp._err)                # w_p^T u_p with u_p fixed
        s = 2.0 * s_half - s_half.detach()                                               # value w^T M w, gradient 2 u^T dw
        yo = y[o]
        div = 1.0 + s
        quad = (bMb + 2.0 * yo * alpha + yo * yo * s - (alpha + yo * s) ** 2 / div) / s2
        outs.append((quad - torch.log(div)) / 2.0 / (num_seen + 1))
    res = torch.stack(outs)                                                               # [out, q]
    return res
