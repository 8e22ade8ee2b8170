"""Batched fantasies -- SURVEY.md 8(f)-3.

What the reference is after (online_gp/models/batched_fixed_noise_online_gp.py:287-332,
online_ski_botorch_model.py:51-61, updated_root_lazy_tensor.py:139-159): expand every cache over
``num_fantasies x batch`` and condition each copy on its own q fantasy points, so that look-ahead acquisition
functions (qKG, qNIPV -- BASELINE config 5) can score a batch of candidate sets at once.  That code is broken at
HEAD, so this is specified from the maths instead and never copies an m x m cache:

    base posterior over the inducing values u:   u ~ N(mu, sigma2 M),   M = (Kt^-1 + A)^-1,  mu = M b
    candidate set j (q points, interpolation rows W_j, per-point noise diag(D_j)), fantasy sample f with targets y_fj:
        P_j   = W_j M                         [q, m]      (q rows of M interpolated: one gather of the dense M on small
                                                           grids, q extra right-hand sides of the PCG otherwise)
        S_j   = D_j + P_j W_j^T               [q, q]      Cholesky per candidate set (batched, q is tiny)
        mu_fj = mu + P_j^T S_j^-1 (y_fj - W_j mu)
        M_j   = M - P_j^T S_j^-1 P_j          (depends on the candidate set only, not on the sampled targets)
    posterior at queries X* with rows W*:
        mean_fj = W* mu + (W* P_j^T) S_j^-1 (y_fj - W_j mu)
        cov_j   = sigma2 [ W* M W*^T - (W* P_j^T) S_j^-1 (W* P_j^T)^T ]

which is exactly ``condition_on_observations`` applied to every (f, j) copy (tests/test_fantasy_gpu.py checks it
against a per-fantasy data-space oracle).  Everything that touches m goes through the HIP operators (wt_columns, the
dense M gather / PCG solves); what remains are batched q x q and q' x q products.
"""
import torch

from .. import grid_ops
from ..distributions import MultivariateNormal


def _batched_cholesky(S):
    """Lower Cholesky of a batch of small SPD matrices [..., q, q] (q is the size of a candidate set: a short
    column loop vectorised over the batch)."""
    q = S.shape[-1]
    L = torch.zeros_like(S)
    for j in range(q):
        d = S[..., j, j] - (L[..., j, :j] ** 2).sum(-1)
        if bool((d <= 0).any()):
            raise RuntimeError("fantasy covariance block is not positive definite")
        L[..., j, j] = d.sqrt()
        if j + 1 < q:
            L[..., j + 1:, j] = (S[..., j + 1:, j] - (L[..., j + 1:, :j] * L[..., j:j + 1, :j]).sum(-1)) / L[..., j:j + 1, j]
    return L


def _chol_solve(L, B):
    """S^-1 B for S = L L^T, batched: L [..., q, q], B [..., q, k]."""
    Y = torch.linalg.solve_triangular(L, B, upper=False)
    return torch.linalg.solve_triangular(L.transpose(-1, -2), Y, upper=True)


class MultiOutputFantasyPosterior:
    """Posterior of a batch of fantasy models with several independent outputs (BFN:37-55 carries `num_outputs` as a
    batch dimension; BoTorch's multi-output convention puts it last): ``mean`` / ``variance`` [*batch, q', out],
    ``mvn`` batch-first like the model's own eval forward ([out, *batch, q'] / [out, *batch, q', q'])."""

    def __init__(self, mvns):
        self.mvns = list(mvns)
        self.mvn = MultivariateNormal(torch.stack([v.mean for v in self.mvns]), torch.stack([v.covariance_matrix for v in self.mvns]))

    @property
    def mean(self):
        return torch.stack([v.mean for v in self.mvns], dim=-1)

    @property
    def variance(self):
        return torch.stack([v.variance for v in self.mvns], dim=-1)

    @property
    def device(self):
        return self.mvns[0].mean.device

    @property
    def dtype(self):
        return self.mvns[0].mean.dtype

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        return torch.stack([v.rsample(sample_shape) for v in self.mvns], dim=-1)


class BatchedFantasyModel:
    """A batch of conditioned copies of a ``FixedNoiseOnlineSKIGP``.

    Single output: ``batch_shape`` = targets.shape[:-1] = [num_fantasies] + inputs.shape[:-2] (or inputs.shape[:-2] when
    the targets carry no extra leading dimension).  Several outputs: targets (and noise) carry them as their LAST
    dimension, [..., q, out]; the outputs are independent GPs (own statistics, hyper-parameters and noise, BFN:37-55), so
    the batch is one single-output core per output on that output's posterior, and ``posterior(X)`` returns a
    :class:`MultiOutputFantasyPosterior`.  ``posterior(X)`` broadcasts X's leading dimensions against the batch."""

    def __new__(cls, base, inputs, targets, noise, _output=None):
        if base.num_outputs > 1 and _output is None:
            return object.__new__(_MultiOutputFantasyModel)
        return object.__new__(cls)

    def __init__(self, base, inputs, targets, noise, _output=None):
        o = 0 if _output is None else _output
        self._o = o
        self.base = base
        grid = base._grid
        dt, dev = base._dtype, base._device
        X = inputs.to(dev, dt)
        Y = targets.to(dev, dt)
        if Y.dim() >= 2 and Y.shape[-1] == 1 and Y.shape[-2] == X.shape[-2]:      # [..., q, 1] -> [..., q]
            Y = Y[..., 0]
        ib, tb = X.shape[:-2], Y.shape[:-1]
        if not (len(tb) == len(ib) + 1 or len(tb) == len(ib)) or tuple(tb[len(tb) - len(ib):]) != tuple(ib):
            raise RuntimeError(f"Unsupported batch shapes: The target batch shape ({tuple(tb)}) must have either the same dimension as or "
                               f"one more dimension than the input batch shape ({tuple(ib)})")       # BFN:292-296
        q = X.shape[-2]
        if Y.shape[-1] != q:
            raise RuntimeError("fantasy targets must have one value per fantasy input")
        N = noise.to(dev, dt) if noise is not None else torch.ones_like(Y)
        if N.dim() >= 2 and N.shape[-1] == 1 and N.shape[-2] == X.shape[-2]:
            N = N[..., 0]
        N = N.expand(Y.shape) if N.shape != Y.shape else N
        # the fixed noise of a candidate set may not depend on the fantasy sample (it enters S_j)
        Nj = N.reshape((-1,) + tuple(ib) + (q,))[0] if len(tb) == len(ib) + 1 else N
        self.batch_shape = torch.Size(tb)
        self.input_batch_shape = torch.Size(ib)
        self.q = q
        self.num_data = base.num_data + q

        pc = base.prediction_cache
        self._post = pc["pred_cov"].ops[o] if base.num_outputs > 1 else pc["pred_cov"]
        self._mu = pc["pred_mean"][o, :, 0].contiguous()                                   # [m]
        self._sigma2 = base._sigma2(o)
        Xf = X.reshape(-1, grid.d).contiguous()
        W = grid_ops.wt_columns(grid, Xf, base._err)                                       # [Bq, m] dense rows (small grids) ...
        P = self._apply_M(W)                                                               # ... and W M, row by row
        base.check_bounds()
        Bn = Xf.shape[0] // q
        self._W = W.reshape(Bn, q, grid.m)
        self._P = P.reshape(Bn, q, grid.m)
        S = torch.matmul(self._P, self._W.transpose(-1, -2))                               # W_j M W_j^T
        S = 0.5 * (S + S.transpose(-1, -2)) + torch.diag_embed(Nj.reshape(Bn, q).clamp_min(1e-7))
        self._L = _batched_cholesky(S)                                                     # [Bn, q, q]
        resid = Y - torch.matmul(self._W, self._mu).reshape(ib + (q,))                      # y_fj - W_j mu   [*tb, q]
        R = resid.reshape((-1, Bn, q)).permute(1, 2, 0)                                    # [Bn, q, F]
        self._alpha = _chol_solve(self._L, R)                                              # S_j^-1 resid    [Bn, q, F]
        self._F = R.shape[-1]
        self.train_inputs = [X]
        self.train_targets = Y

    def _apply_M(self, rows):
        """rows [k, m] -> rows M (M symmetric): the dense factor's M when it exists, chunked PCG solves otherwise."""
        out = []
        for s in range(0, rows.shape[0], 64):
            U, _ = self._post.solve_columns(rows[s:s + 64].contiguous())
            out.append(U)
        return torch.cat(out) if len(out) > 1 else out[0]

    def _lead(self):
        """(has a fantasy dimension, F, Bn)"""
        return len(self.batch_shape) == len(self.input_batch_shape) + 1, self._F, self._W.shape[0]

    def posterior(self, X, observation_noise=False, **kwargs):
        """X: [q', d] or [1.., q', d] shared by the whole batch, or [*input_batch_shape, q', d] per candidate set.  Returns a
        posterior whose mean / variance have shape [*batch_shape, q', 1] and whose covariance is [*batch_shape, q', q']."""
        from .online_ski_botorch_model import WiskiPosterior

        base, grid = self.base, self.base._grid
        dt, dev = base._dtype, base._device
        X = X.to(dev, dt)
        qq = X.shape[-2]
        lead = tuple(X.shape[:-2])
        shared = all(s == 1 for s in lead)
        has_f, F, Bn = self._lead()
        if not shared:
            lead_eff = lead[1:] if (has_f and len(lead) == len(self.batch_shape) and lead[0] == 1) else lead
            if tuple(lead_eff) != tuple(self.input_batch_shape):
                raise RuntimeError(f"query batch shape {lead} does not broadcast against the fantasy batch shape {tuple(self.batch_shape)}")
        Xf = X.reshape(-1, grid.d).contiguous()
        Wq = grid_ops.wt_columns(grid, Xf, base._err)                                      # [Bq * q', m]
        flag = grid_ops.read_flag(base._err)
        if flag:
            base._raise_out_of_bounds(flag)
        MWq = self._apply_M(Wq)                                                            # rows W* M
        Bq = Xf.shape[0] // qq
        Wq = Wq.reshape(Bq, qq, grid.m)
        MWq = MWq.reshape(Bq, qq, grid.m)
        mean0 = torch.matmul(Wq, self._mu)                                                 # [Bq, q']
        prior = torch.matmul(MWq, Wq.transpose(-1, -2))                                    # W* M W*^T      [Bq, q', q']
        K = torch.matmul(MWq, self._W.transpose(-1, -2)) if Bq == Bn else torch.matmul(MWq[0], self._W.transpose(-1, -2))   # W* M W_j^T [Bn, q', q]
        mean = mean0.unsqueeze(-1) + torch.matmul(K, self._alpha)                          # [Bn, q', F]
        KS = _chol_solve(self._L, K.transpose(-1, -2))                                     # S^-1 K^T       [Bn, q, q']
        cov = self._sigma2 * (prior - torch.matmul(K, KS))                                 # [Bn, q', q']
        cov = 0.5 * (cov + cov.transpose(-1, -2))
        if observation_noise:
            cov = cov + self._sigma2 * torch.eye(qq, dtype=dt, device=dev)
        ib = tuple(self.input_batch_shape)
        mean = mean.permute(2, 0, 1).reshape(((F,) if has_f else ()) + ib + (qq,))
        cov = cov.reshape(ib + (qq, qq))
        if has_f:
            cov = cov.unsqueeze(0).expand((F,) + ib + (qq, qq))
        return WiskiPosterior(MultivariateNormal(mean, cov))

    def __call__(self, X):
        return self.posterior(X).mvn

    def eval(self):
        return self

    def train(self, mode=True):
        return self


class _MultiOutputFantasyModel(BatchedFantasyModel):
    """num_outputs > 1: one single-output core per output (see BatchedFantasyModel)."""

    def __init__(self, base, inputs, targets, noise, _output=None):
        out = base.num_outputs
        Y = targets
        if Y.shape[-1] != out:
            raise RuntimeError(f"multi-output fantasy targets must be [..., q, {out}] (outputs last), got {tuple(Y.shape)}")
        N = noise
        if N is not None and (N.dim() == 0 or N.shape[-1] != out):
            N = N.unsqueeze(-1).expand(N.shape + (out,)) if N.dim() == Y.dim() - 1 else N.expand(Y.shape)
        self.base = base
        self.cores = [BatchedFantasyModel(base, inputs, Y[..., o], None if N is None else N[..., o], _output=o) for o in range(out)]
        c0 = self.cores[0]
        self.batch_shape, self.input_batch_shape, self.q, self.num_data = c0.batch_shape, c0.input_batch_shape, c0.q, c0.num_data
        self.train_inputs = c0.train_inputs
        self.train_targets = torch.stack([c.train_targets for c in self.cores], dim=-1)

    def posterior(self, X, observation_noise=False, **kwargs):
        return MultiOutputFantasyPosterior([c.posterior(X, observation_noise=observation_noise, **kwargs).mvn for c in self.cores])

    def __call__(self, X):
        return self.posterior(X).mvn
