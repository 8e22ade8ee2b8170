"""TEST INFRASTRUCTURE -- baseline "B1": dense, op-for-op restatement of the
reference WISKI algorithm in numpy fp64 (valid for m <= ~4096).

Follows, formula by formula (citations relative to /root/reference):
  * caches            online_gp/models/batched_fixed_noise_online_gp.py:31-60
  * cache update      .../batched_fixed_noise_online_gp.py:155-171 (clamp_min(1e-7)**0.5 at :163)
  * root update       online_gp/lazy/updated_root_lazy_tensor.py:53-119 (SVD, some=False)
  * posterior algebra .../batched_fixed_noise_online_gp.py:334-404
  * eval forward      .../batched_fixed_noise_online_gp.py:204-228
  * MLL               online_gp/mlls/batched_woodbury_marginal_log_likelihood.py:19-51
  * stem loss         online_gp/mlls/streaming_partial_mll.py:6-62
The gpytorch pieces (interp weights, Kuu, Cholesky jitter escalation) are
restated from the published algorithm -- PARITY UNPINNED, see spec.py.
Single-output only (the batch-of-outputs layout is a python loop upstream).
"""
import numpy as np
import scipy.linalg as sla

from . import spec


def psd_safe_cholesky(A, jitter=1e-8, max_tries=6):
    """gpytorch.utils.cholesky.psd_safe_cholesky restated: try plain, then add
    jitter * 10**i to the diagonal (imported at updated_root_lazy_tensor.py:5)."""
    try:
        return np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        pass
    prev = 0.0
    Aj = A.copy()
    for i in range(max_tries):
        new = jitter * (10 ** i)
        Aj[np.diag_indices_from(Aj)] += new - prev
        prev = new
        try:
            return np.linalg.cholesky(Aj)
        except np.linalg.LinAlgError:
            continue
    raise np.linalg.LinAlgError("matrix not PSD even with jitter")


class DenseWISKI:
    def __init__(self, grid_bounds, grid_size, kind="rbf", lengthscale=spec.SOFTPLUS0,
                 outputscale=spec.SOFTPLUS0, sigma2=1.0, learn_additional_noise=True,
                 chol_jitter=1e-8):
        self.g0, self.h, self.g = spec.make_grid(grid_bounds, grid_size)
        self.d = len(self.g)
        self.m = int(np.prod(self.g))
        cols = spec.toeplitz_columns(kind, self.h, self.g, lengthscale, outputscale)
        Kuu = np.ones((1, 1))
        for c in cols:  # dim 0 slowest in the flat index
            Kuu = np.kron(Kuu, sla.toeplitz(c))
        self.Kuu_raw = Kuu
        self.sigma2 = float(sigma2) if learn_additional_noise else 1.0
        self.chol_jitter = chol_jitter
        self.root = self.inv_root = None
        self.num_data = 0

    # -- a1/a2: dense W^T (m x n), batched_fixed_noise_online_gp.py:22-28
    def wmat(self, X):
        X = np.asarray(X, dtype=np.float64).reshape(-1, self.d)
        W = np.ones((X.shape[0], 1))
        for i in range(self.d):
            Wi = spec.interp_1d_dense(X[:, i], self.g0[i], self.h[i], int(self.g[i]))
            W = (W[:, :, None] * Wi[:, None, :]).reshape(X.shape[0], -1)
        return W.T

    # -- a3: _initialize_caches, :31-60
    def set_train_data(self, X, y, noise):
        y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        noise = np.asarray(noise, dtype=np.float64).reshape(-1)
        wm = self.wmat(X)
        dinv_y = y / noise[:, None]
        self.response_cache = float((y.T @ dinv_y)[0, 0])
        self.interpolation_cache = wm @ dinv_y
        self.WtW = wm @ (wm.T / noise[:, None])
        self.D_logdet = float(np.sum(np.log(noise)))
        self.root = self.inv_root = None
        self.num_data = y.shape[0]

    # -- a4..a7: condition_on_observations(inplace=True), :258-273 + :155-171
    def condition_on_observations(self, X, y, noise=None):
        y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        noise = np.ones(y.shape[0]) if noise is None else np.asarray(noise, dtype=np.float64).reshape(-1)
        wm = self.wmat(X)
        dinv_y = y / noise[:, None]
        self.response_cache += float((y.T @ dinv_y)[0, 0])
        self.interpolation_cache = self.interpolation_cache + wm @ dinv_y
        self.D_logdet += float(np.sum(np.log(noise)))
        V = wm / np.sqrt(np.clip(noise, 1e-7, None))[None, :]
        self._root_update(V)                      # updated_root_lazy_tensor.py:62 (uses the OLD tensor's roots)
        self.WtW = self.WtW + V @ V.T             # updated_root_lazy_tensor.py:58
        self.num_data += y.shape[0]

    def _ensure_roots(self):
        if self.root is None:  # updated_root_lazy_tensor.py:121-133 -> Cholesky branch (m <= max_cholesky_size)
            L = psd_safe_cholesky(self.WtW, jitter=self.chol_jitter)
            self.root = L
            self.inv_root = sla.solve_triangular(L, np.eye(self.m), lower=True).T

    # -- a6: collect_vector, updated_root_lazy_tensor.py:69-119
    def _root_update(self, V):
        self._ensure_roots()
        p = self.inv_root.T @ V
        U, S, _ = np.linalg.svd(p, full_matrices=True)
        r = U.shape[0]
        s_plus = np.sqrt(S ** 2 + 1.0)
        stacked = np.concatenate([s_plus, np.ones(r - S.shape[0])])
        self.root = self.root @ (U * stacked[None, :])
        stacked_inv = np.concatenate([1.0 / s_plus, np.ones(r - S.shape[0])])
        self.inv_root = self.inv_root @ (U * stacked_inv[None, :])

    # -- a8..a13
    def _posterior_pieces(self):
        self._ensure_roots()
        Kt = self.Kuu_raw / self.sigma2                         # :338-340
        L = self.root
        KL = Kt @ L                                            # :343-348
        Q = L.T @ KL + np.eye(L.shape[1])                      # :350-355 (add_jitter(1.0))
        Kb = Kt @ self.interpolation_cache                     # :363-366
        proj = L.T @ Kb                                        # :357-361
        cq = sla.cho_factor(Q, lower=True)
        return Kt, L, KL, Q, Kb, proj, cq

    def pred_mean_cache(self):
        Kt, L, KL, Q, Kb, proj, cq = self._posterior_pieces()
        return Kb - KL @ sla.cho_solve(cq, proj)                # :375-376

    def pred_cov_cache(self):
        Kt, L, KL, Q, Kb, proj, cq = self._posterior_pieces()
        return Kt - KL @ sla.cho_solve(cq, KL.T)                # :399-403

    # -- a14: eval forward, :204-228 (+ sigma2 at :227-228)
    def predict(self, Xs, full_cov=False):
        wm = self.wmat(Xs)
        mean = (wm.T @ self.pred_mean_cache())[:, 0]
        cov = wm.T @ self.pred_cov_cache() @ wm * self.sigma2
        return (mean, cov) if full_cov else (mean, np.diag(cov).copy())

    # -- a17: BatchedWoodburyMarginalLogLikelihood.__call__, :19-51
    def mll(self):
        Kt, L, KL, Q, Kb, proj, cq = self._posterior_pieces()
        inner_qform = float((proj.T @ sla.cho_solve(cq, proj))[0, 0])
        inner_logdet = 2.0 * float(np.sum(np.log(np.diag(cq[0]))))
        inducing_qform = float((self.interpolation_cache.T @ Kb)[0, 0])
        inv_quad = (self.response_cache - inducing_qform) + inner_qform
        logdet = inner_logdet + self.D_logdet
        n = self.num_data
        final = n * np.log(2 * np.pi)
        if self.sigma2 != 1.0 or True:
            inv_quad = inv_quad / self.sigma2
            final = n * np.log(self.sigma2) + final
        return -0.5 * (inv_quad + logdet + final) / n

    # -- streaming_partial_mll.py:6-62 (one new point x', y')
    def sm_partial_mll(self, x_new, y_new):
        M = self.pred_cov_cache()
        w = self.wmat(x_new)[:, :1]
        new_Wy = self.interpolation_cache + w * float(y_new)
        v = M @ w
        MWy = M @ new_Wy
        div = 1.0 + float((v.T @ w)[0, 0])
        quad = float((new_Wy.T @ MWy)[0, 0]) - float((v.T @ new_Wy)[0, 0]) ** 2 / div
        quad /= self.sigma2
        return (quad - np.log(div)) / 2.0 / (self.num_data + 1)
