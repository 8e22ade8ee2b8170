"""TEST INFRASTRUCTURE -- oracle "O": exact GP in data space on the SKI kernel.

    K = W Kuu W^T,   y ~ N(0, K + sigma2 * D)

This is the identity the reference's own tests pin WISKI against
(tests/mlls/test_batched_woodbury_marginal_log_likelihood.py:55-73 for the MLL,
tests/models/test_woodbury_gp_model.py:260-289 for mean and covariance); it is
independent of every WISKI cache formula.  Uses the Kronecker structure
K[i,j] = prod_dim (W_dim K_dim W_dim^T)[i,j], so it is valid for any grid size
(n x n work only).  numpy fp64; PARITY UNPINNED for W / Kuu (see spec.py).
"""
import numpy as np
import scipy.linalg as sla

from . import spec


class DataSpaceGP:
    def __init__(self, grid_bounds, grid_size, kind="rbf", lengthscale=spec.SOFTPLUS0,
                 outputscale=spec.SOFTPLUS0, sigma2=1.0, cols=None):
        self.g0, self.h, self.g = spec.make_grid(grid_bounds, grid_size)
        self.d = len(self.g)
        # cols: explicit Toeplitz columns per dim (kernels that are not of the k(r / lengthscale) form, e.g. the spectral mixture)
        self.cols = cols if cols is not None else spec.toeplitz_columns(kind, self.h, self.g, lengthscale, outputscale)
        self.Kd = [sla.toeplitz(c) for c in self.cols]
        self.sigma2 = float(sigma2)

    def _WK(self, X):
        X = np.asarray(X, dtype=np.float64).reshape(-1, self.d)
        Wd = [spec.interp_1d_dense(X[:, i], self.g0[i], self.h[i], int(self.g[i])) for i in range(self.d)]
        return Wd, [Wd[i] @ self.Kd[i] for i in range(self.d)]

    def cross(self, XA, XB):
        WA, WKA = self._WK(XA)
        WB, _ = self._WK(XB)
        K = np.ones((WA[0].shape[0], WB[0].shape[0]))
        for i in range(self.d):
            K *= WKA[i] @ WB[i].T
        return K

    def fit(self, X, y, noise):
        self.X = np.asarray(X, dtype=np.float64).reshape(-1, self.d)
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.noise = np.asarray(noise, dtype=np.float64).reshape(-1)
        K = self.cross(self.X, self.X)
        K[np.diag_indices_from(K)] += self.sigma2 * self.noise
        self.chol = sla.cho_factor(K, lower=True)
        self.alpha = sla.cho_solve(self.chol, self.y)
        return self

    def predict(self, Xs, full_cov=False):
        Ks = self.cross(Xs, self.X)
        mean = Ks @ self.alpha
        V = sla.solve_triangular(self.chol[0], Ks.T, lower=True)
        if full_cov:
            cov = self.cross(Xs, Xs) - V.T @ V
            return mean, cov
        _, WKs = self._WK(Xs)
        Ws, _ = self._WK(Xs)
        prior = np.ones(Ks.shape[0])
        for i in range(self.d):
            prior *= np.einsum("ij,ij->i", WKs[i], Ws[i])
        return mean, prior - np.einsum("ij,ij->j", V, V)

    def mll(self):
        """-(1/2)[quad + logdet + n log 2pi] / n  (online_gp/mlls/
        batched_woodbury_marginal_log_likelihood.py:26-51 computes the same value)."""
        n = self.y.shape[0]
        quad = float(self.y @ self.alpha)
        logdet = 2.0 * float(np.sum(np.log(np.diag(self.chol[0]))))
        return -0.5 * (quad + logdet + n * np.log(2.0 * np.pi)) / n
