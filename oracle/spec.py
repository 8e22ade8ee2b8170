"""TEST INFRASTRUCTURE -- grid / kernel specification shared by the oracles.

PARITY UNPINNED: these pieces restate un-vendored gpytorch (reference
requirements.txt:7, unpinned, absent from this image) from its published
algorithm:

* grid: per dim ``delta = (hi-lo)/(g-2)``, ``grid = linspace(lo-delta, hi+delta, g)``
  (gpytorch GridInterpolationKernel; call site
  online_gp/models/batched_fixed_noise_online_gp.py:114-120);
* Kuu = kron_i K_i with K_i[a,b] = s * k(|a-b| h_i / ell_i)  (ScaleKernel
  inside GridInterpolationKernel => overall s**d, batched_fixed_noise_online_gp.py:107-120);
* default raw hyper-parameters 0 => softplus(0) = 0.6931.
"""
import numpy as np

SOFTPLUS0 = float(np.log(2.0))


def make_grid(grid_bounds, grid_size):
    """Return (g0[d], h[d], g[d]) : first grid point, spacing, points per dim."""
    gb = np.asarray(grid_bounds, dtype=np.float64).reshape(-1, 2)
    d = gb.shape[0]
    g = np.asarray([grid_size] * d if np.isscalar(grid_size) else grid_size, dtype=np.int32)
    assert g.shape[0] == d
    delta = (gb[:, 1] - gb[:, 0]) / (g - 2)
    g0 = gb[:, 0] - delta
    h = (gb[:, 1] + delta - g0) / (g - 1)
    return g0, h, g


def stationary_profile(kind, r):
    """k(r) for r = |x-x'|/ell >= 0."""
    r = np.asarray(r, dtype=np.float64)
    if kind == "rbf":
        return np.exp(-0.5 * r * r)
    if kind == "matern52":
        s = np.sqrt(5.0) * r
        return (1.0 + s + s * s / 3.0) * np.exp(-s)
    if kind == "matern32":
        s = np.sqrt(3.0) * r
        return (1.0 + s) * np.exp(-s)
    if kind == "matern12":
        return np.exp(-r)
    raise ValueError(kind)


def toeplitz_columns(kind, h, g, lengthscale, outputscale):
    """First columns of the d symmetric-Toeplitz Kronecker factors, concatenated."""
    d = len(g)
    ell = np.broadcast_to(np.asarray(lengthscale, dtype=np.float64), (d,))
    cols = []
    for i in range(d):
        lag = np.arange(int(g[i]), dtype=np.float64) * h[i]
        cols.append(outputscale * stationary_profile(kind, lag / ell[i]))
    return cols


def keys_cubic(s):
    a = np.abs(s)
    near = ((1.5 * a - 2.5) * a) * a + 1.0
    far = ((-0.5 * a + 2.5) * a - 4.0) * a + 2.0
    return np.where(a <= 1.0, near, np.where(a < 2.0, far, 0.0))


def interp_1d_dense(x, g0, h, g):
    """Dense per-dim interpolation matrix [n, g] (independent numpy statement of
    the same rule as the C oracle, used by the data-space oracle)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if np.any(x < g0) or np.any(x > g0 + h * (g - 1)):
        raise RuntimeError("Received data that was out of bounds for the specified grid.")
    u = (x - g0) / h
    fl = np.floor(u)
    t = u - fl
    j0 = fl.astype(np.int64) - 1
    w = np.stack([keys_cubic(t + 1), keys_cubic(t), keys_cubic(t - 1), keys_cubic(t - 2)], axis=1)
    Wd = np.zeros((n, g))
    for p in range(n):
        if j0[p] < 0 or j0[p] > g - 4:
            base = 0 if j0[p] < 0 else g - 4
            dist = np.abs(g0 + h * (base + np.arange(4)) - x[p])
            Wd[p, base + int(np.argmin(dist))] = 1.0
        else:
            Wd[p, j0[p]:j0[p] + 4] = w[p]
    return Wd


def spectral_mixture_columns(h, g, weights, means, scales):
    """First columns of the per-dim Toeplitz factors of gpytorch's SpectralMixtureKernel on the inducing grid
    (restated from its published formula; the reference's 1-D notebook, notebooks/regression_viz_1D.ipynb, uses
    ``SpectralMixtureKernel(num_mixtures=3)``): column_d(tau) = sum_q w_q exp(-2 pi^2 tau^2 s_qd^2) cos(2 pi tau mu_qd).
    weights [Q]; means, scales [Q, d] (or [Q, 1], shared by the dims)."""
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    mu = np.asarray(means, dtype=np.float64).reshape(len(w), -1)
    sc = np.asarray(scales, dtype=np.float64).reshape(len(w), -1)
    cols = []
    for i in range(len(g)):
        tau = np.arange(int(g[i]), dtype=np.float64) * h[i]
        dq = i if mu.shape[1] > 1 else 0
        cols.append((w[:, None] * np.exp(-2.0 * np.pi ** 2 * (tau[None] * sc[:, dq:dq + 1]) ** 2) * np.cos(2.0 * np.pi * tau[None] * mu[:, dq:dq + 1])).sum(0))
    return cols
