"""Minimal stationary-kernel modules with gpytorch's parameterisation, enough for
the WISKI hot path to run without gpytorch (absent from this image).

Only what the path reads is reproduced: raw parameters under a softplus
(`Positive`) constraint initialised at 0 (=> 0.6931), ``lengthscale`` /
``outputscale`` properties with setters, ``batch_shape`` for independent
hyper-parameters per output (reference
online_gp/models/batched_fixed_noise_online_gp.py:107-120), and the
``GridInterpolationKernel`` attributes the reference touches (``grid``,
``grid_bounds``, ``grid_sizes``, ``base_kernel``, ``num_dims``).
The grid kernel itself is never formed: :meth:`GridInterpolationKernel.toeplitz_columns`
returns the first columns of the d symmetric-Toeplitz Kronecker factors
(ScaleKernel inside => outputscale enters once per dim, i.e. s**d overall).
"""
import math

import torch

from .constraints import Positive
from .grid_ops import GridSpec
from .settings import fused_hyper_columns


def inv_softplus(x):
    x = torch.as_tensor(x, dtype=torch.float64)
    return x + torch.log(-torch.expm1(-x))


class Kernel(torch.nn.Module):
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), lengthscale_prior=None, lengthscale_constraint=None, **kwargs):
        super().__init__()
        if kwargs:       # an option silently dropped here would change the fit without a trace
            raise TypeError(f"{type(self).__name__}: unsupported keyword argument(s) {sorted(kwargs)}")
        self.ard_num_dims = ard_num_dims
        self.batch_shape = torch.Size(batch_shape) if not isinstance(batch_shape, int) else torch.Size([batch_shape])
        self._wiski_priors = {}
        if self.has_lengthscale:
            nd = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(self.batch_shape + torch.Size([1, nd]))))
            self.raw_lengthscale_constraint = lengthscale_constraint if lengthscale_constraint is not None else Positive()
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda: self.lengthscale)
        elif lengthscale_prior is not None or lengthscale_constraint is not None:
            raise TypeError(f"{type(self).__name__} has no lengthscale")

    def register_prior(self, name, prior, closure):
        """`closure()` -> the constrained value `prior` scores; the MLL adds prior.log_prob(closure()).sum() (BWM:48-49)."""
        self.add_module(name, prior)
        self._wiski_priors[name] = (prior, closure)
        from .priors import note_registration

        note_registration()

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        v = torch.as_tensor(value, dtype=torch.float64).expand(self.raw_lengthscale.shape)
        with torch.no_grad():
            self.raw_lengthscale.copy_(self.raw_lengthscale_constraint.inverse_transform(v).to(self.raw_lengthscale))

    # k(r), r = |x - x'| / lengthscale
    def profile(self, r):
        raise NotImplementedError

    def lag_column(self, dim, lags, batch_index=None):
        """Covariance between two points `lags` apart along input dim `dim`
        (what gpytorch evaluates with last_dim_is_batch=True on the grid)."""
        ls = self.lengthscale
        if batch_index is not None and ls.dim() > 2:
            ls = ls[batch_index]
        ls = ls.reshape(-1)
        ell = ls[dim] if ls.numel() > 1 else ls[0]
        return self.profile(lags / ell)

    def lag_columns_cat(self, lags_cat, dim_index, batch_index=None):
        """All d Toeplitz columns in one pass: `lags_cat` [sum g] holds the lags of every dim back to back, `dim_index` [sum g]
        the dim each entry belongs to (a handful of launches instead of ~10 per dim: the hyper-parameter graph is on the
        critical path of every MLL step)."""
        ls = self.lengthscale
        if batch_index is not None and ls.dim() > 2:
            ls = ls[batch_index]
        ls = ls.reshape(-1)
        ell = ls[dim_index] if ls.numel() > 1 else ls[0]
        return self.profile(lags_cat / ell)


class RBFKernel(Kernel):
    has_lengthscale = True

    def profile(self, r):
        return torch.exp(-0.5 * r * r)


class MaternKernel(Kernel):
    has_lengthscale = True

    def __init__(self, nu=2.5, **kwargs):
        if nu not in (0.5, 1.5, 2.5):
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    def profile(self, r):
        if self.nu == 0.5:
            return torch.exp(-r)
        if self.nu == 1.5:
            s = math.sqrt(3.0) * r
            return (1.0 + s) * torch.exp(-s)
        s = math.sqrt(5.0) * r
        return (1.0 + s + s * s / 3.0) * torch.exp(-s)


class SpectralMixtureKernel(Kernel):
    """gpytorch.kernels.SpectralMixtureKernel (the covariance of the reference's 1-D notebook,
    notebooks/regression_viz_1D.ipynb: ``SpectralMixtureKernel(num_mixtures=3)`` handed to ``OnlineSKIRegression``):

        k(tau) = sum_q w_q  prod_dims exp(-2 pi^2 tau_d^2 s_qd^2) cos(2 pi tau_d mu_qd)

    with ``mixture_weights`` w [Q], ``mixture_means`` mu [Q, 1, d], ``mixture_scales`` s [Q, 1, d], all positive (softplus
    of raw parameters initialised at 0, as upstream).  On an inducing grid gpytorch evaluates it per dimension
    (``last_dim_is_batch``) and multiplies the per-dimension Toeplitz factors, so the column of dimension d is
    sum_q w_q exp(-2 pi^2 tau^2 s_qd^2) cos(2 pi tau mu_qd) -- a sum over mixtures inside every factor, which is what
    ``lag_column`` returns.  Not of the k(r / lengthscale) form: it has no ``lengthscale`` and no ``profile``."""

    has_lengthscale = False

    def __init__(self, num_mixtures=None, ard_num_dims=1, batch_shape=torch.Size([]), mixture_scales_constraint=None,
                 mixture_means_constraint=None, mixture_weights_constraint=None, **kwargs):
        if num_mixtures is None:
            raise RuntimeError("num_mixtures is a required argument")
        super().__init__(ard_num_dims=ard_num_dims, batch_shape=batch_shape, **kwargs)
        self.num_mixtures = int(num_mixtures)
        nd = 1 if ard_num_dims is None else int(ard_num_dims)
        bs = self.batch_shape
        self.register_parameter("raw_mixture_weights", torch.nn.Parameter(torch.zeros(bs + torch.Size([self.num_mixtures]))))
        self.register_parameter("raw_mixture_means", torch.nn.Parameter(torch.zeros(bs + torch.Size([self.num_mixtures, 1, nd]))))
        self.register_parameter("raw_mixture_scales", torch.nn.Parameter(torch.zeros(bs + torch.Size([self.num_mixtures, 1, nd]))))
        self.raw_mixture_weights_constraint = mixture_weights_constraint if mixture_weights_constraint is not None else Positive()
        self.raw_mixture_means_constraint = mixture_means_constraint if mixture_means_constraint is not None else Positive()
        self.raw_mixture_scales_constraint = mixture_scales_constraint if mixture_scales_constraint is not None else Positive()

    def _get(self, name):
        return getattr(self, f"raw_{name}_constraint").transform(getattr(self, f"raw_{name}"))

    def _set(self, name, value):
        raw = getattr(self, f"raw_{name}")
        v = torch.as_tensor(value, dtype=torch.float64).expand(raw.shape)
        with torch.no_grad():
            raw.copy_(getattr(self, f"raw_{name}_constraint").inverse_transform(v).to(raw))

    mixture_weights = property(lambda self: self._get("mixture_weights"), lambda self, v: self._set("mixture_weights", v))
    mixture_means = property(lambda self: self._get("mixture_means"), lambda self, v: self._set("mixture_means", v))
    mixture_scales = property(lambda self: self._get("mixture_scales"), lambda self, v: self._set("mixture_scales", v))

    def _params(self, batch_index):
        w, mu, sc = self.mixture_weights, self.mixture_means, self.mixture_scales
        if batch_index is not None and w.dim() > 1:
            w, mu, sc = w[batch_index], mu[batch_index], sc[batch_index]
        return w.double(), mu[:, 0, :].double(), sc[:, 0, :].double()          # [Q], [Q, d], [Q, d]

    def lag_column(self, dim, lags, batch_index=None):
        w, mu, sc = self._params(batch_index)
        dq = dim if mu.shape[1] > 1 else 0
        tau = lags.double()[None, :]                                             # [1, g]
        terms = torch.exp(-2.0 * math.pi ** 2 * (tau * sc[:, dq:dq + 1]) ** 2) * torch.cos(2.0 * math.pi * tau * mu[:, dq:dq + 1])
        return (w[:, None] * terms).sum(0)

    def lag_columns_cat(self, lags_cat, dim_index, batch_index=None):
        w, mu, sc = self._params(batch_index)
        di = dim_index if mu.shape[1] > 1 else torch.zeros_like(dim_index)
        tau = lags_cat.double()[None, :]
        terms = torch.exp(-2.0 * math.pi ** 2 * (tau * sc[:, di]) ** 2) * torch.cos(2.0 * math.pi * tau * mu[:, di])
        return (w[:, None] * terms).sum(0)


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, batch_shape=torch.Size([]), outputscale_prior=None, outputscale_constraint=None, **kwargs):
        super().__init__(batch_shape=batch_shape, **kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(self.batch_shape)))
        self.raw_outputscale_constraint = outputscale_constraint if outputscale_constraint is not None else Positive()
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda: self.outputscale)

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        v = torch.as_tensor(value, dtype=torch.float64).expand(self.raw_outputscale.shape)
        with torch.no_grad():
            self.raw_outputscale.copy_(self.raw_outputscale_constraint.inverse_transform(v).to(self.raw_outputscale))

    def lag_column(self, dim, lags, batch_index=None):
        s = self.outputscale
        if batch_index is not None and s.dim() > 0:
            s = s[batch_index]
        return s * self.base_kernel.lag_column(dim, lags, batch_index)

    def lag_columns_cat(self, lags_cat, dim_index, batch_index=None):
        s = self.outputscale
        if batch_index is not None and s.dim() > 0:
            s = s[batch_index]
        return s * self.base_kernel.lag_columns_cat(lags_cat, dim_index, batch_index)


class _StationaryColumns(torch.autograd.Function):
    """Toeplitz columns of S * k(lag / ell) on the grid and their gradient w.r.t. (ell, S), one launch each
    (csrc/hyper_columns.hip) instead of ~10 broadcasting ops forward and ~12 autograd nodes backward."""

    @staticmethod
    def forward(ctx, ell, scale, grid_spec, kind):
        from . import grid_ops

        ell_c = ell.detach().reshape(-1).contiguous()
        scale_c = None if scale is None else scale.detach().reshape(-1).to(ell_c.dtype).contiguous()
        ctx.grid_spec, ctx.kind = grid_spec, kind
        ctx.shapes = (ell.shape, None if scale is None else (scale.shape, scale.dtype))
        ctx.save_for_backward(ell_c, scale_c)
        return grid_ops.stationary_columns(grid_spec, kind, ell_c, scale_c)

    @staticmethod
    def backward(ctx, gout):
        from . import grid_ops

        ell_c, scale_c = ctx.saved_tensors
        g_ell, g_scale = grid_ops.stationary_columns_grad(ctx.grid_spec, ctx.kind, ell_c, scale_c, gout.contiguous())
        ell_shape, sc = ctx.shapes
        return g_ell.reshape(ell_shape), (None if sc is None else g_scale.reshape(sc[0]).to(sc[1])), None, None


def _fused_stationary(kernel, batch_index):
    """(ell [1] or [d], scale [1] or None, kind) when `kernel` is (Scale of)* RBF | Matern of this module, else None."""
    scale = None
    while isinstance(kernel, ScaleKernel):
        s = kernel.outputscale
        if batch_index is not None and s.dim() > 0:
            s = s[batch_index]
        if s.numel() != 1:
            return None
        scale = s.reshape(1) if scale is None else scale * s.reshape(1)
        kernel = kernel.base_kernel
    if type(kernel) is RBFKernel:
        kind = 0
    elif type(kernel) is MaternKernel:
        kind = {0.5: 1, 1.5: 2, 2.5: 3}[kernel.nu]
    else:
        return None
    ls = kernel.lengthscale
    if batch_index is not None and ls.dim() > 2:
        ls = ls[batch_index]
    return ls.reshape(-1), scale, kind


def _native_stationary(kernel):
    """Scale(...(RBF | Matern)) chains of this module: their columns have the closed form used by lag_columns_cat."""
    while isinstance(kernel, ScaleKernel):
        kernel = kernel.base_kernel
    return type(kernel) in (RBFKernel, MaternKernel, SpectralMixtureKernel)


def _lag_column_any(kernel, dim, lags, num_dims, batch_index=None):
    """Toeplitz column of `kernel` along `dim`.  Native kernels use their closed
    form; foreign (e.g. gpytorch) kernels are evaluated on grid points directly."""
    if isinstance(kernel, Kernel):
        return kernel.lag_column(dim, lags, batch_index)
    # duck-typed stationary kernel object: evaluate k(0, lag) with last_dim_is_batch
    x1 = torch.zeros(1, num_dims, dtype=lags.dtype, device=lags.device)
    x2 = torch.zeros(lags.numel(), num_dims, dtype=lags.dtype, device=lags.device)
    x2[:, dim] = lags
    out = kernel(x1, x2, last_dim_is_batch=True)
    out = out.evaluate() if hasattr(out, "evaluate") else (out.to_dense() if hasattr(out, "to_dense") else out)
    out = out[..., dim, 0, :]
    if batch_index is not None and out.dim() > 1:
        out = out[batch_index]
    return out.reshape(-1)


class GridInterpolationKernel(Kernel):
    """Structured-kernel-interpolation wrapper: holds the inducing grid and the
    base kernel.  ``grid_size`` counts gpytorch's two extension points per dim."""

    def __init__(self, base_kernel, grid_size, num_dims=None, grid_bounds=None, **kwargs):
        super().__init__(**kwargs)
        if grid_bounds is None:
            raise RuntimeError("grid_bounds must be given (the data are not kept)")
        gb = torch.as_tensor(grid_bounds, dtype=torch.float64).reshape(-1, 2)
        if num_dims is None:
            num_dims = gb.shape[0]
        if gb.shape[0] == 1 and num_dims > 1:
            gb = gb.expand(num_dims, 2)
        self.base_kernel = base_kernel
        self.num_dims = num_dims
        self.grid_spec = GridSpec(gb.tolist(), grid_size)
        self.grid_bounds = tuple((float(lo), float(hi)) for lo, hi in gb.tolist())
        self.grid_sizes = list(self.grid_spec.g)
        self.grid_is_dynamic = False

    @property
    def grid(self):
        p = next(self.base_kernel.parameters(), None)
        dev = p.device if p is not None else "cpu"
        dt = p.dtype if p is not None else torch.float32
        return self.grid_spec.grid_points(dtype=dt, device=dev)

    def toeplitz_columns(self, batch_index=None, dtype=None, device=None):
        """Concatenated first columns [sum g] of the per-dim Kronecker factors
        (differentiable w.r.t. the hyper-parameters)."""
        p = next(self.base_kernel.parameters(), None)
        device = device if device is not None else (p.device if p is not None else "cpu")
        gs = self.grid_spec
        if torch.device(device).type == "cuda" and fused_hyper_columns.on():
            fs = _fused_stationary(self.base_kernel, batch_index)
            if fs is not None and fs[0].is_cuda and fs[0].numel() in (1, gs.d) and fs[0].dtype in (torch.float32, torch.float64):
                out = _StationaryColumns.apply(fs[0], fs[1], gs, fs[2])
                return out if dtype is None else out.to(dtype)
        if _native_stationary(self.base_kernel):
            key = str(device)
            cached = self.__dict__.setdefault("_lag_cat", {}).get(key)
            if cached is None:
                lags = torch.cat([gs.h[q] * torch.arange(gs.g[q], dtype=torch.float64) for q in range(gs.d)]).to(device)
                didx = torch.cat([torch.full((gs.g[q],), q, dtype=torch.long) for q in range(gs.d)]).to(device)
                cached = self.__dict__["_lag_cat"][key] = (lags, didx)
            out = self.base_kernel.lag_columns_cat(cached[0], cached[1], batch_index).to(torch.float64)
            return out if dtype is None else out.to(dtype)
        cols = []
        for q in range(gs.d):
            lags = gs.h[q] * torch.arange(gs.g[q], dtype=torch.float64, device=device)
            cols.append(_lag_column_any(self.base_kernel, q, lags, gs.d, batch_index).to(torch.float64))
        out = torch.cat(cols)
        return out if dtype is None else out.to(dtype)
