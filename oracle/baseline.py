"""TEST INFRASTRUCTURE -- ctypes front-end of the multi-threaded CPU baseline (oracle/wiski_baseline_omp.c).

The timed ``cpu_baseline`` of bench.py (kind "port"): the matrix-free WISKI streaming step -- predictive mean of the
incoming batch, absorb, warm-started refresh of the inducing posterior mean -- on all host cores, at the same grid /
batch / init / tolerance as the GPU leg.  Checked against the scalar oracle (cport.py) in tests/test_oracle.py.
Never imported by online_gp_amd."""
import ctypes
import os
import subprocess

import numpy as np

from . import spec

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwiski_baseline.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("wiski_baseline_omp.c", "wiski_baseline_omp_impl.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libwiski_baseline.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def num_threads():
    return int(lib().wb_num_threads())


def set_num_threads(n):
    lib().wb_set_num_threads(ctypes.c_int(int(n)))
    return num_threads()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class StreamingBaseline:
    """Single-output matrix-free WISKI state; full 7^d block stencil, warm-started CG preconditioned by Kt (refresh) or by the
    separable density-profile model of the GPU library (refresh_profile)."""

    def __init__(self, grid_bounds, grid_size, kind="rbf", lengthscale=spec.SOFTPLUS0, outputscale=spec.SOFTPLUS0, sigma2=1.0,
                 dtype=np.float32):
        self.dt = np.dtype(dtype)
        self.sfx = "_f64" if self.dt == np.float64 else "_f32"
        self.creal = ctypes.c_double if self.dt == np.float64 else ctypes.c_float
        g0, h, g = spec.make_grid(grid_bounds, grid_size)
        self.g0, self.h, self.g = g0.astype(self.dt), h.astype(self.dt), g.astype(np.int32)
        self.d, self.m = len(g), int(np.prod(g))
        self.tcol = np.concatenate(spec.toeplitz_columns(kind, h, g, lengthscale, outputscale)).astype(self.dt)
        self.sigma2 = float(sigma2)
        self.b = np.zeros(self.m, self.dt)
        self.A = np.zeros((7 ** self.d, self.m), self.dt)
        self.c_ld = np.zeros(2, np.float64)
        self.u = np.zeros(self.m, self.dt)
        self.z = np.zeros(self.m, self.dt)
        self.solved = False
        self.num_data = 0
        self.wsum = 0.0

    def _fn(self, name):
        f = getattr(lib(), name + self.sfx)
        f.restype = ctypes.c_int
        return f

    def absorb(self, X, y, noise=None):
        X = np.ascontiguousarray(X, self.dt).reshape(-1, self.d)
        y = np.ascontiguousarray(y, self.dt).reshape(-1)
        n = X.shape[0]
        noise = np.ones(n, self.dt) if noise is None else np.ascontiguousarray(noise, self.dt).reshape(-1)
        rc = self._fn("wb_absorb")(_p(X), _p(y), _p(noise), ctypes.c_long(n), self.d, _p(self.g0), _p(self.h), _p(self.g),
                                   ctypes.c_long(self.m), _p(self.b), _p(self.A), _p(self.c_ld))
        if rc:
            raise RuntimeError("Received data that was out of bounds for the specified grid.")
        self.num_data += n
        self.wsum += float(np.sum(1.0 / noise.astype(np.float64)))

    def refresh(self, tol, max_iter=5000):
        """u = (Kt^-1 + A)^-1 b, warm-started from the previous (u, z); returns (iterations, relative residual)."""
        res = ctypes.c_double(0)
        it = self._fn("wb_pcg")(_p(self.A), _p(self.tcol), self.d, _p(self.g), ctypes.c_long(self.m), self.creal(1.0 / self.sigma2),
                                _p(self.b), int(self.solved), ctypes.c_double(tol), max_iter, _p(self.u), _p(self.z), ctypes.byref(res))
        self.solved = True
        return it, float(res.value)

    def _profile_basis(self):
        """Generalized eigenbasis of the density-profile preconditioner, re-solved when the data volume has
        doubled: t_q = per-dim marginal of the row sums of A, normalised to max 1 and clipped at 1e-2;
        K_q = X_q D_q X_q^T with X_q^T diag(t_q) X_q = I, Z_q = diag(t_q) X_q; shift a = (sum_p 1/noise_p) / prod_q sum(t_q)."""
        wsum = float(self.wsum)
        st = getattr(self, "_basis", None)
        if st is None or wsum > 2.0 * st["wsum"]:
            cnt = self.A.sum(axis=0, dtype=np.float64).reshape(tuple(int(v) for v in self.g))
            X, Z, D, norm, off = [], [], [], 1.0, 0
            tc = self.tcol.astype(np.float64)
            for q, gq in enumerate(int(v) for v in self.g):
                marg = cnt.sum(axis=tuple(r for r in range(self.d) if r != q)) if self.d > 1 else cnt
                t = np.clip(marg / marg.max(), 1e-2, None)
                norm *= float(t.sum())
                c = tc[off:off + gq]
                K = c[np.abs(np.arange(gq)[:, None] - np.arange(gq)[None, :])]
                rt = np.sqrt(t)
                w, U = np.linalg.eigh(rt[:, None] * K * rt[None, :])
                X.append((U / rt[:, None]).reshape(-1)); Z.append((U * rt[:, None]).reshape(-1)); D.append(np.clip(w, 0.0, None))
                off += gq
            st = {"wsum": wsum, "norm": norm, "X": np.concatenate(X).astype(self.dt), "Z": np.concatenate(Z).astype(self.dt),
                  "D": np.concatenate(D).astype(self.dt)}
            self._basis = st
        return st, wsum / st["norm"]

    def refresh_profile(self, tol, max_iter=5000):
        """As refresh(), with the separable density-profile preconditioner the GPU library uses (wb_pcg_profile)."""
        st, shift = self._profile_basis()
        res = ctypes.c_double(0)
        it = self._fn("wb_pcg_profile")(_p(self.A), self.d, _p(self.g), ctypes.c_long(self.m), self.creal(1.0 / self.sigma2), _p(st["X"]), _p(st["Z"]),
                                        _p(st["D"]), self.creal(shift), _p(self.b), int(self.solved), ctypes.c_double(tol), max_iter, _p(self.u),
                                        _p(self.z), ctypes.byref(res))
        self.solved = True
        return it, float(res.value)

    def variance(self, Xs, tol=1e-9, max_iter=5000):
        """Predictive variances sigma2 * w^T (Kt^-1 + A)^-1 w of the rows of W(Xs) (BFN:222-228), one cold solve per query
        with the density-profile preconditioner.  Checker for the GPU paths at full size; not timed."""
        Xs = np.asarray(Xs, np.float64).reshape(-1, self.d)
        st, shift = self._profile_basis()
        out = np.empty(Xs.shape[0])
        gi = [int(v) for v in self.g]
        for p in range(Xs.shape[0]):
            w = None
            for q in range(self.d):
                wq = spec.interp_1d_dense(Xs[p:p + 1, q], float(self.g0[q]), float(self.h[q]), gi[q])[0]
                w = wq if w is None else np.multiply.outer(w, wq)
            rhs = np.ascontiguousarray(w.reshape(-1), self.dt)
            u, z = np.zeros(self.m, self.dt), np.zeros(self.m, self.dt)
            res = ctypes.c_double(0)
            self._fn("wb_pcg_profile")(_p(self.A), self.d, _p(self.g), ctypes.c_long(self.m), self.creal(1.0 / self.sigma2), _p(st["X"]), _p(st["Z"]),
                                       _p(st["D"]), self.creal(shift), _p(rhs), 0, ctypes.c_double(tol), max_iter, _p(u), _p(z), ctypes.byref(res))
            out[p] = self.sigma2 * float(rhs.astype(np.float64) @ u.astype(np.float64))
        return out

    def predict_mean(self, Xs):
        Xs = np.ascontiguousarray(Xs, self.dt).reshape(-1, self.d)
        out = np.empty(Xs.shape[0], self.dt)
        rc = self._fn("wb_gather")(_p(Xs), ctypes.c_long(Xs.shape[0]), self.d, _p(self.g0), _p(self.h), _p(self.g), _p(self.u), _p(out))
        if rc:
            raise RuntimeError("Received data that was out of bounds for the specified grid.")
        return out
