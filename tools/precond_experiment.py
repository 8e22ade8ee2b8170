"""CPU experiment (numpy, small grid): does a diagonal scaling S M_s S of the spectral preconditioner M_s = (Kt^-1 + a I)^-1
cut the CG iterations on a line-clustered stream?  Iterations to 1e-4 from a zero start, and from a warm start."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from oracle import spec
g, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 3
kind = sys.argv[2] if len(sys.argv) > 2 else "clustered"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3 * g ** 3
g0, h, gg = spec.make_grid([[-1.1, 1.1]] * d, g)
X, y = bench.synth_stream(n, d, 0, torch.device("cpu"), torch.float64, kind)
X, y = X.numpy(), y.numpy()[:, 0]
Wd = [spec.interp_1d_dense(X[:, q], g0[q], h[q], g) for q in range(d)]
W = np.einsum("pi,pj,pk->pijk", *Wd).reshape(n, -1)
A = W.T @ W; b = W.T @ y
m = g ** d
ell, osc, s2 = 0.6931, 0.6931, 0.6931
cols = spec.toeplitz_columns("rbf", h, gg, ell, osc)
Ts = [np.array([[c[abs(i - j)] for j in range(g)] for i in range(g)]) for c in cols]
ev, Q = zip(*[np.linalg.eigh(T) for T in Ts])
lam = np.einsum("i,j,k->ijk", *ev).reshape(-1) / s2
lam = np.maximum(lam, 1e-12 * lam.max())
def Qt(v): return np.einsum("ai,bj,ck,abc->ijk", Q[0], Q[1], Q[2], v.reshape(g, g, g)).reshape(-1)
def Qf(v): return np.einsum("ia,jb,kc,abc->ijk", Q[0], Q[1], Q[2], v.reshape(g, g, g)).reshape(-1)
def Kinv(v): return Qf(Qt(v) / lam)
def H(v): return Kinv(v) + A @ v
a_i = A.sum(1); abar = a_i.sum() / m
def Ms(v, a=abar): return Qf(Qt(v) * lam / (1 + a * lam))
def pcg(M, x0=None, tol=1e-4, maxit=200):
    x = np.zeros(m) if x0 is None else x0.copy()
    r = b - H(x); z = M(r); p = z.copy(); rz = r @ z; r0 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        Hp = H(p); al = rz / (p @ Hp); x += al * p; r -= al * Hp
        if np.linalg.norm(r) <= tol * r0: return it, x
        z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, x
print(f"grid {g}^3, {kind}, n = {n}: rows with data {np.mean(a_i > 0):.2f}, a_i max / mean {a_i.max() / abar:.1f}")
res = {}
res["M_s (mean density)"] = pcg(lambda v: Ms(v))[0]
for eps_f in (1.0, 0.3, 0.1):
    eps = eps_f * abar
    s = np.sqrt((abar + eps) / (a_i + eps))
    res[f"S M_s S, s = sqrt((abar+e)/(a_i+e)), e = {eps_f} abar"] = pcg(lambda v: s * Ms(s * v))[0]
    sc = np.minimum(s, 1.0)
    res[f"  same, s clipped <= 1, e = {eps_f} abar"] = pcg(lambda v: sc * Ms(sc * v))[0]
# Jacobi-blend: M = M_s(a_lo) restricted ... additive: M_s + diag(1/(a_i + kdiag)) on heavy nodes
kd = np.array([Kinv(np.eye(m)[i])[i] for i in range(0, m, max(1, m // 64))]).mean()
for w in (0.5, 1.0):
    dj = w / (a_i + kd)
    res[f"M_s + {w} diag(1/(a_i + kinv_ii))"] = pcg(lambda v: Ms(v) + dj * v)[0]
for k_, v_ in res.items(): print(f"{v_:4d}  {k_}")
