"""CPU prototype (numpy / scipy.sparse) of the two-level preconditioner of DESIGN.md 3.3b, written before the HIP version: streaming
protocol of bench.py scaled to a g^3 grid (init 5 %, ~100 warm-started steps, tolerance 1e-4), CG iterations per step for
    sep          the separable density-profile preconditioner alone (production until round 3)
    two r        exact block on the r top modes of the PLAIN Kronecker eigenbasis, scalar density model elsewhere
    gen r        exact block on the r top modes of the GENERALIZED eigenbasis of the separable model (what was built)
with the block fresh every step (lag 0), `lag` steps old, or refreshed when the data have grown by `growth`.

    python tools/two_level_prototype.py 24 clustered 128,256 gen

Road-like stream, mean iterations per step (first 20 steps / last 50):
    16^3  sep 5.01 | two r=64 7.8, r=256 2.00 | gen r=32 4.28, r=64 2.86, r=128 2.00 (lag 2: 2.58; growth 1.1 + lag 2: 2.77)
    24^3  sep 5.45 | two r=128 3.99          | gen r=128 2.00 (lag 2: 2.45 = 3.50 / 2.00; growth 1.1 + lag 2: 2.63; growth 1.25: 2.93)
    32^3  sep 5.60
i.e. the generalized basis needs a quarter of the modes of the plain one, a fresh block brings every step to 2 iterations, and what
a realistic pipeline loses is the block's age in the first steps, when each batch is 10-20 % of the data.  (MI355X, 50^3, rank 192,
block aimed at the middle of its service life: 2.26-2.5 against 6.06.)"""
import sys, time, numpy as np, scipy.sparse as sp, scipy.linalg as sla
sys.path.insert(0, '/root/repo')
import bench, torch
from oracle import spec
g = int(sys.argv[1]); kind = sys.argv[2]; d = 3; m = g ** 3
N = int(434874 * m / 125000); q = max(32, int(4096 * m / 125000)); n0 = int(0.05 * N)
g0, h, gg = spec.make_grid([[-1.1, 1.1]] * d, g)
X, y = bench.synth_stream(N, d, 0, torch.device('cpu'), torch.float64, kind); X = X.numpy(); y = y.numpy()[:, 0]
def Wsp(Xp):
    n = len(Xp); idx = np.zeros((n, 1), np.int64); val = np.ones((n, 1))
    for qq in range(d):
        u = (Xp[:, qq] - g0[qq]) / h[qq]; fl = np.floor(u); t = u - fl; j0 = fl.astype(np.int64) - 1
        w = np.stack([spec.keys_cubic(t + 1), spec.keys_cubic(t), spec.keys_cubic(t - 1), spec.keys_cubic(t - 2)], 1)
        j = j0[:, None] + np.arange(4)[None]
        idx = (idx[:, :, None] * g + j[:, None, :]).reshape(n, -1); val = (val[:, :, None] * w[:, None, :]).reshape(n, -1)
    return sp.csr_matrix((val.ravel(), (np.repeat(np.arange(n), idx.shape[1]), idx.ravel())), shape=(n, m))
ell, osc, s2 = 0.6931, 0.6931, 0.6931
cols_ = spec.toeplitz_columns('rbf', h, gg, ell, osc)
Ks = [sla.toeplitz(c) for c in cols_]
def T3(Ms, v, tr=False):
    t = v.reshape(g, g, g)
    if tr: Ms = [M.T for M in Ms]
    t = np.tensordot(Ms[0], t, (1, 0)); t = np.tensordot(Ms[1], t, (1, 1)).transpose(1, 0, 2); t = np.tensordot(t, Ms[2], (2, 1))
    return t.ravel()
def kmv(v): return T3(Ks, v) / s2
# plain eigenbasis
evs, Vs = zip(*[np.linalg.eigh(K) for K in Ks])
lam = np.clip(np.einsum('i,j,k->ijk', *evs).ravel(), 0, None) / s2
def profile_setup(cnt3):
    Xq = []; Dq = []; norm = 1.0
    for qq in range(d):
        marg = cnt3.sum(axis=tuple(r for r in range(d) if r != qq)); t = np.clip(marg / marg.max(), 1e-2, None); norm *= t.sum()
        rt = np.sqrt(t); w, U = np.linalg.eigh(rt[:, None] * Ks[qq] * rt[None, :]); Xq.append(U / rt[:, None]); Dq.append(np.clip(w, 0, None))
    a = cnt3.sum() / norm
    D = np.einsum('i,j,k->ijk', *Dq).ravel() / s2
    Zq = [np.linalg.inv(x).T for x in Xq]   # Z = T X = X^-T
    return Xq, Zq, D, a
class PrecSep:
    def __init__(s, cnt): s.Xq, s.Zq, s.D, s.a = profile_setup(cnt.reshape(g, g, g))
    def __call__(s, r):
        c = T3(s.Xq, r, True); return T3(s.Xq, c * s.D / (1 + s.a * s.D)), T3(s.Zq, c / (1 + s.a * s.D))
class PrecTwo:
    """plain K eigenbasis; S = top-r modes; exact block N = (lam_S^-1 + G_S)^-1, scalar a elsewhere"""
    def __init__(s, r, a): s.S = np.argsort(-lam)[:r]; s.a = a; s.N = None; s.r = r
    def set_block(s, G, a): s.N = np.linalg.inv(np.diag(1 / lam[s.S]) + G); s.a = a
    def __call__(s, rr):
        c = T3(Vs, rr, True); cy = c * lam / (1 + s.a * lam); ct = c / (1 + s.a * lam)
        if s.r: cs = s.N @ c[s.S]; cy[s.S] = cs; ct[s.S] = cs / lam[s.S]
        return T3(Vs, cy), T3(Vs, ct)
class PrecGen:
    """generalized eigenbasis of the separable profile model (computed once, kept); exact block on its top-r modes"""
    def __init__(s, r, cnt):
        s.Xq, s.Zq, s.D, s.a = profile_setup(cnt.reshape(g, g, g)); s.r = r; s.S = np.argsort(-s.D)[:r]; s.norm = cnt.sum() / s.a
    def basis(s):
        i0, i1, i2 = np.unravel_index(s.S, (g, g, g))
        return np.einsum('ar,br,cr->abcr', s.Xq[0][:, i0], s.Xq[1][:, i1], s.Xq[2][:, i2]).reshape(m, -1)
    def set_block(s, G, n): s.N = np.linalg.inv(np.diag(1 / s.D[s.S]) + G); s.a = n / s.norm
    def __call__(s, rr):
        c = T3(s.Xq, rr, True); cy = c * s.D / (1 + s.a * s.D); ct = c / (1 + s.a * s.D)
        if s.r: cs = s.N @ c[s.S]; cy[s.S] = cs; ct[s.S] = cs / s.D[s.S]
        return T3(s.Xq, cy), T3(s.Zq, ct)
def pcg(A, b, u, z, P, tol=1e-4, maxit=200):
    r = b - z - A @ u; r0 = np.linalg.norm(b)
    if np.linalg.norm(r) <= tol * r0: return 0, u, z
    yv, t = P(r); p = yv.copy(); pt = t.copy(); rho = r @ yv
    for it in range(1, maxit + 1):
        Hp = pt + A @ p; al = rho / (p @ Hp); u = u + al * p; z = z + al * pt; r = r - al * Hp
        if np.linalg.norm(r) <= tol * r0: return it, u, z
        yv, t = P(r); rho2 = r @ yv; be = rho2 / rho; rho = rho2; p = yv + be * p; pt = t + be * pt
    return maxit, u, z
def basis_cols(S):  # V_S (m x r) via kron columns
    i0, i1, i2 = np.unravel_index(S, (g, g, g))
    return np.einsum('ar,br,cr->abcr', Vs[0][:, i0], Vs[1][:, i1], Vs[2][:, i2]).reshape(m, -1)
W0 = Wsp(X[:n0]); A0 = (W0.T @ W0).tocsr(); b0 = W0.T @ y[:n0]
batches = [(Wsp(X[p:p + q]), y[p:p + q]) for p in range(n0, N - q + 1, q)]
print(f'g={g} {kind} N={N} q={q} steps={len(batches)}', flush=True)
def run(mode, r=0, lag=0, growth=None):
    A = A0.copy(); b = b0.copy(); n = n0
    if mode == 'sep':
        P = PrecSep(np.asarray(A.sum(1)).ravel()); n_at = n
    elif mode == 'gen':
        P = PrecGen(r, np.asarray(A.sum(1)).ravel()); B = P.basis(); G = B.T @ (A @ B); P.set_block(G, n); hist = [G.copy()]; n_blk = n
    else:
        P = PrecTwo(r, n / m)
        if r: B = basis_cols(P.S); G = B.T @ (A @ B); P.set_block(G, n / m); hist = [G.copy()]
        n_blk = n
    it, u, z = pcg(A, b, np.zeros(m), np.zeros(m), P, maxit=500); cold = it
    its = []
    for Wn, yn in batches:
        A = (A + Wn.T @ Wn).tocsr(); b = b + Wn.T @ yn; n += len(yn)
        if mode == 'sep':
            if n >= 2 * n_at: P = PrecSep(np.asarray(A.sum(1)).ravel()); n_at = n
        elif r:
            F = Wn @ B; G = G + F.T @ F; hist.append(G.copy())
            nn = n if mode == 'gen' else n / m
            if growth is None:
                P.set_block(hist[max(0, len(hist) - 1 - lag)], nn)
            elif n >= growth * n_blk:
                P.set_block(hist[max(0, len(hist) - 1 - lag)], nn); n_blk = n
            else: P.a = n / P.norm if mode == 'gen' else n / m
        else: P.a = n / m
        it, u, z = pcg(A, b, u, z, P); its.append(it)
    its = np.array(its)
    print(f'{mode:4s} r={r:4d} lag={lag} growth={growth}: cold {cold}, mean {its.mean():.2f}, first20 {its[:20].mean():.2f}, last50 {its[-50:].mean():.2f}', flush=True)
run('sep')
modes = sys.argv[4].split(',') if len(sys.argv) > 4 else ['two']
for r in [int(a) for a in sys.argv[3].split(',')]:
    for md in modes:
        run(md, r, 0); run(md, r, 2); run(md, r, 0, 1.1); run(md, r, 2, 1.1); run(md, r, 2, 1.25)
