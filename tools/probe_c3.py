"""Ad-hoc probe of the C3 workload (d=3, 50^3, fp32) kernel timings."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
from oracle import spec

dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.float64
dev = torch.device('cuda:0')
d, g = 3, 50
grid = grid_ops.GridSpec([[-1.1, 1.1]] * d, g)
N = 434874
torch.manual_seed(0)
X = (torch.rand(N, d, device=dev, dtype=dt) * 2 - 1)
y = torch.sin(2 * np.pi * X[:, 0]) * torch.cos(np.pi * X[:, 1]) + 0.5 * X[:, 2] + 0.1 * torch.randn(N, device=dev, dtype=dt)
y = (y - y.mean()) / y.std()
ones = torch.ones(N, device=dev, dtype=dt)
b = torch.zeros(grid.m, device=dev, dtype=dt)
A = torch.zeros(grid.R, grid.m, device=dev, dtype=dt)
stats = torch.zeros(2, device=dev, dtype=torch.float64)
err = grid_ops.new_err_flag(dev)
tcol = torch.as_tensor(np.concatenate(spec.toeplitz_columns('rbf', grid.h, grid.g, spec.SOFTPLUS0, spec.SOFTPLUS0)), device=dev, dtype=dt)
s2 = spec.SOFTPLUS0

def timeit(f, reps=5):
    torch.cuda.synchronize(); f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

for q in (1024, 16384, N):
    def f():
        grid_ops.scatter_stats(grid, X[:q], y[:q], ones[:q], ones[:q], ones[:q], b, A, stats, err)
    t = timeit(f)
    print(f'scatter q={q}: {t*1e6:.1f} us  {q/t:.3e} pt/s')
b.zero_(); A.zero_(); stats.zero_()
grid_ops.scatter_stats(grid, X, y, ones, ones, ones, b, A, stats, err)
for k in (1, 4, 16, 64):
    V = torch.randn(k, grid.m, device=dev, dtype=dt)
    t = timeit(lambda: grid_ops.stencil_spmv(grid, A, V), 20)
    print(f'spmv k={k}: {t*1e6:.1f} us  {A.numel()*A.element_size()/t/1e9:.1f} GB/s (A bytes)')
    t = timeit(lambda: grid_ops.kron_toeplitz_mm(grid, tcol, V), 20)
    print(f'kron k={k}: {t*1e6:.1f} us')
ws = grid_ops.PCGWorkspace()
EIG = grid_ops.kron_eigen(grid, tcol) if (len(sys.argv) < 3 or sys.argv[2] != "plain") else None
for tol in (1e-2, 1e-4, 1e-6):
    torch.cuda.synchronize(); t = time.perf_counter()
    U, Z, it, res = grid_ops.pcg(grid, A, tcol, 1 / s2, b[None], tol=tol, max_iter=1000, check_every=5, workspace=ws, eigen=EIG, shift=N / grid.m)
    torch.cuda.synchronize(); t = time.perf_counter() - t
    print(f'pcg tol={tol}: iters {it} res {res[0]:.2e} time {t*1e3:.2f} ms  ({t/it*1e6:.1f} us/iter)')
for nq in (4096, 1 << 20):
    Xs = torch.rand(nq, d, device=dev, dtype=dt) * 2 - 1
    t = timeit(lambda: grid_ops.gather(grid, Xs, U, err), 10)
    print(f'gather fused n={nq}: {t*1e6:.1f} us {nq/t:.3e} rows/s')
    idx, val = grid_ops.interp(grid, Xs, err)
    t = timeit(lambda: grid_ops.gather_ell(idx, val, U[0]), 10)
    print(f'gather ell n={nq}: {t*1e6:.1f} us {nq*(64*(4+val.element_size())+val.element_size())/t/1e9:.1f} GB/s')
print('err', int(err.item()))
