import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from online_gp_amd import grid_ops
dev = torch.device('cuda:0'); dt = torch.float32
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
q = 4096
b = torch.zeros(grid.m, device=dev, dtype=dt); H = (grid.R + 1) // 2
half = torch.zeros(H, grid.m, device=dev, dtype=dt); full = torch.zeros(grid.R, grid.m, device=dev, dtype=dt)
stats = torch.zeros(2, device=dev, dtype=torch.float64); err = grid_ops.new_err_flag(dev); ones = torch.ones(q, device=dev, dtype=dt)
def tm(f, reps=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
for name, X in (("uniform", torch.rand(q, 3, device=dev) * 2 - 1), ("one cell", torch.rand(q, 3, device=dev) * 0.04),
                ("64 cells", (torch.randint(0, 4, (q, 3), device=dev).float() * 0.045 + torch.rand(q, 3, device=dev) * 0.04)),
                ("sorted uniform", None)):
    if X is None:
        X = torch.rand(q, 3, device=dev) * 2 - 1
        key = ((X[:, 0] + 1.1) / 0.048).floor() * 2500 + ((X[:, 1] + 1.1) / 0.048).floor() * 50 + ((X[:, 2] + 1.1) / 0.048).floor()
        X = X[key.argsort()].contiguous()
    X = X.to(dt).contiguous(); y = torch.randn(q, device=dev, dtype=dt)
    th = tm(lambda: grid_ops.scatter_stats_sym(grid, X, y, ones, ones, ones, b, half, stats, err))
    tf = tm(lambda: grid_ops.scatter_stats(grid, X, y, ones, ones, ones, b, full, stats, err))
    print(f"{name:15s} half {th:7.1f} us   full {tf:7.1f} us")
print("expand", tm(lambda: grid_ops.stencil_expand_add(grid, half, full)), "us (after zeroing: nothing to fold)")

