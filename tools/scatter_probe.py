"""Time the pieces of the absorb (wiski_scatter_stats_cnt) on the bench geometry: q = 4096 uniform points, 50^3 grid, fp32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
dt = torch.float32
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X = torch.rand((q, 3), device="cuda", dtype=dt) * 2 - 1
y = torch.randn(q, device="cuda", dtype=dt); one = torch.ones(q, device="cuda", dtype=dt)
b = torch.zeros(grid.m, device="cuda", dtype=dt); cnt = torch.zeros_like(b); u = torch.randn_like(b); res = torch.zeros_like(b)
A = torch.zeros(((grid.R + 1) // 2, grid.m), device="cuda", dtype=dt)
stats = torch.zeros(2, device="cuda", dtype=torch.float64); err = grid_ops.new_err_flag("cuda")
def run(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us")
run("all (b, cnt, res, A)", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, A, True, cnt, stats, err, u, res))
run("b, cnt, A", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, A, True, cnt, stats, err))
run("b, A", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, A, True, None, stats, err))
run("b, cnt, res (A = NULL)", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, None, True, cnt, stats, err, u, res))
run("b, cnt (A = NULL)", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, None, True, cnt, stats, err))
run("b (A = NULL)", lambda: grid_ops.scatter_stats_cnt(grid, X, y, one, one, one, b, None, True, None, stats, err))
