"""How sparse is the half stencil on a clustered (road-like) stream?  Fraction of all-zero 256-row x 1-group tiles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import grid_ops
dev, dt = torch.device("cuda:0"), torch.float32
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
for kind in ("clustered", "uniform"):
    for n in (21743, 100000, 434874):
        X, y = bench.synth_stream(n, 3, 0, dev, dt, kind)
        half = torch.zeros(((grid.R + 1) // 2, grid.m), device=dev, dtype=dt)
        b = torch.zeros(grid.m, device=dev, dtype=dt); st = torch.zeros(2, device=dev, dtype=torch.float64); err = grid_ops.new_err_flag(dev)
        w = torch.ones(n, device=dev, dtype=dt)
        grid_ops.scatter_stats_sym(grid, X, y[:, 0].contiguous(), w, w, w, b, half, st, err)
        flat = half.reshape(-1)
        m = grid.m
        rows_nz = (b != 0).float().mean().item()
        # group 0: 4 reals per row; groups g >= 1: 7 reals per row at (7 g - 3) m
        nrb = (m + 255) // 256
        tiles = []
        g0 = flat[:4 * m].reshape(m, 4)
        pad = nrb * 256 - m
        def tile_nz(a):
            a = torch.nn.functional.pad(a.abs().sum(1), (0, pad)).reshape(nrb, 256).sum(1)
            return (a != 0)
        nz = [tile_nz(g0)]
        for g in range(1, 25):
            nz.append(tile_nz(flat[(7 * g - 3) * m:(7 * g + 4) * m].reshape(m, 7)))
        nz = torch.stack(nz)
        print(f"{kind:9s} n={n:6d}: rows with data {rows_nz:.3f}; nonzero tiles {nz.float().mean().item():.3f}; nonzero row blocks {nz.any(0).float().mean().item():.3f}")
