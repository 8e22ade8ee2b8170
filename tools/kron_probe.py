"""Time Kuu . V (wiski_kron_toeplitz_mm: d mode products) at a given grid; torch events on the launch stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
from online_gp_amd.kernels import GridInterpolationKernel, RBFKernel, ScaleKernel

import os
CASES = ((3, 50, torch.float32),) if os.environ.get('KRON_ONLY_C3') else ((3, 50, torch.float64), (3, 50, torch.float32), (3, 50, torch.float64), (4, 30, torch.float64), (2, 30, torch.float32))
for (d, g, dt) in CASES:
    cov = GridInterpolationKernel(ScaleKernel(RBFKernel(ard_num_dims=d)), grid_size=g, num_dims=d, grid_bounds=[[-1.1, 1.1]] * d)
    grid = cov.grid_spec
    tcol = cov.toeplitz_columns(device="cuda").to(dt).contiguous()
    for k in (1, 8):
        V = torch.randn(k, grid.m, device="cuda", dtype=dt)
        grid_ops.kron_toeplitz_mm(grid, tcol, V)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            grid_ops.kron_toeplitz_mm(grid, tcol, V)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"d={d} g={g} {str(dt)[6:]} k={k}: {us:8.1f} us per product ({us / k / d:6.1f} us per column-mode), min traffic {2 * d * grid.m * V.element_size() * k / 1e6:.1f} MB")
