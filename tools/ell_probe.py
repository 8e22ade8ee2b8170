"""Time the ELL-form predictive interpolated MVM (wiski_gather_ell) on 2^20 query rows of the 50^3 grid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
for dt in (torch.float32, torch.float64):
    nq = 1 << 20
    X = torch.rand((nq, 3), device="cuda", dtype=dt) * 2.2 - 1.1          # includes boundary cells
    err = grid_ops.new_err_flag("cuda")
    idx, val = grid_ops.interp(grid, X, err)
    v = torch.randn(grid.m, device="cuda", dtype=dt)
    out = grid_ops.gather_ell(idx, val, v)
    ref = (val.double() * v.double()[idx.long()]).sum(1)
    print("max rel err", float((out.double() - ref).abs().max() / ref.abs().max()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        grid_ops.gather_ell(idx, val, v)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = nq * (64 * (4 + val.element_size()) + val.element_size())
    print(f"{str(dt)[6:]}: {us:.1f} us for {by / 1e6:.0f} MB -> {by / us / 1e6:.2f} TB/s = {by / us / 1e6 / 8:.2f} of 8 TB/s")
