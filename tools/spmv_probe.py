"""Time the stencil SpMV (full vs symmetric half layout) at a given grid with HIP-event brackets
(wiski_prof_*), on a stencil filled by the real scatter kernel."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import _hip, grid_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    dev = "cuda"
    grid = grid_ops.GridSpec([[-1.1, 1.1]] * a.dim, a.grid)
    gen = torch.Generator(device=dev).manual_seed(0)
    X = (torch.rand((a.n, a.dim), device=dev, dtype=dt, generator=gen) * 2 - 1)
    y = torch.randn(a.n, device=dev, dtype=dt, generator=gen)
    w = torch.ones(a.n, device=dev, dtype=dt)
    err = grid_ops.new_err_flag(dev)
    H = (grid.R + 1) // 2
    half = torch.zeros((H, grid.m), device=dev, dtype=dt)
    full = torch.zeros((grid.R, grid.m), device=dev, dtype=dt)
    b = torch.zeros(grid.m, device=dev, dtype=dt)
    stats = torch.zeros(2, device=dev, dtype=torch.float64)
    grid_ops.scatter_stats_sym(grid, X, y, w, w, w, b, half, stats, err)
    b.zero_()
    grid_ops.scatter_stats(grid, X, y, w, w, w, b, full, stats, err)
    V = torch.randn((a.k, grid.m), device=dev, dtype=dt, generator=gen)
    lib = _hip.lib()
    es = 4 if dt == torch.float32 else 8
    res = {}
    for name, A in (("full", full), ("half", half)):
        out = grid_ops.stencil_spmv(grid, A, V)
        torch.cuda.synchronize()
        lib.wiski_prof_start(ctypes.c_int32(4096))
        for _ in range(a.reps):
            out = grid_ops.stencil_spmv(grid, A, V)
        torch.cuda.synchronize()
        tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
        lib.wiski_prof_stop(ctypes.byref(tms), ctypes.byref(nl))
        us = tms.value * 1e3 / max(nl.value, 1)
        byts = A.numel() * es
        res[name] = out
        print(f"{name}: {us:8.2f} us/launch ({nl.value} launches)  {byts / 1e6:8.1f} MB  {byts / us / 1e6:6.2f} TB/s", flush=True)
    diff = (res["full"] - res["half"]).abs().max().item() / res["full"].abs().max().item()
    print(f"max rel diff half vs full: {diff:.2e}")


if __name__ == "__main__":
    main()


def timing_dump():
    """With a -DWISKI_SYM_TIMING build (WISKI_HIP_SO=...): phase stamps (100 MHz wall clock) of one mid-grid wave."""
    lib = _hip.lib()
    if not hasattr(lib, "wiski_sym_dbg"):
        return
    buf = (ctypes.c_longlong * 32)()
    lib.wiski_sym_dbg(buf)
    v = list(buf)
    t0 = v[0]
    names = {0: "start", 1: "LDS zero + barrier", 20: "loop end", 21: "flush", 22: "end"}
    for i in range(4):
        names[2 + 3 * i] = f"g{i} fetch"; names[3 + 3 * i] = f"g{i} fma"; names[4 + 3 * i] = f"g{i} lds rmw"
    prev = t0
    for i in sorted(names):
        if v[i] >= t0 and v[i] > 0:
            print(f"  {names[i]:20s} +{(v[i] - prev) * 10:6d} ns   (t = {(v[i] - t0) * 10} ns)")
            prev = v[i]


timing_dump()
