"""k_gather_ell_dma (csrc/gather_ell_dma.h) against the register-staged k_gather_ell: equality with a torch gather and microseconds per
launch over the tuning grid (passes per tile x waves per CU), for the geometries the C ABI accepts.  GB/s = algorithmic bytes
(T (4 + s) + s per row) / time."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import _hip, grid_ops  # noqa: E402


def event_us(fn, n, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--policy", action="store_true")
    ap.add_argument("--final", action="store_true", help="the library's defaults only (for a rocprofv3 run): d = 3 fp32, 20 launches each")
    a = ap.parse_args()
    dev = torch.device("cuda")
    lib = _hip.lib()
    tune = lib.wiski_gather_ell_tune
    for d, g, dt in ((3, 50, torch.float32), (3, 50, torch.float64), (2, 200, torch.float32), (4, 20, torch.float32), (1, 4000, torch.float32), (4, 20, torch.float64)):
        if a.quick and not (d == 3 or (d == 4 and dt == torch.float32)):
            continue
        grid = grid_ops.GridSpec([[-1.1, 1.1]] * d, g)
        T = 4 ** d
        n = a.rows if d < 4 else a.rows // 4
        n += 37                                        # a ragged last tile
        es = 4 if dt == torch.float32 else 8
        X = torch.rand((n, d), device=dev, dtype=dt) * 2 - 1
        err = grid_ops.new_err_flag(dev)
        idx, val = grid_ops.interp(grid, X, err)
        v = torch.randn(grid.m, device=dev, dtype=dt)
        want = (val.double() * v.double()[idx.long()]).sum(1)
        byts = n * (T * (4 + es) + es)
        print(f"d={d} g={g} {str(dt)[6:]} rows={n} T={T}  ({byts / 1e6:.0f} MB)", flush=True)
        if a.final:
            if not (d == 3 and dt == torch.float32):
                continue
            for name, fn, arg in (("k_gather_ell (registers)", lambda: grid_ops.gather_ell(idx, val, v), (0, 0, 1, 0)),
                                  ("k_gather_ell_dma, plain (library default)", lambda: grid_ops.gather_ell(idx, val, v), (0, 0, 0, 0)),
                                  ("k_gather_ell_dma, grid-aware (library default)", lambda: grid_ops.gather_ell(idx, val, v, grid=grid), (0, 0, 0, 0))):
                tune(*(ctypes.c_int32(x) for x in arg))
                fn(); torch.cuda.synchronize()
                us = event_us(fn, 10, reps=2)
                print(f"   {name:52s} {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s  frac {byts / us / 1e6 / 8.0:5.3f}", flush=True)
            tune(ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0))
            continue
        if d >= 2:          # the grid-aware form on the blocked copy of v
            tune(ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0))
            og = grid_ops.gather_ell(idx, val, v, grid=grid)
            print(f"   grid-aware form: max rel dev {float((og.double() - want).abs().max() / want.abs().max()):.1e}", flush=True)
            for p_ in (4, 8):
                for wv in (512, 768, 1024, 2048):
                    for ab in (0,):
                        tune(ctypes.c_int32(p_), ctypes.c_int32(wv), ctypes.c_int32(0), ctypes.c_int32(ab))
                        us = event_us(lambda: grid_ops.gather_ell(idx, val, v, grid=grid), 6)
                        print(f"   {'grid-aware (pack + blocked gather) P=%d waves=%d%s' % (p_, wv, ' ABLATED second block' if ab else ''):72s} {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s  frac {byts / us / 1e6 / 8.0:5.3f}", flush=True)
        cfgs = [("k_gather_ell (registers)", 0, 0, 1, 0)]
        for p in (4, 8, 16):
            if p == 16 and dt == torch.float64:
                continue
            for wv in (128, 192, 256, 320, 384, 512, 768, 1024, 2048):
                for cg in (0, 1):
                    cfgs.append((f"dma P={p} waves={wv}{' contiguous' if cg else ''}", p, wv, 0, cg))
        if a.ablate:
            cfgs = [(f"dma P=8 waves={wv} {nm}", 8, wv, 0, cg) for wv in (512, 1024) for nm, cg in (("", 0), ("ABLATED gather (L1 hits)", 2), ("ABLATED stream (no refill)", 4), ("ABLATED both", 6))]
        if a.policy:
            cfgs = [(f"dma P=8 waves={wv} {nm}", 8, wv, 0, cg) for wv in (512, 768, 1024, 2048) for nm, cg in (("", 0), ("ABLATED: a quad gathers ONE 16-byte group (1/4 of the lines)", 8))]
        for name, p, wv, off, cg in cfgs:
            tune(ctypes.c_int32(p), ctypes.c_int32(wv), ctypes.c_int32(off), ctypes.c_int32(cg))
            out = grid_ops.gather_ell(idx, val, v)
            torch.cuda.synchronize()
            dev_ = float((out.double() - want).abs().max() / want.abs().max())
            us = event_us(lambda: grid_ops.gather_ell(idx, val, v), 6)
            print(f"   {name:52s} {us:9.1f} us  {byts / us / 1e6:6.2f} TB/s  frac {byts / us / 1e6 / 8.0:5.3f}   max rel dev {dev_:.1e}", flush=True)
        # arbitrary (non-consecutive) indices take the slow path: same answer
        tune(ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0))
        idx2 = idx.clone()
        idx2[::3, 1] = idx2[::3, 0]                    # break the run of 4 in every third row
        want2 = (val.double() * v.double()[idx2.long()]).sum(1)
        out2 = grid_ops.gather_ell(idx2, val, v)
        print(f"   arbitrary indices (slow path): max rel dev {float((out2.double() - want2).abs().max() / want2.abs().max()):.1e}", flush=True)
        # a sliced (16-byte aligned at row granularity, still contiguous) view and a small batch
        out3 = grid_ops.gather_ell(idx[5:5 + 4096], val[5:5 + 4096], v)
        print(f"   4096-row slice: max rel dev {float((out3.double() - want[5:5 + 4096]).abs().max() / want.abs().max()):.1e}", flush=True)
    tune(ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0))


if __name__ == "__main__":
    main()
