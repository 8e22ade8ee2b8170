"""Cold-started variance solves (right-hand sides = columns of W*^T) on the road-like 50^3 stream: CG iterations and time with the
separable preconditioner alone and with the two-level block, one column and 64 columns.  python tools/tl_cold_probe.py [kernel]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import grid_ops, settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP

dev = torch.device("cuda")
N, q, g = 434874, 4096, 50
n0 = int(0.05 * N)
steps = int(os.environ.get("STEPS", (N - n0) // q))
gb = torch.tensor([[-1.1, 1.1]] * 3)
TOL = float(os.environ.get("TOL", "1e-4"))
X, y = bench.synth_stream(n0 + steps * q, 3, 0, dev, torch.float32, "clustered")
with settings.cg_tolerance(TOL), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
    m = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
    m.prediction_cache
    for s in range(steps):
        sl = slice(n0 + s * q, n0 + (s + 1) * q)
        m.stream_step(X[sl], y[sl])
    m._finish_pending()
    torch.cuda.synchronize()
    print("stream iterations of the last step:", m._last_iters)
    tr = m.__dict__.get("_two_level")
    blk = tr.block if tr is not None else None
    if blk is None:
        print("no block")
        sys.exit(0)
    blk.finish()
    tl = blk.struct if blk.active >= 0 else None
    print("block r =", blk.r, "slabs", blk.nslab, "refreshes", blk.refreshes, "active", blk.active)
    post = m._posterior_op(0)
    post.two_level_provider = None                      # the rows below name their block (or none) themselves
    Xq, _ = bench.synth_stream(1152, 3, 99, dev, torch.float32, "clustered")
    Wt = grid_ops.wt_columns(m._grid, Xq[:64].contiguous(), m._err)        # [64, m]
    for tol in (1e-4, 3e-3):
        post.tol = tol
        for name, arg in (("separable", None), ("two-level", tl)):
            its, ts = [], []
            for c in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                U, Z = post.solve_columns(Wt[c][None].contiguous(), two_level=arg)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                its.append(post.last_iters)
            print(f"tol {tol:g} one column, {name:9s}: iterations {its}  median {np.median(ts) * 1e3:.3f} ms")
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            U, Z = post.solve_columns(Wt)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"tol {tol:g} 64 columns, separable: iterations {post.last_iters}  median {np.median(ts) * 1e3:.3f} ms")
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            U2, Z2 = post.solve_columns(Wt, two_level=tl)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"tol {tol:g} 64 columns, two-level: iterations {post.last_iters}  median {np.median(ts) * 1e3:.3f} ms   max |dU| / max |U| = {float((U2 - U).abs().max() / U.abs().max()):.2e}")
    # what a rebuild from the statistics costs (settings.two_level_rebuild: after a hyper-parameter step)
    pst = m._memo["precond"][0]
    for rep in range(3):
        tr.lose()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tl2 = tr.rebuild(m._grid, dev, pst, post.kscale, post.wtw.stencil, float(m._wsum[0]), m._err)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"rebuild from the stencil: host {1e3 * (t1 - t0):.3f} ms, until the device is done {1e3 * (t2 - t0):.3f} ms")
