"""Sweep of the two-level preconditioner's knobs on one 3droad-sized pass of the bench stream (50^3, fp32, q = 4096): CG iterations per
step and wall time per step for (rank, lag, growth), road-like and uniform streams.  python tools/two_level_probe.py [grid]"""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP

g = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
N, q = 434874, 4096
TOL = float(os.environ.get('TOL', '1e-4'))
n0 = int(0.05 * N)
steps = (N - n0) // q
gb = torch.tensor([[-1.1, 1.1]] * 3)


def one(kind, on, rank, lag, growth, reps=2):
    X, y = bench.synth_stream(n0 + steps * q, 3, 0, dev, torch.float32, kind)
    best = None
    for rep in range(reps):
        with settings.two_level_subsample(int(os.environ.get('SUB', '1'))), settings.two_level_preconditioner(on), settings.two_level_rank(rank), settings.two_level_lag(lag), settings.two_level_growth(growth), \
                settings.cg_tolerance(TOL), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
            m = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
            m.prediction_cache
            torch.cuda.synchronize()
            its = []
            t0 = time.perf_counter()
            for s in range(steps):
                sl = slice(n0 + s * q, n0 + (s + 1) * q)
                m.stream_step(X[sl], y[sl])
                its.append(m._last_iters[0])
            m._finish_pending()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            tr = m.__dict__.get("_two_level")
            nref = tr.block.refreshes if (tr is not None and tr.block is not None) else 0
            if best is None or dt < best[0]:
                best = (dt, np.mean(its), np.mean(its[:20]), np.mean(its[-50:]), nref)
                if os.environ.get('TRACE'): print('   its', its[:40])
    return best


print(f"grid {g}^3, {steps} steps of q = {q} after {n0} init points")
for kind in ("clustered", "uniform"):
    dt, it, it0, it1, _ = one(kind, False, 0, 0, 1.1)
    print(f"{kind:9s} separable only       : {dt * 1e3:.4f} ms/step  {q / dt:.3e} upd/s  iters {it:.2f} (first20 {it0:.2f}, last50 {it1:.2f})", flush=True)
    cfgs = [(128, 2, 1.1), (256, 2, 1.1), (384, 2, 1.1), (256, 1, 1.1), (256, 3, 1.1), (256, 2, 1.0), (256, 2, 1.25), (256, 1, 1.0), (192, 2, 1.1)]
    if len(sys.argv) > 2:
        cfgs = [tuple(float(v) if "." in v else int(v) for v in c.split(",")) for c in sys.argv[2:]]
    for rank, lag, growth in cfgs:
        dt, it, it0, it1, nref = one(kind, True, rank, lag, growth)
        print(f"{kind:9s} r={rank:3d} lag={lag} growth={growth:4.2f}: {dt * 1e3:.4f} ms/step  {q / dt:.3e} upd/s  iters {it:.2f} (first20 {it0:.2f}, last50 {it1:.2f})  refreshes {nref}",
              flush=True)
