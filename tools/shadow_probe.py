"""What would a shadow stencil cost the solve?  The bench's streaming loop (50^3, q = 4096, road-like stream) runs on the main stream
while a side stream does, per step, what a double-buffered absorb would do beside the solve: copy the 86 MB half stencil and scatter
one batch into the copy.  Reported: ms per step alone / with the side traffic, and the side stream's own time per step.
python tools/shadow_probe.py [copy|scatter|both]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import grid_ops, settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP

what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda")
N, q, g = 434874, 4096, 50
n0 = int(0.05 * N)
steps = (N - n0) // q
gb = torch.tensor([[-1.1, 1.1]] * 3)
X, y = bench.synth_stream(n0 + steps * q, 3, 0, dev, torch.float32, "clustered")
side = torch.cuda.Stream(device=dev)


def run(side_mode):
    with settings.cg_tolerance(1e-4), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
        m = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        m.prediction_cache
        A = m._kernel_cache["WtW"].stencil
        Y = A.clone()
        b2 = torch.zeros(m._grid.m, dtype=torch.float32, device=dev)
        cnt2 = torch.zeros_like(b2)
        st2 = torch.zeros(2, dtype=torch.float64, device=dev)
        err2 = torch.zeros(1, dtype=torch.int32, device=dev)
        ones = torch.ones(q, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        its = []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        for s in range(steps):
            sl = slice(n0 + s * q, n0 + (s + 1) * q)
            if side_mode:
                with torch.cuda.stream(side):
                    if s == 20:
                        ev0.record(side)
                    if side_mode in ("copy", "both"):
                        Y.copy_(A)
                    if side_mode in ("scatter", "both"):
                        grid_ops.scatter_stats_cnt(m._grid, X[sl], y[sl, 0].contiguous(), ones, ones, ones, b2, Y, True, cnt2, st2, err2)
                    if s == steps - 1:
                        ev1.record(side)
            m.stream_step(X[sl], y[sl])
            its.append(m._last_iters[0])
        m._finish_pending()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        sd = ev0.elapsed_time(ev1) / (steps - 20) if side_mode else 0.0
        return dt * 1e3, float(np.mean(its)), sd


for mode in (None, what, None, what):
    dt, it, sd = run(mode)
    print(f"side traffic {str(mode):8s}: {dt:.4f} ms per step ({q / dt * 1e3:.3e} updates/s), {it:.2f} iterations, side stream busy-span {sd:.4f} ms per step", flush=True)
