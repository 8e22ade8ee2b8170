"""Iterations of the cold solve on the init data and of the first streaming steps, uniform and road-like (is the cold count a usable gate for
building the two-level block before the first streamed step?)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from online_gp_amd import settings  # noqa: E402
from online_gp_amd.models import FixedNoiseOnlineSKIGP  # noqa: E402


def run(kind, steps=30, q=4096, n_init=21743, **ctx):
    dev = torch.device("cuda")
    X0, y0 = bench.synth_stream(n_init, 3, 0, dev, torch.float32, kind)
    Xs, ys = bench.synth_stream(steps * q, 3, 1000, dev, torch.float32, kind)
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
        m = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
        m.prediction_cache
        cold = list(m._last_iters)
        its = []
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for t in range(steps):
            m.stream_step(Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q])
            its.append(m._last_iters[0])
        m._finish_pending()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
    return cold, its, dt


if __name__ == "__main__":
    for kind in ("uniform", "clustered"):
        for n_init in (21743, 5000):
            cold, its, dt = run(kind, n_init=n_init)
            print(f"{kind:10s} init {n_init:6d}: cold solve {cold} iterations; steps (reported one call late): {its}  mean {np.mean(its):.2f}  {dt:.4f} ms/step")
