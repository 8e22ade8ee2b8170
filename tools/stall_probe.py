"""Per-step wall times of the headline stream (sync after every step): where do the rare ~80 ms stalls fall?"""
import os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
if os.environ.get("NOGC"): gc.disable()
dev, dt = torch.device("cuda:0"), torch.float32
q, n = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 600
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
T0 = time.perf_counter()
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    for p in range(6):
        model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
        model.prediction_cache
        if os.environ.get("FREEZE"):
            gc.unfreeze(); gc.collect(); gc.freeze()
        Xr, yr = bench.synth_stream(q * 100, 3, 7 + p, dev, dt, "uniform")
        ts = []
        for i in range(100):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
            model._finish_pending(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            if ts[-1] > 2.0: print(f"pass {p} step {i}: {ts[-1]:.1f} ms at t = {time.perf_counter() - T0:.2f} s, num_data {model.num_data}")
        ts.sort(); print(f"pass {p}: median {ts[50]:.3f} ms, max {ts[-1]:.1f} ms")
