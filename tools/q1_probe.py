"""Stream q-point batches through model.stream_step (deferred poll) on the bench geometry; prints ms per step.  Under
`rocprofv3 --kernel-trace` + tools/step_timeline.py this shows where a small-batch step spends its time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(q * (n + 20), 3, 7, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    for i in range(20):
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    model._finish_pending(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20, 20 + n):
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    model._finish_pending(); torch.cuda.synchronize()
    print(f"q = {q}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step, last iters {model._last_iters}")
