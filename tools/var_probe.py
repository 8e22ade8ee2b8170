"""Predictive variance of 64 / 1024 query points on the bench geometry (ms per call); under rocprofv3 shows the kernel mix."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
if os.environ.get("WISKI_NO_SPECTRAL") == "1":
    settings.spectral_factor._set_state(False)
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(100000, 3, 0, dev, dt, "uniform")
Xv, _ = bench.synth_stream(4096, 3, 99, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with settings.cg_tolerance(1e-4), settings.variance_cg_tolerance(3e-3), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    model(Xv[:nq]).variance
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = model(Xv[(rep + 1) * nq:(rep + 2) * nq]).variance
        torch.cuda.synchronize()
        print(f"variance of {nq} queries: {(time.perf_counter() - t0) * 1e3:.3f} ms")
