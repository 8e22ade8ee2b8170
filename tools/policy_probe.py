"""Per-step (first poll, probe?, iterations) of the streaming loop with deferred refreshes, as bench.py runs it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = 4096
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    n = 100
    Xr, yr = bench.synth_stream(q * n, 3, 1000, dev, dt, "uniform")
    out = []
    for i in range(n):
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
        fc, probe = model._pending_fc
        out.append(f"{fc}{'p' if probe else ''}>{model._last_iters[0]}")     # iterations: those of the PREVIOUS step (deferred)
        if len(sys.argv) > 1 and (i + 1) % int(sys.argv[1]) == 0:
            model._finish_pending(); torch.cuda.synchronize(); out.append("|")
    model._finish_pending()
    print(" ".join(out))
