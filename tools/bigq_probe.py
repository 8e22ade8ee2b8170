import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = int(sys.argv[1])
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    n = 24
    Xr, yr = bench.synth_stream(q * n, 3, 1000, dev, dt, "uniform")
    for i in range(4): model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    model._finish_pending(); torch.cuda.synchronize(); t0 = time.perf_counter(); its = []
    for i in range(4, n):
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q]); its.append(model._last_iters[0])
    model._finish_pending(); torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / (n - 4)
    print(f"q={q} owner_min={os.environ.get('WISKI_OWNER_MIN_POINTS','default')}: {dtm*1e6:.1f} us/step, {q/dtm:.3e} updates/s, iters {sum(its)/len(its):.2f}")
