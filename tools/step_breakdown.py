"""Host-side breakdown of one bench step (evaluate / absorb / refresh), synchronising between phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; q = 4096; d = 3
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(40 * q, d, 1000, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True)
model.eval()
acc = np.zeros(4); n = 0
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), torch.no_grad():
    model.prediction_cache
    for t in range(30):
        xb, yb = Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = model(xb).mean
        torch.cuda.synchronize(); t1 = time.perf_counter()
        model.condition_on_observations(xb, yb, inplace=True)
        t2h = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        pc = model.prediction_cache
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if t >= 5:
            acc += [t1 - t0, t2 - t1, t3 - t2, t2h - t1]; n += 1
print('evaluate %.3f ms | absorb %.3f ms (host-only %.3f) | refresh %.3f ms  iters %s' % tuple(list(acc[[0, 1, 3, 2]] / n * 1e3) + [pc['cg_iters']]))
