"""cProfile of the host side of a streaming step (evaluate + absorb), GPU idle excluded as far as possible."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; d = 3; q = 4096
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(140 * q, d, 1000, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True)
model.eval()
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), torch.no_grad():
    model.prediction_cache
    te = ta = tr = 0.0
    pr = cProfile.Profile()
    for i in range(60):
        xq, yq = Xs[i * q:(i + 1) * q], ys[i * q:(i + 1) * q]
        torch.cuda.synchronize()
        if i >= 10: pr.enable()
        t0 = time.perf_counter(); model(xq).mean; t1 = time.perf_counter()
        model.condition_on_observations(xq, yq, inplace=True); t2 = time.perf_counter()
        if i >= 10: pr.disable()
        model.prediction_cache
        if i >= 10: te += t1 - t0; ta += t2 - t1
    print('host evaluate %.1f us  absorb %.1f us (with profiler overhead)' % (te / 50 * 1e6, ta / 50 * 1e6))
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
