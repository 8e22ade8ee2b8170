import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; d = 3
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(4096, d, 1000, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True)
model.eval()
def loop(n, qs):
    for i in range(n):
        xq, yq = Xs[i * qs:(i + 1) * qs], ys[i * qs:(i + 1) * qs]
        model(xq).mean
        model.condition_on_observations(xq, yq, inplace=True)
        model.prediction_cache
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), torch.no_grad():
    model.prediction_cache
    loop(20, 1)
    torch.cuda.synchronize(); t = time.perf_counter(); loop(100, 1); torch.cuda.synchronize(); print('q=1 ms/step', (time.perf_counter() - t) * 10)
    print('iters', model.prediction_cache['cg_iters'])
    pr = cProfile.Profile(); pr.enable(); loop(100, 1); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
