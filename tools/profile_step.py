import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; d = 3; q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(140 * q, d, 1000, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True)
model.eval()
def loop(n0, n):
    for i in range(n0, n0 + n):
        xq, yq = Xs[i * q:(i + 1) * q], ys[i * q:(i + 1) * q]
        model(xq).mean
        model.condition_on_observations(xq, yq, inplace=True)
        model.prediction_cache
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), torch.no_grad():
    model.prediction_cache
    loop(0, 20)
    torch.cuda.synchronize(); t = time.perf_counter(); loop(20, 50); torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t) * 20, 'iters', model.prediction_cache['cg_iters'])
    pr = cProfile.Profile(); pr.enable(); loop(70, 50); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
