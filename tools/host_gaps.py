"""Host-side cost (no profiler) of the three calls of a streaming step, measured as wall time until the call returns
(GPU work is asynchronous except for the solver's final poll)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; d = 3; q = 4096
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(140 * q, d, 1000, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True).eval()
te = ta = tr = 0.0; n = 0
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), torch.no_grad():
    model.prediction_cache
    for i in range(100):
        xq, yq = Xs[i * q:(i + 1) * q], ys[i * q:(i + 1) * q]
        torch.cuda.synchronize()
        t0 = time.perf_counter(); model(xq).mean
        t1 = time.perf_counter(); model.condition_on_observations(xq, yq, inplace=True)
        t2 = time.perf_counter(); model.prediction_cache
        t3 = time.perf_counter()
        if i >= 20: te += t1 - t0; ta += t2 - t1; tr += t3 - t2; n += 1
print('host time per call: evaluate %.1f us | absorb %.1f us | refresh (incl. waiting for the GPU) %.1f us' % (te / n * 1e6, ta / n * 1e6, tr / n * 1e6))
