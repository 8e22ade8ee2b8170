"""Randomised soak of the streaming path at the BASELINE size (50^3, fp32): 150 updates with random batch sizes,
then (a) the carried residual against the true residual, (b) the streamed posterior mean against a model rebuilt
from scratch on all data, (c) the scattered stencil against a single-shot scatter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_gp_amd import grid_ops, settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP

dev = torch.device("cuda:0"); dt = torch.float32; d = 3
gen = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sizes = [1, 2, 3, 7, 64, 100, 1000, 4096, 5000]
qs = [sizes[int(torch.randint(0, len(sizes), (1,), generator=gen))] for _ in range(150)]
n0 = 20000
X, y = bench.synth_stream(n0 + sum(qs), d, 7, dev, dt)
gb = torch.tensor([[-1.1, 1.1]] * d)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-5), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], None, grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    lo = n0
    its = []
    for q in qs:
        model(X[lo:lo + min(q, 64)]).mean
        model.condition_on_observations(X[lo:lo + q], y[lo:lo + q], inplace=True)
        its.append(model.prediction_cache["cg_iters"][0])
        lo += q
    pc = model.prediction_cache
    ms, c = model._mean_state, model._kernel_cache
    true_r = c["interpolation_cache"][0, :, 0] - ms["Z"][0] - grid_ops.stencil_spmv(model._grid, c["WtW"].stencil, ms["U"][0:1])[0]
    bn = float(c["interpolation_cache"].norm())
    print("steps", len(qs), "points", lo, "cg iters min/mean/max", min(its), sum(its) / len(its), max(its))
    print("carried residual vs true residual: max |diff| / |b| =", float((ms["R"][0] - true_r).abs().max()) / bn,
          "  |true r| / |b| =", float(true_r.norm()) / bn)
    ref = FixedNoiseOnlineSKIGP(X[:lo], y[:lo], None, grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    a, b = c["WtW"].stencil, ref._kernel_cache["WtW"].stencil
    print("stencil streamed vs single shot: max rel diff", float((a - b).abs().max() / b.abs().max()))
    Xs = X[:4096]
    m1, m2 = model(Xs).mean, ref(Xs).mean
    print("posterior mean streamed vs rebuilt: max rel diff", float((m1 - m2).abs().max() / m2.abs().max()))
