"""cProfile of the spectral factor's host path: state refresh after a hyper-parameter step, MLL step."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xv, _ = bench.synth_stream(2048, 3, 99, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
which = sys.argv[1] if len(sys.argv) > 1 else "state"
with settings.cg_tolerance(1e-4):
    m = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    with torch.no_grad():
        m.prediction_cache; m(Xv[:64]).variance
    k = m.covar_module.base_kernel
    mll = BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)
    def hyp2():
        with torch.no_grad():
            k.base_kernel.lengthscale = k.base_kernel.lengthscale * 1.001
        m._dump_caches()
        return m._spectral_state(0)
    def mstep():
        m.train()
        with settings.skip_logdet_forward(True):
            for p in m.parameters(): p.grad = None
            m._dump_caches()
            v = mll(m(None), None); (-v).sum().backward()
        m.eval()
    fn = hyp2 if which == "state" else mstep
    for _ in range(3): fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print("ms per call", (time.perf_counter() - t0) * 100)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
