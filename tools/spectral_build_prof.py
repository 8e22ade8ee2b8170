"""Where the spectral factor's reference build (projection of the stencil onto the reduced basis) spends its time."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xv, _ = bench.synth_stream(256, 3, 99, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.cg_tolerance(1e-4), torch.no_grad():
    m = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    m(Xv[:64]).variance
    fac = m._spectral[0]
    for rep in range(3):
        m.__dict__.setdefault("_spectral_dirty", {})[0] = True
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m(Xv[:64]).variance
        torch.cuda.synchronize(); print(f"rebuild + 64 variances: {(time.perf_counter() - t0) * 1e3:.1f} ms  (r_ref {fac.ref.r}, r {fac.cur['basis'].r})")
    pr = cProfile.Profile(); pr.enable()
    m.__dict__.setdefault("_spectral_dirty", {})[0] = True
    m(Xv[:64]).variance; torch.cuda.synchronize()
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
