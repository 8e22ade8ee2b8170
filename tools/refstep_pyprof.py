"""cProfile of the reference-fidelity step (evaluate mean + variance -> Adam step on the MLL -> condition) at q = 1."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(8192, 3, 31337, dev, dt, "uniform")
qs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with settings.cg_tolerance(1e-4), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    def step(i):
        xb, yb = Xr[i * qs:(i + 1) * qs], yr[i * qs:(i + 1) * qs]
        reg.evaluate(xb, yb); reg.update(xb, yb)
    for i in range(4): step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
    for i in range(4, 24): step(i)
    torch.cuda.synchronize(); print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
    pr.disable(); pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(70)
