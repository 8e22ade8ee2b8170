"""Time N Adam steps on the Woodbury MLL (StreamingSKIWrapper._hyper_step) on the bench geometry."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
if os.environ.get("WISKI_NO_SPECTRAL") == "1":
    settings.spectral_factor._set_state(False)
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with settings.cg_tolerance(float(os.environ.get("CGTOL", "1e-4"))), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    for _ in range(3): reg._hyper_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if os.environ.get("PYPROF"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    for _ in range(n): reg._hyper_step()
    torch.cuda.synchronize()
    if os.environ.get("PYPROF"):
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    print(f"_hyper_step: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
