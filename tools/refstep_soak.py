"""4000 reference-fidelity steps (q = 8, lr 1e-3): time per step, device memory, graph captures / replays, eigenvector refreshes,
reference rebuilds and the rank of the spectral basis every 500 steps (is anything growing or re-capturing in a long run?)."""
import os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(40000, 3, 31337, dev, dt, "uniform")
with settings.cg_tolerance(1e-4), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    t0 = tw = time.perf_counter()
    fast = [0]
    orig = reg._evaluate_from_factor
    def counted(x, y):
        r = orig(x, y)
        fast[0] += r is not None
        return r
    reg._evaluate_from_factor = counted
    for i in range(4001):
        xb, yb = Xr[i * 8:(i + 1) * 8], yr[i * 8:(i + 1) * 8]
        rm, nl = reg.evaluate(xb, yb); reg.update(xb, yb)
        if i % 500 == 0:
            torch.cuda.synchronize()
            gs, fac = reg._graphed, reg.gp._spectral[0]
            k = reg.gp.covar_module.base_kernel
            now = time.perf_counter()
            print(i, "ms/step %.3f (last 500: %.3f)" % ((now - t0) / max(i, 1) * 1e3, (now - tw) / 500 * 1e3), "fused evaluates", fast[0], "mean_ok", fac.mean_ok, "mean bound", fac.last_mean_bound, "measured", fac.last_measured, "x", fac.measurements, "fused graph", gs.fused, "alloc MB %.1f reserved MB %.1f" % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6),
                  "captures", gs.captures, "replays", gs.replays, "dev refreshes", fac.device_refreshes, "rebuilds", fac.rebuilds, "rank", fac.cur["basis"].r,
                  "rmse %.4f" % rm, "ls", [round(float(v), 4) for v in k.base_kernel.lengthscale.detach().reshape(-1)], flush=True)
            tw = time.perf_counter()
