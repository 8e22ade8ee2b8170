"""The reference-fidelity step (evaluate -> Adam step on the MLL -> condition) in the DENSE regime (small inducing grids, the reference's own
regime): d = 1 / 64 nodes (BASELINE config 1's size), d = 2 / 30^2 (config 5's), d = 3 / 10^3 (config 4's); ms per step at q = 1 and 8.
python tools/dense_refstep_probe.py [d ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd.kernels import MaternKernel, ScaleKernel
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float64
cases = ((1, 64, "rbf"), (2, 30, "rbf"), (3, 10, "rbf"), (2, 30, "matern12"), (3, 10, "matern52"))     # the last two: BASELINE config 5's / 4's kernels (full rank)
if len(sys.argv) > 1:
    cases = tuple(c for c in cases if str(c[0]) in sys.argv[1:])
for d, g, kind in cases:
    cov = None if kind == "rbf" else ScaleKernel(MaternKernel(nu={"matern12": 0.5, "matern52": 2.5}[kind], ard_num_dims=d)).to(dev)
    X0, y0 = bench.synth_stream(200, d, 0, dev, dt, "uniform")
    Xr, yr = bench.synth_stream(4096, d, 31337, dev, dt, "uniform")
    reg = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, g, 1.0, covar_module=cov)
    lo = 0
    for qs, nst in ((1, 60), (8, 40)):
        tot = []
        for i in range(nst):
            xb, yb = Xr[lo:lo + qs], yr[lo:lo + qs]; lo += qs
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reg.evaluate(xb, yb)
            reg.update(xb, yb)
            torch.cuda.synchronize(); tot.append(time.perf_counter() - t0)
        fac = reg.gp.__dict__.get("_spectral", {}).get(0)
        gs = reg.__dict__.get("_graphed")
        info = "" if fac is None or fac.cur is None else " | spectral rank %d (ref %d), device refreshes %d, rebuilds %d, graph replays %s fused %s disabled %s" % (
            fac.cur["basis"].r, fac.ref.r, fac.device_refreshes, fac.rebuilds, getattr(gs, "replays", None), getattr(gs, "fused", None), getattr(gs, "disabled", None))
        print("d = %d, grid %d^%d (m = %d) %s, q = %d: %.3f ms per step%s" % (d, g, d, g ** d, kind, qs, float(np.median(tot[8:])) * 1e3, info), flush=True)
