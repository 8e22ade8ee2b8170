"""The reference-fidelity step (evaluate mean + variance -> Adam step on the MLL -> condition) at q = 1 / 64 on the bench geometry:
median ms per step after the graph capture, split into the evaluate and update halves (each synchronised), and the host time the
two calls take to ISSUE their work (no synchronisation inside) -- the step is host-bound where that approaches the total."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(49152, 3, 31337, dev, dt, "uniform")
with settings.cg_tolerance(1e-4), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    lo = 0
    for qs, nst in ((1, 40), (64, 30)):
        tot, tev, tup, thost = [], [], [], []
        for i in range(nst):
            xb, yb = Xr[lo:lo + qs], yr[lo:lo + qs]; lo += qs
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reg.evaluate(xb, yb)
            if i % 2:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            reg.update(xb, yb)
            t2 = time.perf_counter()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            if i % 2:
                tev.append(t1 - t0); tup.append(t3 - t1)
            else:
                tot.append(t3 - t0); thost.append(t2 - t0)
        m = lambda v: float(np.median(v[4:])) * 1e3
        print("q = %4d: step %.3f ms (host issue %.3f ms); synchronised halves: evaluate %.3f + update %.3f ms" % (qs, m(tot), m(thost), m(tev), m(tup)), flush=True)
