import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev = torch.device('cuda:0'); dt = torch.float32; d = 3
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(2000, d, 1000, dev, dt)
model = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, 50, 1.0)
def loop(a, b):
    for i in range(a, b):
        xb, yb = Xs[i:i + 1], ys[i:i + 1]
        model.evaluate(xb, yb); model.update(xb, yb, update_gp=True)
with settings.cg_tolerance(1e-4):
    loop(0, 12)
    pr = cProfile.Profile(); pr.enable(); loop(12, 112); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
