"""The reference driver's online loop (experiments/regression.py:48-54) at full fidelity on the 50^3 grid:
   for each incoming batch: evaluate (predictive mean AND variance -> rmse, nll), then update (one Adam step on the
   Woodbury MLL under skip_logdet_forward + condition_on_observations).  Reports ms per step for several batch sizes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev = torch.device('cuda:0'); dt = torch.float32; d = 3
X0, y0 = bench.synth_stream(21743, d, 0, dev, dt)
Xs, ys = bench.synth_stream(200000, d, 1000, dev, dt)
for q, steps in ((1, 30), (64, 30), (1024, 12)):
    model = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, 50, 1.0)
    with settings.cg_tolerance(1e-4):
        res = {}
        for mode in ("evaluate+update(hyper step)", "evaluate+update(no hyper step)"):
            t_ev = t_up = 0.0
            for i in range(steps):
                xb, yb = Xs[i * q:(i + 1) * q], ys[i * q:(i + 1) * q]
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rmse, nll = model.evaluate(xb, yb)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                model.update(xb, yb, update_gp=mode.endswith("(hyper step)"))
                torch.cuda.synchronize(); t2 = time.perf_counter()
                if i >= 3:
                    t_ev += t1 - t0; t_up += t2 - t1
            n = steps - 3
            res[mode] = {"evaluate_ms": t_ev / n * 1e3, "update_ms": t_up / n * 1e3}
    print(json.dumps({"q": q, **res, "rmse": rmse, "nll": nll}))
