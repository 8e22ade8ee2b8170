"""Where the reference's timed online step (evaluate mean+variance -> Adam step on the Woodbury MLL -> condition) spends its
time on the bench geometry, per q."""
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
Xr, yr = bench.synth_stream(8192, 3, 31337, dev, dt, "uniform")
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
with settings.cg_tolerance(float(os.environ.get("CGTOL", "1e-4"))), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    for qs in (1, 64, 1024):
        for i in range(4):
            xb, yb = Xr[i * qs:(i + 1) * qs], yr[i * qs:(i + 1) * qs]
            te, _ = T(lambda: reg.evaluate(xb, yb))
            tu, _ = T(lambda: reg.update(xb, yb))
            if i:
                print(f"q={qs:5d} evaluate {te:7.2f} ms   update {tu:7.2f} ms")
    xb, yb = Xr[5000:5001], yr[5000:5001]
    for rep in range(3):
        t, _ = T(lambda: reg._hyper_step())
        print(f"_hyper_step {t:.2f} ms")
    gp = reg.gp
    with torch.no_grad():
        for rep in range(3):
            t, _ = T(lambda: gp.condition_on_observations(xb, yb, None, inplace=True))
            t2, _ = T(lambda: gp.prediction_cache)
            t3, _ = T(lambda: gp(xb).mean)
            t4, _ = T(lambda: gp(xb).variance)
            print(f"condition {t:.2f}  prediction_cache {t2:.2f}  mean {t3:.2f}  variance(1 query) {t4:.2f} ms")
