"""The device pipeline of the hyper step (graphed Adam step, device-side eigenvector refresh, fused kernel columns) against the plain
path on other grids / dtypes / kernels: 60 steps each, per-step (rmse, nll, loss) traces and final predictions compared."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
from online_gp_amd import kernels
dev = "cuda"
def run(d, g, dtype, kern, fast, steps=60, q=3, lr=5e-3):
    rng = np.random.default_rng(d * 100 + g)
    n0 = 800
    X = rng.uniform(-1, 1, (n0 + steps * q, d)); y = np.sin(2 * X[:, 0]) * (X[:, 1] if d > 1 else 1.0) + 0.05 * rng.standard_normal(len(X))
    Xt = torch.as_tensor(X, device=dev, dtype=dtype); yt = torch.as_tensor(y, device=dev, dtype=dtype)[:, None]
    cov = None
    if kern == "matern":
        cov = kernels.ScaleKernel(kernels.MaternKernel(nu=2.5, ard_num_dims=d))
    with settings.graphed_hyper_step(fast), settings.spectral_device_refresh(fast), settings.fused_hyper_columns(fast):
        reg = OnlineSKIRegression(Identity(d), Xt[:n0], yt[:n0], lr, g, 1.0, covar_module=cov)
        out = []
        t0 = time.perf_counter()
        for i in range(steps):
            lo = n0 + q * i
            out.append(reg.evaluate(Xt[lo:lo + q], yt[lo:lo + q]) + (reg.update(Xt[lo:lo + q], yt[lo:lo + q])[1],))
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / steps * 1e3
        Xq = torch.as_tensor(rng.uniform(-1, 1, (40, d)), device=dev, dtype=dtype)
        m, v = reg.predict(Xq)
        fac = reg.gp.__dict__.get("_spectral", {}).get(0)
        gs = reg.__dict__.get("_graphed")
        info = (None if fac is None or fac.cur is None else (fac.cur["basis"].r, fac.cur["basis"].kmax, fac.device_refreshes, fac.rebuilds), None if gs is None else (gs.captures, gs.replays, gs.disabled))
    return np.array(out), m.double().cpu().numpy(), v.double().cpu().numpy(), dt_, info
for d, g, dtype, kern in [(2, 64, torch.float64, "rbf"), (2, 50, torch.float32, "rbf"), (3, 16, torch.float64, "matern"), (3, 20, torch.float32, "rbf"), (4, 8, torch.float64, "rbf"), (4, 9, torch.float32, "rbf")]:
    a = run(d, g, dtype, kern, True); b = run(d, g, dtype, kern, False)
    tol = 1e-7 if dtype == torch.float64 else 2e-3
    e_tr = np.abs(a[0] - b[0]).max() / max(1.0, np.abs(b[0]).max()); e_m = np.abs(a[1] - b[1]).max() / max(1.0, np.abs(b[1]).max()); e_v = (np.abs(a[2] - b[2]) / b[2]).max()
    print(f"d={d} g={g} {str(dtype)[6:]} {kern}: fast {a[3]:.2f} ms/step {a[4]} | plain {b[3]:.2f} ms/step {b[4]} | trace {e_tr:.1e} mean {e_m:.1e} var {e_v:.1e}", "OK" if max(e_tr, e_m, e_v) < tol * 50 else "MISMATCH", flush=True)
