"""Long run of the reference-fidelity step on a SMALL grid through the device pipeline (settings.spectral_dense_regime): 3000 steps of q = 2 at lr 1e-3,
per 500 steps the time per step, graph captures / replays, eigenvector refreshes, reference rebuilds, rank, memory -- and at the end the posterior
against the data-space oracle (oracle/dataspace.py, n x n Cholesky on all 6 200 points) at the drifted hyper-parameters.
python tools/dense_refstep_soak.py d g [rbf|matern52|matern12]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dataspace
from online_gp_amd.kernels import MaternKernel, ScaleKernel
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float64
d, g = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "rbf"
rng = np.random.default_rng(3)
n0, q, steps = 200, 2, 3000
X = rng.uniform(-1, 1, (n0 + q * steps, d)); y = np.sin(2.0 * X.sum(1)) * np.cos(1.5 * X[:, 0]) + 0.1 * rng.standard_normal(len(X)); y = (y - y.mean()) / y.std()
Xg, yg = torch.as_tensor(X, device=dev, dtype=dt), torch.as_tensor(y, device=dev, dtype=dt)[:, None]
cov = None if kind == "rbf" else ScaleKernel(MaternKernel(nu={"matern12": 0.5, "matern52": 2.5}[kind], ard_num_dims=d)).to(dev)
reg = OnlineSKIRegression(Identity(d), Xg[:n0], yg[:n0], 1e-3, g, 1.0, covar_module=cov)
t0 = tw = time.perf_counter()
for i in range(steps):
    sl = slice(n0 + i * q, n0 + (i + 1) * q)
    reg.evaluate(Xg[sl], yg[sl]); reg.update(Xg[sl], yg[sl])
    if (i + 1) % 500 == 0:
        torch.cuda.synchronize()
        gs, fac = reg._graphed, reg.gp._spectral[0]
        now = time.perf_counter()
        print(i + 1, "ms/step last 500: %.3f" % ((now - tw) / 500 * 1e3), "captures", gs.captures, "replays", gs.replays, "disabled", gs.disabled, "device refreshes", fac.device_refreshes,
              "rebuilds", fac.rebuilds, "rank", fac.cur["basis"].r, "mean_ok", fac.mean_ok, "reserved MB %.0f" % (torch.cuda.memory_reserved() / 1e6), flush=True)
        tw = time.perf_counter()
k = reg.gp.covar_module.base_kernel
ell = k.base_kernel.lengthscale.detach().cpu().numpy().reshape(-1).astype(np.float64); s = float(k.outputscale.detach()); s2 = float(reg.gp.likelihood.second_noise.detach())
Xs = rng.uniform(-1, 1, (64, d))
mean, var = reg.predict(torch.as_tensor(Xs, device=dev, dtype=dt))
O = dataspace.DataSpaceGP([[-1.1, 1.1]] * d, g, kind, ell, s, s2).fit(X, y, np.ones(len(X)))
mo, vo = O.predict(Xs)
print("after %d steps: ell %s outputscale %.4f sigma2 %.4f; mean dev %.2e, variance dev %.2e (data-space oracle on %d points)" % (
    steps, ell.round(4), s, s2, np.abs(mean.cpu().numpy().reshape(-1) - mo).max() / np.abs(mo).max(), np.max(np.abs(var.cpu().numpy().reshape(-1) - (vo + s2)) / (vo + s2)), len(X)))
