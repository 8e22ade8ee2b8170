"""BASELINE configs[1] at full size: powerplant-like stream (N = 9568, d = 4), 30^4 inducing grid (m = 810000), fp64;
parity of predictive mean / variance against the data-space CPU oracle (rtol 1e-4) + timings."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dataspace
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
rng = np.random.default_rng(1)
N, d, g = 9568, 4, 30
X = rng.uniform(-1, 1, (N, d)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.5 * X[:, 2] * X[:, 3] + 0.1 * rng.standard_normal(N)
y = (y - y.mean()) / y.std()
dev = 'cuda'
Xt, yt = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)[:, None]
n0 = N // 20
torch.cuda.synchronize(); t0 = time.perf_counter()
model = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=g, learn_additional_noise=True)
model.eval()
for s in range(n0, N, 1024):
    model.condition_on_observations(Xt[s:s + 1024], yt[s:s + 1024], inplace=True)
torch.cuda.synchronize(); t_abs = time.perf_counter() - t0
with settings.variance_chunk(32):
    t0 = time.perf_counter(); pc = model.prediction_cache; torch.cuda.synchronize(); t_ref = time.perf_counter() - t0
    Xs = Xt[:64]
    t0 = time.perf_counter(); mvn = model(Xs); mean = mvn.mean.cpu().numpy(); var = mvn.variance.cpu().numpy(); t_pred = time.perf_counter() - t0
s2 = float(model.likelihood.second_noise.detach())
t0 = time.perf_counter()
O = dataspace.DataSpaceGP([[-1.1, 1.1]] * d, g, sigma2=s2).fit(X, y, np.ones(N))
mo, vo = O.predict(X[:64]); t_or = time.perf_counter() - t0
print(json.dumps({"config": "C2 d=4 30^4 fp64 N=9568", "absorb_s": t_abs, "refresh_ms": t_ref * 1e3, "cg_iters": pc["cg_iters"],
                  "predict_64_mean_var_ms": t_pred * 1e3, "rel_err_mean": float(np.abs(mean - mo).max() / np.abs(mo).max()),
                  "rel_err_var": float(np.abs(var - vo).max() / np.abs(vo).max()), "oracle_cpu_s": t_or,
                  "hbm_GB": torch.cuda.max_memory_allocated() / 1e9}))
