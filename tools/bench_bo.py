"""Small-grid (dense MFMA path) timings for the BO / active-learning configurations of BASELINE.json:
C4  Ackley d=3, grid 10^3 over the raw Ackley bounds (the reference's quirk), q=3 per step, UCB-style posterior calls
C5  d=2 grid 30^2, Matern-1/2, heteroscedastic noise, q=6 per step
Per step (the three timers of experiments/bayesopt/bayesopt.py:181-236): fit = one MLL value+gradient,
acqf = 20 posterior calls on 512 x q candidate batches (mean + q x q covariance), condition = non-inplace update."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, ScaleKernel
from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
from online_gp_amd.models import OnlineSKIBotorchModel

dev = torch.device('cuda:0')

def run(name, d, g, gb, nu, q, steps, hetero):
    torch.manual_seed(0)
    X = torch.rand(10 + q * steps, d, device=dev, dtype=torch.float32)
    y = -(X - 0.5).norm(dim=1, keepdim=True) + 0.1 * torch.randn(X.shape[0], 1, device=dev)
    noise = (torch.rand_like(y) * 0.05 + 1e-3) if hetero else torch.ones_like(y)
    cov = GridInterpolationKernel(ScaleKernel(MaternKernel(nu=nu, ard_num_dims=d)), grid_size=g, num_dims=d, grid_bounds=gb)
    model = OnlineSKIBotorchModel(X[:10], y[:10], noise[:10], covar_module=cov, learn_additional_noise=True)
    t_fit = t_acq = t_cond = 0.0
    for s in range(steps):
        lo = 10 + s * q
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mll = BatchedWoodburyMarginalLogLikelihood(model.likelihood, model)
        model.train(); v = mll(model(None), None); (-v).backward(); model.zero_grad()
        for p_ in model.parameters(): p_.grad = None
        torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.no_grad():
            for _ in range(20):
                post = model.posterior(torch.rand(512, q, d, device=dev))
                ucb = post.mean[..., 0] + 2.0 * post.variance[..., 0].sqrt()
                _ = post.mvn.covariance_matrix
        torch.cuda.synchronize(); t2 = time.perf_counter()
        with torch.no_grad():
            model = model.condition_on_observations(X[lo:lo + q], y[lo:lo + q], noise[lo:lo + q])
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if s >= 2:
            t_fit += t1 - t0; t_acq += t2 - t1; t_cond += t3 - t2
    # fantasy-style use (look-ahead acquisitions, qNIPV): condition a sibling on q candidate points and read its posterior,
    # hyper-parameters fixed -- rank-q Woodbury update of the cached posterior matrix vs a fresh factor per fantasy
    from online_gp_amd import settings
    fant = {}
    with torch.no_grad():
        model.eval(); model.posterior(torch.rand(4, q, d, device=dev)).mean
        for name_, flag in (("rank_update", True), ("fresh_factor", False)):
            with settings.dense_rank_updates(flag):
                torch.cuda.synchronize(); tf = time.perf_counter()
                for _ in range(10):
                    xf = torch.rand(q, d, device=dev)
                    fm = model.condition_on_observations(xf, torch.zeros(q, 1, device=dev), noise[:q])
                    _ = fm.posterior(torch.rand(64, q, d, device=dev)).variance
                torch.cuda.synchronize(); fant[name_] = (time.perf_counter() - tf) / 10 * 1e3
    n = steps - 2
    print(json.dumps({"config": name + " -- fantasy + posterior, ms each", **fant}))
    print(json.dumps({"config": name, "m": g ** d, "q": q, "fit_ms": t_fit / n * 1e3, "acqf_20_posteriors_ms": t_acq / n * 1e3,
                      "condition_ms": t_cond / n * 1e3, "num_data": model.num_data}))

run("C4 BO Ackley d=3 10^3 Matern-5/2", 3, 10, torch.tensor([[-32.768, 32.768]] * 3), 2.5, 3, 12, False)
run("C5 AL d=2 30^2 Matern-1/2 heteroscedastic", 2, 30, torch.tensor([[0.0, 1.0]] * 2), 0.5, 6, 12, True)
