"""BASELINE config 4 (BayesOpt, Ackley d = 3, q = 3, 10^3 Matern-5/2 grid, fp64: the dense regime) for a few hundred steps: the harness's
three timers per step; under rocprofv3 --kernel-trace --stats the kernel mix.  python tools/c4_probe.py [steps]"""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import harness
from online_gp_amd.constraints import Interval
from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, ScaleKernel
from online_gp_amd.models import OnlineSKIBotorchModel
from online_gp_amd.priors import GammaPrior

DEV = torch.device("cuda:0")
d, q = 3, 3
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bounds = torch.tensor([[-32.768, 32.768]] * d, dtype=torch.float64)
gen = torch.Generator(device="cpu").manual_seed(0)


def ackley(X, noise_std=4.0):
    a, b, c = 20.0, 0.2, 2 * math.pi
    v = -a * torch.exp(-b * (X.pow(2).sum(-1) / d).sqrt()) - torch.exp(torch.cos(c * X).sum(-1) / d) + a + math.e
    return -v + noise_std * torch.randn(v.shape, generator=gen, dtype=torch.float64).to(v)


init_x = torch.rand(10, d, generator=gen, dtype=torch.float64).to(DEV)
init_y = ackley(bounds[:, 0].to(DEV) + (bounds[:, 1] - bounds[:, 0]).to(DEV) * init_x).reshape(-1, 1)


def make_model(train_x, train_y, old):
    if old is None:
        cov = GridInterpolationKernel(
            ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=d, lengthscale_prior=GammaPrior(3.0, 6.0), lengthscale_constraint=Interval(1e-4, 12.0)),
                        outputscale_prior=GammaPrior(2.0, 0.15), outputscale_constraint=Interval(1e-4, 12.0)),
            grid_size=10, num_dims=d, grid_bounds=bounds)
        return OnlineSKIBotorchModel(train_x, train_y, None, covar_module=cov, learn_additional_noise=True)
    return OnlineSKIBotorchModel(covar_module=old.covar_module, kernel_cache=old._kernel_cache, learn_additional_noise=True,
                                 likelihood=old.likelihood, num_data=old.num_data)


t0 = time.perf_counter()
rows, _, _, last = harness.bayesopt(ackley, bounds, make_model, init_x, init_y, steps, batch_size=q, fit_iters=4, num_candidates=128)
tot = time.perf_counter() - t0
r = rows[20:]
print(f"{steps} steps in {tot:.2f} s; per step after warm-up: fit {1e3 * np.mean([x['fit_time'] for x in r]):.2f} ms (4 Adam steps on the MLL), "
      f"acqf {1e3 * np.mean([x['acqf_time'] for x in r]):.2f} ms (128 candidate sets of q = 3), condition {1e3 * np.mean([x['condition_time'] for x in r]):.2f} ms, "
      f"total {1e3 * np.mean([x['total'] for x in r]):.2f} ms")
fac = last.__dict__.get("_spectral", {}).get(0)
if fac is not None and fac.cur is not None:
    print(f"spectral factor: rank {fac.cur['basis'].r} (reference {fac.ref.r}), device refreshes {fac.device_refreshes}, rebuilds from the stencil {fac.rebuilds}")
