"""GPU: batched fantasies / batch-expanded conditioning (reference BFN:287-332, OSB:51-61,
URLT:139-159 -- broken at HEAD, so the expectation is the per-fantasy data-space oracle)."""
import numpy as np
import pytest
import torch

from oracle import dataspace

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Sampler:
    """What BoTorch's MC samplers do for fantasize(): sampler(posterior) -> [num_fantasies, *batch, q, 1]."""

    def __init__(self, num):
        self.sample_shape = torch.Size([num])

    def __call__(self, posterior):
        torch.manual_seed(3)
        return posterior.rsample(self.sample_shape)


def _model(kind, d, g, n, dense):
    from online_gp_amd import settings
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, RBFKernel, ScaleKernel
    from online_gp_amd.models import OnlineSKIBotorchModel

    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (n, d)); y = np.sin(4 * X[:, 0]) * np.cos(3 * X[:, -1]) + 0.05 * rng.standard_normal(n)
    nz = rng.uniform(0.3, 0.8, n)
    base = RBFKernel(ard_num_dims=d) if kind == "rbf" else MaternKernel(nu=0.5, ard_num_dims=d)
    gb = torch.tensor([[0.0, 1.0]] * d, dtype=torch.float64)
    cov = GridInterpolationKernel(ScaleKernel(base), grid_size=g, num_dims=d, grid_bounds=gb)
    with settings.dense_small_grids(dense):
        m = OnlineSKIBotorchModel(torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None], torch.as_tensor(nz, device=DEV)[:, None],
                                  covar_module=cov, learn_additional_noise=True)
    m.eval()
    return m, X, y, nz, gb.numpy(), kind


@pytest.mark.parametrize("kind,d,g,dense", [("rbf", 2, 8, True), ("matern12", 2, 30, True), ("rbf", 3, 12, False)])
def test_fantasize_matches_per_fantasy_oracle(kind, d, g, dense):
    """fantasize(X [b, q, d]) -> batch [num_fantasies, b]; posterior at shared and per-candidate queries equals an exact GP
    re-fitted on data + each fantasy, for the dense factor (BO / malaria grids: 8^2, the C5 geometry 30^2 Matern-1/2 with
    heteroscedastic noise) and for the PCG path (12^3)."""
    from online_gp_amd import settings

    with settings.dense_small_grids(dense), settings.cg_tolerance(1e-11):
        m, X, y, nz, gb, kind = _model(kind, d, g, 40, dense)
        rng = np.random.default_rng(1)
        b, q, F = 3, 2, 2
        Xc = torch.as_tensor(rng.uniform(0.1, 0.9, (b, q, d)), device=DEV)
        fm = m.fantasize(Xc, _Sampler(F))
        assert tuple(fm.batch_shape) == (F, b) and fm.num_data == 42
        Yf = fm.train_targets                                   # [F, b, q]
        s2 = float(m._sigma2(0))
        fnoise = float(m.likelihood.noise.detach().mean())
        ell = m.covar_module.base_kernel.base_kernel.lengthscale.detach().double().cpu().numpy().reshape(-1)
        osc = float(m.covar_module.base_kernel.outputscale.detach().double())
        Xq = torch.as_tensor(rng.uniform(0.05, 0.95, (5, d)), device=DEV)
        post = fm.posterior(Xq)                                 # shared queries
        assert post.mean.shape == (F, b, 5, 1) and post.variance.shape == (F, b, 5, 1) and post.mvn.covariance_matrix.shape == (F, b, 5, 5)
        Xqb = torch.as_tensor(rng.uniform(0.05, 0.95, (b, 4, d)), device=DEV)
        postb = fm.posterior(Xqb)                               # one query set per candidate set
        assert postb.mean.shape == (F, b, 4, 1)
        for f in range(F):
            for j in range(b):
                Xa = np.concatenate([X, Xc[j].cpu().numpy()]); ya = np.concatenate([y, Yf[f, j].cpu().numpy()])
                na = np.concatenate([nz, np.full(q, fnoise)])
                O = dataspace.DataSpaceGP(gb, g, kind, ell, osc, s2).fit(Xa, ya, na)
                mo, co = O.predict(Xq.cpu().numpy(), full_cov=True)
                assert np.abs(post.mean[f, j, :, 0].cpu().numpy() - mo).max() < 1e-6 * max(np.abs(mo).max(), 1e-2)
                assert np.abs(post.mvn.covariance_matrix[f, j].cpu().numpy() - co).max() < 1e-6 * np.abs(co).max()
                mo2, vo2 = O.predict(Xqb[j].cpu().numpy())
                assert np.abs(postb.mean[f, j, :, 0].cpu().numpy() - mo2).max() < 1e-6 * max(np.abs(mo2).max(), 1e-2)
                assert np.abs(postb.variance[f, j, :, 0].cpu().numpy() - vo2).max() < 1e-6 * vo2.max()
        # variances do not depend on the sampled targets (what qNIPV integrates)
        assert torch.equal(post.variance[0], post.variance[1])


def test_get_fantasy_model_shapes_and_errors():
    from online_gp_amd import settings

    with settings.cg_tolerance(1e-11):
        m, X, y, nz, gb, kind = _model("rbf", 2, 8, 30, True)
        x1 = torch.rand(2, 2, device=DEV, dtype=torch.float64) * 0.8 + 0.1
        plain = m.get_fantasy_model(x1, torch.zeros(2, device=DEV, dtype=torch.float64))            # unbatched: a sibling model
        assert type(plain) is type(m) and plain.num_data == 32
        fb = m.get_fantasy_model(x1.expand(4, 2, 2), torch.zeros(4, 2, device=DEV, dtype=torch.float64))        # same batch dims
        assert tuple(fb.batch_shape) == (4,)
        ff = m.get_fantasy_model(x1, torch.zeros(3, 2, device=DEV, dtype=torch.float64))            # one more target dim (BFN:292)
        assert tuple(ff.batch_shape) == (3,) and ff.posterior(x1).mean.shape == (3, 2, 1)
        with pytest.raises(RuntimeError, match="Unsupported batch shapes"):
            m.get_fantasy_model(x1.expand(4, 2, 2), torch.zeros(5, 3, 7, 2, device=DEV, dtype=torch.float64))
        with pytest.raises(RuntimeError, match="in place"):
            m.condition_on_observations(x1.expand(4, 2, 2), torch.zeros(4, 2, 1, device=DEV, dtype=torch.float64), inplace=True)
        # a conditioned copy of the unbatched kind equals one member of the batch
        one = m.condition_on_observations(x1, torch.ones(2, 1, device=DEV, dtype=torch.float64), torch.full((2, 1), 0.5, device=DEV, dtype=torch.float64))
        fbm = m.get_fantasy_model(x1[None], torch.ones(1, 2, device=DEV, dtype=torch.float64), torch.full((1, 2), 0.5, device=DEV, dtype=torch.float64))
        xq = torch.rand(6, 2, device=DEV, dtype=torch.float64) * 0.8 + 0.1
        a, bb = one(xq), fbm.posterior(xq).mvn
        assert torch.allclose(a.mean, bb.mean[0], rtol=1e-7, atol=1e-10) and torch.allclose(a.variance, bb.variance[0], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("g,dense", [(8, True), (14, False)])
def test_multi_output_fantasies_match_per_output_oracle(g, dense):
    """num_outputs = 2 (the Dirichlet classifier's layout: per-output targets AND per-output fixed noise): a batch of
    fantasy models with targets [F, b, q, out] equals, output by output, an exact GP re-fitted on that output's data plus
    the fantasy -- dense factor (8^2) and PCG path (14^3 is past max_cholesky_size)."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    d = 2 if dense else 3
    rng = np.random.default_rng(4)
    n, out = 50, 2
    X = rng.uniform(0, 1, (n, d))
    Y = np.stack([np.sin(4 * X[:, 0]) * np.cos(3 * X[:, -1]), np.cos(5 * X[:, 0]) + X[:, -1]], axis=1) + 0.05 * rng.standard_normal((n, out))
    NZ = rng.uniform(0.3, 0.9, (n, out))
    gb = [[0.0, 1.0]] * d
    with settings.dense_small_grids(dense), settings.cg_tolerance(1e-11), settings.spectral_factor(False):
        m = FixedNoiseOnlineSKIGP(torch.as_tensor(X, device=DEV), torch.as_tensor(Y, device=DEV), torch.as_tensor(NZ, device=DEV),
                                  grid_bounds=torch.tensor(gb, dtype=torch.float64), grid_size=g, learn_additional_noise=True).eval()
        b, q, F = 2, 3, 2
        Xc = torch.as_tensor(rng.uniform(0.1, 0.9, (b, q, d)), device=DEV)
        Yf = torch.as_tensor(rng.standard_normal((F, b, q, out)), device=DEV)
        Nf = torch.as_tensor(rng.uniform(0.2, 0.6, (b, q, out)), device=DEV)
        fm = m.get_fantasy_model(Xc, Yf, Nf)
        assert tuple(fm.batch_shape) == (F, b) and fm.num_data == n + q
        Xq = torch.as_tensor(rng.uniform(0.05, 0.95, (5, d)), device=DEV)
        post = fm.posterior(Xq)
        assert post.mean.shape == (F, b, 5, out) and post.variance.shape == (F, b, 5, out)
        assert post.mvn.mean.shape == (out, F, b, 5) and post.mvn.covariance_matrix.shape == (out, F, b, 5, 5)
        k = m.covar_module.base_kernel
        for o in range(out):
            ell = k.base_kernel.lengthscale.detach().double().cpu().numpy()[o].reshape(-1)
            osc = float(k.outputscale.detach().double()[o])
            s2 = float(m._sigma2(o))
            for f in range(F):
                for j in range(b):
                    Xa = np.concatenate([X, Xc[j].cpu().numpy()]); ya = np.concatenate([Y[:, o], Yf[f, j, :, o].cpu().numpy()])
                    na = np.concatenate([NZ[:, o], Nf[j, :, o].cpu().numpy()])
                    O = dataspace.DataSpaceGP(gb, g, "rbf", ell, osc, s2).fit(Xa, ya, na)
                    mo, co = O.predict(Xq.cpu().numpy(), full_cov=True)
                    assert np.abs(post.mean[f, j, :, o].cpu().numpy() - mo).max() < 1e-6 * max(np.abs(mo).max(), 1e-2)
                    assert np.abs(post.mvn.covariance_matrix[o, f, j].cpu().numpy() - co).max() < 1e-6 * np.abs(co).max()
