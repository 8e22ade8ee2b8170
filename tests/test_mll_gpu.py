"""GPU: the Woodbury MLL (value and hyper-parameter gradients) equals the exact-GP MLL on the
same SKI kernel -- the reference's one numeric parity test
(tests/mlls/test_batched_woodbury_marginal_log_likelihood.py:55-73), on its inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import dataspace, spec

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
TH = float(np.log(2.0))   # softplus(0); d theta / d raw = 0.5 there


def _dlog(grad_raw):
    return grad_raw / 0.5 * TH


def test_mll_value_and_gradients_match_exact_gp_single_and_batched():
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    G = np.load(os.path.join(GOLD, "case2_mll_2d.npz"))
    X, Y, N = [torch.as_tensor(G[k], device=DEV) for k in ("x", "y", "noise")]
    gb = torch.tensor([[0.0, 1.0], [0.0, 1.0]])
    # single output (test_fixed_noise)
    m = FixedNoiseOnlineSKIGP(X, Y[:, :1], N[:, :1], grid_bounds=gb, grid_size=5, learn_additional_noise=False)
    mll = BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)
    m.train()
    v = mll(m(X), Y[:, 0])
    v.sum().backward()
    assert abs(float(v.detach()) - float(G["mll_0"])) < 1e-7 * abs(float(G["mll_0"]))
    k = m.covar_module.base_kernel
    got = np.concatenate([_dlog(k.base_kernel.raw_lengthscale.grad.cpu().numpy().reshape(-1)), [_dlog(float(k.raw_outputscale.grad))]])
    assert np.abs(got - G["dmll_dlog_0"]).max() < 1e-5 * max(np.abs(G["dmll_dlog_0"]).max(), 1e-2)
    # 3-output batch (test_batched_fixed_noise): independent hyper-parameters per output
    mb = FixedNoiseOnlineSKIGP(X, Y, N, grid_bounds=gb, grid_size=5, learn_additional_noise=False)
    mllb = BatchedWoodburyMarginalLogLikelihood(mb.likelihood, mb)
    mb.train()
    vb = mllb(mb(X), Y)
    assert vb.shape == (3,)
    vb.sum().backward()
    kb = mb.covar_module.base_kernel
    for o in range(3):
        assert abs(float(vb[o].detach()) - float(G[f"mll_{o}"])) < 1e-7 * abs(float(G[f"mll_{o}"]))
        got = np.concatenate([_dlog(kb.base_kernel.raw_lengthscale.grad[o].cpu().numpy().reshape(-1)), [_dlog(float(kb.raw_outputscale.grad[o]))]])
        assert np.abs(got - G[f"dmll_dlog_{o}"]).max() < 1e-5 * max(np.abs(G[f"dmll_dlog_{o}"]).max(), 1e-2)


def test_mll_learnable_noise_gradient_matches_finite_difference():
    """The variant the reference leaves commented out (test file :84-86)."""
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (40, 2)); y = np.sin(2 * X[:, 0]) + X[:, 1] + 0.1 * rng.standard_normal(40); nz = rng.uniform(0.5, 1.5, 40)
    Xt, yt, nt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None], torch.as_tensor(nz, device=DEV)[:, None]
    m = FixedNoiseOnlineSKIGP(Xt, yt, nt, grid_bounds=torch.tensor([[-1.1, 1.1]] * 2), grid_size=8, learn_additional_noise=True)
    m.likelihood.second_noise = 0.37
    mll = BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)
    m.train()
    v = mll(m(Xt), yt)
    v.backward()
    s2 = float(m.likelihood.second_noise.detach())

    def ref(s):
        return dataspace.DataSpaceGP([[-1.1, 1.1]] * 2, 8, sigma2=s).fit(X, y, nz).mll()

    assert abs(float(v.detach()) - ref(s2)) < 1e-7 * abs(ref(s2))
    fd = (ref(s2 * (1 + 1e-5)) - ref(s2 * (1 - 1e-5))) / (2e-5 * s2)       # d/d sigma2
    raw = m.likelihood.second_noise_covar.raw_noise
    dsig_draw = float(torch.sigmoid(raw.detach()))                              # d softplus
    assert abs(float(raw.grad) / dsig_draw - fd) < 1e-4 * max(abs(fd), 1e-2)


def test_mll_stochastic_trace_and_slq_on_a_larger_grid():
    """Matrix-free branch (dense path disabled): Hutchinson gradient + stochastic-Lanczos logdet (loose
    tolerances), and the dense branch on the same model (exact) for comparison."""
    from online_gp_amd import settings
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
    from online_gp_amd.mlls.batched_woodbury_marginal_log_likelihood import num_trace_samples
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (300, 3)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2] + 0.1 * rng.standard_normal(300)
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    m = FixedNoiseOnlineSKIGP(Xt, yt, None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 3), grid_size=10, learn_additional_noise=True)
    mll = BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)
    m.train()
    with num_trace_samples(64), settings.dense_small_grids(False), settings.spectral_factor(False):
        v = mll(m(Xt), yt)
        v.backward()
        # fixed probes + warm starts: a second step after a streamed batch re-uses the probe vectors and starts their solves
        # from the first step's solutions; its gradient is compared with the exact (dense) one on the same statistics
        with num_trace_samples(48):
            for p_ in m.parameters():
                p_.grad = None
            mll(m(Xt), yt).backward()
            Z_first = m._mll_probes[0]["Z"].clone()
            m.condition_on_observations(Xt[:20] * 0.9, yt[:20], None, inplace=True)
            for p_ in m.parameters():
                p_.grad = None
            mll(m(Xt), yt).backward()
            g_second = m.covar_module.base_kernel.base_kernel.raw_lengthscale.grad.clone()
            assert m._mll_probes[0]["Z"].shape == (48, 1000) and not torch.equal(m._mll_probes[0]["Z"], Z_first)
        with settings.dense_small_grids(True):
            for p_ in m.parameters():
                p_.grad = None
            m.zero_grad()
            mll(m(Xt), yt).backward()
            g_second_dense = m.covar_module.base_kernel.base_kernel.raw_lengthscale.grad.clone()
        assert torch.isfinite(g_second).all() and (g_second - g_second_dense).abs().max() < 0.35 * g_second_dense.abs().max()
        m.set_train_data(Xt, yt, torch.ones_like(yt))           # back to the original statistics for the comparisons below
        m.train()
        for p_ in m.parameters():
            p_.grad = None
        v = mll(m(Xt), yt)
        v.backward()
    g_stoch = m.covar_module.base_kernel.base_kernel.raw_lengthscale.grad.clone()
    m.zero_grad()
    for p_ in m.parameters():
        p_.grad = None
    v_dense = mll(m(Xt), yt)
    v_dense.backward()
    g_dense = m.covar_module.base_kernel.base_kernel.raw_lengthscale.grad.clone()
    s2 = float(m.likelihood.second_noise.detach())
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * 3, 10, sigma2=s2).fit(X, y, np.ones(300))
    assert abs(float(v.detach()) - O.mll()) < 0.05 * abs(O.mll())
    assert abs(float(v_dense.detach()) - O.mll()) < 1e-6 * abs(O.mll())
    assert torch.isfinite(g_stoch).all() and g_stoch.abs().sum() > 0
    assert (g_stoch - g_dense).abs().max() < 0.3 * g_dense.abs().max()          # 64 Rademacher probes


def test_update_with_hyperparameter_step_and_fit():
    """OnlineSKIRegression.update(update_gp=True) = one Adam step on -MLL under skip_logdet_forward,
    then condition_on_observations (OSR:113-146); fit() runs batch epochs (OSR:80-111)."""
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (260, 2)); y = np.sin(3 * X[:, :1]) + 0.1 * rng.standard_normal((260, 1))
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=torch.float32), torch.as_tensor(y, device=DEV, dtype=torch.float32)
    r = OnlineSKIRegression(Identity(2), Xt[:100], yt[:100], 5e-2, 12, 1.0)
    recs = r.fit(Xt[:100], yt[:100], 3)
    assert len(recs) == 3 and all(np.isfinite(rec["train_loss"]) for rec in recs)
    assert recs[-1]["train_loss"] < recs[0]["train_loss"] + 1e-3
    before = r.gp.covar_module.base_kernel.base_kernel.raw_lengthscale.detach().clone()
    for s in range(100, 260, 40):
        stem_loss, gp_loss = r.update(Xt[s:s + 40], yt[s:s + 40])
        assert stem_loss == 0 and np.isfinite(gp_loss)
    assert r.gp.num_data == 260
    assert (r.gp.covar_module.base_kernel.base_kernel.raw_lengthscale.detach() - before).abs().max() > 0
    rmse, nll = r.evaluate(Xt[:100], yt[:100])
    assert rmse < 0.5 and np.isfinite(nll)


def test_gather_grad_matches_finite_differences():
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(0)
    for d, g in [(1, 12), (2, 9), (3, 8)]:
        grid = grid_ops.GridSpec([[-1.1, 1.1]] * d, g)
        X = rng.uniform(-0.95, 0.95, (20, d))
        V = rng.standard_normal((20, grid.m))
        err = grid_ops.new_err_flag(DEV)
        Xt, Vt = torch.as_tensor(X, device=DEV), torch.as_tensor(V, device=DEV)
        for diag in (False, True):
            Vuse = Vt if diag else Vt[:1]
            got = grid_ops.gather_grad(grid, Xt, Vuse, diag=diag).cpu().numpy()
            fd = np.zeros((20, d))
            for q in range(d):
                for sgn in (+1, -1):
                    Xp = X.copy(); Xp[:, q] += sgn * 1e-6
                    out = grid_ops.gather(grid, torch.as_tensor(Xp, device=DEV), Vuse, err, diag=diag)
                    out = out if diag else out[:, 0]
                    fd[:, q] += sgn * out.cpu().numpy() / 2e-6
            assert np.abs(got - fd).max() < 1e-5 * max(np.abs(fd).max(), 1.0)


def test_sm_partial_mll_value_matches_dense_restatement_and_is_differentiable():
    """Value vs the op-for-op restatement of streaming_partial_mll.py (oracle/dense_reference.py);
    gradient w.r.t. the new input vs finite differences of the same restatement."""
    from oracle import dense_reference
    from online_gp_amd.mlls import sm_partial_mll
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(2)
    X = rng.uniform(-0.9, 0.9, (60, 2)); y = np.sin(2 * X[:, 0]) + 0.5 * X[:, 1] + 0.1 * rng.standard_normal(60)
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    m = FixedNoiseOnlineSKIGP(Xt, yt, None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 2), grid_size=9, learn_additional_noise=True)
    m.eval()
    s2 = float(m.likelihood.second_noise.detach())
    B1 = dense_reference.DenseWISKI([[-1.1, 1.1]] * 2, 9, sigma2=s2, chol_jitter=1e-10)
    B1.set_train_data(X, y, np.ones(60))
    xn = np.array([[0.31, -0.42]]); yn = 0.7
    xt = torch.tensor(xn, device=DEV, requires_grad=True)
    val = sm_partial_mll(m, xt, torch.tensor([[yn]], device=DEV), 60)
    ref = B1.sm_partial_mll(xn, yn)
    assert abs(float(val.detach()) - ref) < 1e-6 * abs(ref)
    val.sum().backward()
    fd = np.zeros(2)
    for q in range(2):
        for sgn in (+1, -1):
            xp = xn.copy(); xp[0, q] += sgn * 1e-5
            fd[q] += sgn * B1.sm_partial_mll(xp, yn) / 2e-5
    assert np.abs(xt.grad.cpu().numpy()[0] - fd).max() < 1e-4 * max(np.abs(fd).max(), 1e-3)


@pytest.mark.parametrize("kind", ["linear", "mlp"])
def test_learned_stem_streaming_update_runs(kind):
    from online_gp_amd.models import MLP, LinearStem, OnlineSKIRegression

    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (400, 5)); y = np.sin(X[:, :1] + X[:, 1:2]) + 0.1 * rng.standard_normal((400, 1))
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=torch.float32), torch.as_tensor(y, device=DEV, dtype=torch.float32)
    stem = (LinearStem(5, 2) if kind == "linear" else MLP(5, 2, 2, "16,8")).to(DEV)     # (the reference's config strings: "hidden_dims: 16,8")
    r = OnlineSKIRegression(stem, Xt[:200], yt[:200], 1e-2, 16, 1.0)
    w0 = stem[0].weight.detach().clone()
    for s in range(200, 260, 20):
        stem_loss, gp_loss = r.update(Xt[s:s + 20], yt[s:s + 20])
        assert np.isfinite(stem_loss) and np.isfinite(gp_loss) and stem_loss != 0
    assert (stem[0].weight.detach() - w0).abs().max() > 0
    rmse, nll = r.evaluate(Xt[300:], yt[300:])
    assert np.isfinite(rmse) and np.isfinite(nll)


def test_mll_feature_gradient_matches_finite_differences_and_fit_trains_the_stem():
    """d(-MLL)/d features (mlls/feature_gradient.py: joint stem + GP training of OSR.fit, OSR:80-112) against central
    differences of the MLL itself, single- and multi-output with heteroscedastic noise; then fit() moves a LinearStem."""
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood, mll_feature_surrogate
    from online_gp_amd.models import FixedNoiseOnlineSKIGP, LinearStem, OnlineSKIRegression

    rng = np.random.default_rng(5)
    n, d, out = 36, 2, 2
    X = torch.as_tensor(rng.uniform(-0.9, 0.9, (n, d)), device=DEV)
    Y = torch.as_tensor(np.stack([np.sin(2 * X[:, 0].cpu().numpy()) + X[:, 1].cpu().numpy(), np.cos(X[:, 0].cpu().numpy() * X[:, 1].cpu().numpy())], 1)
                        + 0.1 * rng.standard_normal((n, out)), device=DEV)
    N = torch.as_tensor(rng.uniform(0.5, 1.5, (n, out)), device=DEV)
    gb = torch.tensor([[-1.1, 1.1]] * d)

    def neg_mll(Xq):
        m = FixedNoiseOnlineSKIGP(Xq, Y, N, grid_bounds=gb, grid_size=8, learn_additional_noise=True).double()
        mll = BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)
        m.train()
        return m, -mll(m(Xq), Y).sum()

    Xg = X.clone().requires_grad_(True)
    m, loss = neg_mll(Xg.detach())
    surr = mll_feature_surrogate(m, Xg, Y, N)
    assert abs(float(surr.detach())) == 0.0
    surr.backward()
    g = Xg.grad.cpu().numpy()
    eps = 1e-5
    for (i, j) in [(0, 0), (7, 1), (20, 0), (35, 1)]:
        Xp, Xm = X.clone(), X.clone()
        Xp[i, j] += eps; Xm[i, j] -= eps
        fd = (float(neg_mll(Xp)[1]) - float(neg_mll(Xm)[1])) / (2 * eps)
        assert abs(g[i, j] - fd) < 1e-5 * max(1.0, abs(fd)), (i, j, g[i, j], fd)

    # fit(): the stem parameters receive gradients and move; the training loss goes down
    torch.manual_seed(0)
    Xh = torch.randn(120, 5, device=DEV, dtype=torch.float32)
    yh = (torch.tanh(Xh[:, :1] - 0.5 * Xh[:, 1:2]) + 0.05 * torch.randn(120, 1, device=DEV))
    stem = LinearStem(5, 2)
    model = OnlineSKIRegression(stem, Xh, yh, 5e-2, 8, 1.0)
    w0 = [p.detach().clone() for p in model.stem.parameters()]
    rec = model.fit(Xh, yh, 15)
    assert any((p.detach() - q).abs().max() > 1e-4 for p, q in zip(model.stem.parameters(), w0))
    assert rec[-1]["train_loss"] < rec[0]["train_loss"]


def test_mll_feature_gradient_matrix_free_regime():
    """Beyond max_cholesky_size the trace part of d(-MLL)/d features is a Hutchinson estimate over probe solves
    (mlls/feature_gradient.py).  With the identity as probes (weight 1) the estimate is exact: it must equal the
    dense-regime gradient / central differences of the exact MLL up to the CG tolerance; with Rademacher probes it must
    agree statistically; and fit() on a grid past the dense regime now moves the stem."""
    from online_gp_amd import settings
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood, mll_feature_surrogate
    from online_gp_amd.models import FixedNoiseOnlineSKIGP, LinearStem, OnlineSKIRegression

    rng = np.random.default_rng(6)
    n, d = 40, 3
    X = torch.as_tensor(rng.uniform(-0.9, 0.9, (n, d)), device=DEV)
    Y = torch.as_tensor(np.sin(2 * X[:, :1].cpu().numpy()) + 0.1 * rng.standard_normal((n, 1)), device=DEV)
    N = torch.as_tensor(rng.uniform(0.5, 1.5, (n, 1)), device=DEV)
    gb = torch.tensor([[-1.1, 1.1]] * d)

    def build(Xq):
        return FixedNoiseOnlineSKIGP(Xq, Y, N, grid_bounds=gb, grid_size=6, learn_additional_noise=True).double()

    # dense-regime gradient (checked against central differences by the test above)
    Xd = X.clone().requires_grad_(True)
    mll_feature_surrogate(build(X), Xd, Y, N).backward()
    g_dense = Xd.grad.clone()
    with settings.max_cholesky_size(10), settings.cg_tolerance(1e-10):
        m = build(X)
        assert not m._use_dense()
        Xg = X.clone().requires_grad_(True)
        mll_feature_surrogate(m, Xg, Y, N, probes=torch.eye(m._grid.m, device=DEV, dtype=torch.float64), probe_weight=1.0).backward()
        assert (Xg.grad - g_dense).abs().max() < 1e-6 * g_dense.abs().max()
        from online_gp_amd.mlls.batched_woodbury_marginal_log_likelihood import num_trace_samples

        Xr = X.clone().requires_grad_(True)
        with num_trace_samples(4000):
            mll_feature_surrogate(build(X), Xr, Y, N).backward()
        assert (Xr.grad - g_dense).norm() < 0.1 * g_dense.norm()
        # fit(): a stem on a grid past the dense regime receives gradients and moves
        torch.manual_seed(0)
        Xh = torch.randn(96, 5, device=DEV, dtype=torch.float32)
        yh = torch.tanh(Xh[:, :1] - 0.5 * Xh[:, 1:2]) + 0.05 * torch.randn(96, 1, device=DEV)
        model = OnlineSKIRegression(LinearStem(5, 2), Xh, yh, 5e-2, 8, 1.0)
        assert not model.gp._use_dense()
        w0 = [p.detach().clone() for p in model.stem.parameters()]
        model.fit(Xh, yh, 5)
        assert any((p.detach() - q).abs().max() > 1e-4 for p, q in zip(model.stem.parameters(), w0))


def test_mll_adds_registered_priors_before_dividing_by_n():
    """BWM:48-51: res += sum_priors log p(theta); res /= n.  The reference's BO kernel (Matern-5/2, Gamma priors, Interval
    constraints, experiments/bayesopt/bayesopt.py:69-77): value = prior-free MLL at the same hyper-parameters + log-priors / n,
    and the hyper-parameter gradients pick up the prior term."""
    from online_gp_amd.constraints import Interval
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, ScaleKernel
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
    from online_gp_amd.models import FixedNoiseOnlineSKIGP
    from online_gp_amd.priors import GammaPrior

    rng = np.random.default_rng(4)
    X = rng.uniform(0, 1, (30, 2)); y = np.cos(3 * X[:, 0]) * X[:, 1]
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    gb = torch.tensor([[0.0, 1.0]] * 2)

    def build(with_priors):
        kw_l = dict(lengthscale_prior=GammaPrior(3.0, 6.0)) if with_priors else {}
        kw_s = dict(outputscale_prior=GammaPrior(2.0, 0.15)) if with_priors else {}
        base = ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=2, lengthscale_constraint=Interval(1e-4, 12.0), **kw_l),
                           outputscale_constraint=Interval(1e-4, 12.0), **kw_s)
        cov = GridInterpolationKernel(base, grid_size=8, num_dims=2, grid_bounds=gb)
        base.outputscale = 1.3
        base.base_kernel.lengthscale = torch.tensor([0.4, 0.9])
        m = FixedNoiseOnlineSKIGP(Xt, yt, torch.ones_like(yt), covar_module=cov, learn_additional_noise=True).to(DEV)
        m.train()
        return m, BatchedWoodburyMarginalLogLikelihood(m.likelihood, m)

    m0, mll0 = build(False)
    m1, mll1 = build(True)
    v0, v1 = mll0(None, None), mll1(None, None)
    ls, osc = m1.covar_module.base_kernel.base_kernel.lengthscale.detach().double(), m1.covar_module.base_kernel.outputscale.detach().double()
    lp = torch.distributions.Gamma(3.0, 6.0).log_prob(ls.cpu()).sum() + torch.distributions.Gamma(2.0, 0.15).log_prob(osc.cpu()).sum()
    assert abs(float(v1.detach()) - float(v0.detach()) - float(lp) / 30) < 1e-6      # raw parameters are fp32
    v0.backward(); v1.backward()
    g0 = m0.covar_module.base_kernel.raw_outputscale.grad; g1 = m1.covar_module.base_kernel.raw_outputscale.grad
    # d/d raw of log Gamma(2, 0.15)(s) / n with s = lo + (hi - lo) sigmoid(raw)
    s = float(osc); dsdraw = (s - 1e-4) * (12.0 - s) / (12.0 - 1e-4)
    want = ((2.0 - 1.0) / s - 0.15) * dsdraw / 30
    assert abs(float(g1 - g0) - want) < 1e-6


@pytest.mark.parametrize("nout", [1, 2])
def test_dense_mll_reuses_the_posterior_factor_evaluate_has_just_built(nout):
    """Dense regime, the reference's step order (evaluate -> MLL step at the same hyper-parameters and data): the MLL's forward takes the
    model's current posterior factor instead of building its own (mlls/...: _current_dense_posterior).  Value and gradients equal
    those of a model that has no prediction cache; after zero_grad() (the reference's signal that the hyper-parameters moved), after
    new data and after a parameter write the cached factor is NOT taken."""
    from online_gp_amd.lazy import dense_woodbury
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(4)
    n = 60
    X = rng.uniform(-1, 1, (n, 2)); Y = np.stack([np.sin(2 * X[:, 0]) + X[:, 1], np.cos(X[:, 0] * X[:, 1])], 1)[:, :nout] + 0.1 * rng.standard_normal((n, nout))
    Xt, Yt = torch.as_tensor(X, device=DEV), torch.as_tensor(Y, device=DEV)
    gb = torch.tensor([[-1.1, 1.1]] * 2, dtype=torch.float64)

    def build():
        return FixedNoiseOnlineSKIGP(Xt[:40], Yt[:40], None, grid_bounds=gb, grid_size=8, learn_additional_noise=True)

    def value_and_grads(model):
        mll = BatchedWoodburyMarginalLogLikelihood(model.likelihood, model)
        for p in model.parameters():
            p.grad = None
        v = mll(None, None).sum()
        v.backward()
        return float(v.detach()), [p.grad.detach().clone() for p in model.parameters() if p.requires_grad and p.grad is not None]

    built = []
    orig = dense_woodbury.DenseInducingPosterior.__init__

    def counting(self, *a, **k):
        built.append(1)
        return orig(self, *a, **k)

    dense_woodbury.DenseInducingPosterior.__init__ = counting
    try:
        ref = build(); ref.train()
        v0, g0 = value_and_grads(ref)                                  # no cache: builds its own factor(s)
        assert len(built) == nout
        m = build(); m.eval()
        m(Xt[40:44]).variance                                          # evaluate(): the model builds and caches its factor(s)
        nb = len(built)
        v1, g1 = value_and_grads(m)                                    # the MLL step takes them
        assert len(built) == nb
        assert abs(v1 - v0) <= 1e-12 * abs(v0) and len(g0) == len(g1)
        for a, b in zip(g0, g1):
            assert (a - b).abs().max().item() <= 1e-10 * max(a.abs().max().item(), 1e-3)
        m.zero_grad()                                                  # "the hyper-parameters may have moved": cache gone
        nb = len(built); value_and_grads(m); assert len(built) == nb + nout
        m.eval(); m(Xt[40:44]).variance
        m.condition_on_observations(Xt[40:48], Yt[40:48], inplace=True)  # new data: the cached factor is stale (or a pending rank update)
        nb = len(built); v2, _ = value_and_grads(m)
        r2 = FixedNoiseOnlineSKIGP(Xt[:48], Yt[:48], None, grid_bounds=gb, grid_size=8, learn_additional_noise=True); r2.train()
        v3, _ = value_and_grads(r2)
        assert abs(v2 - v3) <= 1e-9 * abs(v3)
    finally:
        dense_woodbury.DenseInducingPosterior.__init__ = orig
