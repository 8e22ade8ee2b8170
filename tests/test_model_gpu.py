"""GPU: model-surface parity.  The HIP path (through the reference-compatible classes)
against the golden fixtures and the data-space oracle; shape contracts of the
reference's live test (tests/models/test_batched_online_ski_gp_model.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import dataspace, spec

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"

RTOL = {torch.float64: 1e-4, torch.float32: 1e-2}   # north-star tolerances (BASELINE.json)
# fp32 assertions INSIDE the north-star bar: about 3x the deviation measured on MI355X (printed by the tests as "MEASURED ...";
# profiles/r06_parity_measured.txt), so that a regression that stays below 1e-2 is still caught
TIGHT_MEAN_6K, TIGHT_VAR_6K = 5e-5, 3e-5                        # measured 1.3e-5 / 7.5e-6
TIGHT_VAR_C3 = {"default path": 6e-3, "pcg path": 3e-3}         # measured 5.4e-4 / 5.2e-4 (uniform), 2.1e-3 / 1.1e-3 (road-like: 434 874 points of fp32 statistics)
TIGHT_CLF_MEAN = 1e-5                                           # measured 2.7e-6


def _mk_kernel(kind, d, gb, g, ell, osc):
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, RBFKernel, ScaleKernel

    base = RBFKernel(ard_num_dims=d) if kind == "rbf" else MaternKernel(nu={"matern52": 2.5, "matern32": 1.5, "matern12": 0.5}[kind], ard_num_dims=d)
    k = GridInterpolationKernel(ScaleKernel(base), grid_size=g, num_dims=d, grid_bounds=gb)
    k.base_kernel.outputscale = osc
    k.base_kernel.base_kernel.lengthscale = torch.as_tensor(np.broadcast_to(ell, (d,)).copy())
    return k


def _stream_files():
    return sorted(glob.glob(os.path.join(GOLD, "case1_*.npz")) + glob.glob(os.path.join(GOLD, "case4_*.npz")) +
                  glob.glob(os.path.join(GOLD, "case5_*.npz")))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("path", _stream_files(), ids=os.path.basename)
def test_streaming_posterior_matches_golden(path, dtype):
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    G = np.load(path, allow_pickle=True)
    gb, g, kind = G["grid_bounds"], G["grid_size"].tolist(), str(G["kind"])
    d = gb.shape[0]
    if dtype == torch.float32 and "case1" in path:
        pytest.skip("sigma2 = 0.01 with lengthscale 10: conditioning beyond fp32 (the reference test is fp64 too)")
    Xs = torch.as_tensor(G["test_x"], device=DEV, dtype=dtype)
    model = None
    for i in range(int(G["n_chunks"])):
        x = torch.as_tensor(G[f"x_{i}"], device=DEV, dtype=dtype)
        y = torch.as_tensor(G[f"y_{i}"], device=DEV, dtype=dtype)[:, None]
        nz = torch.as_tensor(G[f"noise_{i}"], device=DEV, dtype=dtype)[:, None]
        if model is None:
            model = FixedNoiseOnlineSKIGP(x, y, nz, covar_module=_mk_kernel(kind, d, gb, g, G["lengthscale"], float(G["outputscale"])),
                                          learn_additional_noise=True)
            model.likelihood.second_noise = float(G["sigma2"])
            model.eval()
        else:
            model.condition_on_observations(x, y, nz, inplace=True)
        mvn = model(Xs)
        mean, cov = G[f"mean_{i}"], G[f"cov_{i}"]
        rt = RTOL[dtype]
        assert np.abs(mvn.mean.double().cpu().numpy() - mean).max() <= rt * np.abs(mean).max()
        assert np.abs(mvn.variance.double().cpu().numpy() - np.diag(cov)).max() <= rt * np.abs(np.diag(cov)).max()
        if i == int(G["n_chunks"]) - 1:
            assert np.abs(mvn.covariance_matrix.double().cpu().numpy() - cov).max() <= rt * np.abs(cov).max()
    assert model.num_data == sum(len(G[f"y_{i}"]) for i in range(int(G["n_chunks"])))


def test_batch_outputs_golden_and_cache_shapes():
    """Shapes asserted by the reference's live test (test_batched_online_ski_gp_model.py:96-104)
    plus values for the 2-output heteroscedastic case."""
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    G = np.load(os.path.join(GOLD, "case3_batch_1d.npz"))
    X, Y, N = [torch.as_tensor(G[k], device=DEV) for k in ("x", "y", "noise")]
    model = FixedNoiseOnlineSKIGP(X[:5], Y[:5], N[:5], grid_bounds=torch.tensor([[0.0, 1.0]]), grid_size=10)
    model.train()
    dist = model(*model.train_inputs)
    assert dist.mean.shape == torch.Size((2, 5)) and dist.mean.norm() <= 1e-5
    assert dist.covariance_matrix.shape == torch.Size((2, 5, 5))
    c = model._kernel_cache
    assert c["WtW"].shape == torch.Size((2, 10, 10))
    assert c["D_logdet"].shape == torch.Size((2,))
    assert c["response_cache"].shape == torch.Size((2, 1, 1))
    assert c["interpolation_cache"].shape == torch.Size((2, 10, 1))
    new_model = model.condition_on_observations(X[5:], Y[5:], N[5:], inplace=False)
    assert new_model.num_data == 10 and model.num_data == 5
    model.condition_on_observations(X[5:], Y[5:], N[5:], inplace=True)
    assert model.num_data == 10
    Xs = torch.as_tensor(G["test_x"], device=DEV)
    for mm in (model, new_model):
        mm.eval()
        d = mm(Xs)
        assert d.mean.shape == torch.Size((2, 5)) and d.covariance_matrix.shape == torch.Size((2, 5, 5))
        for o in range(2):
            assert np.abs(d.mean[o].cpu().numpy() - G[f"mean_{o}"]).max() < 1e-4 * np.abs(G[f"mean_{o}"]).max()
            assert np.abs(d.covariance_matrix[o].cpu().numpy() - G[f"cov_{o}"]).max() < 1e-4 * np.abs(G[f"cov_{o}"]).max()


def test_nonbatch_shapes_and_dense_wtw_view():
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    torch.manual_seed(0)
    x = torch.rand(10, 1, device=DEV, dtype=torch.float64)
    y = torch.sin(3.0 * x)
    model = FixedNoiseOnlineSKIGP(x[:5], y[:5], 0.01 * y[:5] ** 2 + 1e-4, grid_bounds=torch.tensor([[0.0, 1.0]]), grid_size=10)
    model.train()
    d = model(*model.train_inputs)
    assert d.mean.shape == torch.Size((5,)) and d.covariance_matrix.shape == torch.Size((5, 5))
    model.eval()
    d = model(x[5:])
    assert d.mean.shape == torch.Size((5,)) and d.covariance_matrix.shape == torch.Size((5, 5))
    # LazyTensor contract of the WtW operator: matmul / evaluate / symmetric
    A = model._kernel_cache["WtW"]
    dense = A.evaluate()
    assert dense.shape == (10, 10) and torch.allclose(dense, dense.t(), atol=1e-12)
    v = torch.randn(10, 3, device=DEV, dtype=torch.float64)
    assert torch.allclose(A @ v, dense @ v, atol=1e-12)
    K = model.Kuu.evaluate()
    assert torch.allclose(K, K.t(), atol=1e-12) and torch.allclose(model.Kuu_response[0], K @ model._kernel_cache["interpolation_cache"][0])


def test_out_of_bounds_raises_like_gpytorch():
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    x = torch.rand(8, 2, device=DEV, dtype=torch.float64)
    model = FixedNoiseOnlineSKIGP(x, x.sum(1, keepdim=True), None, grid_bounds=torch.tensor([[0.0, 1.0]] * 2), grid_size=8)
    model.eval()
    model.condition_on_observations(torch.tensor([[5.0, 0.5]], device=DEV, dtype=torch.float64), torch.ones(1, 1, device=DEV, dtype=torch.float64),
                                    inplace=True)
    with pytest.raises(RuntimeError, match="out of bounds"):
        model(x[:2]).mean


def test_float32_grid_quirk_does_not_move_the_posterior():
    """gpytorch creates the grid in float32 and promotes it; the spec'd geometry is float64 (VERDICT r1
    'Parity').  With the quirk switched on the grid moves by ~1e-7 relative and the posterior by far less than the
    1e-4 parity tolerance, in the dense and in the PCG regime."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(9)
    for d, g in ((2, 10), (3, 14)):
        X = rng.uniform(-1, 1, (300, d)); y = np.sin(2 * X.sum(1)) + 0.1 * rng.standard_normal(300)
        Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
        res = []
        for quirk in (False, True):
            with settings.float32_grid(quirk):
                m = FixedNoiseOnlineSKIGP(Xt[:200], yt[:200], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64), grid_size=g,
                                          learn_additional_noise=True).eval()
            m.condition_on_observations(Xt[200:], yt[200:], inplace=True)
            mvn = m(Xt[:40])
            res.append((mvn.mean.clone(), mvn.variance.clone(), m._grid.g0[0], m._grid.h[0]))
        (m0, v0, g00, h0), (m1, v1, g01, h1) = res
        assert (g00, h0) != (g01, h1) and abs(g00 - g01) < 5e-7 and abs(h0 - h1) < 5e-6 * h0
        assert float((m0 - m1).abs().max()) < 1e-5 * float(m0.abs().max())
        assert float((v0 - v1).abs().max()) < 1e-5 * float(v0.abs().max())


@pytest.mark.parametrize("g", [8, 24])     # 8^2: dense factor; 24^3 > max_cholesky_size: PCG path (flag rides on the solver poll)
def test_out_of_grid_points_leave_consistent_statistics(g):
    """Advisor finding r1: (a) a query outside the grid raises inside the posterior call, not at the next update;
    (b) a training point outside the grid is dropped whole -- no y^2 / log-noise / num_data / weight-sum residue -- so
    a caller that catches the error keeps a model that equals the oracle on the points inside the grid."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    d = 2 if g == 8 else 3
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, (60, d)); y = np.sin(3 * X.sum(1)); nz = rng.uniform(0.5, 1.5, 60)
    Xt = torch.as_tensor(X, device=DEV); yt = torch.as_tensor(y, device=DEV)[:, None]; nt = torch.as_tensor(nz, device=DEV)[:, None]
    gb = [[0.0, 1.0]] * d
    model = FixedNoiseOnlineSKIGP(Xt[:40], yt[:40], nt[:40], grid_bounds=torch.tensor(gb), grid_size=g, learn_additional_noise=True)
    model.eval()
    # (a) query outside: raises here, and the model is untouched
    bad_q = Xt[:3].clone(); bad_q[1, 0] = 7.0
    with pytest.raises(RuntimeError, match="out of bounds"):
        model(bad_q)
    m_ok = model(Xt[40:45]).mean
    assert torch.isfinite(m_ok).all() and model.num_data == 40
    with settings.deferred_bounds_check(True):          # deferred mode: the flag waits for the next poll / check_bounds
        model(bad_q)
        with pytest.raises(RuntimeError, match="out of bounds"):
            model.check_bounds()
    # (b) a batch with two points outside: the error surfaces at the next posterior request, the 18 good points stay in
    xb = Xt[40:].clone(); xb[3, 1] = -4.0; xb[11, 0] = 2.5
    model.condition_on_observations(xb, yt[40:], nt[40:], inplace=True)
    with pytest.raises(RuntimeError, match="out of bounds"):
        model(Xt[:2]).mean
    assert model.num_data == 58
    keep = np.ones(60, bool); keep[[43, 51]] = False
    s2 = float(model.likelihood.second_noise.detach())
    ell = model.covar_module.base_kernel.base_kernel.lengthscale.detach().cpu().numpy().reshape(-1)
    osc = float(model.covar_module.base_kernel.outputscale)
    O = dataspace.DataSpaceGP(gb, g, "rbf", ell, osc, s2).fit(X[keep], y[keep], nz[keep])
    mo, vo = O.predict(X[:7])
    mvn = model(Xt[:7])
    assert np.abs(mvn.mean.cpu().numpy() - mo).max() < 1e-6 * np.abs(mo).max()
    assert np.abs(mvn.variance.cpu().numpy() - vo).max() < 1e-6 * np.abs(vo).max()
    stats = model._kernel_cache["response_cache"].reshape(-1).cpu().numpy()
    assert abs(stats[0] - np.sum(y[keep] ** 2 / nz[keep])) < 1e-9 * abs(stats[0])
    assert abs(float(model._kernel_cache["D_logdet"].reshape(-1)[0]) - np.log(nz[keep]).sum()) < 1e-9 * 60
    assert abs(model._wsum[0] - np.sum(1.0 / nz[keep])) < 1e-9 * 60


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_online_ski_regression_wrapper(dtype):
    """OnlineSKIRegression surface (OSR:16-197): grid bounds +0.1, noise == 1, predict adds sigma2."""
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (300, 2)); y = np.sin(3 * X[:, :1]) * np.cos(2 * X[:, 1:]) + 0.05 * rng.standard_normal((300, 1))
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=dtype), torch.as_tensor(y, device=DEV, dtype=dtype)
    r = OnlineSKIRegression(Identity(2), Xt[:100], yt[:100], 1e-2, 16, 1.0)
    assert r.gp.covar_module.grid_bounds[0] == pytest.approx((-1.1, 1.1))
    for s in range(100, 300, 50):
        rmse, nll = r.evaluate(Xt[s:s + 50], yt[s:s + 50])
        assert np.isfinite(rmse) and np.isfinite(nll)
        r.update(Xt[s:s + 50], yt[s:s + 50], update_gp=False)
    assert r.gp.num_data == 300
    pm, pv = r.predict(Xt[:7])
    assert pm.shape == (7, 1) and pv.shape == (7, 1)
    s2 = float(r.gp.likelihood.second_noise.detach())
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * 2, 16, sigma2=s2).fit(X, y[:, 0], np.ones(300))
    mo, vo = O.predict(X[:7])
    rt = RTOL[dtype]
    assert np.abs(pm[:, 0].double().cpu().numpy() - mo).max() <= rt * np.abs(mo).max()
    assert np.abs(pv[:, 0].double().cpu().numpy() - s2 - vo).max() <= rt * vo.max()
    # set_train_data rebuilds the statistics from scratch
    r.set_train_data(Xt[:50], yt[:50])
    O2 = dataspace.DataSpaceGP([[-1.1, 1.1]] * 2, 16, sigma2=s2).fit(X[:50], y[:50, 0], np.ones(50))
    pm2, _ = r.predict(Xt[:7])
    assert np.abs(pm2[:, 0].double().cpu().numpy() - O2.predict(X[:7])[0]).max() <= rt * np.abs(mo).max()


def test_botorch_adaptor_batched_posterior_and_cache_handover():
    """OnlineSKIBotorchModel.posterior on b x q x d inputs (OSB:63-68) and the kernel_cache
    hand-over path of experiments/bayesopt/bayesopt.py:86-96."""
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, ScaleKernel
    from online_gp_amd.models import OnlineSKIBotorchModel

    rng = np.random.default_rng(2)
    X = rng.uniform(0, 1, (40, 3)); y = -np.linalg.norm(X - 0.5, axis=1, keepdims=True)
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)
    gb = torch.tensor([[-32.768, 32.768]] * 3)   # the reference's BO quirk: raw Ackley bounds, unit-cube inputs
    cov = GridInterpolationKernel(ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=3)), grid_size=10, num_dims=3, grid_bounds=gb)
    m0 = OnlineSKIBotorchModel(Xt[:10], yt[:10], None, covar_module=cov, learn_additional_noise=True)
    m1 = m0.condition_on_observations(Xt[10:], yt[10:])
    m2 = OnlineSKIBotorchModel(covar_module=m1.covar_module, kernel_cache=m1._kernel_cache, learn_additional_noise=True,
                               likelihood=m1.likelihood, num_data=m1.num_data)
    assert abs(m2._wsum[0] - m1._wsum[0]) < 1e-6 * m1._wsum[0]        # recovered from the row-sum statistic of the cache
    Xq = torch.as_tensor(rng.uniform(0, 1, (4, 3, 3)), device=DEV)
    p1, p2 = m1.posterior(Xq), m2.posterior(Xq)
    assert p1.mean.shape == (4, 3, 1) and p1.variance.shape == (4, 3, 1) and p1.mvn.covariance_matrix.shape == (4, 3, 3)
    assert torch.allclose(p1.mean, p2.mean, rtol=1e-6, atol=1e-9)
    s2 = float(m1.likelihood.second_noise.detach())
    ell = float(cov.base_kernel.base_kernel.lengthscale.detach()[0, 0]); osc = float(cov.base_kernel.outputscale.detach())
    O = dataspace.DataSpaceGP(gb.numpy(), 10, "matern52", ell, osc, s2).fit(X, y[:, 0], np.ones(40))
    for b in range(4):
        mo, co = O.predict(Xq[b].cpu().numpy(), full_cov=True)
        assert np.abs(p1.mean[b, :, 0].cpu().numpy() - mo).max() < 1e-4 * max(np.abs(mo).max(), 1e-3)
        assert np.abs(p1.mvn.covariance_matrix[b].cpu().numpy() - co).max() < 1e-4 * np.abs(co).max()
    assert p1.rsample(torch.Size([5])).shape == (5, 4, 3, 1)


def test_c3_scale_properties_50cubed_fp32():
    """BASELINE size (d=3, 50^3, fp32): size-independent properties of the HIP operators and
    parity of a streamed posterior against the data-space oracle on a 6k-point prefix."""
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    torch.manual_seed(0)
    n = 6000
    X = torch.rand(n, 3, device=DEV, dtype=torch.float32) * 2 - 1
    y = (torch.sin(2 * np.pi * X[:, 0]) * torch.cos(np.pi * X[:, 1]) + 0.5 * X[:, 2] + 0.1 * torch.randn(n, device=DEV))[:, None]
    model = FixedNoiseOnlineSKIGP(X[:2000], y[:2000], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 3), grid_size=50, learn_additional_noise=True)
    model.eval()
    for s in range(2000, n, 1000):
        model.condition_on_observations(X[s:s + 1000], y[s:s + 1000], inplace=True)
    A, K = model._kernel_cache["WtW"], model.Kuu
    grid = model._grid
    u, v = torch.randn(grid.m, device=DEV), torch.randn(grid.m, device=DEV)
    for op in (A, K):                                        # symmetry + linearity
        assert abs(float(u @ (op @ v)) - float(v @ (op @ u))) < 2e-4 * float(u.norm() * (op @ v).norm())
        assert torch.allclose(op @ (2 * u + v), 2 * (op @ u) + (op @ v), rtol=1e-4, atol=1e-4 * float((op @ u).abs().max()))
    # W^T W 1 = W^T 1  (rows of W sum to one): stencil row sums == scattered counts
    ones = torch.ones(grid.m, device=DEV)
    b1 = torch.zeros(grid.m, device=DEV); st = torch.zeros(2, device=DEV, dtype=torch.float64)
    on = torch.ones(n, device=DEV)
    grid_ops.scatter_stats(grid, X, on, on, on, on, b1, None, st, model._err)
    assert torch.allclose(A @ ones, b1, rtol=1e-4, atol=1e-4)
    assert abs(float(st[0]) - n) < 1e-6
    # posterior parity vs the exact data-space GP on the same 6k points (rtol 1e-2, fp32)
    Xs = X[:64]
    with settings.cg_tolerance(1e-6):
        mvn = model(Xs)
        mean, var = mvn.mean.double().cpu().numpy(), mvn.variance.double().cpu().numpy()
    s2 = float(model.likelihood.second_noise.detach())
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * 3, 50, sigma2=s2).fit(X.double().cpu().numpy(), y[:, 0].double().cpu().numpy(), np.ones(n))
    mo, vo = O.predict(Xs.double().cpu().numpy())
    dm, dv = np.abs(mean - mo).max() / np.abs(mo).max(), np.abs(var - vo).max() / np.abs(vo).max()
    print(f"MEASURED 50^3 fp32, 6k points vs the data-space GP: mean {dm:.2e} variance {dv:.2e}")
    assert dm <= 1e-2 and dv <= 1e-2          # the north-star's fp32 bar
    assert dm <= TIGHT_MEAN_6K and dv <= TIGHT_VAR_6K   # ~3x what is measured (fp32 statistics, 1e-6 solves): a regression inside the bar is still caught


@pytest.mark.parametrize("kind", ["uniform", "clustered"])
def test_c3_full_stream_checkpoints_vs_cpu_port(kind):
    """BASELINE metric config at FULL size: the whole 3droad-sized synthetic stream (434 874 points, d = 3, 50^3 grid, fp32,
    q = 4096 per step through the one-call streaming step with the deferred poll) against the matrix-free CPU port of the same
    algorithm in fp64 (oracle/baseline.py, OpenMP), predictive means at 4096 fixed test points at three checkpoints: after
    the 5 % init, after 25 % and after 100 % of the stream ("reported numbers" (3)).  Bar: the north-star's fp32
    rtol of 1e-2 on the largest mean; the measured deviation is printed."""
    import bench
    from oracle import baseline
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    N, n0, q = 434874, 21743, 4096
    Xc, yc = bench.synth_stream(N, 3, 0, torch.device("cpu"), torch.float64, kind)     # "clustered": the road-like variant of SURVEY 8d
    Xt, _ = bench.synth_stream(4096, 3, 99, torch.device("cpu"), torch.float64, kind)
    Xg, yg = Xc.to(DEV, torch.float32), yc.to(DEV, torch.float32)
    Xtg = Xt.to(DEV, torch.float32)
    marks = [n0, n0 + ((N // 4 - n0) // q) * q, N]
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    got = []
    with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-5), settings.deferred_bounds_check(True), settings.deferred_refresh(True), \
            torch.no_grad():
        model = FixedNoiseOnlineSKIGP(Xg[:n0], yg[:n0], torch.ones_like(yg[:n0]), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
        model.prediction_cache
        done = n0
        for mark in marks:
            while done < mark:
                hi = min(done + q, mark)
                model.stream_step(Xg[done:hi], yg[done:hi], want_mean=False)
                done = hi
            model._finish_pending()
            got.append(model(Xtg).mean.double().cpu().numpy())
        s2 = float(model._sigma2(0))
    assert model.num_data == N
    B = baseline.StreamingBaseline([[-1.1, 1.1]] * 3, 50, sigma2=s2, dtype=np.float64)
    Xn, yn = Xc.numpy(), yc.numpy()[:, 0]
    done = 0
    for mark, g in zip(marks, got):
        B.absorb(Xn[done:mark], yn[done:mark])
        done = mark
        B.refresh(1e-9)
        want = B.predict_mean(Xt.numpy()).astype(np.float64)
        dev = np.abs(g - want).max() / np.abs(want).max()
        print(f"{kind} checkpoint {mark:6d} points: max |mean - cpu| / max |cpu| = {dev:.2e}")
        assert dev <= 1e-2
        assert dev <= 2e-4        # what fp32 statistics + a 1e-5 solve actually give (about 1e-5)
    # predictive VARIANCE at the end of the full stream (64 of the fixed test points) against the fp64 port: the product's default
    # path (the spectral Woodbury factor on this grid and kernel) and the PCG path of the same model
    want_v = B.variance(Xt.numpy()[:64])
    with settings.cg_tolerance(1e-5), torch.no_grad():
        v_def = model(Xtg[:64]).variance.double().cpu().numpy()
        with settings.spectral_factor(False):
            v_pcg = model(Xtg[:64]).variance.double().cpu().numpy()
    for name, v in (("default path", v_def), ("pcg path", v_pcg)):
        dv = np.max(np.abs(v - want_v) / want_v)
        print(f"MEASURED {kind} end of stream, 64 variances, {name}: max rel dev vs the fp64 port = {dv:.2e}")
        assert dv <= 1e-2
        assert dv <= TIGHT_VAR_C3[name]      # ~3x what is measured
    if kind == "clustered":
        # the road-like stream keeps a two-level block, and the 64-column variance solve above went through it (multi-column form:
        # k_tl_coef_mc / k_tl_apply_mc / k_spec_slab_mfma_mc<.., TL>): 4-5 iterations where the separable model alone takes 15
        post = model.prediction_cache["pred_cov"]
        assert post.last_two_level is not None and post.last_iters <= 8, post.last_iters


def test_c2_scale_30pow4_fp64_parity():
    """BASELINE configs[1]: d=4, 30^4 grid (m=810000), fp64, rtol 1e-4 vs the CPU oracle."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(1)
    n = 1500
    X = rng.uniform(-1, 1, (n, 4)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.5 * X[:, 2] * X[:, 3] + 0.1 * rng.standard_normal(n)
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    model = FixedNoiseOnlineSKIGP(Xt[:1000], yt[:1000], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 4), grid_size=30, learn_additional_noise=True)
    model.eval()
    model.condition_on_observations(Xt[1000:], yt[1000:], inplace=True)
    s2 = float(model.likelihood.second_noise.detach())
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * 4, 30, sigma2=s2).fit(X, y, np.ones(n))
    Xs = X[:16]
    mo, vo = O.predict(Xs)
    with settings.variance_chunk(16):
        mvn = model(Xt[:16])
        mean, var = mvn.mean.cpu().numpy(), mvn.variance.cpu().numpy()
    assert np.abs(mean - mo).max() <= 1e-4 * np.abs(mo).max()
    assert np.abs(var - vo).max() <= 1e-4 * np.abs(vo).max()


def test_dirichlet_classifier_wrapper_banana_like():
    """OnlineSKIClassifier (2-output batch, heteroscedastic Dirichlet noise): batch fit + streaming updates reach the
    reference's accuracy thresholds (tests/classification/test_ski_classifier.py:33,59: >= 0.85 batch, >= 0.75 online)
    on a synthetic two-moons-like problem (the Banana data needs the network)."""
    from online_gp_amd.models import Identity, OnlineSKIClassifier

    rng = np.random.default_rng(0)
    n = 600
    lab = rng.integers(0, 2, n)
    ang = rng.uniform(0, np.pi, n)
    X = np.stack([np.cos(ang) * (1 - 2 * lab) * 0.5 + 0.25 * (2 * lab - 1), np.sin(ang) * (1 - 2 * lab) * 0.5 + 0.15 * (2 * lab - 1)], 1)
    X += 0.08 * rng.standard_normal(X.shape)
    X = np.clip(X, -0.95, 0.95)
    Xt = torch.as_tensor(X, device=DEV, dtype=torch.float32); yt = torch.as_tensor(lab, device=DEV)
    clf = OnlineSKIClassifier(Identity(2), Xt[:200], yt[:200], 1e-2, 1e-2, 16, 1.0)
    recs = clf.fit(Xt[:200], yt[:200], 3)
    assert len(recs) == 3 and np.isfinite(recs[-1]["train_loss"])
    acc_batch = clf.predict(Xt[400:]).eq(yt[400:]).float().mean().item()
    assert acc_batch >= 0.85
    correct = 0
    for s in range(200, 400, 20):
        correct += clf.predict(Xt[s:s + 20]).eq(yt[s:s + 20]).sum().item()
        clf.update(Xt[s:s + 20], yt[s:s + 20], update_gp=(s % 60 == 0))
    assert correct / 200 >= 0.75
    assert clf.gp.num_data == 400 and clf.gp.num_outputs == 2
    assert clf.predict(Xt[400:]).eq(yt[400:]).float().mean().item() >= 0.85
    # the two regression outputs behind the argmax, against the data-space oracle on the Dirichlet-transformed problem
    # (gp_dirichlet_classification.py:15-21): per-output targets log(alpha) - sigma2_i / 2 with fixed noise sigma2_i, no learned noise
    from online_gp_amd.models.online_ski_classifier import dirichlet_transform

    ty, _, s2i = dirichlet_transform(yt[:400], 1e-2)
    k = clf.gp.covar_module.base_kernel
    means = clf.gp(Xt[400:420]).mean.double().cpu().numpy()                       # [2, 20]
    for o in range(2):
        ell = k.base_kernel.lengthscale.detach().double()[o].cpu().numpy().reshape(-1) if k.base_kernel.lengthscale.dim() > 2 else \
            k.base_kernel.lengthscale.detach().double().cpu().numpy().reshape(-1)
        osc = float(k.outputscale.detach().double().reshape(-1)[o if k.outputscale.numel() > 1 else 0])
        O = dataspace.DataSpaceGP([[-1.0, 1.0]] * 2, 16, "rbf", ell, osc, 1.0).fit(X[:400], ty[:, o].double().cpu().numpy(), s2i[:, o].double().cpu().numpy())
        mo, _ = O.predict(X[400:420])
        dmo = np.abs(means[o] - mo).max() / np.abs(mo).max()
        print(f"MEASURED classifier output {o}: mean dev {dmo:.2e}")
        assert dmo <= 1e-2 and dmo <= TIGHT_CLF_MEAN


def test_c2_full_stream_30pow4_fp64_parity():
    """BASELINE configs[1] as stated, at full size: powerplant-like stream N = 9568, d = 4, 30^4 grid (m = 810 000), fp64,
    streamed in 1024-point batches; predictive mean AND variance against the data-space oracle (n x n Cholesky on the host) at
    rtol 1e-4 (was tools/check_c2_full.py)."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(1)
    N, d, g = 9568, 4, 30
    X = rng.uniform(-1, 1, (N, d)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.5 * X[:, 2] * X[:, 3] + 0.1 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    n0 = N // 20
    model = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64), grid_size=g,
                                  learn_additional_noise=True)
    model.eval()
    for s in range(n0, N, 1024):
        model.condition_on_observations(Xt[s:s + 1024], yt[s:s + 1024], inplace=True)
    assert model.num_data == N
    with settings.variance_chunk(32):
        mvn = model(Xt[:32])
        mean, var = mvn.mean.cpu().numpy(), mvn.variance.cpu().numpy()
    s2 = float(model.likelihood.second_noise.detach())
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * d, g, sigma2=s2).fit(X, y, np.ones(N))
    mo, vo = O.predict(X[:32])
    assert np.abs(mean - mo).max() <= 1e-4 * np.abs(mo).max()
    assert np.abs(var - vo).max() <= 1e-4 * np.abs(vo).max()
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-4)])
def test_residual_carry_over_tracks_true_residual(dtype, tol):
    """Streaming refreshes that start from the scatter-maintained residual (warm = 2) stay on the true residual
    b - z - A u and give the same posterior as refreshes that recompute it (and as the data-space oracle)."""
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    torch.manual_seed(1)
    d, g, n0, q, steps = 3, 12, 300, 64, 20      # 20 > 16: crosses one from-scratch recomputation
    X = torch.rand(n0 + q * steps, d, device=DEV, dtype=dtype) * 2 - 1
    y = (torch.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2] + 0.1 * torch.randn(X.shape[0], device=DEV, dtype=dtype))[:, None]
    noise = torch.rand_like(y) + 0.5

    def run(carry):
        with settings.residual_carry_over(carry), settings.cg_tolerance(1e-10 if dtype == torch.float64 else 1e-6), \
                settings.dense_small_grids(False), torch.no_grad():
            model = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], noise[:n0], grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=g,
                                          learn_additional_noise=True).eval()
            used = 0
            for s in range(steps):
                sl = slice(n0 + s * q, n0 + (s + 1) * q)
                model.prediction_cache                                   # refresh (warm from the 2nd step on)
                used += int(bool(model._mean_state["R_ok"]))
                model.condition_on_observations(X[sl], y[sl], noise[sl], inplace=True)
                assert model._mean_state["R_ok"] == carry
            pc = model.prediction_cache
            ms = model._mean_state
            c = model._kernel_cache
            true_r = c["interpolation_cache"][0, :, 0] - ms["Z"][0] - grid_ops.stencil_spmv(model._grid, c["WtW"].stencil, ms["U"][0:1])[0]
            return model, pc["pred_mean"][0, :, 0].clone(), ms["R"][0].clone(), true_r

    m1, mean1, r1, true1 = run(True)
    m0, mean0, _, _ = run(False)
    scale = float(m1._kernel_cache["interpolation_cache"].abs().max())
    assert float((r1 - true1).abs().max()) < tol * scale * 50          # the carried residual is the true residual
    assert float((mean1 - mean0).abs().max()) < tol * 50 * float(mean0.abs().max())
    Xs = X[:32]
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * d, g, sigma2=float(m1.likelihood.second_noise.detach())).fit(
        X.double().cpu().numpy(), y[:, 0].double().cpu().numpy(), noise[:, 0].double().cpu().numpy())
    mo, _ = O.predict(Xs.double().cpu().numpy())
    with settings.skip_posterior_variances(True), settings.dense_small_grids(False), torch.no_grad():
        mh = m1(Xs).mean.double().cpu().numpy()
    assert np.abs(mh - mo).max() <= RTOL[dtype] * np.abs(mo).max()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_empty_batches_and_colliding_points(dtype):
    """Edge cases of the streaming update: an empty batch changes nothing; a batch of identical points (every atomic of
    the scatter collides) and a batch on the grid-box corners (one-hot boundary cells) match the data-space oracle."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    torch.manual_seed(3)
    d, g = 3, 12
    gb = [[-1.1, 1.1]] * d
    X0 = torch.rand(200, d, device=DEV, dtype=dtype) * 2 - 1
    f = lambda X: (torch.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2])[:, None]
    y0 = f(X0) + 0.1 * torch.randn(200, 1, device=DEV, dtype=dtype)
    with settings.dense_small_grids(False), settings.cg_tolerance(1e-10 if dtype == torch.float64 else 1e-6), torch.no_grad():
        model = FixedNoiseOnlineSKIGP(X0, y0, None, grid_bounds=torch.tensor(gb), grid_size=g, learn_additional_noise=True).eval()
        model.prediction_cache
        before = {k: v.clone() for k, v in (("b", model._kernel_cache["interpolation_cache"]), ("A", model._kernel_cache["WtW"].stencil),
                                            ("s", model._kernel_cache["_stats"]))}
        model.condition_on_observations(X0[:0], y0[:0], inplace=True)                 # empty batch
        assert model.num_data == 200
        assert torch.equal(before["b"], model._kernel_cache["interpolation_cache"]) and torch.equal(before["A"], model._kernel_cache["WtW"].stencil)
        assert torch.equal(before["s"], model._kernel_cache["_stats"])
        xd = torch.tensor([[0.3, -0.2, 0.55]], device=DEV, dtype=dtype).repeat(777, 1)   # 777 copies of one point
        yd = f(xd) + 0.1 * torch.randn(777, 1, device=DEV, dtype=dtype)
        corners = torch.tensor([[sx * 1.1, sy * 1.1, sz * 1.1] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], device=DEV, dtype=dtype)
        yc = f(corners)
        model.condition_on_observations(xd, yd, inplace=True)
        model.prediction_cache
        model.condition_on_observations(corners, yc, inplace=True)
        Xs = torch.cat([X0[:24], xd[:1], corners[:2]])
        mvn = model(Xs)
        mean, var = mvn.mean.double().cpu().numpy(), mvn.variance.double().cpu().numpy()
    Xall = torch.cat([X0, xd, corners]).double().cpu().numpy()
    yall = torch.cat([y0, yd, yc])[:, 0].double().cpu().numpy()
    O = dataspace.DataSpaceGP(gb, g, sigma2=float(model.likelihood.second_noise.detach())).fit(Xall, yall, np.ones(Xall.shape[0]))
    mo, vo = O.predict(Xs.double().cpu().numpy())
    assert model.num_data == 200 + 777 + 8
    assert np.abs(mean - mo).max() <= RTOL[dtype] * np.abs(mo).max()
    assert np.abs(var - vo).max() <= RTOL[dtype] * np.abs(vo).max()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-8)])
def test_stream_step_one_call_equals_the_three_python_calls(dtype, tol):
    """`stream_step` (one C-ABI call: wiski_stream_step = gather + scatter with residual carry + warm PCG) against
    evaluate -> condition_on_observations(inplace) -> prediction_cache, over a stream long enough to hit the periodic residual
    recomputation, and against the data-space oracle at the end; falls back to the generic path where the fast path does not
    apply (dense grids, out-of-date preconditioner profile) and reports out-of-grid points from inside the call."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(12)
    d, g, n0, q, steps = 3, 20, 400, 64, 40
    X = rng.uniform(-1, 1, (n0 + q * steps, d)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2] + 0.1 * rng.standard_normal(X.shape[0])
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=dtype), torch.as_tensor(y, device=DEV, dtype=dtype)[:, None]
    gb = torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64)
    with settings.cg_tolerance(1e-5 if dtype == torch.float32 else 1e-11), torch.no_grad():
        a = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        b = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        a.prediction_cache; b.prediction_cache
        fast = 0
        for s in range(steps):
            sl = slice(n0 + s * q, n0 + (s + 1) * q)
            fast += a._stream_fast_state(Xt[sl], yt[sl]) is not None
            ma = a.stream_step(Xt[sl], yt[sl])
            with settings.skip_posterior_variances(True):
                mb = b(Xt[sl]).mean
            b.condition_on_observations(Xt[sl], yt[sl], inplace=True)
            b.prediction_cache
            assert float((ma - mb).abs().max()) <= 10 * tol * float(mb.abs().max())
        assert fast >= steps - 6                      # the one-call path carried (nearly) every step
        assert a.num_data == b.num_data == n0 + q * steps and abs(a._wsum[0] - b._wsum[0]) < 1e-9
        sa, sb = a._kernel_cache["WtW"].stencil, b._kernel_cache["WtW"].stencil
        assert float((sa - sb).abs().max()) <= tol * float(sb.abs().max())
        assert torch.allclose(a._kernel_cache["_stats"], b._kernel_cache["_stats"], rtol=1e-9)
        mv = a(Xt[:32])
        s2 = float(a.likelihood.second_noise.detach())
        O = dataspace.DataSpaceGP([[-1.1, 1.1]] * d, g, sigma2=s2).fit(X, y, np.ones(X.shape[0]))
        mo, vo = O.predict(X[:32])
        rt = RTOL[dtype]
        assert np.abs(mv.mean.double().cpu().numpy() - mo).max() <= rt * np.abs(mo).max()
        assert np.abs(mv.variance.double().cpu().numpy() - vo).max() <= rt * vo.max()
        # an out-of-grid point inside the batch is reported by the same call; the model stays consistent
        bad = Xt[:q].clone(); bad[5, 1] = 3.0
        with pytest.raises(RuntimeError, match="out of bounds"):
            a.stream_step(bad, yt[:q])
        assert a.num_data == n0 + q * steps + q - 1
        # dense regime: plain fallback
        c = FixedNoiseOnlineSKIGP(Xt[:50, :2], yt[:50], None, grid_bounds=gb[:2], grid_size=8, learn_additional_noise=True).eval()
        mc = c.stream_step(Xt[50:60, :2].contiguous(), yt[50:60])
        assert mc.shape == (10,) and c.num_data == 60


def test_deferred_refresh_gives_the_same_stream_and_reports_one_call_later():
    """settings.deferred_refresh: stream_step leaves the first convergence poll in flight (wiski_stream_step with a
    wiski_pcg_async handle); the next consumer resumes it.  Same posterior as the synchronous stream, `prediction_cache` /
    `posterior` finish a pending solve, an out-of-grid batch is reported by the following call and leaves the model
    consistent."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(13)
    d, g, n0, q, steps = 3, 20, 400, 128, 24
    X = rng.uniform(-1, 1, (n0 + q * steps, d)); y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1]) + 0.3 * X[:, 2] + 0.1 * rng.standard_normal(X.shape[0])
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=torch.float32), torch.as_tensor(y, device=DEV, dtype=torch.float32)[:, None]
    gb = torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64)
    with settings.cg_tolerance(1e-5), torch.no_grad():
        a = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        b = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        a.prediction_cache; b.prediction_cache
        pend = 0
        for s in range(steps):
            sl = slice(n0 + s * q, n0 + (s + 1) * q)
            with settings.deferred_refresh(True):
                ma = a.stream_step(Xt[sl], yt[sl])
            pend += a.__dict__.get("_pending_step") is not None
            mb = b.stream_step(Xt[sl], yt[sl])
            assert float((ma - mb).abs().max()) <= 2e-3 * float(mb.abs().max())
            if s % 7 == 3:
                va = a(Xt[:8]).mean                      # a consumer in between: finishes the pending solve first
                assert a.__dict__.get("_pending_step") is None
                assert float((va - b(Xt[:8]).mean).abs().max()) <= 2e-3 * float(va.abs().max())
        assert pend >= steps - 4
        pa, pb = a.prediction_cache["pred_mean"], b.prediction_cache["pred_mean"]
        assert float((pa - pb).abs().max()) <= 2e-3 * float(pb.abs().max())
        assert a.num_data == b.num_data and torch.allclose(a._kernel_cache["_stats"], b._kernel_cache["_stats"], rtol=1e-9)
        # out-of-grid batch: the error arrives with the next call, which is then not absorbed
        bad = Xt[:q].clone(); bad[7, 0] = -5.0
        n_before = a.num_data
        with settings.deferred_refresh(True):
            a.stream_step(bad, yt[:q])
            with pytest.raises(RuntimeError, match="out of bounds"):
                a.stream_step(Xt[q:2 * q], yt[q:2 * q])
        assert a.num_data == n_before + q - 1
        assert torch.isfinite(a(Xt[:8]).mean).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.float64, 1e-7)])
def test_deferred_stream_of_random_batch_sizes_matches_a_rebuilt_model(dtype, tol):
    """60 deferred streaming steps of random size 1..3000 (speculative absorb ahead of the pending poll, guard hits and
    misses, poll placement with probes, periodic residual recomputation) against the batch means of a synchronous twin and,
    at the end, a model built from scratch on all the data: same statistics, same posterior mean."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(29)
    d, g, n0 = 3, 24, 600
    sizes = [int(rng.choice([1, 2, 5, 17, 64, 300, 1000, 3000])) for _ in range(60)]
    n = n0 + sum(sizes)
    X = rng.uniform(-1, 1, (n, d)); y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + 0.5 * X[:, 2] ** 2 + 0.1 * rng.standard_normal(n)
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=dtype), torch.as_tensor(y, device=DEV, dtype=dtype)[:, None]
    gb = torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64)
    cg = 1e-5 if dtype == torch.float32 else 1e-10
    with settings.cg_tolerance(cg), settings.skip_posterior_variances(True), torch.no_grad():
        a = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        b = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        a.prediction_cache; b.prediction_cache
        lo, fast = n0, 0
        for q in sizes:
            sl = slice(lo, lo + q)
            with settings.deferred_refresh(True):
                ma = a.stream_step(Xt[sl], yt[sl])
            fast += a.__dict__.get("_pending_step") is not None
            mb = b(Xt[sl]).mean
            b.condition_on_observations(Xt[sl], yt[sl], None, inplace=True)
            b.prediction_cache
            assert float((ma.reshape(-1) - mb.reshape(-1)).abs().max()) <= tol * max(float(mb.abs().max()), 1.0), q
            lo += q
        assert fast >= 50                                 # the one-call deferred path really ran
        c = FixedNoiseOnlineSKIGP(Xt, yt, None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        pa, pc = a.prediction_cache["pred_mean"], c.prediction_cache["pred_mean"]
        assert a.num_data == c.num_data == n
        assert float((pa - pc).abs().max()) <= tol * float(pc.abs().max())
        sa, sc = a._kernel_cache["_stats"], c._kernel_cache["_stats"]
        assert torch.allclose(sa, sc, rtol=1e-5 if dtype == torch.float32 else 1e-11)


def test_c1_notebook_setup_with_the_spectral_mixture_kernel():
    """BASELINE config 1 as the reference's notebook builds it (notebooks/regression_viz_1D.ipynb cell 22:
    ``OnlineSKIRegression(stem, init_x, init_y, lr=1e-1, grid_size=12, grid_bound=1, covar_module=SpectralMixtureKernel(3))``):
    streamed predictions equal the data-space oracle on the same spectral-mixture Toeplitz column, and fit() moves the
    mixture parameters downhill."""
    from online_gp_amd.kernels import SpectralMixtureKernel
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(0)
    x = np.linspace(-1, 1, 39); y = np.sin(4 * x) + 0.4 * rng.standard_normal(39)
    perm = rng.permutation(39)
    x, y = x[perm], y[perm]
    Xt = torch.as_tensor(x, device=DEV)[:, None]; yt = torch.as_tensor(y, device=DEV)[:, None]
    sm = SpectralMixtureKernel(num_mixtures=3)
    sm.mixture_weights = [0.6, 0.3, 0.1]
    sm.mixture_means = torch.tensor([[[0.3]], [[0.7]], [[1.4]]])
    sm.mixture_scales = torch.tensor([[[0.2]], [[0.4]], [[0.8]]])
    reg = OnlineSKIRegression(Identity(1), Xt[:10], yt[:10], 1e-1, 12, 1, covar_module=sm)
    for t in range(10, 39):
        reg.update(Xt[t:t + 1], yt[t:t + 1], update_stem=False, update_gp=False)     # hyper-parameters held: pure streaming
    s2 = float(reg.gp.likelihood.second_noise.detach())
    g0, h, g = spec.make_grid([[-1.1, 1.1]], 12)
    w = sm.mixture_weights.detach().double().cpu().numpy(); mu = sm.mixture_means.detach().double().cpu().numpy().reshape(3, 1)
    sc = sm.mixture_scales.detach().double().cpu().numpy().reshape(3, 1)
    O = dataspace.DataSpaceGP([[-1.1, 1.1]], 12, sigma2=s2, cols=spec.spectral_mixture_columns(h, g, w, mu, sc)).fit(x[:, None], y, np.ones(39))
    xs = np.linspace(-0.95, 0.95, 17)
    mo, vo = O.predict(xs[:, None])
    mean, var = reg.predict(torch.as_tensor(xs, device=DEV)[:, None])
    assert np.abs(mean[:, 0].detach().cpu().numpy() - mo).max() < 1e-6 * np.abs(mo).max()
    assert np.abs(var[:, 0].detach().cpu().numpy() - (vo + s2)).max() < 1e-6 * (vo + s2).max()
    # the notebook's pre-training: the MLL gradient reaches the mixture parameters
    before = [p.detach().clone() for p in sm.parameters()]
    recs = reg.fit(Xt[:10], yt[:10], 15)
    assert recs[-1]["train_loss"] < recs[0]["train_loss"]
    assert any((p.detach() - b).abs().max() > 1e-3 for p, b in zip(sm.parameters(), before))
    # updates with hyper-parameter steps run (the notebook's online loop)
    reg.set_lr(1e-2)
    for t in range(10, 14):
        reg.update(Xt[t:t + 1], yt[t:t + 1])
    m2, v2 = reg.predict(torch.as_tensor(xs, device=DEV)[:, None])
    assert torch.isfinite(m2).all() and (v2 > 0).all()


def test_hyperparameter_caches_follow_a_fused_optimizer():
    """torch.optim.Adam(fused=True) does not advance the parameters' autograd version counters; the model's caches key on
    zero_grad() calls as well (the reference calls gp.zero_grad() after every step) and on hyperparameters_changed()."""
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(2)
    X = torch.as_tensor(rng.uniform(-1, 1, (120, 2)), device=DEV); y = torch.sin(3 * X[:, :1])
    m = FixedNoiseOnlineSKIGP(X, y, None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 2), grid_size=12, learn_additional_noise=True)
    m.eval()
    Xs = X[:7]
    v0 = m(Xs).variance.clone()
    ls = m.covar_module.base_kernel.base_kernel.raw_lengthscale
    with torch.no_grad():
        ls.data.mul_(1.0).add_(0.5)                     # a write the version counter does not see
    m.hyperparameters_changed()
    v1 = m(Xs).variance.clone()
    assert (v1 - v0).abs().max() > 1e-6 * v0.abs().max()
    opt = torch.optim.Adam([ls], lr=0.3, fused=True)
    ver = ls._version
    ls.grad = torch.ones_like(ls)
    opt.step()
    assert ls._version == ver                           # the premise: a fused step leaves the counter alone
    m.zero_grad()
    v2 = m(Xs).variance.clone()
    assert (v2 - v1).abs().max() > 1e-6 * v1.abs().max()
    O = dataspace.DataSpaceGP([[-1.1, 1.1]] * 2, 12, "rbf", m.covar_module.base_kernel.base_kernel.lengthscale.detach().cpu().numpy().reshape(-1),
                              float(m.covar_module.base_kernel.outputscale.detach()), float(m.likelihood.second_noise.detach())).fit(
        X.cpu().numpy(), y[:, 0].cpu().numpy(), np.ones(120))
    _, vo = O.predict(Xs.cpu().numpy())
    assert np.abs(v2.cpu().numpy() - vo).max() < 1e-6 * vo.max()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("d,g", [(2, 12), (3, 14)])
def test_multi_output_absorb_in_one_launch_equals_the_per_output_loop(dtype, d, g):
    """num_outputs as a kernel batch dimension (BFN:37-55): the statistics of all outputs are absorbed by ONE launch
    (wiski_scatter_stats_multi, packed half stencils) -- same caches and posterior as the per-output launches of a model whose
    stencils were un-packed, for heteroscedastic per-output noise, unit noise, and with the carried residual."""
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.lazy.operators import StencilWtW
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(d + g)
    n0, q, out = 200, 64, 3
    X = torch.as_tensor(rng.uniform(-1, 1, (n0 + 3 * q, d)), device=DEV, dtype=dtype)
    Y = torch.as_tensor(rng.standard_normal((n0 + 3 * q, out)), device=DEV, dtype=dtype) + torch.sin(2 * X[:, :1])
    NZ = torch.as_tensor(rng.uniform(0.4, 1.6, (n0 + 3 * q, out)), device=DEV, dtype=dtype)
    gb = torch.tensor([[-1.1, 1.1]] * d)
    calls = {"multi": 0}
    orig = grid_ops.scatter_stats_multi

    def counting(*a, **k):
        calls["multi"] += 1
        return orig(*a, **k)

    grid_ops.scatter_stats_multi = counting
    try:
        with settings.cg_tolerance(1e-10 if dtype == torch.float64 else 1e-6), settings.spectral_factor(False):
            a = FixedNoiseOnlineSKIGP(X[:n0], Y[:n0], NZ[:n0], grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
            assert calls["multi"] == 1 and a._stencil_pack(a._kernel_cache["WtW"].ops) is not None
            b = FixedNoiseOnlineSKIGP(X[:n0], Y[:n0], NZ[:n0], grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
            for op in b._kernel_cache["WtW"].ops:                    # un-pack: every stencil its own tensor -> per-output launches
                op.stencil = op.stencil.clone()
            assert b._stencil_pack(b._kernel_cache["WtW"].ops) is None
            a.prediction_cache; b.prediction_cache                   # posterior-mean state: the next absorbs carry the residual
            before = calls["multi"]
            for i in range(3):
                sl = slice(n0 + i * q, n0 + (i + 1) * q)
                nz = NZ[sl] if i < 2 else None                       # the last batch with unit noise
                a.condition_on_observations(X[sl], Y[sl], nz, inplace=True)
                b.condition_on_observations(X[sl], Y[sl], nz, inplace=True)
                a.prediction_cache; b.prediction_cache
            assert calls["multi"] == before + 3
            ca, cb = a._kernel_cache, b._kernel_cache
            tol = 1e-12 if dtype == torch.float64 else 2e-5
            for o in range(out):
                sa, sb = ca["WtW"].ops[o].stencil, cb["WtW"].ops[o].stencil
                assert (sa - sb).abs().max() <= tol * sb.abs().max()
            for key in ("interpolation_cache", "_cnt", "_stats"):
                assert (ca[key] - cb[key]).abs().max() <= tol * cb[key].abs().max()
            if a._mean_state is not None:                            # matrix-free regime (14^3): the carried residuals agree too
                assert (a._mean_state["R"] - b._mean_state["R"]).abs().max() <= 50 * tol * max(1.0, float(b._mean_state["R"].abs().max()))
            else:
                assert d == 2                                        # 12^2 is the dense regime: no solver state
            ma, mb = a(X[:9]), b(X[:9])
            assert torch.allclose(ma.mean, mb.mean, rtol=1e-4, atol=1e-6) and torch.allclose(ma.variance, mb.variance, rtol=1e-4, atol=1e-8)
    finally:
        grid_ops.scatter_stats_multi = orig


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-9)])
def test_graphed_hyper_step_equals_the_eager_step(dtype, tol):
    """The per-batch Adam step on the MLL recorded into one HIP graph (models/_graphed_step.py) walks the same hyper-parameter
    trajectory as the op-by-op step: same losses, same parameters, same predictions, step after step -- across the eager warm-up,
    the capture, a re-capture forced by a re-selection of the factor's index set, and a learning-rate change."""
    from online_gp_amd import settings
    from online_gp_amd.lazy import spectral_woodbury as sw
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(5)
    d, n0, steps = 3, 600, 16
    X = rng.uniform(-1, 1, (n0 + steps * 4, d)); y = np.sin(2 * X[:, 0]) * X[:, 1] + 0.3 * X[:, 2] + 0.05 * rng.standard_normal(len(X))
    Xt = torch.as_tensor(X, device=DEV, dtype=dtype); yt = torch.as_tensor(y, device=DEV, dtype=dtype)[:, None]
    old_every = sw.RESELECT_EVERY
    sw.RESELECT_EVERY = 6                      # the factor re-selects its index set every 6 device refreshes: forces one re-capture
    try:
        runs = {}
        for graphed in (True, False):
            with settings.graphed_hyper_step(graphed):
                reg = OnlineSKIRegression(Identity(d), Xt[:n0], yt[:n0], 1e-2, 16, 1.0)
                trace = []
                for i in range(steps):
                    lo = n0 + 4 * i
                    if i == 11:
                        if graphed:
                            first = (reg._graphed.replays, reg._graphed.captures, reg._graphed.disabled)
                        reg.set_lr(5e-3)             # a new optimiser: a new recorder, which warms up again
                    rmse, nll = reg.evaluate(Xt[lo:lo + 4], yt[lo:lo + 4])
                    _, loss = reg.update(Xt[lo:lo + 4], yt[lo:lo + 4])
                    k = reg.gp.covar_module.base_kernel
                    trace.append((rmse, nll, loss, k.base_kernel.lengthscale.detach().double().cpu().numpy().reshape(-1).copy(),
                                  float(k.outputscale.detach()), float(reg.gp.likelihood.second_noise.detach())))
                runs[graphed] = (trace, reg)
        gs = runs[True][1]._graphed
        assert gs.disabled is None, gs.disabled
        assert gs.fused and gs.fused_captures == gs.captures          # Scale(RBF) + homoskedastic noise + Adam: recorded without autograd
        assert first[2] is None and first[0] == 11 - 3 and first[1] >= 2, first     # 3 eager warm-up steps, then replays; >= 1 re-capture
        assert gs.replays == steps - 11 - 3 and gs.captures == 1
        assert runs[False][1]._graphed.replays == 0
        for i, (a, b) in enumerate(zip(runs[True][0], runs[False][0])):
            for u, v in zip(a[:3], b[:3]):
                assert abs(u - v) <= tol * 50 * max(1.0, abs(v)), (i, a[:3], b[:3])
            # (the raw parameters are fp32 tensors whatever the model dtype: the recording without autograd -- csrc/hyper_step.hip -- follows
            #  the framework's operation order, but a last-bit difference of an fp32 parameter now and then is not an error)
            ptol = max(tol * 10, 5e-7)
            assert np.abs(a[3] - b[3]).max() <= ptol and abs(a[4] - b[4]) <= ptol and abs(a[5] - b[5]) <= ptol, (i, a[3:], b[3:])
    finally:
        sw.RESELECT_EVERY = old_every


@pytest.mark.parametrize("kernel", ["matern_interval", "rbf_iso_two_scales"])
def test_fused_hyper_step_equals_the_autograd_recording(kernel):
    """The captured Adam step recorded WITHOUT autograd (csrc/hyper_step.hip: constraint transforms, MLL tail, chain rule, Adam) against
    the same step recorded through autograd + torch.optim.Adam, for the parameterisations the reference's drivers use besides the default:
    Matern-5/2 with ARD lengthscales under Interval constraints (experiments/bayesopt/bayesopt.py:72-76), and an isotropic RBF under two
    nested output scales."""
    from online_gp_amd import settings
    from online_gp_amd.constraints import Interval
    from online_gp_amd.kernels import MaternKernel, RBFKernel, ScaleKernel
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(11)
    d, n0, steps = 3, 600, 10
    X = rng.uniform(-1, 1, (n0 + steps * 4, d)); y = np.sin(2 * X[:, 0]) * X[:, 1] + 0.3 * X[:, 2] + 0.05 * rng.standard_normal(len(X))
    Xt = torch.as_tensor(X, device=DEV, dtype=torch.float64); yt = torch.as_tensor(y, device=DEV, dtype=torch.float64)[:, None]

    def make():
        if kernel == "matern_interval":
            return ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=d, lengthscale_constraint=Interval(1e-4, 12.0)), outputscale_constraint=Interval(1e-4, 12.0))
        return ScaleKernel(ScaleKernel(RBFKernel()))

    runs = {}
    for fused in (True, False):
        with settings.fused_hyper_step(fused):
            reg = OnlineSKIRegression(Identity(d), Xt[:n0], yt[:n0], 1e-2, 16, 1.0, covar_module=make())
            trace = []
            for i in range(steps):
                lo = n0 + 4 * i
                rmse, nll = reg.evaluate(Xt[lo:lo + 4], yt[lo:lo + 4])
                _, loss = reg.update(Xt[lo:lo + 4], yt[lo:lo + 4])
                trace.append([rmse, nll, loss] + [float(v) for p_ in reg.gp.parameters() for v in p_.detach().double().reshape(-1).cpu()])
            runs[fused] = (np.asarray(trace), reg._graphed)
    assert runs[True][1].disabled is None and runs[True][1].fused and runs[True][1].replays == steps - 3
    assert runs[False][1].disabled is None and not runs[False][1].fused and runs[False][1].replays == steps - 3
    # (metrics and loss are fp64 quantities; the raw parameters are fp32 tensors: a last-bit difference now and then)
    diff = np.abs(runs[True][0] - runs[False][0]).max(0)
    assert diff[:3].max() < 1e-7 and diff[3:].max() < 5e-7, diff


def test_fused_evaluate_equals_the_posterior_path():
    """evaluate() of <= 64 points from the spectral factor in one launch (wiski_spectral_evaluate) against the general path (posterior
    object, wiski_spectral_var, wiski_gaussian_metrics), step after step in the reference's evaluate -> update loop, several batch sizes."""
    from online_gp_amd import settings
    from online_gp_amd.models import Identity, OnlineSKIRegression

    rng = np.random.default_rng(3)
    d, n0 = 3, 800
    X = rng.uniform(-1, 1, (n0 + 400, d)); y = np.sin(2 * X[:, 0]) * X[:, 1] + 0.3 * X[:, 2] + 0.05 * rng.standard_normal(len(X))
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 1e-12)):
        Xt = torch.as_tensor(X, device=DEV, dtype=dtype); yt = torch.as_tensor(y, device=DEV, dtype=dtype)[:, None]
        with settings.spectral_tail(1e-6):             # (the one-launch kernel takes ranks <= 512: fp64's default tail of 1e-9 needs ~800 here)
            reg = OnlineSKIRegression(Identity(d), Xt[:n0], yt[:n0], 1e-3, 16, 1.0)
            lo, took_fast = n0, 0
            for qs in (1, 1, 1, 1, 1, 7, 64, 33, 130, 1, 100, 8):            # (<= 64: one launch; beyond: GEMM + one launch)
                xb, yb = Xt[lo:lo + qs], yt[lo:lo + qs]; lo += qs
                fast = reg._evaluate_from_factor(xb, yb.reshape(-1, 1))
                with settings.fused_evaluate(False):
                    ref = reg.evaluate(xb, yb)
                if fast is not None:
                    took_fast += 1
                    assert abs(fast[0] - ref[0]) <= tol * max(1.0, abs(ref[0])) and abs(fast[1] - ref[1]) <= tol * max(1.0, abs(ref[1])), (qs, fast, ref)
                reg.update(xb, yb)
            assert took_fast >= 10         # (before the first hyper step the PCG state is current: the general path serves that one)
            with pytest.raises(RuntimeError):
                reg.evaluate(torch.full((1, d), 5.0, device=DEV, dtype=dtype), yt[:1])      # outside the grid: raised from the fused path as well


def test_classifier_means_after_an_mll_step_come_from_the_factor_and_match_pcg():
    """predict -> update loops with an MLL step per batch (the Dirichlet classifier, two outputs with per-point noise): once the
    hyper-parameters have moved since the last mean solve and the spectral factors are being kept current for the MLL, a
    means-only request is answered from them -- same means as the PCG solve it replaces."""
    from online_gp_amd import settings
    from online_gp_amd.models import Identity, OnlineSKIClassifier

    rng = np.random.default_rng(9)
    d, n0 = 3, 900
    X = rng.uniform(-1, 1, (n0 + 80, d)); lab = (np.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2] > 0).astype(np.int64)
    Xt = torch.as_tensor(X, device=DEV, dtype=torch.float64); lt = torch.as_tensor(lab, device=DEV)
    with settings.cg_tolerance(1e-10):
        clf = OnlineSKIClassifier(Identity(d), Xt[:n0], lt[:n0], 0.01, 1e-2, 14, 1.1)
        for i in range(6):
            lo = n0 + 8 * i
            clf.predict(Xt[lo:lo + 8]); clf.update(Xt[lo:lo + 8], lt[lo:lo + 8])
        gp = clf.gp
        assert gp._spectral_in_use() and gp._memo.get("prediction_cache") is None
        Xq = Xt[n0 + 60:n0 + 80]
        clf.eval()
        with settings.skip_posterior_variances(True):
            m_fac = clf(Xq).mean
            assert gp._memo.get("prediction_cache") is None               # no PCG solve was made for it
            with settings.spectral_factor(False):
                m_pcg = clf(Xq).mean
        assert m_fac.shape == m_pcg.shape == (2, 20)
        assert (m_fac - m_pcg).abs().max().item() < 1e-5 * m_pcg.abs().max().item()


def test_graphed_hyper_step_with_two_outputs_equals_the_eager_step():
    """The Dirichlet classifier's per-batch Adam step (two outputs: two factors, per-point noise, no learnable sigma2) as one graph."""
    from online_gp_amd import settings
    from online_gp_amd.models import Identity, OnlineSKIClassifier

    rng = np.random.default_rng(11)
    d, n0, steps = 3, 900, 10
    X = rng.uniform(-1, 1, (n0 + 8 * steps, d)); lab = (np.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2] > 0).astype(np.int64)
    Xt = torch.as_tensor(X, device=DEV, dtype=torch.float64); lt = torch.as_tensor(lab, device=DEV)
    runs = {}
    for graphed in (True, False):
        with settings.graphed_hyper_step(graphed), settings.cg_tolerance(1e-10):
            clf = OnlineSKIClassifier(Identity(d), Xt[:n0], lt[:n0], 0.01, 1e-2, 14, 1.1)
            trace = []
            for i in range(steps):
                lo = n0 + 8 * i
                clf.predict(Xt[lo:lo + 8])
                _, loss = clf.update(Xt[lo:lo + 8], lt[lo:lo + 8])
                k = clf.gp.covar_module.base_kernel
                trace.append((loss, k.base_kernel.lengthscale.detach().cpu().numpy().reshape(-1).copy(), k.outputscale.detach().cpu().numpy().reshape(-1).copy()))
            clf.eval()
            with settings.skip_posterior_variances(True):
                runs[graphed] = (trace, clf(Xt[:40]).mean.cpu().numpy(), clf._graphed)
    gs = runs[True][2]
    assert gs.disabled is None and gs.captures >= 1 and gs.replays == steps - 3, (gs.disabled, gs.captures, gs.replays)
    assert runs[False][2].replays == 0
    for i, (a, b) in enumerate(zip(runs[True][0], runs[False][0])):
        assert abs(a[0] - b[0]) <= 1e-8 * max(1.0, abs(b[0])), (i, a[0], b[0])
        assert np.abs(a[1] - b[1]).max() <= 1e-9 and np.abs(a[2] - b[2]).max() <= 1e-9, (i, a, b)
    assert np.abs(runs[True][1] - runs[False][1]).max() <= 1e-7 * np.abs(runs[False][1]).max()


def test_reference_step_pipeline_at_the_bench_size_matches_the_plain_path():
    """BASELINE's configuration (d = 3, 50^3 grid, fp32, 21 743 initial points): 30 evaluate -> Adam-on-MLL -> condition steps with the
    device pipeline (graphed Adam step, eigenvectors refined on the device, fused kernel columns, factor means) against the same steps
    op by op with host-side eigen-decompositions -- per-step metrics, losses, the hyper-parameter trajectory and final predictions."""
    import bench
    from online_gp_amd import settings
    from online_gp_amd.models import Identity, OnlineSKIRegression

    dev = torch.device(DEV)
    X0, y0 = bench.synth_stream(21743, 3, 0, dev, torch.float32, "uniform")
    Xr, yr = bench.synth_stream(4096, 3, 31337, dev, torch.float32, "uniform")
    runs = {}
    for fast in (True, False):
        with settings.cg_tolerance(1e-4), settings.graphed_hyper_step(fast), settings.spectral_device_refresh(fast), settings.fused_hyper_columns(fast):
            reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
            tr = []
            for i in range(30):
                xb, yb = Xr[16 * i:16 * (i + 1)], yr[16 * i:16 * (i + 1)]
                rmse, nll = reg.evaluate(xb, yb)
                _, loss = reg.update(xb, yb)
                k = reg.gp.covar_module.base_kernel
                tr.append([rmse, nll, loss] + k.base_kernel.lengthscale.detach().reshape(-1).tolist() + [float(k.outputscale.detach()), float(reg.gp.likelihood.second_noise.detach())])
            m, v = reg.predict(Xr[2048:2112])
            runs[fast] = (np.array(tr), m.double().cpu().numpy(), v.double().cpu().numpy(), reg)
    gs, fac = runs[True][3]._graphed, runs[True][3].gp._spectral[0]
    assert gs.disabled is None and gs.replays == 27 and fac.device_refreshes >= 29 and fac.rebuilds == 1
    a, b = runs[True], runs[False]
    assert np.abs(a[0] - b[0]).max() < 2e-3 * max(1.0, np.abs(b[0]).max()), np.abs(a[0] - b[0]).max(0)
    assert np.abs(a[0][:, 3:] - b[0][:, 3:]).max() < 1e-5                       # the hyper-parameters themselves: fp32 rounding of a 1e-3 Adam step
    assert np.abs(a[1] - b[1]).max() < 1e-3 * np.abs(b[1]).max() and (np.abs(a[2] - b[2]) / b[2]).max() < 1e-2


def test_preconditioner_eigenbasis_is_resolved_for_every_hyperparameter_change():
    """The CG preconditioner's generalized eigenbasis also carries the u = Kt z image of the search directions inside the fused solver
    kernels, so it has to be re-solved for every hyper-parameter change, however small: after a 1 % lengthscale step the warm-started
    solve of the streaming model equals a cold solve of a model built at the new hyper-parameters (fp64, tolerance 1e-10)."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(4)
    d, g, n = 3, 16, 1500
    X = torch.as_tensor(rng.uniform(-1, 1, (n, d)), device=DEV); y = torch.sin(2 * X.sum(1, keepdim=True))
    Xq = torch.as_tensor(rng.uniform(-1, 1, (64, d)), device=DEV)

    def build():
        return FixedNoiseOnlineSKIGP(X, y, torch.ones_like(y), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=g, learn_additional_noise=True).eval()

    with settings.spectral_factor(False), settings.cg_tolerance(1e-10), settings.skip_posterior_variances(True):
        m = build()
        m(Xq).mean
        eig0 = m._memo["precond"][0]["eig"]
        k = m.covar_module.base_kernel.base_kernel
        with torch.no_grad():
            k.lengthscale = k.lengthscale * 1.01
        m.hyperparameters_changed()
        mean_warm = m(Xq).mean.clone()
        assert m._memo["precond"][0]["eig"] is not eig0
        m2 = build()
        with torch.no_grad():
            m2.covar_module.base_kernel.base_kernel.lengthscale = k.lengthscale.detach().clone()
        mean_cold = m2(Xq).mean
        assert (mean_warm - mean_cold).abs().max().item() < 1e-9 * mean_cold.abs().max().item()


@pytest.mark.parametrize("case", ["spectral", "dense", "pcg"])
def test_root_form_of_the_predictive_covariance_and_fast_pred_samples(case):
    """BFN:229-243 (`fast_pred_samples`): the reference hands out a RootLazyTensor from a Lanczos root of the inducing posterior.  Here
    a factor provides the root without a solve: the spectral Woodbury factor (large grid, smooth kernel: root + left-out prior
    variance as a diagonal term) and the dense factor (small grid: W* L_M) -- and the root's covariance IS the covariance the
    default path returns, which is checked against the data-space oracle; sampling goes through the root (no n x n factorisation).
    On the PCG path (Matern-1/2 on a large grid) no root exists: the flag changes nothing and sampling factorises the exact matrix."""
    from online_gp_amd import settings
    from online_gp_amd.distributions import RootLazyTensor
    from online_gp_amd.kernels import MaternKernel, ScaleKernel
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(5)
    d, g = (3, 14) if case != "dense" else (3, 8)
    n, ns = 600, 24
    X = rng.uniform(-1, 1, (n, d)); y = np.sin(2 * X.sum(1)) + 0.1 * rng.standard_normal(n)
    Xs = rng.uniform(-1, 1, (ns, d))
    gb = [[-1.1, 1.1]] * d
    cov_mod = ScaleKernel(MaternKernel(nu=0.5, ard_num_dims=d)).to(DEV) if case == "pcg" else None
    Xt, yt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None]
    with settings.cg_tolerance(1e-10):
        m = FixedNoiseOnlineSKIGP(Xt, yt, torch.ones_like(yt), grid_bounds=torch.tensor(gb), grid_size=g, learn_additional_noise=True, covar_module=cov_mod).eval()
        Xst = torch.as_tensor(Xs, device=DEV)
        exact = m(Xst)
        C = exact.covariance_matrix.double()
        with settings.fast_pred_samples(True):
            fast = m(Xst)
        lc = fast.lazy_covariance_matrix
        if case == "pcg":
            assert not isinstance(lc, RootLazyTensor) and exact.lazy_covariance_matrix.root_decomposition() is None
        else:
            assert isinstance(lc, RootLazyTensor) and lc.root.shape[0] == ns
            assert (lc.extra is not None) == (case == "spectral")
            scale = float(C.diagonal().max())
            assert float((lc.evaluate().double() - C).abs().max()) < 1e-7 * scale
            assert torch.allclose(fast.variance.double(), exact.variance.double(), rtol=1e-7, atol=1e-12 * scale)
            # ... and that covariance is the oracle's (kernel hyper-parameters at their defaults)
            k = m.covar_module.base_kernel
            O = dataspace.DataSpaceGP(gb, g, "rbf", k.base_kernel.lengthscale.detach().cpu().numpy().reshape(-1), float(k.outputscale),
                                      float(m.likelihood.second_noise)).fit(X, y, np.ones(n))
            _, Co = O.predict(Xs, full_cov=True)
            assert np.abs(C.cpu().numpy() - Co).max() < 1e-4 * np.abs(Co).max()
        # sampling: through the root where there is one (also with the flag off), Cholesky of the exact matrix otherwise
        torch.manual_seed(0)
        S = (fast if case != "pcg" else exact).rsample(torch.Size([40000])).double()
        assert S.shape == (40000, ns)
        emp = torch.cov(S.t())
        assert float((emp - C).abs().max()) < 0.05 * float(C.diagonal().max())
        assert float((S.mean(0) - exact.mean.double()).abs().max()) < 0.05 * float(C.diagonal().max().sqrt())
        if case != "pcg":
            torch.manual_seed(0)
            S2 = exact.rsample(torch.Size([8])).double()          # flag off: the same root route
            torch.manual_seed(0)
            S3 = fast.rsample(torch.Size([8])).double()
            assert torch.allclose(S2, S3, rtol=1e-9, atol=1e-9)
