"""GPU: dense MFMA building blocks against plain PyTorch references of the same ops
(asymmetric operands so a transposed C-write cannot pass)."""
import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (130, 70, 45), (257, 301, 129), (1000, 96, 1000), (1000, 1000, 1000), (1111, 900, 257), (1500, 1500, 100)])
def test_gemm_all_transposes(dtype, tol, ta, tb, M, N, K):
    """Small outputs and the dense regime's mid sizes (up to 500 / 300 tiles of 64 x 64 in fp64 / fp32) run on the 32 x 32-tile kernel,
    (1500, 1500, 100) on the 64 x 64-tile one (too many tiles for the small kernel, too little K for the 128 x 128 one)."""
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=dtype).to(DEV)
    B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=dtype).to(DEV)
    C0 = torch.randn((M, N), generator=g, dtype=dtype).to(DEV)
    ref = 0.7 * ((A.t() if ta else A).double() @ (B.t() if tb else B).double()) - 0.3 * C0.double()
    C = grid_ops.gemm(A, B, ta=ta, tb=tb, alpha=0.7, beta=-0.3, C=C0.clone())
    assert (C.double() - ref).abs().max().item() <= tol * K ** 0.5 * 10
    # identity check with an asymmetric B: catches a row/column swap in the C write
    if not ta and not tb and M == K:
        I = torch.eye(M, dtype=dtype, device=DEV)
        assert torch.equal(grid_ops.gemm(I, B), B)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K,pad", [(2048, 2048, 256, 0), (2100, 1937, 333, 0), (1929, 2201, 290, 3)])
def test_gemm_large_tile_kernel(dtype, tol, ta, tb, M, N, K, pad):
    """The 128 x 128 register-prefetch kernel (grids of >= 128 blocks): aligned interior tiles (vector loads), ragged edges in
    all three dimensions (scalar predicated loads), and leading dimensions that defeat the 16-byte loads (`pad`: operands are
    column slices of wider matrices)."""
    from online_gp_amd import _hip, grid_ops
    import ctypes

    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    Ash = (K, M) if ta else (M, K)
    Bsh = (N, K) if tb else (K, N)
    Aw = torch.randn(Ash[0], Ash[1] + pad, generator=g, dtype=dtype).to(DEV)
    Bw = torch.randn(Bsh[0], Bsh[1] + pad, generator=g, dtype=dtype).to(DEV)
    A, B = Aw[:, :Ash[1]], Bw[:, :Bsh[1]]
    C0 = torch.randn((M, N), generator=g, dtype=dtype).to(DEV)
    ref = 0.7 * ((A.t() if ta else A).double() @ (B.t() if tb else B).double()) - 0.3 * C0.double()
    C = C0.clone()
    cr = _hip.creal(dtype)
    rc = _hip.fn("wiski_gemm", dtype)(ctypes.c_int32(int(ta)), ctypes.c_int32(int(tb)), ctypes.c_int32(M), ctypes.c_int32(N), ctypes.c_int32(K), cr(0.7),
                                      ctypes.c_void_p(Aw.data_ptr()), ctypes.c_int32(Aw.shape[1]), ctypes.c_void_p(Bw.data_ptr()), ctypes.c_int32(Bw.shape[1]),
                                      cr(-0.3), _hip.dptr(C), ctypes.c_int32(N), _hip.stream_ptr(C.device))
    assert rc == 0
    assert (C.double() - ref).abs().max().item() <= tol * K ** 0.5 * 10


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("n", [17, 64, 200, 513, 897, 1000, 1345, 2048])
def test_cholesky_trsm_logdet(dtype, tol, n):
    """n > 512: the two-level blocked factorisation (dense.hip: potrf_two_level -- big diagonal blocks through the one-launch Cholesky +
    inverse, panel and trailing update as two GEMMs): 2 blocks with a short second one (513), 3 / 4 equal blocks, 5 blocks at the model's
    largest size."""
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(n)
    R = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64)).to(DEV, dtype)
    L = A.clone()
    info = grid_ops.potrf_(L)
    assert int(info.item()) == 0
    assert torch.equal(L, torch.tril(L))
    assert ((L @ L.t()).double() - A.double()).abs().max().item() < tol * 10
    Lref = torch.linalg.cholesky(A.double())
    assert (L.double() - Lref).abs().max().item() < tol * 10
    assert abs(float(grid_ops.chol_logdet(L)) - float(2 * Lref.diagonal().log().sum())) < tol * n
    B = torch.randn(n, 37, generator=g, dtype=torch.float64).to(DEV, dtype)
    X = grid_ops.trsm_(L, B.clone(), trans=False)
    assert ((L @ X).double() - B.double()).abs().max().item() < tol * 50
    Y = grid_ops.trsm_(L, B.clone(), trans=True)
    assert ((L.t() @ Y).double() - B.double()).abs().max().item() < tol * 50


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("n", [1, 5, 31, 32, 33, 63, 65, 100, 327, 449, 480, 481, 500, 512, 513, 700, 897, 1000, 1345, 2048])
def test_cholesky_with_explicit_inverse(dtype, tol, n):
    """wiski_potrf_inverse: factor in place + X = L^-1.  n <= 480 is the one-launch path of cooperating workgroups (dense_coop.h: every
    panel boundary case -- partial last block, exactly full blocks, one block, no owner tiles at all for <= 2 blocks); up to 512 the factor
    still is, with a blocked triangular solve for the inverse; beyond that the two-level blocked factorisation, whose diagonal-block inverses
    are assembled into X by two GEMMs per block row (dense.hip: potrf_two_level)."""
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(1000 + n)
    R = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64)).to(DEV, dtype)
    pad = torch.full((n, n + 3), float("nan"), device=DEV, dtype=dtype)      # a leading dimension larger than n is honoured
    L = A.clone()
    X, info = grid_ops.potrf_inverse_(L)
    assert int(info.item()) == 0
    Lref = torch.linalg.cholesky(A.double())
    assert torch.equal(L, torch.tril(L)) and torch.equal(X, torch.tril(X))
    assert (L.double() - Lref).abs().max().item() < tol * 10
    Xref = torch.linalg.inv(Lref)
    assert (X.double() - Xref).abs().max().item() < tol * 10 * max(1.0, Xref.abs().max().item())
    assert ((X @ L).double() - torch.eye(n, dtype=torch.float64, device=DEV)).abs().max().item() < tol * 50
    # plain potrf takes the same small path
    L2 = A.clone()
    assert int(grid_ops.potrf_(L2).item()) == 0
    assert torch.equal(L2, L)
    del pad


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.float64, 1e-10)])
def test_cooperative_cholesky_back_to_back_and_on_two_streams(dtype, tol):
    """The cooperating-workgroups factorisation keeps its hand-over flags in a persistent block per stream that every launch leaves
    zeroed: many launches back to back (different sizes, no synchronisation in between) and launches on a second stream give the same
    factors and inverses as the host reference."""
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(77)
    mats = []
    for n in (327, 64, 480, 97, 327, 33, 449):
        R = torch.randn(n, n, generator=g, dtype=torch.float64)
        mats.append((R @ R.t() / n + torch.eye(n, dtype=torch.float64)).to(DEV, dtype))
    outs = []
    for rep in range(3):
        for A in mats:
            L = A.clone()
            X, info = grid_ops.potrf_inverse_(L)
            outs.append((A, L, X, info))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for A in mats[:3]:
            L = A.clone()
            X, info = grid_ops.potrf_inverse_(L)
            outs.append((A, L, X, info))
    torch.cuda.synchronize()
    for A, L, X, info in outs:
        n = A.shape[0]
        assert int(info.item()) == 0
        Lref = torch.linalg.cholesky(A.double())
        assert (L.double() - Lref).abs().max().item() < tol * 10
        assert ((X @ L).double() - torch.eye(n, dtype=torch.float64, device=DEV)).abs().max().item() < tol * 50


def test_one_workgroup_cholesky_path_still_factorises():
    """The one-workgroup factorisation + separate inverse (dense_small.h) is what a stream under graph capture takes, and what
    WISKI_POTRF_COOP=0 selects: run it in a child process (the switch is read once per process) against the host reference."""
    import subprocess
    import sys

    code = (
        "import torch\n"
        "from online_gp_amd import grid_ops\n"
        "for dt, tol in ((torch.float64, 1e-9), (torch.float32, 5e-3)):\n"
        "    for n in (31, 64, 200, 327, 480, 512):\n"
        "        g = torch.Generator(device='cpu').manual_seed(n)\n"
        "        R = torch.randn(n, n, generator=g, dtype=torch.float64)\n"
        "        A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64)).to('cuda', dt)\n"
        "        L = A.clone(); X, info = grid_ops.potrf_inverse_(L)\n"
        "        assert int(info.item()) == 0\n"
        "        Lref = torch.linalg.cholesky(A.double())\n"
        "        assert (L.double() - Lref).abs().max().item() < tol, (dt, n)\n"
        "        assert ((X @ L).double() - torch.eye(n, dtype=torch.float64, device='cuda')).abs().max().item() < 10 * tol, (dt, n)\n"
        "print('one-workgroup path ok')\n"
    )
    env = dict(os.environ, WISKI_POTRF_COOP="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "one-workgroup path ok" in out.stdout, out.stderr[-2000:]


def test_cholesky_with_inverse_inside_a_captured_graph():
    """A stream under graph capture takes the two-launch path (the cooperative kernel's flag block is allocated on first use, which a
    capture forbids): the recorded factorisation replays correctly."""
    from online_gp_amd import grid_ops

    n = 200
    g = torch.Generator(device="cpu").manual_seed(1)
    R = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64)).to(DEV)
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    grid_ops.potrf_inverse_(A.clone(), info=info)                  # (warm-up outside the capture)
    torch.cuda.synchronize()
    graph, work = torch.cuda.CUDAGraph(), A.clone()
    with torch.cuda.graph(graph):
        work.copy_(A)
        X, _ = grid_ops.potrf_inverse_(work, info=info)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    assert ((X @ work) - torch.eye(n, dtype=torch.float64, device=DEV)).abs().max().item() < 1e-12


def test_multi_copy_is_one_launch_for_all_segments():
    from online_gp_amd import grid_ops

    src = [torch.randn(k, device=DEV, dtype=torch.float64) for k in (1, 7, 327 * 327, 4800, 1, 33)]
    dst = [torch.zeros_like(s_) for s_ in src]
    n_dev = torch.zeros(1, device=DEV, dtype=torch.float64)
    grid_ops.multi_copy(list(zip(dst, src)), scalar=12345.0, scalar_dst=n_dev)
    assert all(torch.equal(d_, s_) for d_, s_ in zip(dst, src)) and float(n_dev) == 12345.0


def test_small_cholesky_reports_a_non_positive_pivot():
    from online_gp_amd import grid_ops

    v = torch.randn(200, 3, device=DEV, dtype=torch.float64)
    A = v @ v.t()
    X, info = grid_ops.potrf_inverse_(A.clone())
    assert int(info.item()) != 0


def test_not_pd_sets_info_and_psd_safe_adds_jitter():
    from online_gp_amd import grid_ops

    w = torch.randn(900, 700, device=DEV, dtype=torch.float64)          # two-level path: rank 700 < 900, the failure is in the SECOND big block
    assert int(grid_ops.potrf_((w @ w.t()).clone()).item()) != 0

    v = torch.randn(50, 3, device=DEV, dtype=torch.float64)
    A = v @ v.t()                                   # rank 3: not PD
    assert int(grid_ops.potrf_(A.clone()).item()) != 0
    L = grid_ops.psd_safe_cholesky(A)
    assert ((L @ L.t()) - A).abs().max().item() < 1e-3


def test_updated_root_lazy_tensor_contract():
    """UpdatedRootLazyTensor (URLT:9-159): functional update, roots, matmul, evaluate."""
    from online_gp_amd.lazy import UpdatedRootLazyTensor

    torch.manual_seed(0)
    m, q = 40, 3
    W = torch.randn(m, 60, device=DEV, dtype=torch.float64)
    A = W @ W.t()
    lt = UpdatedRootLazyTensor(A, initial_is_root=False)
    assert lt.shape == (m, m) and torch.equal(lt.evaluate(), A)
    v = torch.randn(m, 2, device=DEV, dtype=torch.float64)
    assert torch.allclose(lt @ v, A @ v, atol=1e-10) and torch.allclose(lt._matmul(v), A @ v, atol=1e-10)
    L = lt.root_decomposition().root.evaluate()
    R = lt.root_inv_decomposition().root.evaluate()
    assert torch.allclose(L @ L.t(), A, atol=1e-8) and torch.allclose(R @ R.t(), torch.linalg.inv(A), atol=1e-8)
    V = torch.randn(m, q, device=DEV, dtype=torch.float64)
    lt2 = lt.update(V)
    assert lt2 is not lt and torch.equal(lt.evaluate(), A)                 # functional: the old object is untouched
    A2 = A + V @ V.t()
    assert torch.allclose(lt2.evaluate(), A2, atol=1e-10)
    L2, R2 = lt2.root, lt2.inv_root
    assert torch.allclose(L2 @ L2.t(), A2, atol=1e-8)                      # L~ L~^T = A + V V^T   (URLT:70-100)
    assert torch.allclose(R2.t() @ L2, torch.eye(m, device=DEV, dtype=torch.float64), atol=1e-8)   # R~ = L~^-T  (URLT:111-117)
    lt3 = lt2.update(V[:, 0])                                               # 1-D vector form (URLT:54-55)
    assert torch.allclose(lt3.evaluate(), A2 + V[:, :1] @ V[:, :1].t(), atol=1e-10)
    root_init = UpdatedRootLazyTensor(W.t().contiguous(), initial_is_root=True)      # tensor = root^T root (URLT:29-30)
    assert torch.allclose(root_init.evaluate(), A, atol=1e-9)
    assert lt.expand(2, m, m).shape == (2, m, m)


@pytest.mark.parametrize("dtype,rt", [(torch.float64, 1e-8), (torch.float32, 2e-3)])
def test_dense_path_matches_pcg_path_and_exposes_root_space(dtype, rt):
    """m = 1000 (the BO grid): dense MFMA factor == matrix-free PCG posterior; Q, K L, L^T K b shapes."""
    from online_gp_amd import settings
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (400, 3)); y = np.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(400)
    Xt, yt = torch.as_tensor(X, device=DEV, dtype=dtype), torch.as_tensor(y, device=DEV, dtype=dtype)[:, None]
    Xs = Xt[:32]
    outs = []
    for dense in (True, False):
        with settings.dense_small_grids(dense), settings.cg_tolerance(1e-11 if dtype == torch.float64 else 1e-6):
            m = FixedNoiseOnlineSKIGP(Xt[:300], yt[:300], None, grid_bounds=torch.tensor([[-1.1, 1.1]] * 3), grid_size=10, learn_additional_noise=True)
            m.eval()
            m.condition_on_observations(Xt[300:], yt[300:], inplace=True)
            d = m(Xs)
            outs.append((d.mean.double(), d.variance.double(), d.covariance_matrix.double()))
            if dense:
                assert m.prediction_cache["cg_iters"] == [0]
                assert m.current_qmatrix.shape == (1000, 1000) and m.current_inducing_compression_matrix.shape == (1000, 1000)
                assert m.root_space_projection.shape == (1000, 1)
                Q = m.current_qmatrix
                assert torch.allclose(Q, Q.t(), atol=1e-3 if dtype == torch.float32 else 1e-9)
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= rt * b.abs().max().item()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
def test_rank_q_posterior_updates_match_a_fresh_factor(dtype, tol):
    """Dense regime: condition_on_observations (in place and functional, single and multi-output, heteroscedastic)
    updates the cached posterior matrix by Woodbury rank-q steps; mean, variance and logdet equal a fresh factorisation."""
    from online_gp_amd import settings
    from online_gp_amd.lazy.dense_woodbury import DenseInducingPosterior
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    torch.manual_seed(2)
    d, g, n0, q, steps, out = 2, 12, 60, 5, 6, 2
    X = torch.rand(n0 + q * steps, d, device="cuda", dtype=dtype) * 2 - 1
    Y = torch.stack([torch.sin(2 * X[:, 0]) + X[:, 1], torch.cos(X[:, 0] * X[:, 1])], 1) + 0.1 * torch.randn(X.shape[0], out, device="cuda", dtype=dtype)
    N = torch.rand_like(Y) + 0.5
    gb = torch.tensor([[-1.1, 1.1]] * d)
    Xs = X[:40]

    def moments(model):
        with torch.no_grad():
            mvn = model(Xs)
        return mvn.mean, mvn.variance

    with torch.no_grad():
        fun = FixedNoiseOnlineSKIGP(X[:n0], Y[:n0], N[:n0], grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        inp = FixedNoiseOnlineSKIGP(X[:n0], Y[:n0], N[:n0], grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        moments(fun); moments(inp)                                   # build the dense caches
        for s in range(steps):
            sl = slice(n0 + s * q, n0 + (s + 1) * q)
            fun = fun.condition_on_observations(X[sl], Y[sl], N[sl])            # functional (fantasy-style)
            inp.condition_on_observations(X[sl], Y[sl], N[sl], inplace=True)
            for m_ in (fun, inp):
                assert "pending_rank_update" in m_._memo                                          # applied lazily ...
                pc = m_.prediction_cache
                assert all(p.updates == s + 1 for p in pc["pred_cov"].ops)                        # ... by a rank update, not a fresh factor
        with settings.dense_rank_updates(False):
            ref = FixedNoiseOnlineSKIGP(X, Y, N, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        mr, vr = moments(ref)
        for m_ in (fun, inp):
            mm, vv = moments(m_)
            assert float((mm - mr).abs().max()) < tol * float(mr.abs().max())
            assert float((vv - vr).abs().max()) < tol * float(vr.abs().max())
            for o, p in enumerate(m_._memo["prediction_cache"]["pred_cov"].ops):
                fresh = ref._memo["prediction_cache"]["pred_cov"].ops[o]
                assert isinstance(p, DenseInducingPosterior) and abs(float(p.logdet) - float(fresh.logdet)) < max(tol, 1e-6) * abs(float(fresh.logdet)) * 10


def test_root_space_objects_match_the_dense_reference_b1():
    """the reference-named root-space objects against the op-for-op dense restatement B1
    (oracle/dense_reference.py), value by value.
      * fresh model, A of full rank (no jitter, the Cholesky root is unique): Q = I + L^T Kt L, Kt L and L^T Kt b entry-wise;
      * after streaming updates the model's (L, R) are carried by the rank-q root update (wiski_root_update, URLT:69-119)
        like B1's SVD update; roots then differ by a right orthogonal factor, so: L L^T = A, R^T L = I, spec(Q), the
        quadratic form proj^T Q^-1 proj of the MLL (BWM:27), and the posterior mean / covariance the reference algebra
        gives from them (BFN:375-376, 399-403) against B1's."""
    from oracle import dense_reference
    from online_gp_amd import grid_ops
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(11)
    d, g = 2, 6
    gb = [[-1.1, 1.1]] * d
    n0 = 400
    X = rng.uniform(-1.1, 1.1, (n0 + 50, d)); y = np.sin(2 * X[:, 0]) * X[:, 1] + 0.1 * rng.standard_normal(n0 + 50)
    nz = rng.uniform(0.5, 1.5, n0 + 50)
    Xt, yt, nt = torch.as_tensor(X, device=DEV), torch.as_tensor(y, device=DEV)[:, None], torch.as_tensor(nz, device=DEV)[:, None]
    m = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], nt[:n0], grid_bounds=torch.tensor(gb, dtype=torch.float64), grid_size=g, learn_additional_noise=True)
    m.eval()
    s2 = float(m.likelihood.second_noise.detach())
    ell = m.covar_module.base_kernel.base_kernel.lengthscale.detach().double().cpu().numpy().reshape(-1)    # fp32 parameters: hand B1 the same values
    osc = float(m.covar_module.base_kernel.outputscale.detach().double())
    B1 = dense_reference.DenseWISKI(gb, g, lengthscale=ell, outputscale=osc, sigma2=s2)
    B1.set_train_data(X[:n0], y[:n0], nz[:n0])
    Kt, L1, KL1, Q1, Kb1, proj1, _ = B1._posterior_pieces()
    assert np.linalg.matrix_rank(B1.WtW) == g ** d          # full rank: both sides take the plain Cholesky root
    Q, KL, proj = (t.cpu().numpy() for t in (m.current_qmatrix, m.current_inducing_compression_matrix, m.root_space_projection))
    assert np.abs(Q - Q1).max() < 1e-8 * np.abs(Q1).max()
    assert np.abs(KL - KL1).max() < 1e-8 * np.abs(KL1).max()
    assert np.abs(proj - proj1).max() < 1e-8 * np.abs(proj1).max()
    wtw = m._kernel_cache["WtW"]
    L0 = wtw.root_decomposition().root.evaluate().clone()
    assert torch.allclose(L0, torch.tril(L0))
    # streaming: q = 1, 4, 12 -- the root pair follows by rank-q updates on both sides (a batch of more than m / 2 points
    # drops the pair and the next request re-factorises: q = 25 below)
    lo = n0
    for q in (1, 4, 12, 25):
        sl = slice(lo, lo + q); lo += q
        m.condition_on_observations(Xt[sl], yt[sl], nt[sl], inplace=True)
        B1.condition_on_observations(X[sl], y[sl], nz[sl])
        L = wtw.root_decomposition().root.evaluate()
        R = wtw.root_inv_decomposition().root.evaluate()
        assert torch.allclose(L, torch.tril(L)) == (q == 25)    # updated, not re-factorised
        A = wtw.evaluate()
        assert (L @ L.t() - A).abs().max() < 1e-10 * A.abs().max()
        assert (torch.as_tensor(B1.WtW, device=DEV) - A).abs().max() < 1e-10 * A.abs().max()
        assert (R.t() @ L - torch.eye(g ** d, device=DEV, dtype=L.dtype)).abs().max() < 1e-8
        Kt, L1, KL1, Q1, Kb1, proj1, cq = B1._posterior_pieces()
        Q, KL, proj = (t.cpu().numpy() for t in (m.current_qmatrix, m.current_inducing_compression_matrix, m.root_space_projection))
        ev, ev1 = np.linalg.eigvalsh(0.5 * (Q + Q.T)), np.linalg.eigvalsh(0.5 * (Q1 + Q1.T))
        assert np.abs(ev - ev1).max() < 1e-8 * ev1.max()
        qf, qf1 = (proj.T @ np.linalg.solve(Q, proj)).item(), (proj1.T @ np.linalg.solve(Q1, proj1)).item()
        assert abs(qf - qf1) < 1e-9 * abs(qf1)
        Kb = m.Kuu_response[0].cpu().numpy().reshape(-1, 1)
        mean_ref_alg = Kb - KL @ np.linalg.solve(Q, proj)                       # BFN:375-376 from the model's own pieces
        assert np.abs(mean_ref_alg - B1.pred_mean_cache()).max() < 1e-8 * np.abs(B1.pred_mean_cache()).max()
        cov_ref_alg = Kt - KL @ np.linalg.solve(Q, KL.T)                         # BFN:399-403
        assert np.abs(cov_ref_alg - B1.pred_cov_cache()).max() < 1e-8 * np.abs(B1.pred_cov_cache()).max()
        mu = m.prediction_cache["pred_mean"][0].cpu().numpy()
        assert np.abs(mu - B1.pred_mean_cache()).max() < 1e-7 * np.abs(mu).max()
    # the standalone operator: UpdatedRootLazyTensor.update against B1's SVD update on the same V
    from online_gp_amd.lazy import UpdatedRootLazyTensor

    A0 = torch.as_tensor(B1.WtW, device=DEV)
    U = UpdatedRootLazyTensor(A0, initial_is_root=False)
    V = torch.as_tensor(B1.wmat(X[lo:lo + 5]) / np.sqrt(nz[lo:lo + 5])[None, :], device=DEV)
    U2 = U.update(V)
    B1.condition_on_observations(X[lo:lo + 5], y[lo:lo + 5], nz[lo:lo + 5])
    L2, R2 = U2.root_decomposition().root.evaluate(), U2.root_inv_decomposition().root.evaluate()
    assert (U2.evaluate() - torch.as_tensor(B1.WtW, device=DEV)).abs().max() < 1e-10 * A0.abs().max()
    assert (L2 @ L2.t() - U2.evaluate()).abs().max() < 1e-9 * A0.abs().max()
    assert (R2.t() @ L2 - torch.eye(g ** d, device=DEV, dtype=L2.dtype)).abs().max() < 1e-8
    LB, RB = torch.as_tensor(B1.root, device=DEV), torch.as_tensor(B1.inv_root, device=DEV)
    assert (LB @ LB.t() - L2 @ L2.t()).abs().max() < 1e-8 * A0.abs().max()     # same Gram matrix as B1's L U S~
    assert (RB @ RB.t() - R2 @ R2.t()).abs().max() < 1e-6 * (RB @ RB.t()).abs().max()
    # fp32 root update, C ABI directly
    Lf, Rf, Vf = L0.float().contiguous(), torch.linalg.inv(L0).t().float().contiguous(), V.float().contiguous()
    Af = Lf @ Lf.t()
    grid_ops.root_update_(Lf, Rf, Vf)
    assert (Lf @ Lf.t() - (Af + Vf @ Vf.t())).abs().max() < 1e-4 * Af.abs().max()
    assert (Rf.t() @ Lf - torch.eye(g ** d, device=DEV)).abs().max() < 1e-2


def test_root_pair_is_dropped_when_the_stencil_is_rebuilt_or_all_reduced():
    """set_train_data zeroes the stencil in place; a root pair carried from before must not be
    rank-updated onto it (L L^T would be A_old + A_new).  Same for the statistics all-reduce of the data-parallel path."""
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    g = torch.Generator(device="cpu").manual_seed(5)
    X = (torch.rand(300, 2, generator=g, dtype=torch.float64) * 2 - 1).to(DEV)
    y = torch.sin(3 * X.sum(1, keepdim=True))
    gb = torch.tensor([[-1.1, 1.1]] * 2)
    model = FixedNoiseOnlineSKIGP(X, y, torch.ones_like(y), grid_bounds=gb, grid_size=10, learn_additional_noise=True)
    op = model._kernel_cache["WtW"]
    L0 = op.root_decomposition().root.evaluate().clone()
    assert op.root is not None
    X2 = (torch.rand(40, 2, generator=g, dtype=torch.float64) * 2 - 1).to(DEV)       # n <= m / 2: the rank-update branch
    y2 = torch.cos(2 * X2.sum(1, keepdim=True))
    model.set_train_data(X2, y2, torch.ones_like(y2))
    assert op.root is None and op.inv_root is None
    A = op.evaluate()
    L = op.root_decomposition().root.evaluate()
    assert ((L @ L.t()) - A).abs().max().item() < 1e-6 * max(1.0, A.abs().max().item())
    assert ((L0 @ L0.t()) - A).abs().max().item() > 1e-3          # the old root really described another matrix
    # data-parallel increment (half_delta path): the roots are dropped as well
    op.root_decomposition()
    halves = model._half_buffers()
    model._absorb(model._kernel_cache, X[:8], y[:8], torch.ones_like(y[:8]), init=False, half_delta=halves)
    assert op.root is None
    halves[0].zero_()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
@pytest.mark.parametrize("q", [1, 7, 32, 40])
def test_root_update_device_and_host_eigensolve(dtype, tol, q):
    """wiski_root_update for batch sizes on both sides of the device-Jacobi limit (q <= 32: one-workgroup parallel Jacobi,
    asynchronous; beyond: host Jacobi), including odd q and a rank-deficient V."""
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(q)
    m = 96
    Rm = torch.randn(m, m, generator=g, dtype=torch.float64)
    A = Rm @ Rm.t() / m + torch.eye(m, dtype=torch.float64)
    L0 = torch.linalg.cholesky(A)
    V = torch.randn(m, q, generator=g, dtype=torch.float64)
    if q >= 7:
        V[:, -1] = V[:, 0] + V[:, 1]                        # a direction already spanned by the others
    L = L0.to(DEV, dtype).contiguous(); R = torch.linalg.inv(L0).t().to(DEV, dtype).contiguous(); Vd = V.to(DEV, dtype).contiguous()
    grid_ops.root_update_(L, R, Vd)
    want = A + V @ V.t()
    assert ((L @ L.t()).double().cpu() - want).abs().max() < tol * want.abs().max()
    assert ((R.t() @ L).double().cpu() - torch.eye(m, dtype=torch.float64)).abs().max() < 50 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
@pytest.mark.parametrize("d,g", [(1, 64), (2, 30), (3, 10), (2, 7)])
def test_dense_factor_one_call_equals_the_stepwise_build(dtype, tol, d, g):
    """wiski_dense_factor (the dense regime's posterior factor as one C-ABI call: G = Kt^(1/2), B = I + sym(G A G), C, C^-1, T = C^-1 G,
    M = T^T T, logdet) against the same build one library call at a time (lazy/dense_woodbury.py: _build_stepwise)."""
    from online_gp_amd.lazy.dense_woodbury import DenseInducingPosterior
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(d * 100 + g)
    n = 150
    X = torch.as_tensor(rng.uniform(-1, 1, (n, d)), device=DEV, dtype=dtype)
    y = torch.as_tensor(np.sin(2 * rng.uniform(-1, 1, n)), device=DEV, dtype=dtype)[:, None]
    gb = torch.tensor([[-1.1, 1.1]] * d, dtype=torch.float64)
    model = FixedNoiseOnlineSKIGP(X, y, None, grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
    assert model._use_dense()
    post = model._posterior_op(0)
    assert isinstance(post, DenseInducingPosterior) and post.chol is not None
    alt = object.__new__(DenseInducingPosterior)
    for k in ("grid", "wtw", "tcol", "kscale", "eigen", "shape", "dtype", "device"):
        setattr(alt, k, getattr(post, k))
    alt._build_stepwise(post.grid, post.wtw, post.eigen, post.grid.m)
    sc = alt.dense.abs().max().item()
    assert (post.dense - alt.dense).abs().max().item() <= tol * sc
    assert (post.chol - alt.chol).abs().max().item() <= tol * alt.chol.abs().max().item()
    assert abs(float(post.logdet) - float(alt.logdet)) <= tol * max(abs(float(alt.logdet)), 1.0)
    assert torch.equal(post.chol, torch.tril(post.chol))
