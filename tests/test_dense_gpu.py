"""GPU: dense MFMA building blocks against plain PyTorch references of the same ops
(asymmetric operands so a transposed C-write cannot pass)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (130, 70, 45), (257, 301, 129), (1000, 96, 1000)])
def test_gemm_all_transposes(dtype, tol, ta, tb, M, N, K):
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


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("n", [17, 64, 200, 1000])
def test_cholesky_trsm_logdet(dtype, tol, n):
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


def test_not_pd_sets_info_and_psd_safe_adds_jitter():
    from online_gp_amd import grid_ops

    v = torch.randn(50, 3, device=DEV, dtype=torch.float64)
    A = v @ v.t()                                   # rank 3: not PD
    assert int(grid_ops.potrf_(A.clone()).item()) != 0
    L = grid_ops.psd_safe_cholesky(A)
    assert ((L @ L.t()) - A).abs().max().item() < 1e-3
