"""Dense Woodbury factor for small inducing grids (m <= settings.max_cholesky_size):
the regime the reference runs in (BFN:343-404 with a full Cholesky root).

Instead of a root of A = W^T D^-1 W (rank deficient until n >= m, hence the
reference's Cholesky jitter) the symmetric root of Kt is used, which is exact
and jitter-free:  with G = Kt^(1/2) (Kronecker eigenbasis),
    B = I + G A G  (SPD, eigenvalues >= 1),   B = C C^T   (wiski_potrf)
    M = (Kt^-1 + A)^-1 = G B^-1 G = T^T T,  T = C^-1 G     (wiski_trsm + MFMA wiski_gemm)
    logdet(I + L^T Kt L) = logdet(B) = 2 sum log diag C     (Sylvester; BWM:27)
M (m x m, a few MB) is then cached: every later posterior call is two gathers.
"""
import torch

from .. import grid_ops
from .operators import _Operator


class DenseInducingPosterior(_Operator):
    def __init__(self, grid, wtw, tcol, kscale, eigen):
        self.grid, self.wtw, self.tcol, self.kscale, self.eigen = grid, wtw, tcol, float(kscale), eigen
        m = grid.m
        self.shape = torch.Size([m, m])
        self.dtype, self.device = tcol.dtype, tcol.device
        eye = torch.eye(m, dtype=self.dtype, device=self.device)
        G = grid_ops.kron_spectral_mm(grid, eigen, eye, kscale=self.kscale, power=0.5)          # Kt^(1/2), symmetric [m, m]
        AG = grid_ops.stencil_spmv(grid, wtw.stencil, G)                                         # rows = columns of A G (A, G symmetric)
        B = grid_ops.kron_spectral_mm(grid, eigen, AG, kscale=self.kscale, power=0.5)            # G A G
        B = 0.5 * (B + B.t())
        B.diagonal().add_(1.0)
        self.chol = grid_ops.psd_safe_cholesky(B.contiguous())                                   # C
        T = grid_ops.trsm_(self.chol, G.clone(), trans=False)                                    # C^-1 G
        self.dense = grid_ops.gemm(T, T, ta=True)                                                # M = T^T T
        self.logdet = grid_ops.chol_logdet(self.chol)
        self.last_iters, self.last_relres = 0, []

    def solve_columns(self, RHS, U=None, Z=None, warm=False):
        """RHS [k, m] -> U = M RHS (rows), Z = Kt^-1 U is not tracked in the dense path."""
        U = grid_ops.gemm(RHS.contiguous(), self.dense)      # M symmetric: (M RHS^T)^T = RHS M
        return U, None

    def _matmul(self, rhs):
        return grid_ops.gemm(self.dense, rhs.contiguous())

    def evaluate(self):
        return self.dense

    def diag(self):
        return self.dense.diagonal()
