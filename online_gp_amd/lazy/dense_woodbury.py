"""Dense Woodbury factor for small inducing grids (m <= settings.max_cholesky_size):
the regime the reference runs in (BFN:343-404 with a full Cholesky root).

Instead of a root of A = W^T D^-1 W (rank deficient until n >= m, hence the
reference's Cholesky jitter) the symmetric root of Kt is used, which is exact
and jitter-free:  with G = Kt^(1/2) (Kronecker eigenbasis),
    B = I + G A G  (SPD, eigenvalues >= 1),   B = C C^T   (wiski_potrf)
    M = (Kt^-1 + A)^-1 = G B^-1 G = T^T T,  T = C^-1 G     (explicit C^-1 from wiski_potrf_inverse + MFMA wiski_gemm)
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
        if wtw.stencil.is_contiguous() and grid_ops.is_half_stencil(grid, wtw.stencil) and len(eigen) == 2:
            # the whole build queued by ONE C call (wiski_dense_factor: G = Kt^(1/2), B = I + sym(G A G), C and C^-1, T = C^-1 G, M = T^T T, logdet)
            M, C0, logdet, info = grid_ops.dense_factor(grid, wtw.stencil, eigen, self.kscale)
            if bool(((info[0] == 0) & torch.isfinite(C0.diagonal()).all()).item()):             # one host read
                self.chol, self.dense, self.logdet = C0, M, logdet
                self.last_iters, self.last_relres = 0, []
                self.updates = 0
                return
        self._build_stepwise(grid, wtw, eigen, m)
        self.last_iters, self.last_relres = 0, []
        self.updates = 0          # rank-q updates applied since the last fresh factorisation

    def _build_stepwise(self, grid, wtw, eigen, m):
        """The same factor one library call at a time: full-stencil layouts, and the jitter escalation after a non-positive pivot."""
        eye = torch.eye(m, dtype=self.dtype, device=self.device)
        G = grid_ops.kron_spectral_mm(grid, eigen, eye, kscale=self.kscale, power=0.5)          # Kt^(1/2), symmetric [m, m]
        AG = grid_ops.stencil_spmv(grid, wtw.stencil, G)                                         # rows = columns of A G (A, G symmetric)
        B = grid_ops.kron_spectral_mm(grid, eigen, AG, kscale=self.kscale, power=0.5)            # G A G
        B = 0.5 * (B + B.t())
        B.diagonal().add_(1.0)
        # C and C^-1 from one call (wiski_potrf_inverse: one launch for m <= 480, the two-level blocked form beyond), T = C^-1 G as ONE
        # GEMM instead of a blocked triangular solve with m right-hand sides; a non-positive pivot takes the jitter escalation
        Bc = B.contiguous()
        C0 = Bc.clone()
        Cinv, info = grid_ops.potrf_inverse_(C0)
        if bool(((info[0] == 0) & torch.isfinite(C0.diagonal()).all()).item()):                 # one host read
            self.chol = C0                                                                       # C
            T = grid_ops.gemm(Cinv, G)                                                           # C^-1 G
        else:
            self.chol = grid_ops.psd_safe_cholesky(Bc)
            T = grid_ops.trsm_(self.chol, G.clone(), trans=False)
        del Cinv
        self.dense = grid_ops.gemm(T, T, ta=True)                                                # M = T^T T
        self.logdet = grid_ops.chol_logdet(self.chol)

    def rank_update(self, wtw_new, x, wa, err):
        """Posterior for the statistics A + W(x)^T diag(wa) W(x), from this one, in O(m^2 q):
            M' = M - M W^T (diag(1/wa) + W M W^T)^-1 W M,
            logdet(I + Kt A') = logdet(I + Kt A) + logdet(diag(wa)) + logdet(diag(1/wa) + W M W^T)
        -- the posterior-space counterpart of the reference's rank-q root update (URLT:69-119), used by
        condition_on_observations / fantasies in the dense regime instead of a fresh O(m^3) factor."""
        grid = self.grid
        new = object.__new__(DenseInducingPosterior)
        new.grid, new.wtw, new.tcol, new.kscale, new.eigen = grid, wtw_new, self.tcol, self.kscale, self.eigen
        new.shape, new.dtype, new.device = self.shape, self.dtype, self.device
        new.chol = None
        MW = grid_ops.gather_rows(grid, x, self.dense, err)               # [q, m]: rows w_p^T M
        W = grid_ops.wt_columns(grid, x, err)                             # [q, m]
        S = grid_ops.gemm(MW, W, tb=True)                                 # W M W^T
        S = 0.5 * (S + S.t())
        S.diagonal().add_(1.0 / wa)
        L = grid_ops.psd_safe_cholesky(S.contiguous())
        C = grid_ops.trsm_(L, MW.clone(), trans=False)                    # L^-1 W M
        new.dense = self.dense.clone()
        grid_ops.gemm(C, C, ta=True, alpha=-1.0, beta=1.0, C=new.dense)   # M - (W M)^T S^-1 (W M)
        new.logdet = self.logdet + grid_ops.chol_logdet(L) + torch.log(wa.double()).sum()
        new.last_iters, new.last_relres = 0, []
        new.updates = self.updates + 1
        return new

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
