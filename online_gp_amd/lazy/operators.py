"""Matrix-free operators of the WISKI hot path with the LazyTensor matmul
contract the reference relies on (``_size/_matmul/_transpose_nonbatch/evaluate/
matmul/@``; online_gp/lazy/updated_root_lazy_tensor.py:44-51,135-137).

All arithmetic is done by the HIP kernels behind :mod:`online_gp_amd.grid_ops`.
Each operator covers ONE output (a [m, m] or [n, n] matrix); the model stacks
them per output.
"""
import torch

from .. import grid_ops, settings
from ..distributions import LazyCovariance


class _Operator(LazyCovariance):
    def _size(self):
        return self.shape

    def _transpose_nonbatch(self):
        return self  # every operator here is symmetric

    def transpose(self, *a):
        return self

    def matmul(self, rhs):
        squeeze = rhs.dim() == 1
        rhs2 = rhs[:, None] if squeeze else rhs
        out = self._matmul(rhs2)
        return out[:, 0] if squeeze else out

    __matmul__ = matmul

    def evaluate(self):
        n = self.shape[-1]
        if n > 8192:
            raise RuntimeError(f"refusing to densify a {n} x {n} operator")
        eye = torch.eye(n, dtype=self.dtype, device=self.device)
        return self._matmul(eye)

    def diag(self):
        return self.evaluate().diagonal()


class _RootHolder:
    """``.root`` as gpytorch's RootLazyTensor exposes it (URLT:74-76)."""

    def __init__(self, root):
        from ..distributions import DenseLazyTensor

        self.root = DenseLazyTensor(root)


class StencilWtW(_Operator):
    """W^T D^-1 W in block-stencil form.  Native storage is the symmetric half: the
    (7^d + 1)/2 relative offsets >= the centre (A is symmetric: every stored entry
    stands for A[i, j] and A[j, i]) as ``(7^d + 1)/2 * m`` reals in the
    row-interleaved layout of ``wiski_scatter_stats_sym`` (include/wiski.h) -- the
    ``[(7^d + 1)/2, m]`` shape of ``stencil`` only carries the size, its rows are
    not offsets; ``grid_ops.half_stencil_to_offset_major`` converts.  A full
    offset-major ``[7^d, m]`` stencil is accepted too.  Replaces the dense m x m
    tensor held by the reference's UpdatedRootLazyTensor
    (updated_root_lazy_tensor.py:42,58)."""

    def __init__(self, grid, stencil, root=None, inv_root=None):
        self.grid = grid
        self.stencil = stencil
        self.shape = torch.Size([grid.m, grid.m])
        self.dtype = stencil.dtype
        self.device = stencil.device
        # the reference's own representation of this matrix (UpdatedRootLazyTensor: a root L and an inverse root
        # R = L^-T, URLT:36-42), kept on small grids (m <= settings.max_cholesky_size) from the first time it is
        # asked for and then carried through streaming updates by rank-q root updates (URLT:62-119)
        self.root = root
        self.inv_root = inv_root

    # -- root API of UpdatedRootLazyTensor (URLT:121-133): Cholesky branch of gpytorch's root_decomposition
    def _ensure_roots(self):
        if self.root is not None and self.inv_root is not None:
            return
        if self.grid.m > settings.max_cholesky_size.value():
            raise NotImplementedError("a dense root of W^T D^-1 W is only kept for grids of <= settings.max_cholesky_size nodes")
        L = grid_ops.psd_safe_cholesky(self.evaluate().contiguous(), jitter=settings.cholesky_jitter.value())
        eye = torch.eye(self.grid.m, dtype=self.dtype, device=self.device)
        self.root = L
        self.inv_root = grid_ops.trsm_(L, eye, trans=False).t().contiguous()      # R = L^-T

    def root_decomposition(self, **kwargs):
        self._ensure_roots()
        return _RootHolder(self.root)

    def root_inv_decomposition(self, **kwargs):
        self._ensure_roots()
        return _RootHolder(self.inv_root)

    def update_roots_(self, V):
        """A += V V^T has just been scattered into the stencil: carry (L, R) along (no-op while no root exists)."""
        if self.root is None or self.inv_root is None:
            return
        if V.shape[1] > self.grid.m // 2:          # a batch this large: cheaper to re-factorise when next asked
            self.root = self.inv_root = None
            return
        grid_ops.root_update_(self.root, self.inv_root, V)

    @classmethod
    def zeros(cls, grid, dtype, device):
        return cls(grid, torch.zeros(((grid.R + 1) // 2, grid.m), dtype=dtype, device=device))

    @property
    def is_half(self):
        return grid_ops.is_half_stencil(self.grid, self.stencil)

    def full_stencil(self):
        """The full [7^d, m] stencil (a copy; diagnostics and tests)."""
        if not self.is_half:
            return self.stencil.clone()
        full = torch.zeros((self.grid.R, self.grid.m), dtype=self.dtype, device=self.device)
        grid_ops.stencil_expand_add(self.grid, self.stencil.clone(), full)
        return full

    def clone(self):
        cp = lambda t: None if t is None else t.clone()
        return StencilWtW(self.grid, self.stencil.clone(), cp(self.root), cp(self.inv_root))

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return StencilWtW(self.grid, self.stencil.to(device), mv(self.root), mv(self.inv_root))

    def _matmul(self, rhs):  # rhs [m, k]
        V = rhs.t().contiguous()
        return grid_ops.stencil_spmv(self.grid, self.stencil, V).t()

    def diag(self):
        if self.is_half:                      # group 0 of the interleaved layout: 4 reals per row, the first is A[i, i]
            return self.stencil.reshape(-1)[0:4 * self.grid.m:4].clone()
        return self.stencil[(self.grid.R - 1) // 2].clone()


class KroneckerToeplitz(_Operator):
    """scale * kron_i SymToeplitz(tcol_i): the lazy ``Kuu`` of
    batched_fixed_noise_online_gp.py:334-341 (divided by sigma^2 when the
    second noise is learnable)."""

    def __init__(self, grid, tcol, scale=1.0):
        self.grid = grid
        self.tcol = tcol
        self.scale = float(scale)
        self.shape = torch.Size([grid.m, grid.m])
        self.dtype = tcol.dtype
        self.device = tcol.device

    def _matmul(self, rhs):
        V = rhs.t().contiguous()
        return grid_ops.kron_toeplitz_mm(self.grid, self.tcol, V, self.scale).t()

    def __truediv__(self, s):
        return KroneckerToeplitz(self.grid, self.tcol, self.scale / float(s))

    def diag(self):
        d0 = self.scale
        off = 0
        for g in self.grid.g:
            d0 = d0 * float(self.tcol[off])
            off += g
        return torch.full((self.grid.m,), d0, dtype=self.dtype, device=self.device)


class InducingPosterior(_Operator):
    """M = (Kt^-1 + A)^-1 = Kt - Kt L Q^-1 L^T Kt, the un-scaled posterior
    covariance of the inducing values (batched_fixed_noise_online_gp.py:385-404),
    applied by preconditioned CG (wiski_pcg) instead of through a root."""

    def __init__(self, grid, wtw, tcol, kscale, tol, max_iter, workspace=None, check_every=10, eigen=None, shift=0.0, err=None):
        self.grid = grid
        self.wtw = wtw
        self.tcol = tcol
        self.kscale = float(kscale)
        self.tol = tol
        self.max_iter = max_iter
        self.workspace = workspace
        self.check_every = check_every
        self.eigen = eigen
        self.shift = float(shift)
        self.err = err            # device out-of-grid flag: its value rides on the solver's convergence poll
        self.last_err = 0
        self.shape = torch.Size([grid.m, grid.m])
        self.dtype = tcol.dtype
        self.device = tcol.device
        self.last_iters = 0
        self.last_relres = []
        self.two_level_provider = None   # callable (operator, columns) -> TwoLevelStruct or None for solves that name no block (set by the model)
        self.last_two_level = None       # what the last solve used

    def solve_columns(self, RHS, U=None, Z=None, warm=False, first_check=0, inplace=False, R=None, two_level=None):
        """RHS [k, m] -> (U, Z) with U = M RHS.  inplace: write into the given U, Z even on a cold start.
        R [k, m] (optional): caller-owned residual buffer, left holding RHS - Z - A U; warm=2 starts from it.
        two_level: the exact block of the two-level preconditioner for this solve (lazy/two_level.py; default: the operator's own)."""
        k = RHS.shape[0] if RHS.dim() > 1 else 1
        if two_level is None and self.two_level_provider is not None:
            two_level = self.two_level_provider(self, k)          # the stream's block for THIS eigenbasis as it stands now, or None
        if two_level is not None and k > 1 and not (two_level.d_mc and k <= two_level.mc_cols):
            two_level = None                                      # (a caller's block without room for k columns: the separable model alone)
        self.last_two_level = two_level
        ce = self.check_every
        if k >= 16 and k * self.grid.m >= (1 << 21):
            # a wide solve: one iteration is hundreds of microseconds of kernels, a convergence poll ~10 us of drained pipeline -- look
            # after every iteration from the first one that can have converged (a poll every 3rd iteration stops one iteration late
            # on average: 3-7 % of a 15-iteration solve against ~2 % for the polls).  Documented on settings.cg_check_every, which
            # governs every other solve.
            ce = 1
            first_check = first_check or (2 if two_level is not None else 3)
        U, Z, it, res = grid_ops.pcg(self.grid, self.wtw.stencil, self.tcol, self.kscale, RHS, U=U, Z=Z, warm=warm, inplace=inplace, tol=self.tol,
                                     max_iter=self.max_iter, check_every=ce, workspace=self.workspace, eigen=self.eigen,
                                     shift=self.shift, first_check=first_check, err=self.err, R=R, two_level=two_level)
        self.last_iters, self.last_relres, self.last_err = it, res, grid_ops.pcg.last_err
        self.last_converged = grid_ops.pcg.last_converged
        return U, Z

    def _matmul(self, rhs):
        U, _ = self.solve_columns(rhs.t().contiguous())
        return U.t()


class InterpolatedKernel(_Operator):
    """W Kuu W^T for a set of points (gpytorch InterpolatedLazyTensor as built by
    covar_module(X); batched_fixed_noise_online_gp.py:143,178,205)."""

    def __init__(self, grid, x, tcol, scale, err):
        self.grid = grid
        self.x = x.contiguous()
        self.tcol = tcol
        self.scale = float(scale)
        self.err = err
        n = x.shape[0]
        self.shape = torch.Size([n, n])
        self.dtype = x.dtype
        self.device = x.device

    def _matmul(self, rhs):  # [n, k]
        n, k = rhs.shape
        ones = torch.ones(n, dtype=self.dtype, device=self.device)
        WtV = torch.zeros((k, self.grid.m), dtype=self.dtype, device=self.device)
        stats = torch.zeros(2, dtype=torch.float64, device=self.device)
        for c in range(k):
            grid_ops.scatter_stats(self.grid, self.x, rhs[:, c].contiguous(), ones, ones, ones, WtV[c], None, stats, self.err)
        KW = grid_ops.kron_toeplitz_mm(self.grid, self.tcol, WtV, self.scale)
        return grid_ops.gather(self.grid, self.x, KW, self.err)

    def diag(self):
        Wt = grid_ops.wt_columns(self.grid, self.x, self.err)
        KW = grid_ops.kron_toeplitz_mm(self.grid, self.tcol, Wt, self.scale)
        return grid_ops.gather(self.grid, self.x, KW, self.err, diag=True)


class PredictiveCovariance(LazyCovariance):
    """sigma2 * W* M W*^T for a query batch (batched_fixed_noise_online_gp.py:222-228),
    evaluated lazily: the k = n* solves U = M W*^T run once, on first use, in
    column chunks; ``diag`` needs only the per-query quadratic forms."""

    def __init__(self, post, x, sigma2, err, chunk=64, block=None, spectral=None):
        self.post = post
        self.spectral = spectral  # callable -> SpectralQuery for x (or None): lazy/spectral_woodbury.py
        self.x = x.contiguous()
        self.sigma2 = float(sigma2)
        self.err = err
        self.chunk = chunk
        self.block = block  # q: covariance only within consecutive blocks of q points
        n = x.shape[0]
        self.shape = torch.Size([n, n]) if block is None else torch.Size([n // block, block, block])
        self.dtype = x.dtype
        self.device = x.device
        self._diag = None
        self._full = None

    def _dense_path(self, want_full):
        """M is cached dense: rows w_p^T M by one row-gather, then quadratic forms by diag-gathers."""
        grid = self.post.grid
        if getattr(self, "_Mg", None) is None:
            self._Mg = grid_ops.gather_rows(grid, self.x, self.post.dense, self.err)    # [n, m]: row p = w_p^T M (coalesced row reads)
        Mg = self._Mg
        if self._diag is None:
            self._diag = grid_ops.gather(grid, self.x, Mg, self.err, diag=True) * self.sigma2
        if want_full:
            n = self.x.shape[0]
            if self.block is None:
                full = grid_ops.gather(grid, self.x, Mg, self.err) * self.sigma2                      # [n, n]
            else:
                # only the q x q blocks: pair (row b*q+i, query b*q+j) -> one diag-gather of n*q pairs
                q = self.block
                base = (torch.arange(n, device=self.device) // q) * q
                rows = torch.arange(n, device=self.device).repeat_interleave(q)
                cols = (base[:, None] + torch.arange(q, device=self.device)[None, :]).reshape(-1)
                full = grid_ops.gather(grid, self.x[cols].contiguous(), Mg[rows].contiguous(), self.err, diag=True)
                full = full.reshape(n // q, q, q) * self.sigma2
            self._full = 0.5 * (full + full.transpose(-1, -2))

    def _spectral_path(self, sp, want_full):
        """Reduced-eigenbasis factor (smooth kernels on large grids): one projection, one triangular solve."""
        if self._diag is None:
            self._diag = (sp.diag() * self.sigma2).to(self.dtype)
        if want_full:
            self._full = (sp.full(self.block) * self.sigma2).to(self.dtype)

    def _solve_chunks(self, want_full):
        if hasattr(self.post, "dense"):
            return self._dense_path(want_full)
        sp = self.spectral() if self.spectral is not None else None
        if sp is not None:
            return self._spectral_path(sp, want_full)
        n = self.x.shape[0]
        grid = self.post.grid
        diag = torch.empty(n, dtype=self.dtype, device=self.device)
        full = None
        if want_full:
            full = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        step = self.chunk if self.block is None else max(self.block, (self.chunk // self.block) * self.block)
        tol_keep = getattr(self.post, "tol", None)
        vt = settings.variance_cg_tolerance.value()
        if vt is not None and not want_full and tol_keep is not None:
            self.post.tol = max(float(vt), float(tol_keep))          # quadratic forms: second order in the residual
        try:
            self._solve_chunk_loop(n, step, grid, diag, full, want_full)
        finally:
            if tol_keep is not None:
                self.post.tol = tol_keep
        self._diag = diag * self.sigma2
        if want_full:
            full = full * self.sigma2
            self._full = 0.5 * (full + full.transpose(-1, -2))

    def _solve_chunk_loop(self, n, step, grid, diag, full, want_full):
        for s in range(0, n, step):
            e = min(s + step, n)
            xs = self.x[s:e]
            RHS = grid_ops.wt_columns(grid, xs, self.err)
            U, _ = self.post.solve_columns(RHS)
            diag[s:e] = grid_ops.gather(grid, xs, U, self.err, diag=True)
            if want_full:
                if self.block is None:
                    full[:, s:e] = grid_ops.gather(grid, self.x, U, self.err)
                else:
                    q = self.block
                    G = grid_ops.gather(grid, xs, U, self.err)  # [e-s, e-s]
                    nb = (e - s) // q
                    G = G.reshape(nb, q, nb, q)
                    idx = torch.arange(nb, device=self.device)
                    full[s // q:e // q] = G[idx, :, idx, :]

    def root_decomposition(self):
        """A root of this covariance where a factor provides one without a solve (the counterpart of the reference's
        ``fast_pred_samples`` branch, BFN:229-243 -- there a Lanczos root of rank <= n*, truncated without an error estimate):
        spectral factor: R = sigma (chol^-1 F*^T)^T [n*, r] plus the left-out prior variance as a diagonal term -- exactly the
        covariance ``evaluate()`` returns; dense factor: R = sigma W* L_M with M = L_M L_M^T (factorised once per posterior) -- exact.
        None on the PCG path (rough kernels on large grids) and for blocked covariances: callers then factorise ``evaluate()``."""
        from ..distributions import RootLazyTensor

        if self.block is not None:
            return None
        if getattr(self, "_root", None) is not None:
            return self._root
        s = self.sigma2 ** 0.5
        if hasattr(self.post, "dense"):
            LM = getattr(self.post, "_chol_M", None)
            if LM is None:
                LM = self.post._chol_M = grid_ops.psd_safe_cholesky(self.post.dense.contiguous())
            R = grid_ops.gather(self.post.grid, self.x, LM.t().contiguous(), self.err) * s        # W* L_M  [n*, m]
            self._root = RootLazyTensor(R)
            return self._root
        sp = self.spectral() if self.spectral is not None else None
        if sp is None:
            return None
        Y = sp._solve()                                                                           # chol^-1 F*^T  [r, n*]
        self._root = RootLazyTensor((Y.t() * s).to(self.dtype).contiguous(), (sp._tail * self.sigma2).to(self.dtype))
        return self._root

    def diag(self):
        if self._diag is None:
            self._solve_chunks(False)
        d = self._diag
        return d if self.block is None else d.reshape(-1, self.block)

    def evaluate(self):
        if self._full is None:
            self._solve_chunks(True)
        return self._full

    def __getitem__(self, item):
        from ..distributions import DenseLazyTensor

        return DenseLazyTensor(self.evaluate()[item])
