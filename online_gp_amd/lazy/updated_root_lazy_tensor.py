"""UpdatedRootLazyTensor -- dense A = W^T D^-1 W with a (root, inverse root) pair and
rank-q updates; host-side mirror of the reference's
online_gp/lazy/updated_root_lazy_tensor.py:9-159 (same constructor, functional
``update``, ``root_decomposition().root``, ``root_inv_decomposition().root``,
``evaluate``, ``_matmul``/``@``, ``expand``) on the dense MFMA kernels.

Root update (URLT:69-119).  The reference forms p = R^T V, a full SVD p = U S V^T with
an r x r U, and L <- L U S~, R <- R U S~^-1 (O(m r^2)).  ``wiski_root_update`` (csrc/dense.hip)
produces the same Gram matrices in O(m r q) with the thin factor U_q = p V_t S^-1 obtained from the
q x q eigen-problem p^T p = V_t S^2 V_t^T (host Jacobi, tiny):
    L <- L + (L U_q) diag(sqrt(S^2+1) - 1) U_q^T,   R <- R + (R U_q) diag(1/sqrt(S^2+1) - 1) U_q^T
so that L L^T = A + V V^T and R = L^-T exactly as in the reference (roots are unique
only up to a right orthogonal factor; L differs from the reference's L U S~ by one).
"""
import torch

from .. import grid_ops, settings
from ..distributions import DenseLazyTensor
from .operators import _Operator


class _Root:
    def __init__(self, root):
        self.root = DenseLazyTensor(root) if torch.is_tensor(root) else root


class UpdatedRootLazyTensor(_Operator):
    def __init__(self, initial_tensor=None, n_shape=None, initial_is_root=True, root=None, inv_root=None):
        if initial_tensor is None:
            initial_tensor = torch.zeros(1, n_shape)
        if initial_is_root:
            initial_tensor = grid_ops.gemm(initial_tensor, initial_tensor, ta=True)     # URLT:29-30
        self.tensor = initial_tensor.contiguous()
        self.root = root
        self.inv_root = inv_root
        self.shape = self.tensor.shape
        self.dtype, self.device = self.tensor.dtype, self.tensor.device

    def _matmul(self, rhs):
        return grid_ops.gemm(self.tensor, rhs.contiguous())

    def evaluate(self):
        return self.tensor

    def diag(self):
        return self.tensor.diagonal()

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return UpdatedRootLazyTensor(self.tensor.to(device), initial_is_root=False, root=mv(self.root), inv_root=mv(self.inv_root))

    # -- roots: Cholesky branch of gpytorch's root_decomposition (m <= max_cholesky_size), URLT:121-133
    def _ensure_roots(self):
        triangular = False
        if self.root is None:
            L = grid_ops.psd_safe_cholesky(self.tensor, jitter=settings.cholesky_jitter.value())
            self.root = L
            triangular = True
        if self.inv_root is None:
            # (a root this object has just factorised is triangular: one blocked substitution; a root handed in by the caller -- any square
            #  root, e.g. one carried through rank-q updates -- takes the general inverse.  No look at the entries: the comparison
            #  `root == tril(root)` this replaced was a host synchronisation per call)
            if triangular:
                eye = torch.eye(self.shape[-1], dtype=self.dtype, device=self.device)
                Linv = grid_ops.trsm_(self.root, eye, trans=False)
            else:
                Linv = torch.linalg.inv_ex(self.root).inverse
            self.inv_root = Linv.t().contiguous()                                        # R = L^-T

    def root_decomposition(self, **kwargs):
        self._ensure_roots()
        return _Root(self.root)

    def root_inv_decomposition(self, **kwargs):
        self._ensure_roots()
        return _Root(self.inv_root)

    def update(self, vector):
        if vector.dim() == 1:
            vector = vector.view(-1, 1)
        V = vector.contiguous()
        tensor = grid_ops.gemm(V, V, tb=True, alpha=1.0, beta=1.0, C=self.tensor.clone())   # A + V V^T, URLT:58
        root, inv_root = self.collect_vector(V)
        return UpdatedRootLazyTensor(tensor, initial_is_root=False, root=root, inv_root=inv_root)

    def collect_vector(self, V):
        """(new root, new inverse root) for A + V V^T (URLT:69-119), on the C ABI's ``wiski_root_update``."""
        self._ensure_roots()
        L, R = self.root.clone(), self.inv_root.clone()
        grid_ops.root_update_(L, R, V.contiguous())
        return L, R

    def expand(self, *sizes):
        return self._expand_batch(torch.Size(sizes[0] if len(sizes) == 1 and not isinstance(sizes[0], int) else sizes)[:-2])

    def _expand_batch(self, batch_shape):
        from ..models.batched_fixed_noise_online_gp import BatchOperator

        n = 1
        for b in batch_shape:
            n *= b
        return BatchOperator([UpdatedRootLazyTensor(self.tensor.clone(), initial_is_root=False,
                                                    root=None if self.root is None else self.root.clone(),
                                                    inv_root=None if self.inv_root is None else self.inv_root.clone()) for _ in range(max(n, 1))])
