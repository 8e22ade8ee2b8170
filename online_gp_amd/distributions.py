"""Minimal MultivariateNormal with a lazily evaluated covariance (the attributes
the reference's callers read: ``mean``, ``variance``, ``stddev``,
``covariance_matrix``, ``lazy_covariance_matrix``, ``rsample``; see
online_gp/models/online_ski_regression.py:56-62 and
online_gp/models/online_ski_botorch_model.py:63-68)."""
import torch


class LazyCovariance:
    """Interface: ``shape``, ``diag()``, ``evaluate()``."""

    shape = None

    def diag(self):
        raise NotImplementedError

    def evaluate(self):
        raise NotImplementedError

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def ndimension(self):
        return len(self.shape)

    dim = ndimension


class ZeroLazyTensor(LazyCovariance):
    def __init__(self, *sizes, dtype=None, device=None):
        self.shape = torch.Size(sizes)
        self.dtype = dtype
        self.device = device

    def diag(self):
        return torch.zeros(self.shape[:-1], dtype=self.dtype, device=self.device)

    def evaluate(self):
        return torch.zeros(self.shape, dtype=self.dtype, device=self.device)


class DenseLazyTensor(LazyCovariance):
    def __init__(self, tensor):
        self.tensor = tensor
        self.shape = tensor.shape

    def diag(self):
        return self.tensor.diagonal(dim1=-2, dim2=-1)

    def evaluate(self):
        return self.tensor

    def __getitem__(self, item):
        return DenseLazyTensor(self.tensor[item])


class RootLazyTensor(LazyCovariance):
    """R R^T (+ diag(extra)) from a root R [n, r] -- what gpytorch's RootLazyTensor is to the reference's ``fast_pred_samples``
    branch (BFN:229-243): a covariance that is sampled in O(n r) without ever factorising an n x n matrix.  ``extra`` (optional,
    [n] >= 0) is a diagonal term the root does not carry (the prior variance a truncated basis leaves out)."""

    def __init__(self, root, extra=None):
        self.root = root
        self.extra = extra
        n = root.shape[-2]
        self.shape = torch.Size(tuple(root.shape[:-2]) + (n, n))
        self.dtype, self.device = root.dtype, root.device

    def diag(self):
        d = (self.root * self.root).sum(-1)
        return d if self.extra is None else d + self.extra

    def evaluate(self):
        full = self.root @ self.root.transpose(-1, -2)
        if self.extra is not None:
            full = full + torch.diag_embed(self.extra)
        return full

    def root_decomposition(self):
        return self

    def __getitem__(self, item):
        return DenseLazyTensor(self.evaluate()[item])


class MultivariateNormal:
    def __init__(self, mean, covariance):
        self.loc = mean
        if torch.is_tensor(covariance):
            covariance = DenseLazyTensor(covariance)
        self._covar = covariance

    @property
    def mean(self):
        return self.loc

    @property
    def lazy_covariance_matrix(self):
        return self._covar

    @property
    def covariance_matrix(self):
        return self._covar.evaluate()

    @property
    def variance(self):
        return self._covar.diag().clamp_min(0).reshape(self.loc.shape)

    @property
    def stddev(self):
        return self.variance.sqrt()

    @property
    def batch_shape(self):
        return self.loc.shape[:-1]

    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    def confidence_region(self):
        s = self.stddev * 2
        return self.mean - s, self.mean + s

    def rsample(self, sample_shape=torch.Size()):
        # a covariance that knows a root of itself (RootLazyTensor; PredictiveCovariance from the spectral or the dense factor) is
        # sampled through it: loc + R z (+ sqrt(extra) z'), exact for R R^T + diag(extra), no n x n factorisation and no jitter
        rd = getattr(self._covar, "root_decomposition", None)
        rt = rd() if rd is not None else None
        if rt is not None and rt.root.dim() == 2 and self.loc.dim() == 1:
            z = torch.randn(*sample_shape, rt.root.shape[-1], dtype=rt.root.dtype, device=rt.root.device)
            out = self.loc + z @ rt.root.t()
            if rt.extra is not None:
                out = out + rt.extra.clamp_min(0).sqrt() * torch.randn(*sample_shape, *self.loc.shape, dtype=rt.root.dtype, device=rt.root.device)
            return out
        cov = self.covariance_matrix
        n = cov.shape[-1]
        jitter = 1e-6 if cov.dtype == torch.float32 else 1e-8
        eye = torch.eye(n, dtype=cov.dtype, device=cov.device)
        L = None
        for i in range(6):
            L, info = torch.linalg.cholesky_ex(cov + (jitter * 10 ** i) * cov.diagonal(dim1=-2, dim2=-1).mean().clamp_min(1e-30) * eye)
            if int(info.max()) == 0:
                break
        z = torch.randn(*sample_shape, *self.loc.shape, dtype=cov.dtype, device=cov.device)
        return self.loc + (L @ z.unsqueeze(-1)).squeeze(-1)

    sample = rsample
