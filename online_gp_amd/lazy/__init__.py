from .operators import (InducingPosterior, InterpolatedKernel, KroneckerToeplitz, PredictiveCovariance, StencilWtW)
from .dense_woodbury import DenseInducingPosterior
from .updated_root_lazy_tensor import UpdatedRootLazyTensor

__all__ = ["StencilWtW", "KroneckerToeplitz", "InducingPosterior", "InterpolatedKernel", "PredictiveCovariance", "DenseInducingPosterior",
           "UpdatedRootLazyTensor"]
