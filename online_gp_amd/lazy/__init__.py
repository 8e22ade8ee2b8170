from .operators import (InducingPosterior, InterpolatedKernel, KroneckerToeplitz, PredictiveCovariance, StencilWtW)

__all__ = ["StencilWtW", "KroneckerToeplitz", "InducingPosterior", "InterpolatedKernel", "PredictiveCovariance"]
