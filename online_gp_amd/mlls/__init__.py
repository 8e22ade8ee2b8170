from .batched_woodbury_marginal_log_likelihood import BatchedWoodburyMarginalLogLikelihood
from .feature_gradient import mll_feature_surrogate
from .streaming_partial_mll import sm_partial_mll

__all__ = ["BatchedWoodburyMarginalLogLikelihood", "mll_feature_surrogate", "sm_partial_mll"]
