from .batched_woodbury_marginal_log_likelihood import BatchedWoodburyMarginalLogLikelihood
from .streaming_partial_mll import sm_partial_mll

__all__ = ["BatchedWoodburyMarginalLogLikelihood", "sm_partial_mll"]
