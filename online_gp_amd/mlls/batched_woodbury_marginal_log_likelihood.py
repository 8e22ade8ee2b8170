"""BatchedWoodburyMarginalLogLikelihood (reference
online_gp/mlls/batched_woodbury_marginal_log_likelihood.py:6-51)."""
import torch


class BatchedWoodburyMarginalLogLikelihood(torch.nn.Module):
    def __init__(self, likelihood, model, clear_caches_every_iteration=False):
        super().__init__()
        self.likelihood = likelihood
        self.model = model
        self.has_learnable_noise = self.likelihood.second_noise_covar is not None
        self.clear_caches_every_iteration = clear_caches_every_iteration

    def forward(self, distro, targets, *args):
        raise NotImplementedError
