"""Hyper-parameter priors (gpytorch.priors, absent from this image): thin modules around torch.distributions with the
``log_prob`` the marginal log-likelihood adds per registered prior (reference
online_gp/mlls/batched_woodbury_marginal_log_likelihood.py:48-49).  The reference's drivers use
``GammaPrior(3.0, 6.0)`` on length-scales and ``GammaPrior(2.0, 0.15)`` on output-scales
(experiments/bayesopt/bayesopt.py:72-76, experiments/active_learning/qnIPV_experiment.py:94-96)."""
import torch


class Prior(torch.nn.Module):
    def _dist(self, like):
        raise NotImplementedError

    def log_prob(self, value):
        return self._dist(value).log_prob(value)


class GammaPrior(Prior):
    def __init__(self, concentration, rate):
        super().__init__()
        self.register_buffer("concentration", torch.as_tensor(float(concentration), dtype=torch.float64))
        self.register_buffer("rate", torch.as_tensor(float(rate), dtype=torch.float64))

    def _dist(self, like):
        return torch.distributions.Gamma(self.concentration.to(like), self.rate.to(like), validate_args=False)


class NormalPrior(Prior):
    def __init__(self, loc, scale):
        super().__init__()
        self.register_buffer("loc", torch.as_tensor(float(loc), dtype=torch.float64))
        self.register_buffer("scale", torch.as_tensor(float(scale), dtype=torch.float64))

    def _dist(self, like):
        return torch.distributions.Normal(self.loc.to(like), self.scale.to(like), validate_args=False)


class LogNormalPrior(NormalPrior):
    def _dist(self, like):
        return torch.distributions.LogNormal(self.loc.to(like), self.scale.to(like), validate_args=False)


class UniformPrior(Prior):
    def __init__(self, a, b):
        super().__init__()
        self.register_buffer("a", torch.as_tensor(float(a), dtype=torch.float64))
        self.register_buffer("b", torch.as_tensor(float(b), dtype=torch.float64))

    def _dist(self, like):
        return torch.distributions.Uniform(self.a.to(like), self.b.to(like), validate_args=False)


REGISTRY_EPOCH = [0]      # bumped by every register_prior call anywhere: a cheap "did the set of priors change" for cached consumers


def note_registration():
    REGISTRY_EPOCH[0] += 1


def named_priors(module):
    """(name, prior, closure) for every prior registered (``register_prior``) on `module` or a sub-module;
    ``closure()`` returns the constrained parameter value the prior scores."""
    seen = set()
    for prefix, sub in module.named_modules():
        reg = getattr(sub, "_wiski_priors", None)
        if not reg or id(sub) in seen:
            continue
        seen.add(id(sub))
        for name, (prior, closure) in reg.items():
            yield (f"{prefix}.{name}" if prefix else name), prior, closure
