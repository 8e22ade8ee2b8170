"""Fixed-noise x learnable multiplicative second-noise Gaussian likelihood.

Mirror of the reference's online_gp/likelihoods/fnmg_likelihood.py:11-18
(``noise = noise_covar.noise * second_noise``), with the two gpytorch noise
modules it builds on restated minimally: ``FixedHeteroskedasticNoise`` (a fixed
per-point tensor) and ``HomoskedasticNoise`` (raw parameter 0 under
softplus + GreaterThan(1e-4), gpytorch's default noise constraint).
"""
import torch
from torch.nn.functional import softplus

from ..kernels import inv_softplus


class FixedNoise(torch.nn.Module):
    def __init__(self, noise):
        super().__init__()
        self.noise = noise

    def _apply(self, fn):
        self.noise = fn(self.noise)
        return super()._apply(fn)


class HomoskedasticNoise(torch.nn.Module):
    LOWER = 1e-4

    def __init__(self, batch_shape=torch.Size([])):
        super().__init__()
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(torch.Size(batch_shape) + torch.Size([1]))))

    @property
    def noise(self):
        return softplus(self.raw_noise) + self.LOWER

    @noise.setter
    def noise(self, value):
        v = torch.as_tensor(value, dtype=torch.float64).to(self.raw_noise.device)
        v = v.expand(self.raw_noise.shape) if v.dim() <= self.raw_noise.dim() else v.reshape(self.raw_noise.shape)
        with torch.no_grad():
            self.raw_noise.copy_(inv_softplus((v - self.LOWER).clamp_min(1e-12)).to(self.raw_noise))


class FNMGLikelihood(torch.nn.Module):
    """``FNMGLikelihood(noise, learn_additional_noise)``: fixed noise [.., n]
    times a learnable scalar (per output) second noise."""

    def __init__(self, noise, learn_additional_noise=False, batch_shape=torch.Size([]), **kwargs):
        super().__init__()
        self.noise_covar = FixedNoise(noise)
        self.second_noise_covar = HomoskedasticNoise(batch_shape=batch_shape) if learn_additional_noise else None

    @property
    def second_noise(self):
        if self.second_noise_covar is None:
            return 0
        return self.second_noise_covar.noise

    @second_noise.setter
    def second_noise(self, value):
        if self.second_noise_covar is None:
            raise RuntimeError("Attempting to set secondary learned noise for FixedNoiseGaussianLikelihood, "
                               "but learn_additional_noise must have been False!")
        self.second_noise_covar.noise = value

    @property
    def noise(self):
        return self.noise_covar.noise * self.second_noise  # fnmg_likelihood.py:16-18

    @noise.setter
    def noise(self, value):
        self.noise_covar.noise = value
