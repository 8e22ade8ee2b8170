"""OnlineSKIBotorchModel -- BoTorch-facing adaptor (reference
online_gp/models/online_ski_botorch_model.py:11-68).  BoTorch is absent from
this image; the posterior object therefore duck-types ``GPyTorchPosterior``
(``mvn``, ``mean``, ``variance``, ``rsample``) and is upgraded to the real class
when botorch is importable."""
import torch

from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP

try:  # pragma: no cover - botorch not installed here
    from botorch.posteriors import GPyTorchPosterior as _GPyTorchPosterior
except Exception:  # noqa: BLE001
    _GPyTorchPosterior = None


class WiskiPosterior:
    def __init__(self, mvn):
        self.mvn = mvn

    @property
    def mean(self):
        return self.mvn.mean.unsqueeze(-1)

    @property
    def variance(self):
        return self.mvn.variance.unsqueeze(-1)

    @property
    def device(self):
        return self.mvn.mean.device

    @property
    def dtype(self):
        return self.mvn.mean.dtype

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        return self.mvn.rsample(sample_shape).unsqueeze(-1)


class OnlineSKIBotorchModel(FixedNoiseOnlineSKIGP):
    def __init__(self, train_inputs=None, train_targets=None, train_noise_term=None, covar_module=None, kernel_cache=None,
                 grid_bounds=None, grid_size=30, learn_additional_noise=False, **kwargs):
        super().__init__(train_inputs=train_inputs, train_targets=train_targets, train_noise_term=train_noise_term,
                         covar_module=covar_module, kernel_cache=kernel_cache, grid_bounds=grid_bounds, grid_size=grid_size,
                         learn_additional_noise=learn_additional_noise, **kwargs)
        self._is_custom_likelihood = True
        # a BO loop refits the hyper-parameters at every step (bayesopt.py:187): its MLL steps take the device pipeline of the spectral
        # factor on small grids as well (settings.spectral_dense_regime)
        self.__dict__["_stream_owner"] = True

    def forward(self, X):
        if X is not None and X.dim() > 2 and X.shape[0] == 1:   # :37-42
            X = X[0]
        return super().forward(X)

    def get_fantasy_model(self, inputs, targets, noise=None, **kwargs):
        if noise is None:
            noise = torch.ones_like(targets) * self.likelihood.noise.mean()
        return super().get_fantasy_model(inputs, targets, noise)

    def fantasize(self, X, sampler, observation_noise=True, **kwargs):
        """OSB:51-61: sample fantasy targets at X [*b, q, d] from the current posterior (``sampler(posterior)`` ->
        [num_fantasies, *b, q, 1]) and condition on them; the noise of the fantasy points is the mean of the likelihood's
        noise, as upstream.  Returns a batch of conditioned models (batch shape [num_fantasies, *b])."""
        kwargs.pop("propagate_grads", None)
        post_X = self.posterior(X, observation_noise=observation_noise, **kwargs)
        Y_fantasized = sampler(post_X)
        noise = self.likelihood.noise.mean().detach().expand(Y_fantasized.shape[1:])
        return self.condition_on_observations(X=X, Y=Y_fantasized, noise=noise)

    def posterior(self, X, observation_noise=False, **kwargs):
        self.eval()
        X = X.to(self._dtype)
        mvn = self(X)
        if observation_noise:          # sigma2 on the diagonal (the fixed per-point part is unknown at new inputs)
            from ..distributions import MultivariateNormal

            cov = mvn.covariance_matrix
            eye = torch.eye(cov.shape[-1], dtype=cov.dtype, device=cov.device)
            mvn = MultivariateNormal(mvn.mean, cov + float(self._sigma2(0)) * eye)
        if _GPyTorchPosterior is not None:  # pragma: no cover
            try:
                import gpytorch

                return _GPyTorchPosterior(gpytorch.distributions.MultivariateNormal(mvn.mean, mvn.covariance_matrix))
            except Exception:  # noqa: BLE001
                pass
        return WiskiPosterior(mvn)
