"""OnlineSKIClassifier -- two-class Dirichlet-transformed WISKI classifier (counterpart of
online_gp/models/online_ski_classifier.py:14-145 and gp_dirichlet_classification.py:5-45):

    OnlineSKIClassifier(stem, init_x, init_y, alpha_eps, lr, grid_size, grid_bound, **kw)
    .fit(x, y, num_epochs, test_dataset=None)   .update(x, y, update_stem=True, update_gp=True)   .predict(x) -> labels
    .set_train_data(features, targets, noise)   .set_lr(...)

Labels y in {0, 1} become two regression outputs with heteroscedastic fixed noise: alpha = alpha_eps + onehot(y),
sigma2_i = log(1/alpha + 1), target = log(alpha) - sigma2_i / 2; the predicted class is the argmax of the two posterior
means.  Exercises the 2-output batch layout and non-unit noise of the GP core; the streaming protocol is shared with
the regression wrapper (``_streaming_wrapper.StreamingSKIWrapper``)."""
import torch

from .. import settings
from ._streaming_wrapper import StreamingSKIWrapper
from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP


def dirichlet_transform(targets, alpha_eps, num_classes=2):
    """labels [n] -> (regression targets [n, C], alpha [n, C], per-point noise variances [n, C])"""
    labels = targets.reshape(-1).long()
    alpha = torch.full((labels.shape[0], num_classes), float(alpha_eps), device=labels.device)
    alpha.scatter_add_(1, labels[:, None], torch.ones_like(alpha[:, :1]))
    sigma2 = torch.log1p(1.0 / alpha)
    return alpha.log() - 0.5 * sigma2, alpha, sigma2


class OnlineSKIClassifier(StreamingSKIWrapper):
    def __init__(self, stem, init_x, init_y, alpha_eps, lr, grid_size, grid_bound, **kwargs):
        super().__init__()
        self.alpha_eps = alpha_eps
        self._target_batch_shape = torch.Size([2])
        stem = stem.to(init_x.device)
        feats = stem(init_x).detach()
        gp_targets, noise = self._encode(init_y)
        gp = FixedNoiseOnlineSKIGP(feats, gp_targets.to(feats.dtype), noise.to(feats.dtype),
                                   grid_bounds=torch.tensor([[-grid_bound, grid_bound]] * stem.output_dim),
                                   grid_size=[grid_size] * stem.output_dim)
        self._setup(stem, gp, lr, init_x)

    # ----- hooks of the streaming protocol
    def _encode(self, targets):
        gp_targets, _, sigma2 = dirichlet_transform(targets, self.alpha_eps)
        return gp_targets, sigma2

    def _transform_targets(self, targets, alpha_eps):
        return dirichlet_transform(targets, alpha_eps)

    def _partial_mll_targets(self, gp_targets, noise):
        return (gp_targets / noise).transpose(-1, -2)            # y / sigma2_i per output

    # ----- prediction / training
    def predict(self, inputs):
        self.eval()
        with settings.skip_posterior_variances(True):
            return self(inputs).mean.argmax(0)

    def set_train_data(self, inputs, targets, noise):
        self.gp.set_train_data(inputs.detach(), targets, noise)

    def fit(self, inputs, targets, num_epochs, test_dataset=None):
        def after_epoch():
            acc = float("nan")
            if test_dataset is not None:
                test_x, test_y = test_dataset[:]
                acc = float(self.predict(test_x).eq(test_y.reshape(-1)).float().mean())
            return {"test_acc": acc}

        return self._fit_loop(inputs, targets, num_epochs, after_epoch)
