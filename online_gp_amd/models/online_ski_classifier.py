"""OnlineSKIClassifier -- two-class Dirichlet-transformed WISKI classifier (host-side mirror of
the reference's online_gp/models/online_ski_classifier.py:14-145 and
gp_dirichlet_classification.py:5-45: same constructor, ``fit/update/predict/set_train_data/set_lr``).

Labels y in {0, 1} become two regression outputs with heteroscedastic fixed noise
(gp_dirichlet_classification.py:15-21): alpha = alpha_eps + onehot(y),
sigma2_i = log(1/alpha + 1), targets = log(alpha) - sigma2_i / 2; the predicted class is the argmax
of the two posterior means.  It exercises the 2-output batch layout and non-unit noise of the GP core."""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from .. import settings
from ..mlls import BatchedWoodburyMarginalLogLikelihood, sm_partial_mll
from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP


def dirichlet_transform(targets, alpha_eps, num_classes=2):
    targets = targets.reshape(-1).long()
    alpha = torch.full((targets.shape[0], num_classes), float(alpha_eps), device=targets.device)
    alpha[torch.arange(targets.shape[0], device=targets.device), targets] += 1.0
    sigma2 = torch.log(1.0 / alpha + 1.0)
    return alpha.log() - 0.5 * sigma2, alpha, sigma2


class OnlineSKIClassifier(torch.nn.Module):
    def __init__(self, stem, init_x, init_y, alpha_eps, lr, grid_size, grid_bound, **kwargs):
        super().__init__()
        self.stem = stem.to(init_x.device)
        self.alpha_eps = alpha_eps
        ty, _, s2 = dirichlet_transform(init_y, alpha_eps)
        features = self.stem(init_x).detach()
        dt = features.dtype
        self.gp = FixedNoiseOnlineSKIGP(features, ty.to(dt), s2.to(dt), grid_bounds=torch.tensor([[-grid_bound, grid_bound]] * stem.output_dim),
                                        grid_size=[grid_size] * stem.output_dim)
        self.mll = BatchedWoodburyMarginalLogLikelihood(self.gp.likelihood, self.gp)
        self.gp_optimizer = torch.optim.Adam(self.gp.parameters(), lr=lr)
        self.stem_optimizer = torch.optim.Adam(self.stem.parameters(), lr=lr)
        self._target_batch_shape = torch.Size([2])
        self._raw_inputs = [init_x]

    def _transform_targets(self, targets, alpha_eps):
        return dirichlet_transform(targets, alpha_eps)

    def forward(self, inputs):
        inputs = inputs.view(-1, self.stem.input_dim)
        return self.gp(self.stem(inputs))

    def predict(self, inputs):
        self.eval()
        with settings.skip_posterior_variances(True):
            return self(inputs).mean.argmax(0)                   # gp_dirichlet_classification.py:23-27

    def set_train_data(self, inputs, targets, noise):
        self.gp.set_train_data(inputs.detach(), targets, noise)

    def _refresh_features(self, inputs, targets):
        features = self.stem(inputs)
        ty, _, s2 = dirichlet_transform(targets, self.alpha_eps)
        self.set_train_data(features, ty.to(features.dtype), s2.to(features.dtype))
        self.gp.zero_grad()
        return features

    def fit(self, inputs, targets, num_epochs, test_dataset=None):
        records = []
        gp_sched = CosineAnnealingLR(self.gp_optimizer, num_epochs, 1e-4)
        stem_sched = CosineAnnealingLR(self.stem_optimizer, num_epochs, 1e-4)
        features = self._refresh_features(inputs, targets)
        for epoch in range(num_epochs):
            self.train()
            self.gp_optimizer.zero_grad()
            self.stem_optimizer.zero_grad()
            loss = -self.mll(self.gp(features), targets).sum()
            loss.backward()
            self.gp_optimizer.step()
            self.stem_optimizer.step()
            gp_sched.step()
            stem_sched.step()
            features = self._refresh_features(inputs, targets)
            test_acc = float("NaN")
            if test_dataset is not None:
                test_x, test_y = test_dataset[:]
                test_acc = self.predict(test_x).eq(test_y).float().mean().item()
            records.append({"train_loss": loss.item(), "test_acc": test_acc, "epoch": epoch + 1})
        self._refresh_features(inputs, targets)
        self.eval()
        return records

    def update(self, inputs, targets, update_stem=True, update_gp=True):
        inputs = inputs.view(-1, self.stem.input_dim)
        ty, _, noise = dirichlet_transform(targets.view(-1), self.alpha_eps)
        stem_loss = self._update_stem(inputs, ty, noise) if update_stem else 0.0
        gp_loss = self._update_gp(inputs, ty) if update_gp else 0.0
        with torch.no_grad():
            features = self.stem(inputs)
            self.gp.condition_on_observations(features, ty.to(features.dtype), noise.to(features.dtype), inplace=True)
            if any(True for _ in self.stem.modules()):
                self._raw_inputs = [torch.cat([*self._raw_inputs, inputs])]
        self.eval()
        return stem_loss, gp_loss

    def _update_gp(self, inputs, targets):
        self.gp_optimizer.zero_grad()
        self.gp.train()
        with settings.skip_logdet_forward(True):
            loss = -self.mll(self.gp(self.stem(inputs).detach()), targets).sum()
        loss.backward()
        self.gp_optimizer.step()
        self.gp.zero_grad()
        self.gp.eval()
        return loss.item()

    def _update_stem(self, inputs, targets, noise):
        self.stem_optimizer.zero_grad()
        new_features = self.stem(inputs)
        if new_features.requires_grad is False:
            return 0
        new_y = (targets / noise).t()                              # online_ski_classifier.py:117 (y / sigma2_i per output)
        loss = -sm_partial_mll(self.gp, new_features, new_y, self.gp.num_data).sum()
        loss.backward()
        self.stem_optimizer.step()
        return loss.item()

    def set_lr(self, gp_lr, stem_lr=None, bn_mom=None):
        stem_lr = gp_lr if stem_lr is None else stem_lr
        self.gp_optimizer = torch.optim.Adam(self.gp.parameters(), lr=gp_lr)
        self.stem_optimizer = torch.optim.Adam(self.stem.parameters(), lr=stem_lr)
        if bn_mom is not None:
            for mod in self.stem.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.momentum = bn_mom
