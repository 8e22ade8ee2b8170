"""OnlineSKIRegression -- stem + WISKI GP + optimisers (host-side mirror of the
reference's online_gp/models/online_ski_regression.py:16-197: same constructor,
``fit/update/evaluate/predict/set_train_data/set_lr/noise``)."""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from .. import settings
from ..mlls import BatchedWoodburyMarginalLogLikelihood, mll_feature_surrogate, sm_partial_mll
from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP


class OnlineSKIRegression(torch.nn.Module):
    def __init__(self, stem, init_x, init_y, lr, grid_size, grid_bound, covar_module=None, **kwargs):
        super().__init__()
        self.stem = stem.to(init_x.device)
        assert init_y.ndim == 2, "targets must have explicit output dimension"
        if init_y.size(-1) == 1:
            target_batch_shape = []
        else:
            target_batch_shape = torch.Size([init_y.size(-1)])
        features = self.stem(init_x).detach()
        noise_term = torch.ones_like(init_y)
        grid_bound += 1e-1                                   # OSR:26
        self.gp = FixedNoiseOnlineSKIGP(
            features,
            init_y,
            noise_term,
            covar_module=covar_module,
            grid_bounds=torch.tensor([[-grid_bound, grid_bound]] * stem.output_dim),
            grid_size=[grid_size] * stem.output_dim,
            learn_additional_noise=True,
        )
        self.mll = BatchedWoodburyMarginalLogLikelihood(self.gp.likelihood, self.gp)
        self.gp_optimizer = torch.optim.Adam(self.gp.parameters(), lr=lr)
        self.stem_optimizer = torch.optim.Adam(self.stem.parameters(), lr=lr)
        self._target_batch_shape = target_batch_shape
        self.target_dim = init_y.size(-1)
        self._raw_inputs = [init_x]

    def forward(self, inputs):
        inputs = inputs.view(-1, self.stem.input_dim)
        features = self.stem(inputs)
        return self.gp(features)

    def predict(self, inputs):
        self.eval()
        pred_dist = self(inputs)
        mean, var = pred_dist.mean, pred_dist.variance
        if self.target_dim > 1:                               # [out, n] -> [n, out]
            mean, var = mean.t(), var.t()
        pred_mean = mean.reshape(-1, self.target_dim)
        pred_var = var.reshape(-1, self.target_dim)
        pred_var = pred_var + self.gp.likelihood.second_noise.detach().reshape(1, -1)   # OSR:61
        return pred_mean, pred_var

    def evaluate(self, inputs, targets):
        inputs = inputs.view(-1, self.stem.input_dim)
        targets = targets.view(-1, self.target_dim)
        self.eval()
        rmse, nll = 0, 0
        batches = list(zip(inputs.split(1024), targets.split(1024)))     # DataLoader(batch_size=1024), OSR:67-68
        num_batches = len(batches)
        for input_batch, target_batch in batches:
            pred_mean, pred_var = self.predict(input_batch)
            rmse += (pred_mean - target_batch).pow(2).mean().sqrt().item() / num_batches
            diag_dist = torch.distributions.Normal(pred_mean, pred_var.sqrt())
            nll += -diag_dist.log_prob(target_batch).mean().item() / num_batches
        return rmse, nll

    def fit(self, inputs, targets, num_epochs, test_dataset=None):
        records = []
        gp_lr_sched = CosineAnnealingLR(self.gp_optimizer, num_epochs, 1e-4)
        stem_lr_sched = CosineAnnealingLR(self.stem_optimizer, num_epochs, 1e-4)
        features = self._refresh_features(inputs, targets)
        for epoch in range(num_epochs):
            self.train()
            self.mll.train()
            self.stem_optimizer.zero_grad()
            self.gp_optimizer.zero_grad()
            train_dist = self.gp(features)
            loss = -self.mll(train_dist, targets).sum()
            if features.requires_grad and self.gp._use_dense():
                # joint stem + GP training (OSR:80-112): d(-MLL)/d features, written out (mlls/feature_gradient.py)
                loss = loss + mll_feature_surrogate(self.gp, features, targets)
            loss.backward()
            self.stem_optimizer.step()
            self.gp_optimizer.step()
            stem_lr_sched.step()
            gp_lr_sched.step()
            features = self._refresh_features(inputs, targets)

            rmse = nll = float("NaN")
            if test_dataset is not None:
                test_x, test_y = test_dataset[:]
                rmse, nll = self.evaluate(test_x, test_y)
            records.append({"epoch": epoch + 1, "train_loss": loss.item(), "test_rmse": rmse, "test_nll": nll,
                            "noise": self.gp.likelihood.second_noise_covar.noise.mean().item()})

        with settings.detach_interp_coeff(True):
            self._refresh_features(inputs, targets)
        self.eval()
        return records

    def update(self, inputs, targets, update_stem=True, update_gp=True):
        inputs = inputs.view(-1, self.stem.input_dim)
        targets = targets.view(-1, self.target_dim)

        stem_loss = self._update_stem(inputs, targets) if update_stem else 0.0
        gp_loss = self._update_gp(inputs, targets) if update_gp else 0.0

        with torch.no_grad():
            features = self.stem(inputs)
            # noise term == 1 (OSR:122): None selects the unit-noise fast path (no weight tensors)
            self.gp.condition_on_observations(features, targets, None, inplace=True)
            if any(True for _ in self.stem.modules()):
                self._raw_inputs = [torch.cat([*self._raw_inputs, inputs])]
                self.stem.train()
                if update_stem:
                    self._get_features(inputs)
        self.eval()
        return stem_loss, gp_loss

    def _update_gp(self, inputs, targets):
        self.gp_optimizer.zero_grad()
        self.gp.train()
        self.mll.train()
        with settings.skip_logdet_forward(True):
            features = self.stem(inputs)
            train_dist = self.gp(features.detach())
            loss = -self.mll(train_dist, targets).sum()
        loss.backward()
        self.gp_optimizer.step()
        self.gp.zero_grad()
        self.gp.eval()
        return loss.item()

    def _update_stem(self, inputs, targets):
        self.stem_optimizer.zero_grad()
        num_seen = self.gp.num_data
        self.stem.eval()  # deterministic features: BatchNorm in eval mode (OSR:152)
        new_features = self.stem(inputs)
        if new_features.requires_grad is False:               # Identity stem, OSR:154-155
            return 0
        loss = -sm_partial_mll(self.gp, new_features, targets.transpose(-1, -2), num_seen).sum()
        loss.backward()
        self.stem_optimizer.step()
        return loss.item()

    def _get_features(self, inputs):
        # refresh the BatchNorm statistics with the new points plus a replay sample (OSR:164-174)
        inputs = inputs.view(-1, self.stem.input_dim)
        num_seen = self._raw_inputs[0].size(0)
        batch_idxs = torch.randint(0, num_seen, (1024,), device=self._raw_inputs[0].device)
        input_batch = torch.cat([inputs, self._raw_inputs[0][batch_idxs]])
        return self.stem(input_batch)[:inputs.size(0)]

    def _refresh_features(self, inputs, targets):
        features = self.stem(inputs)
        self.set_train_data(features, targets)
        self.gp.zero_grad()
        return features

    def set_train_data(self, inputs, targets):
        noise = torch.ones_like(targets)
        self.gp.set_train_data(inputs.detach(), targets, noise)

    def set_lr(self, gp_lr, stem_lr=None, bn_mom=None):
        stem_lr = gp_lr if stem_lr is None else stem_lr
        self.gp_optimizer = torch.optim.Adam(self.gp.parameters(), lr=gp_lr)
        self.stem_optimizer = torch.optim.Adam(self.stem.parameters(), lr=stem_lr)
        if bn_mom is not None:
            for m in self.stem.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.momentum = bn_mom

    @property
    def noise(self):
        return self.gp.likelihood.noise
