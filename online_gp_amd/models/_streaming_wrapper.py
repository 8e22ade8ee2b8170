"""Shared host-side machinery of the two streaming wrappers (regression / Dirichlet classification).

The reference keeps two near-identical classes (online_gp/models/online_ski_regression.py,
online_ski_classifier.py).  Here the streaming protocol lives once, in :class:`StreamingSKIWrapper`, and the
two public classes only say how labels become GP targets (``_encode``) and how a posterior becomes a
prediction.  The protocol, from SURVEY.md 8(b) / 3.1:

``update(x, y)``   1. one Adam step of the stem on the Sherman-Morrison partial MLL of the new batch,
                   2. one Adam step of the GP hyper-parameters on the Woodbury MLL of the data seen so far,
                   3. absorb the batch into the sufficient statistics (in place),
                   4. let BatchNorm layers of the stem see the new inputs plus a replay sample.
``fit(x, y, E)``   E full-batch epochs of joint training with cosine-annealed learning rates; the statistics are
                   rebuilt from the current features after every step.
"""
import math

import torch

from .. import settings
from ..mlls import BatchedWoodburyMarginalLogLikelihood, mll_feature_surrogate, sm_partial_mll

_LR_FLOOR = 1e-4          # eta_min of the cosine schedules
_REPLAY = 1024            # replay sample size of the BatchNorm refresh
EVAL_CHUNK = 1024         # evaluate() batch size


def _cosine_lr(base, floor, step, total):
    """Closed form of torch's CosineAnnealingLR(T_max=total, eta_min=floor) after `step` steps."""
    return floor + 0.5 * (base - floor) * (1.0 + math.cos(math.pi * step / total))


class _NoOptimizer:
    """What a parameter-free stem gets instead of Adam: the optimiser surface the loops below use, doing nothing."""

    param_groups = ()

    def zero_grad(self, set_to_none=True):
        pass

    def step(self):
        pass


class _ReplayBuffer:
    """Raw inputs seen so far (only kept when the stem has sub-modules that carry running statistics)."""

    def __init__(self, first):
        self._chunks = [first]
        self._flat = None

    def append(self, x):
        self._chunks.append(x)
        self._flat = None

    def __len__(self):
        return sum(c.shape[0] for c in self._chunks)

    def sample(self, n):
        if self._flat is None:
            self._flat = torch.cat(self._chunks)
            self._chunks = [self._flat]
        idx = torch.randint(0, self._flat.shape[0], (n,), device=self._flat.device)
        return self._flat[idx]


class StreamingSKIWrapper(torch.nn.Module):
    """stem -> FixedNoiseOnlineSKIGP, two Adam optimisers, the Woodbury MLL.  Subclasses provide ``_encode``."""

    def _setup(self, stem, gp, lr, init_x):
        self.stem = stem.to(init_x.device)
        self.gp = gp
        gp.__dict__["_stream_owner"] = True          # the per-batch loop may take the spectral pipeline on small grids too (settings.spectral_dense_regime)
        self.mll = BatchedWoodburyMarginalLogLikelihood(gp.likelihood, gp)
        self._make_optimizers(lr, lr)
        self._replay = _ReplayBuffer(init_x)

    # ------------------------------------------------------------------ hooks
    def _encode(self, targets):
        """labels -> (GP targets [n, out], fixed noise [n, out] or None for unit noise)"""
        raise NotImplementedError

    def _partial_mll_targets(self, gp_targets, noise):
        """what sm_partial_mll sees as the new responses, laid out [out, n]"""
        raise NotImplementedError

    # ----------------------------------------------------------- small helpers
    def _make_optimizers(self, gp_lr, stem_lr):
        # same Adam; `fused` runs the update of all parameters in one launch instead of ~15 (the handful of GP
        # hyper-parameters make an Adam step pure launch latency)
        def adam(params, lr):
            params = list(params)
            fused = bool(params) and all(p.is_cuda and p.is_floating_point() for p in params)
            try:
                if fused and capturable:              # step count on the device: the update can be recorded into a graph (_graphed_step.py)
                    return torch.optim.Adam(params, lr=lr, fused=True, capturable=True)
                return torch.optim.Adam(params, lr=lr, fused=True) if fused else torch.optim.Adam(params, lr=lr)
            except (RuntimeError, TypeError):
                return torch.optim.Adam(params, lr=lr)

        capturable = True
        self.gp_optimizer = adam(self.gp.parameters(), gp_lr)
        capturable = False
        stem_params = [p for p in self.stem.parameters() if p.requires_grad]
        self.stem_optimizer = adam(stem_params, stem_lr) if stem_params else _NoOptimizer()     # (Identity: nothing to learn)
        self.__dict__["_graphed"] = None              # a captured hyper step belongs to its optimiser

    def set_lr(self, gp_lr, stem_lr=None, bn_mom=None):
        self._make_optimizers(gp_lr, gp_lr if stem_lr is None else stem_lr)
        if bn_mom is not None:
            for layer in self.stem.modules():
                if isinstance(layer, torch.nn.BatchNorm1d):
                    layer.momentum = bn_mom

    def _stem_has_modules(self):
        """True if the stem carries running statistics (batch normalisation) that must keep seeing inputs."""
        return any(isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) for mod in self.stem.modules())

    def _as_rows(self, inputs):
        return inputs.reshape(-1, self.stem.input_dim)

    def forward(self, inputs):
        return self.gp(self.stem(self._as_rows(inputs)))

    def _rebuild_statistics(self, features, labels):
        gp_targets, noise = self._encode(labels)
        dt = features.dtype
        if noise is None:
            noise = torch.ones_like(gp_targets)
        self.gp.set_train_data(features.detach(), gp_targets.to(dt), noise.to(dt))
        self.gp.zero_grad()

    # ------------------------------------------------------------------ update
    def update(self, inputs, targets, update_stem=True, update_gp=True):
        inputs = self._as_rows(inputs)
        gp_targets, noise = self._encode(targets)
        stem_loss = self._stem_step(inputs, gp_targets, noise) if update_stem else 0.0
        gp_loss = self._hyper_step() if update_gp else 0.0
        with torch.no_grad():
            feats = self.stem(inputs)
            dt = feats.dtype
            self.gp.condition_on_observations(feats, gp_targets.to(dt), None if noise is None else noise.to(dt), inplace=True)
            if self._stem_has_modules():
                self._replay.append(inputs)
                self.stem.train()
                if update_stem:                      # running statistics see the new points and a replay sample
                    self.stem(torch.cat([inputs, self._replay.sample(_REPLAY)]))
        self._ensure_eval()
        if torch.is_tensor(gp_loss):                 # the captured step's loss, read only now: the absorb above was queued behind the
            gs = self.__dict__.get("_graphed")       # graph without waiting for it (one GPU idle gap less per step)
            gp_loss = gs.read_loss() if gs is not None else float(gp_loss)
        return stem_loss, gp_loss

    def _ensure_eval(self):
        """``self.eval()`` unless everything already is in eval mode (the module-tree walk costs ~70 us and the streaming loop asks
        three times per step)."""
        if self.training or self.gp.training or self.mll.training or self.stem.training:
            self.eval()

    def _hyper_step(self):
        """One Adam step on -MLL of the statistics absorbed so far (the reference scores the data seen *before*
        the new batch: BWM ignores its arguments and reads the kernel cache)."""
        gs = self.__dict__.get("_graphed")
        if gs is None:
            from ._graphed_step import GraphedHyperStep

            gs = self.__dict__["_graphed"] = GraphedHyperStep(self)
        loss = gs.step(lazy=True)                    # forward + backward + Adam as one captured graph where that applies
        if loss is not None:
            return loss
        opt = self.gp_optimizer
        opt.zero_grad()
        # (the reference toggles gp.train() / mll.train() / gp.eval() around this, OSR:135-147; the Woodbury MLL here reads
        # only the kernel cache and no module looks at its `training` flag on the way, so the four module-tree walks --
        # ~0.4 ms of a ~2 ms step -- are left out)
        with settings.skip_logdet_forward(True):
            loss = -self.mll(None, None).sum()
        loss.backward()
        opt.step()
        self.gp.zero_grad()
        return float(loss.detach())

    def _stem_step(self, inputs, gp_targets, noise):
        if isinstance(self.stem_optimizer, _NoOptimizer):    # parameter-free stem: nothing to differentiate
            return 0
        self.stem.eval()                             # deterministic features while differentiating
        feats = self.stem(inputs)
        if not feats.requires_grad:                  # parameter-free stem
            return 0
        opt = self.stem_optimizer
        opt.zero_grad()
        new_y = self._partial_mll_targets(gp_targets, noise)
        loss = -sm_partial_mll(self.gp, feats, new_y.to(feats.dtype), self.gp.num_data).sum()
        loss.backward()
        opt.step()
        return float(loss.detach())

    # --------------------------------------------------------------------- fit
    def _fit_loop(self, inputs, labels, num_epochs, after_epoch):
        base_gp = [g["lr"] for g in self.gp_optimizer.param_groups]
        base_stem = [g["lr"] for g in self.stem_optimizer.param_groups]
        records = []
        feats = self.stem(inputs)
        self._rebuild_statistics(feats, labels)
        for epoch in range(1, num_epochs + 1):
            self.train()
            self.mll.train()
            self.gp_optimizer.zero_grad()
            self.stem_optimizer.zero_grad()
            loss = -self.mll(None, None).sum()
            if feats.requires_grad:
                loss = loss + self._feature_loss(feats, labels)
            loss.backward()
            self.stem_optimizer.step()
            self.gp_optimizer.step()
            for groups, base in ((self.gp_optimizer.param_groups, base_gp), (self.stem_optimizer.param_groups, base_stem)):
                for grp, b in zip(groups, base):
                    grp["lr"] = _cosine_lr(b, _LR_FLOOR, epoch, num_epochs)
            feats = self.stem(inputs)
            self._rebuild_statistics(feats, labels)
            rec = {"epoch": epoch, "train_loss": float(loss.detach())}
            rec.update(after_epoch())
            records.append(rec)
        with settings.detach_interp_coeff(True):
            self._rebuild_statistics(self.stem(inputs), labels)
        self.eval()
        return records

    def _feature_loss(self, feats, labels):
        """A scalar whose gradient w.r.t. `feats` is d(-MLL)/d features (joint stem + GP training, OSR:80-112): exact in the
        dense regime, with a Hutchinson estimate of the log-determinant part beyond it (mlls/feature_gradient.py)."""
        gp_targets, noise = self._encode(labels)
        return mll_feature_surrogate(self.gp, feats, gp_targets.to(feats.dtype), None if noise is None else noise.to(feats.dtype))
