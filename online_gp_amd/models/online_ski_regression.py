"""OnlineSKIRegression -- stem + WISKI GP behind the streaming-regression surface the reference's drivers use
(counterpart of online_gp/models/online_ski_regression.py:16-197, experiments/regression.py:38-138):

    OnlineSKIRegression(stem, init_x, init_y, lr, grid_size, grid_bound, covar_module=None, **kw)
    .fit(x, y, num_epochs, test_dataset=None) -> list of per-epoch dicts
    .update(x, y, update_stem=True, update_gp=True) -> (stem_loss, gp_loss)
    .evaluate(x, y) -> (rmse, nll)        .predict(x) -> (mean [n, out], var [n, out])
    .set_train_data(features, targets)    .set_lr(gp_lr, stem_lr=None, bn_mom=None)    .noise / .stem / .gp

Conventions it shares with the reference: the inducing grid spans +-(grid_bound + 0.1) in every feature dimension, the
per-point noise term is identically 1 with a learnable homoscedastic sigma^2 on top, ``predict`` returns the
observation variance (latent + sigma^2), ``evaluate`` averages per-batch RMSE / Gaussian NLL over batches of 1024.
The streaming protocol itself lives in ``_streaming_wrapper.StreamingSKIWrapper``."""
import torch

from .. import grid_ops, settings
from ._streaming_wrapper import EVAL_CHUNK, StreamingSKIWrapper
from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP

_GRID_MARGIN = 0.1
_LOG_2PI = 1.8378770664093453


class OnlineSKIRegression(StreamingSKIWrapper):
    def __init__(self, stem, init_x, init_y, lr, grid_size, grid_bound, covar_module=None, **kwargs):
        super().__init__()
        if init_y.dim() != 2:
            raise ValueError("targets must carry an explicit output dimension: [n, out]")
        self.target_dim = init_y.shape[-1]
        self._target_batch_shape = torch.Size([]) if self.target_dim == 1 else torch.Size([self.target_dim])
        stem = stem.to(init_x.device)
        feats = stem(init_x).detach()
        half_width = grid_bound + _GRID_MARGIN
        gp = FixedNoiseOnlineSKIGP(
            feats, init_y, torch.ones_like(init_y),
            covar_module=covar_module,
            grid_bounds=torch.tensor([[-half_width, half_width]] * stem.output_dim),
            grid_size=[grid_size] * stem.output_dim,
            learn_additional_noise=True,
        )
        self._setup(stem, gp, lr, init_x)

    # ----- hooks of the streaming protocol
    def _encode(self, targets):
        return targets.reshape(-1, self.target_dim), None      # unit noise: the scatter's unit-weight fast path

    def _partial_mll_targets(self, gp_targets, noise):
        return gp_targets.transpose(-1, -2)

    # ----- prediction
    def predict(self, inputs):
        self._ensure_eval()
        post = self(inputs)
        mean, var = post.mean, post.variance
        if self.target_dim > 1:                                  # the GP lays several outputs out as [out, n]
            mean, var = mean.transpose(-1, -2), var.transpose(-1, -2)
        sigma2 = self.gp.likelihood.second_noise.detach().reshape(1, -1)
        return mean.reshape(-1, self.target_dim), var.reshape(-1, self.target_dim) + sigma2

    def evaluate(self, inputs, targets):
        """(rmse, nll): means over batches of 1024 of the per-batch RMSE and mean Gaussian NLL; one host sync."""
        inputs = self._as_rows(inputs)
        targets = targets.reshape(-1, self.target_dim)
        self._ensure_eval()
        fast = self._evaluate_from_factor(inputs, targets)
        if fast is not None:
            return fast
        per_batch = []
        fused = self.target_dim == 1 and inputs.is_cuda
        # the out-of-grid check of the queries is read together with the metrics (ONE host sync at the end instead of one inside
        # every posterior call plus one for the metrics: each sync leaves the GPU idle until the host has queued the next launch)
        late_check = fused and settings.deferred_bounds_check.off()
        for lo in range(0, inputs.shape[0], EVAL_CHUNK):
            y = targets[lo:lo + EVAL_CHUNK]
            if fused:
                # one output: both metrics of the batch from (mean, variance, sigma2) in one launch (wiski_gaussian_metrics)
                with settings.deferred_bounds_check(True):
                    post = self(inputs[lo:lo + EVAL_CHUNK])
                mu = post.mean.reshape(-1).contiguous()
                s2 = self.gp.likelihood.second_noise.detach().reshape(-1).to(mu.dtype)
                per_batch.append(grid_ops.gaussian_metrics(mu, post.variance.reshape(-1).to(mu.dtype).contiguous(), y.reshape(-1).to(mu.dtype).contiguous(),
                                                           s2))
                continue
            mean, var = self.predict(inputs[lo:lo + EVAL_CHUNK])
            sq = (mean - y) ** 2
            nll = 0.5 * (sq / var + var.log() + _LOG_2PI)
            per_batch.append(torch.stack([sq.mean().sqrt(), nll.mean()]))
        if late_check:
            vals = torch.cat([torch.stack(per_batch).mean(0).double(), self.gp._err.double()]).tolist()
            if int(vals[2]):
                self.gp._raise_out_of_bounds(int(vals[2]))      # as the posterior call would have (gpytorch raises inside it)
            return vals[0], vals[1]
        rmse, nll = torch.stack(per_batch).mean(0).tolist()
        return rmse, nll

    def _evaluate_from_factor(self, inputs, targets):
        """evaluate() of one chunk (<= 1024 points, one output) straight from the spectral factor: one projection launch and ONE launch
        for means, variances and both metrics (wiski_spectral_evaluate; beyond 64 points or rank 512 the MFMA GEMM chol^-1 F^T + one
        launch), one host read.  Same decisions as the posterior call
        (batched_fixed_noise_online_gp._eval_forward): only where the factor serves the mean as well -- the PCG state is not current and
        the factor's mean monitor is green; otherwise None and the general path runs."""
        gp = self.gp
        n = inputs.shape[0]
        if not (self.target_dim == 1 and inputs.is_cuda and 0 < n <= EVAL_CHUNK and gp.has_learnable_noise and settings.skip_posterior_variances.off()
                and settings.fused_evaluate.on()):
            return None
        ms = gp._mean_state
        if gp._memo.get("prediction_cache") is not None or (ms is not None and ms.get("ver") == gp._hyper_version()):
            return None
        sp = gp._spectral_state(0)
        if sp is None:
            return None
        fac, st, tcol64 = sp
        if fac.measure_due:                                # (the mean monitor wants a measurement: the general path takes it)
            return None
        if not fac.mean_ok or "t" not in st:
            return None
        dt = gp._dtype
        Xf = self.stem(inputs).detach().to(gp._device, dt).reshape(-1, gp._grid.d).contiguous()
        q = fac.query(st, Xf, tcol64)
        ws = fac.__dict__.get("_eval_ws")
        if ws is None:
            ws = fac._eval_ws = torch.zeros(200, dtype=torch.float64, device=gp._device)
        s2c = gp.__dict__.get("_s2_dev")                  # left on the device by the captured hyper step (models/_graphed_step.py)
        s2 = s2c[1] if s2c is not None and s2c[0] == gp._hyper_version() and s2c[1].dtype == dt else gp.likelihood.second_noise.detach().reshape(-1).to(dt)
        out = grid_ops.spectral_evaluate(q.Fs, q.prior, st["Linv"], st["t"], st["kscale"], s2, targets.reshape(-1).to(dt).contiguous(), gp._err, ws)
        fac.mean_monitor(st, q, gp._kernel_cache["interpolation_cache"][0, :, 0], tcol64, out[3])
        if not fac.mean_ok:                               # (a verdict read just now turned the factor's mean off)
            return None
        gs = self.__dict__.get("_graphed")
        if gs is not None:
            gs.prepare()                                  # (the hyper step of this batch: checked and staged before the host waits below)
        vals = out.tolist()
        if int(vals[2]):
            if settings.deferred_bounds_check.off():
                gp._raise_out_of_bounds(int(vals[2]))
        else:
            # the flag was clean after everything absorbed so far: the hyper step that follows need not read it again
            gp.__dict__["_bounds_clean_at"] = (gp.num_data, fac.data_version)
        return vals[0], vals[1]

    # ----- batch training
    def fit(self, inputs, targets, num_epochs, test_dataset=None):
        def after_epoch():
            rmse = nll = float("nan")
            if test_dataset is not None:
                rmse, nll = self.evaluate(*test_dataset[:])
            return {"test_rmse": rmse, "test_nll": nll,
                    "noise": float(self.gp.likelihood.second_noise_covar.noise.detach().mean())}

        return self._fit_loop(inputs, targets, num_epochs, after_epoch)

    def set_train_data(self, inputs, targets):
        """Rebuild the statistics from features (not raw inputs) and targets."""
        self.gp.set_train_data(inputs.detach(), targets, torch.ones_like(targets))

    @property
    def noise(self):
        return self.gp.likelihood.noise
