"""Driver loops around the hot path -- counterparts of the reference's timed experiment loops, so that the
recorded tables (step_time / regret, the three BO timers, the active-learning curve) can be regenerated on this
library:

  online_regression   experiments/regression.py:41-81      evaluate -> update per incoming batch, `online_metrics` rows
  bayesopt            experiments/bayesopt/bayesopt.py:180-236   re-initialise from the kernel cache + fit / acquisition /
                                                            condition, one row of timers per step
  qnipv_active_learning  experiments/active_learning/qnIPV_experiment.py:150-215   look-ahead variance reduction through
                                                            batched fantasies

Hydra, datasets, loggers and BoTorch's optimisers are out of scope: callers pass tensors and get rows
back; ``write_csv`` stores them.  The acquisition optimiser is a random search over candidate sets (BoTorch's
``optimize_acqf`` is not in this image).
"""
import csv
import math
import time

import torch

from . import settings
from .mlls import BatchedWoodburyMarginalLogLikelihood


def write_csv(rows, path):
    if not rows:
        return
    cols = list(rows[0].keys())
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for r in rows:
            w.writerow(r)


def settle_interpreter_heap():
    """Collect, then freeze the interpreter's long-lived object graph (torch + numpy: ~10^6 objects).  A streaming loop of
    0.1 - 0.3 ms steps otherwise meets a FULL cyclic-GC pass every few hundred steps, and that pass costs ~80 ms
    (tools/stall_probe.py) -- several hundred steps' worth.  After the freeze the collector only scans objects created since.
    Call once after the models are built; harmless to call again (it unfreezes, collects and re-freezes)."""
    import gc

    gc.unfreeze()
    gc.collect()
    gc.freeze()


def _sync(t):
    if torch.is_tensor(t) and t.is_cuda:
        torch.cuda.synchronize(t.device)


# ------------------------------------------------------------------------------------------------ streaming regression
def online_regression(online_model, train_x, train_y, test_x, test_y, batch_size=1, logging_freq=1, update_stem=True,
                      batch_model=None, max_steps=None):
    """For every incoming batch: evaluate (rmse, nll of the batch *before* it is absorbed), then update; `step_time`
    covers exactly those two calls.  Every `logging_freq` steps a row with the cumulative online metrics, the regret
    against `batch_model` (a model fitted on the whole stream, optional), test metrics, the learned noise and the step time
    is appended.  Returns the rows of the reference's `online_metrics` table."""
    rows = []
    on_rmse = on_nll = b_rmse_sum = b_nll_sum = 0.0
    settle_interpreter_heap()
    n = train_x.shape[-2]
    steps = n // batch_size if max_steps is None else min(max_steps, n // batch_size)
    for t in range(steps):
        x = train_x[t * batch_size:(t + 1) * batch_size]
        y = train_y[t * batch_size:(t + 1) * batch_size]
        _sync(x)
        t0 = time.perf_counter()
        with settings.detach_interp_coeff(True):
            o_rmse, o_nll = online_model.evaluate(x, y)
        stem_loss, gp_loss = online_model.update(x, y, update_stem=update_stem)
        _sync(x)
        step_time = time.perf_counter() - t0
        on_rmse += o_rmse
        on_nll += o_nll
        if batch_model is not None:
            with torch.no_grad():
                br, bn = batch_model.evaluate(x, y)
            b_rmse_sum += br
            b_nll_sum += bn
        if t % logging_freq == logging_freq - 1:
            rmse, nll = online_model.evaluate(test_x, test_y)
            rows.append({"step": (t + 1) * batch_size, "stem_loss": float(stem_loss), "gp_loss": float(gp_loss),
                         "batch_rmse": b_rmse_sum, "batch_nll": b_nll_sum, "online_rmse": on_rmse, "online_nll": on_nll,
                         "regret": on_rmse - b_rmse_sum, "test_rmse": rmse, "test_nll": nll,
                         "noise": float(online_model.noise.detach().mean()), "step_time": step_time})
    return rows


# ---------------------------------------------------------------------------------------------------------- BayesOpt
def fit_mll(model, num_iter=30, lr=0.1):
    """Maximise the Woodbury MLL (+ registered priors) over the model's hyper-parameters with Adam; the reference calls
    BoTorch's L-BFGS-B wrapper here.  Returns the final MLL value."""
    mll = BatchedWoodburyMarginalLogLikelihood(model.likelihood, model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    model.train()
    val = None
    for _ in range(num_iter):
        opt.zero_grad()
        val = mll(None, None).sum()
        (-val).backward()
        opt.step()
        model.zero_grad()
    model.eval()
    return float(val.detach()) if val is not None else float("nan")


def ucb_random_search(model, q, d, num_candidates=512, beta=2.0, generator=None, device=None, dtype=None):
    """argmax over random candidate sets X [num_candidates, q, d] in the unit cube of mean_q(mu + sqrt(beta) sigma): a
    derivative-free stand-in for optimize_acqf(qUCB)."""
    device = device if device is not None else model._device
    dtype = dtype if dtype is not None else model._dtype
    cand = torch.rand((num_candidates, q, d), generator=generator, device="cpu").to(device, dtype)
    with torch.no_grad():
        post = model.posterior(cand)
        score = (post.mean[..., 0] + math.sqrt(beta) * post.variance[..., 0].clamp_min(0).sqrt()).max(dim=-1).values
    return cand[int(score.argmax())]


def bayesopt(test_function, bounds, make_model, init_x, init_y, num_steps, batch_size=3, noise=None, fit_iters=30,
             num_candidates=512, beta=2.0, seed=0, on_step=None):
    """The reference's BO loop with its three timers.  Per step:
        t0  re-initialise the model from the previous model's kernel cache (``make_model(train_x, train_y, old_model)``,
            bayesopt.py:86-96) and refit the hyper-parameters on the MLL,
        t1  optimise the acquisition over q-batches in the unit cube, evaluate ``test_function`` on the un-normalised points,
        t2  ``condition_on_observations`` (functional: returns the model of the next step).
    `bounds` [d, 2] are the test function's bounds; inputs handed to the model live in the unit cube (and the grid covers
    the raw bounds: the reference's quirk).  Targets are standardised with the initial statistics.  Returns
    (rows, train_x, train_y) with rows = dict(fit_time, acqf_time, condition_time, total, max_achieved)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = bounds.shape[0]
    lo, hi = bounds[:, 0], bounds[:, 1]
    mean, std = init_y.mean(), init_y.std().clamp_min(1e-12)
    train_x, train_y = init_x, (init_y - mean) / std
    model = None
    rows = []
    for step in range(num_steps):
        _sync(train_x); t = time.perf_counter()
        model = make_model(train_x, train_y, model)
        fit_mll(model, fit_iters)
        _sync(train_x); t0 = time.perf_counter() - t
        t = time.perf_counter()
        new_x = ucb_random_search(model, batch_size, d, num_candidates, beta, g)
        raw = test_function(lo.to(new_x) + (hi - lo).to(new_x) * new_x)
        new_y = ((raw.reshape(-1, 1) - mean) / std).to(train_y)
        train_x, train_y = torch.cat([train_x, new_x]), torch.cat([train_y, new_y])
        _sync(train_x); t1 = time.perf_counter() - t
        t = time.perf_counter()
        kw = {} if noise is None else {"noise": torch.full_like(new_y, float(noise))}
        model = model.condition_on_observations(X=new_x, Y=new_y, **kw)
        _sync(train_x); t2 = time.perf_counter() - t
        rows.append({"step": step, "fit_time": t0, "acqf_time": t1, "condition_time": t2, "total": t0 + t1 + t2,
                     "max_achieved": float(train_y.max() * std + mean)})
        if on_step is not None:
            on_step(step, model, train_x, train_y)
    return rows, train_x, train_y, model


# ------------------------------------------------------------------------------------------- qNIPV active learning
def qnipv_select(model, candidate_sets, mc_points, sampler):
    """Negative integrated posterior variance of every candidate set [b, q, d] (BoTorch's qNegIntegratedPosteriorVariance):
    fantasize on the set, average the fantasy posterior variance over `mc_points` and over the fantasies.  Returns the scores [b]
    (higher = better)."""
    with torch.no_grad():
        fm = model.fantasize(candidate_sets, sampler, observation_noise=True)
        var = fm.posterior(mc_points).variance                  # [num_fantasies, b, N, 1]
        return -var.mean(dim=-2).squeeze(-1).mean(dim=0)


def qnipv_active_learning(model, pool_x, observe, mc_points, batch_size=6, num_steps=10, num_candidate_sets=32, num_fantasies=4,
                          noise_fn=None, seed=0, on_step=None):
    """Per step: draw `num_candidate_sets` random q-subsets of the remaining pool, score them by qNIPV through batched
    fantasies, query ``observe`` at the winner, condition the model (functional).  Returns (rows, model, chosen indices);
    rows carry the selection / conditioning times and the integrated posterior variance over `mc_points` after the step."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    class _Sampler:
        def __init__(self, n):
            self.sample_shape = torch.Size([n])

        def __call__(self, posterior):
            return posterior.rsample(self.sample_shape)

    avail = torch.ones(pool_x.shape[0], dtype=torch.bool)
    rows, chosen = [], []
    for step in range(num_steps):
        _sync(pool_x); t = time.perf_counter()
        idx_pool = avail.nonzero()[:, 0]
        sets = torch.stack([idx_pool[torch.randperm(idx_pool.numel(), generator=g)[:batch_size]] for _ in range(num_candidate_sets)])
        cand = pool_x[sets.to(pool_x.device)]
        scores = qnipv_select(model, cand, mc_points, _Sampler(num_fantasies))
        best = int(scores.argmax())
        _sync(pool_x); t_sel = time.perf_counter() - t
        pick = sets[best]
        avail[pick] = False
        chosen.append(pick)
        x_new = pool_x[pick.to(pool_x.device)]
        y_new = observe(x_new).reshape(-1, 1)
        t = time.perf_counter()
        kw = {} if noise_fn is None else {"noise": noise_fn(x_new).reshape(-1, 1)}
        model = model.condition_on_observations(X=x_new, Y=y_new.to(x_new), **kw)
        with torch.no_grad():
            ipv = float(model.posterior(mc_points).variance.mean())
        _sync(pool_x); t_cond = time.perf_counter() - t
        rows.append({"step": step, "select_time": t_sel, "condition_time": t_cond, "integrated_posterior_variance": ipv,
                     "qnipv_best": float(scores[best]), "num_data": int(model.num_data)})
        if on_step is not None:
            on_step(step, model)
    return rows, model, torch.cat(chosen)
