"""Why the stencil-sharded step needs more CG iterations than the point exchange in the N = 2 self-test (3.1-3.3 against 2.8): the two things
a sharded replica does differently to its two-level block -- the lock-step switch-in (exactly `two_level_lag` steps after a refresh was
launched, so that every rank switches at the same step) and the subsampled Gram accumulation (every world-th point of the gathered batch) --
applied one at a time to the SINGLE-process headline stream (the same 2 x 4096 points per step a two-rank job absorbs)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from online_gp_amd import settings  # noqa: E402
from online_gp_amd.models import FixedNoiseOnlineSKIGP  # noqa: E402


def run(q, steps, lockstep, subsample, lag):
    dev = torch.device("cuda")
    X0, y0 = bench.synth_stream(21743, 3, 0, dev, torch.float32, "clustered")
    Xs, ys = bench.synth_stream(steps * q, 3, 1000, dev, torch.float32, "clustered")
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), \
            settings.two_level_lockstep(lockstep), settings.two_level_subsample(subsample), settings.two_level_lag(lag), torch.no_grad():
        m = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
        m.prediction_cache
        its = []
        for t in range(steps):
            m.stream_step(Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q])
            its.append(m._last_iters[0])
        m._finish_pending()
    tr = m.__dict__.get("_two_level")
    return float(np.mean(its)), float(np.mean(its[-10:])), (tr.block.refreshes if tr is not None and tr.block is not None else 0)


if __name__ == "__main__":
    q, steps = 8192, 48          # the gathered batch of two ranks, as many steps as a 3droad-sized pass holds
    base = settings.two_level_subsample.value()
    print(f"road-like 50^3 stream, {steps} steps of q = {q}; default subsample {base}, default lag {settings.two_level_lag.value()}")
    for name, ls, sub, lag in (("as one process runs it", False, base, 2), ("lock-step switch-in only (lag 2)", True, base, 2), ("lock-step, lag 1", True, base, 1),
                               ("subsample max(default, 2) only", False, max(base, 2), 2), ("both (what a sharded replica of 2 does)", True, max(base, 2), 2)):
        mean, last, nref = run(q, steps, ls, sub, lag)
        print(f"   {name:44s} CG iterations per step: mean {mean:.2f}, last 10 steps {last:.2f}; {nref} block refreshes")
