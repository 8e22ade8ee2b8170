"""Which reference cycles does a streaming step leave behind?  (they keep GPU tensors alive until the cyclic GC runs)"""
import os, sys, gc, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = 4096
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    Xr, yr = bench.synth_stream(q * 60, 3, 7, dev, dt, "uniform")
    for i in range(10): model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    model._finish_pending(); gc.collect(); gc.disable()
    gc.set_debug(gc.DEBUG_SAVEALL)
    for i in range(10, 50): model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    model._finish_pending()
    n = gc.collect()
    print("unreachable objects after 40 steps:", n)
    c = collections.Counter(type(o).__name__ for o in gc.garbage)
    print(c.most_common(12))
    for o in gc.garbage:
        if isinstance(o, torch.Tensor): print("tensor", tuple(o.shape), o.dtype, o.device); break
    fr = [o for o in gc.garbage if type(o).__name__ in ("function", "cell", "frame", "dict")][:6]
    for o in fr: print(type(o).__name__, getattr(o, "__qualname__", ""), str(o)[:160])
