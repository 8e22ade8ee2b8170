"""How many steps of a 3droad-sized pass leave the one-call streaming path (generic three-call fallback), and what they cost."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = 4096
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, kind)
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    n = 100
    Xr, yr = bench.synth_stream(q * n, 3, 1000, dev, dt, kind)
    orig = model._stream_fast_state
    misses = []
    def wrapped(X, Y, _i=[0]):
        st = orig(X, Y)
        if st is None: misses.append(_i[0])
        _i[0] += 1
        return st
    model._stream_fast_state = wrapped
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    model._finish_pending()
    print(kind, "steps off the one-call path:", misses)
    print("step ms (synchronised each step): median %.3f, the off-path ones %s, slowest 5 %s" % (sorted(ts)[n // 2], [round(ts[i], 2) for i in misses], [(i, round(t, 2)) for t, i in sorted(((t, i) for i, t in enumerate(ts)), reverse=True)[:5]]))
