"""Is the streaming loop GPU-bound or host-bound?  Needs a -DWISKI_STEP_TIMING build (WISKI_HIP_SO=...): host time spent waiting
for convergence polls per step, how many polls had already arrived when the host looked, and the wall time per step."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import _hip, settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
gb = torch.tensor([[-1.1, 1.1]] * 3)
lib = _hip.lib()
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), settings.deferred_refresh(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    n = 100
    Xr, yr = bench.synth_stream(q * (n + 10), 3, 7, dev, dt, "uniform")
    for i in range(10): model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
    bench_gc = getattr(bench, "gc_settle", None)
    us, cnt, imm = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_longlong()
    for rep in range(3):
        model._finish_pending(); torch.cuda.synchronize()
        lib.wiski_debug_poll_wait(ctypes.byref(us), ctypes.byref(cnt), ctypes.byref(imm))
        t_call = 0.0
        t0 = time.perf_counter()
        for i in range(10, 10 + n):
            a = time.perf_counter()
            model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
            t_call += time.perf_counter() - a
        model._finish_pending(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e6
        lib.wiski_debug_poll_wait(ctypes.byref(us), ctypes.byref(cnt), ctypes.byref(imm))
        print(f"q={q}: {wall:.1f} us/step wall, {t_call / n * 1e6:.1f} us/step inside stream_step (python + C), "
              f"{us.value / n:.1f} us/step waiting for polls ({cnt.value} waits, {imm.value} already there)")
