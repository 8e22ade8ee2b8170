"""wiski_gemm at the dense regime's mid sizes (n = 600 .. 1400), all four transposes: us and TFLOP/s.  WISKI_GEMM32_MAX_TILES moves the
switch between the 32 x 32 and the 64 x 64 tile kernel (in 64 x 64 tiles of the output)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.float64, torch.float32):
    for n in (400, 600, 800, 1000, 1200, 1400, 1600, 2048):
        A = torch.randn(n, n, device="cuda", dtype=dt); B = torch.randn(n, n, device="cuda", dtype=dt); C = torch.empty_like(A)
        out = []
        for ta, tb in ((False, False), (True, False), (False, True)):
            us = bench(lambda: grid_ops.gemm(A, B, ta=ta, tb=tb, C=C))
            out.append("%s%s %6.1f us %5.1f TF" % ("T" if ta else "N", "T" if tb else "N", us, 2 * n ** 3 / us / 1e6))
        print(str(dt)[6:], n, " | ".join(out), flush=True)
