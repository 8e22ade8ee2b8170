"""wiski_potrf / wiski_potrf_inverse at n = 128 .. 480, fp64 and fp32 (the one-workgroup path of dense_small.h; WISKI_POTRF_SMALL=0 for
the blocked path)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops
def bench(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.float64, torch.float32):
    for n in (128, 256, 327, 427, 480):
        R = torch.randn(n, n, dtype=torch.float64, device="cuda")
        A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64, device="cuda")).to(dt)
        buf = A.clone()
        def f_inv():
            buf.copy_(A); grid_ops.potrf_inverse_(buf)
        def f_po():
            buf.copy_(A); grid_ops.potrf_(buf)
        def f_cp():
            buf.copy_(A)
        print(dt, n, "potrf+inv %.1f us  potrf %.1f us  (copy %.1f us)" % (bench(f_inv), bench(f_po), bench(f_cp)), flush=True)
