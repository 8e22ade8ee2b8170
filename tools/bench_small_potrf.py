"""wiski_potrf / wiski_potrf_inverse at n = 128 .. 480, fp64 and fp32 (the one-workgroup path of dense_small.h; WISKI_POTRF_SMALL=0 for
the blocked path); with an argument: those sizes instead (python tools/bench_small_potrf.py 600 1000 1500 2048 -- the two-level blocked
factorisation, WISKI_POTRF_TWO_LEVEL=0 for the 64-wide right-looking loop), plus a triangular solve with n right-hand sides."""
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
    for n in ([int(a) for a in sys.argv[1:]] or (128, 256, 327, 427, 480)):
        R = torch.randn(n, n, dtype=torch.float64, device="cuda")
        A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64, device="cuda")).to(dt)
        buf = A.clone()
        def f_inv():
            buf.copy_(A); grid_ops.potrf_inverse_(buf)
        def f_po():
            buf.copy_(A); grid_ops.potrf_(buf)
        def f_cp():
            buf.copy_(A)
        line = "potrf+inv %.1f us  potrf %.1f us  (copy %.1f us)" % (bench(f_inv), bench(f_po), bench(f_cp))
        if len(sys.argv) > 1:
            L = A.clone(); grid_ops.potrf_(L)
            B = torch.randn(n, n, dtype=dt, device="cuda"); W = B.clone()
            def f_tr():
                W.copy_(B); grid_ops.trsm_(L, W, trans=False)
            def f_tt():
                W.copy_(B); grid_ops.trsm_(L, W, trans=True)
            line += "  trsm n rhs %.1f us  transposed %.1f us" % (bench(f_tr, 20), bench(f_tt, 20))
        print(dt, n, line, flush=True)
