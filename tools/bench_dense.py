"""Dense small-grid building blocks: achieved TFLOP/s of the MFMA GEMM, Cholesky and TRSM."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from online_gp_amd import grid_ops
dev = 'cuda'
def tm(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
for dt, name in ((torch.float32, 'f32'), (torch.float64, 'f64')):
    for n in (1024, 1536, 2048, 4096):
        A = torch.randn(n, n, device=dev, dtype=dt); B = torch.randn(n, n, device=dev, dtype=dt)
        t = tm(lambda: grid_ops.gemm(A, B, ta=True))
        S = (A @ A.t() / n + torch.eye(n, device=dev, dtype=dt)).contiguous()
        tc = tm(lambda: grid_ops.potrf_(S.clone()), 3)
        L = S.clone(); grid_ops.potrf_(L)
        ts = tm(lambda: grid_ops.trsm_(L, B.clone()), 3)
        print(json.dumps({"dtype": name, "n": n, "gemm_ms": t * 1e3, "gemm_tflops": 2 * n ** 3 / t / 1e12, "potrf_ms": tc * 1e3,
                          "potrf_tflops": n ** 3 / 3 / tc / 1e12, "trsm_ms": ts * 1e3, "trsm_tflops": n ** 3 / ts / 1e12}))
