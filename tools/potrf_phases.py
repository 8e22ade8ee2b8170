"""Phase stamps of the one-workgroup Cholesky (dense_small.h) from a -DWISKI_POTRF_TIMING build of the library:
   hipcc ... -DWISKI_POTRF_TIMING -> WISKI_HIP_SO=<that .so> python tools/potrf_phases.py [n] [f32]
Per round: B (wave 0's diagonal step), wait for the other waves (previous trailing tiles + panel load), C (+ factor store), D1; microseconds."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import grid_ops, _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 327
dt = torch.float32 if "f32" in sys.argv else torch.float64
R = torch.randn(n, n, dtype=torch.float64, device="cuda")
A = (R @ R.t() / n + torch.eye(n, dtype=torch.float64, device="cuda")).to(dt)
buf = A.clone()
for _ in range(5):
    buf.copy_(A); (grid_ops.potrf_inverse_(buf) if 'inv' in sys.argv else grid_ops.potrf_(buf))
torch.cuda.synchronize()
st = (ctypes.c_longlong * (17 * 8))()
assert _hip.lib().wiski_potrf_stamps(st) == 0
tot = [0.0] * 4
print("round     B  Xwait  C+st    D1   (us; wall_clock64 at 100 MHz.  B = the diagonal step of workgroup 0 / the one workgroup, Xwait = what it then waits\n"
      "                                  for: the other waves' tile fetches (cooperative kernel) or trailing tiles + panel load (one-workgroup kernel))")
nr = (n + 31) // 32
for r in range(nr):
    s = [st[r * 8 + k] for k in range(5)]
    d = [(s[k + 1] - s[k]) / 100.0 for k in range(4)] if r < nr - 1 else [(s[1] - s[0]) / 100.0, (s[2] - s[1]) / 100.0, 0.0, 0.0]
    tot = [a + b for a, b in zip(tot, d)]
    print("%5d %5.1f %6.1f %5.1f %5.1f" % (r, *d))
print("total %5.1f %6.1f %5.1f %5.1f  sum %.1f" % (*tot, sum(tot)))
