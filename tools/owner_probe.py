"""Absorb of q points at 50^3 through wiski_scatter_stats_step, owner-computes (with a binning workspace) against the atomic
form; run under rocprofv3 for the per-kernel times (tools/trace_medians.py)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import _hip, grid_ops
q = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
dev, dt = torch.device("cuda:0"), torch.float32
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, 50)
X, y = bench.synth_stream(q * 24, 3, 5, dev, dt, kind)
y = y.reshape(-1).contiguous()
ones = torch.ones(q, device=dev, dtype=dt)
H = (grid.R + 1) // 2
A = torch.zeros((H, grid.m), device=dev, dtype=dt); b = torch.zeros(grid.m, device=dev, dtype=dt); cnt = torch.zeros_like(b); res = torch.zeros_like(b)
u = torch.randn(grid.m, device=dev, dtype=dt); stats = torch.zeros(2, device=dev, dtype=torch.float64); err = grid_ops.new_err_flag(dev)
mean = torch.empty(q, device=dev, dtype=dt)
f = _hip.lib().wiski_scatter_bin_bytes; f.restype = ctypes.c_int64
nbytes = int(f(grid.ref, ctypes.c_int64(q), ctypes.c_int32(4)))
binw = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
fn = _hip.fn("wiski_scatter_stats_step", dt)
def call(i, use_bin):
    sl = slice(i * q, (i + 1) * q)
    rc = fn(grid.ref, _hip.dptr(X[sl]), _hip.dptr(y[sl]), _hip.dptr(ones), _hip.dptr(ones), _hip.dptr(ones), ctypes.c_int64(q), _hip.dptr(b), _hip.dptr(A),
            _hip.dptr(cnt), _hip.dptr(u), _hip.dptr(res), _hip.dptr(mean), _hip.dptr(stats), _hip.dptr(err), None, ctypes.c_int64(0), None, ctypes.c_int64(0),
            None, ctypes.c_int64(0), _hip.dptr(binw) if use_bin else None, ctypes.c_int64(nbytes if use_bin else 0), _hip.stream_ptr(dev))
    assert rc == 0
for use_bin in (True, False):
    for i in range(4): call(i, use_bin)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(4, 24): call(i, use_bin)
    torch.cuda.synchronize()
    print(f"q={q} {kind} {'owner-computes' if use_bin else 'atomic form'}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per absorb")
