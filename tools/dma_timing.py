"""Per-wave phase timeline of k_spmv_sym_dma (needs a -DWISKI_DMA_TIMING build passed as WISKI_HIP_SO):
start / prologue issued / prologue landed / tile t landed (x7) / loop end / flush issued / end, 10 ns ticks."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import _hip, grid_ops  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, g)
gen = torch.Generator(device=dev).manual_seed(0)
n = 100000
X = torch.rand((n, 3), device=dev, generator=gen) * 2 - 1
y = torch.randn(n, device=dev, generator=gen)
w = torch.ones(n, device=dev)
err = grid_ops.new_err_flag(dev)
half = torch.zeros(((grid.R + 1) // 2, grid.m), device=dev)
b = torch.zeros(grid.m, device=dev)
stats = torch.zeros(2, device=dev, dtype=torch.float64)
grid_ops.scatter_stats_sym(grid, X, y, w, w, w, b, half, stats, err)
V = torch.randn((1, grid.m), device=dev, generator=gen)
for _ in range(5):
    out = grid_ops.stencil_spmv(grid, half, V)
torch.cuda.synchronize()
lib = _hip.lib()
nrb = (grid.m + 255) // 256
NP = int(os.environ.get("WISKI_SYM_DMA_PARTS", "4"))
nw = nrb * NP
buf = (ctypes.c_longlong * (nw * 16))()
assert lib.wiski_dma_dbg(buf, ctypes.c_int(nw * 16)) == 0
T = np.array(buf, dtype=np.int64).reshape(nw, 16).astype(np.float64)
t0 = T[:, 0].min()
T = (T - t0) * 0.01   # us
names = ["start", "issued", "prologue landed"] + [f"tile{t} landed" for t in range(7)] + ["loop end", "flushed", "end"]
print(f"waves {nw}; kernel span (first start -> last end) {T[:, 12].max():.2f} us")
for y_ in range(NP):
    if NP == 4:
        d0 = (y_ + 1) & 3; nt = 4 if d0 == 0 else 7
    elif NP == 5:
        d0 = [1, 2, 3, 0, 3][y_]; nt = [7, 7, 4, 4, 3][y_]
    elif NP == 6:
        d0 = [1, 2, 3, 0, 2, 3][y_]; nt = [7, 4, 4, 4, 3, 3][y_]
    else:
        d0 = y_ + 1 if y_ < 3 else (0 if y_ == 3 else y_ - 3); nt = 4 if y_ <= 3 else 3
    S = T[y_ * nrb:(y_ + 1) * nrb]
    print(f"-- chunk d0={d0} (blockIdx.y={y_}), {nrb} waves: start min/med/max {S[:,0].min():.2f}/{np.median(S[:,0]):.2f}/{S[:,0].max():.2f}  "
          f"end min/med/max {S[:,12].min():.2f}/{np.median(S[:,12]):.2f}/{S[:,12].max():.2f}")
    cols = [1, 2] + list(range(3, 3 + nt)) + [10, 11, 12]
    prev = 0
    line = []
    for c in cols:
        dt = np.median(S[:, c] - S[:, prev])
        line.append(f"{names[c]} +{dt:.2f}")
        prev = c
    print("   median phase durations (us): " + " | ".join(line))
    print(f"   prologue split: window issued +{np.median(S[:,13]-S[:,0]):.2f} | tiles issued +{np.median(S[:,14]-S[:,13]):.2f} | tw zeroed +{np.median(S[:,1]-S[:,14]):.2f}")
# bytes landed per microsecond (a tile counts at its 'landed' stamp)
land = T[:, 3:10].ravel(); land = land[land > 0]
hist, edges = np.histogram(land, bins=np.arange(0, T[:, 12].max() + 1, 1.0))
print("   tiles landed per us: " + " ".join(f"{int(h)}" for h in hist) + f"   (x 7 KiB; steady 9.3 TB/s = {9.3e6/7168:.0f}/us)")
ends, _ = np.histogram(T[:, 12], bins=np.arange(0, T[:, 12].max() + 1, 1.0))
print("   waves ending per us: " + " ".join(f"{int(h)}" for h in ends))
# by XCD (linear block id % 8) and by CU-ish slot: end-time distribution
lin = np.arange(nw)
for x in range(8):
    S = T[lin % 8 == x]
    print(f"   XCD {x}: waves {len(S)} end med/p90/max {np.median(S[:,12]):.2f}/{np.percentile(S[:,12],90):.2f}/{S[:,12].max():.2f}  bytes-weighted tiles {np.sum(S[:,3:10]>0)}")
late = np.argsort(-T[:, 12])[:12]
print("   latest waves (linear id, y, x, start, end):", [(int(i), int(i // nrb), int(i % nrb), round(float(T[i,0]),2), round(float(T[i,12]),2)) for i in late])
# concurrency profile: how many waves are alive at each microsecond
for t in np.arange(0, T[:, 12].max() + 1, 1.0):
    alive = ((T[:, 0] <= t) & (T[:, 12] > t)).sum()
    print(f"   t={t:5.1f} us alive {alive}")
