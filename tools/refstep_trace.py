"""30 reference-fidelity steps at q = 1 (evaluate mean + variance -> Adam step on the MLL -> condition) for a kernel trace:
`rocprofv3 --kernel-trace --output-format csv -- python tools/refstep_trace.py`, then `python tools/trace_timeline.py <kernel_trace.csv>`
(tools/jobs/r3step.sh does both; profiles/r03_refstep_timeline.txt)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float32
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(8192, 3, 31337, dev, dt, "uniform")
with settings.cg_tolerance(1e-4), settings.variance_cg_tolerance(3e-3):
    reg = OnlineSKIRegression(Identity(3), X0, y0, 1e-3, 50, 1.0)
    for i in range(30):
        xb, yb = Xr[i:i + 1], yr[i:i + 1]
        reg.evaluate(xb, yb); reg.update(xb, yb)
    torch.cuda.synchronize()
