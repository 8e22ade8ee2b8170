"""40 reference-fidelity steps at q = 1 on a SMALL grid (the dense regime) for a kernel trace:
`rocprofv3 --kernel-trace --output-format csv -- python tools/dense_refstep_trace.py d g [kernel]`, then `python tools/trace_timeline.py <kernel_trace.csv>`
(tools/jobs/r5denseref.sh; profiles/r05_dense_refstep.txt).  kernel: rbf (default) | matern52 | matern12."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd.kernels import MaternKernel, RBFKernel, ScaleKernel
from online_gp_amd.models import Identity, OnlineSKIRegression
dev, dt = torch.device("cuda:0"), torch.float64
d, g = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "rbf"
cov = {"rbf": None, "matern52": lambda: ScaleKernel(MaternKernel(nu=2.5, ard_num_dims=d)), "matern12": lambda: ScaleKernel(MaternKernel(nu=0.5, ard_num_dims=d))}[kind]
X0, y0 = bench.synth_stream(200, d, 0, dev, dt, "uniform")
Xr, yr = bench.synth_stream(4096, d, 31337, dev, dt, "uniform")
reg = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, g, 1.0, covar_module=None if cov is None else cov().to(dev))
for i in range(40):
    xb, yb = Xr[i:i + 1], yr[i:i + 1]
    reg.evaluate(xb, yb); reg.update(xb, yb)
torch.cuda.synchronize()
