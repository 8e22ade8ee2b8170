import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev = torch.device('cuda:0'); dt = torch.float32; d = 3
X0, y0 = bench.synth_stream(100000, d, 0, dev, dt)
model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=torch.tensor([[-1.1, 1.1]] * d), grid_size=50, learn_additional_noise=True)
model.eval()
with settings.cg_tolerance(1e-4), torch.no_grad():
    pc = model.prediction_cache; print('mean iters', pc['cg_iters'])
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        mv = model(X0[:64]); v = mv.variance
        torch.cuda.synchronize(); print('var 64: %.2f ms' % ((time.perf_counter() - t) * 1e3), 'iters', pc['pred_cov'].last_iters, max(pc['pred_cov'].last_relres))
