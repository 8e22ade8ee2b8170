"""True number of PCG iterations each streaming step needs (poll after every iteration) along the bench stream."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from online_gp_amd import settings
from online_gp_amd.models import FixedNoiseOnlineSKIGP
dev, dt = torch.device("cuda:0"), torch.float32
q = 4096
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
X0, y0 = bench.synth_stream(21743, 3, 0, dev, dt, kind)
gb = torch.tensor([[-1.1, 1.1]] * 3)
with settings.skip_posterior_variances(True), settings.cg_tolerance(1e-4), settings.deferred_bounds_check(True), torch.no_grad():
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=50, learn_additional_noise=True).eval()
    model.prediction_cache
    n = 100
    Xr, yr = bench.synth_stream(q * n, 3, 1000, dev, dt, kind)
    seq = []
    rels = []
    for i in range(n):
        model._last_iters = [1]
        model._probe_wait = 1000
        model._refresh_count = 1
        model.stream_step(Xr[i * q:(i + 1) * q], yr[i * q:(i + 1) * q])
        seq.append(model._last_iters[0])
        rels.append(model._memo['prediction_cache'].get('relres', None) if False else getattr(model, '_last_rel', None))
    print("need per step:", "".join(str(min(s, 9)) for s in seq))
    print(collections.Counter(seq))
    print("rel at convergence:", " ".join(f"{r:.1e}" if r is not None else "-" for r in rels[::3]))
