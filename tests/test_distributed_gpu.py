"""GPU (one device, two processes, gloo): the real HIP data-parallel paths -- half-stencil delta scatter +
all-reduce, and shard all-gather + redundant scatter -- equal single-process accumulation of the concatenated shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    X = torch.as_tensor(rng.uniform(-1, 1, (40 + 3 * 2048, 3)), device=dev, dtype=torch.float32)
    y = torch.sin(2 * X[:, :1]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=torch.float32)
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    ref = FixedNoiseOnlineSKIGP(X[:40], y[:40], None, grid_bounds=gb, grid_size=12, learn_additional_noise=True)
    ref.eval()
    for s in range(3):
        lo = 40 + s * 2048
        ref.condition_on_observations(X[lo:lo + 2048], y[lo:lo + 2048], inplace=True)
    b = ref._kernel_cache
    sc = float(b["WtW"].stencil.abs().max())
    Xs = X[:16]
    ok = True
    for mode in ("stats", "points"):      # all-reduce of the half-stencil delta / all-gather of the shards + redundant scatter
        model = FixedNoiseOnlineSKIGP(X[:40], y[:40], None, grid_bounds=gb, grid_size=12, learn_additional_noise=True)
        model.eval()
        upd = ShardedStatsUpdater(model, exchange=mode)
        for s in range(3):
            lo = 40 + s * 2048 + rank * 1024
            upd.update(X[lo:lo + 1024], y[lo:lo + 1024])                   # unit noise
            model.prediction_cache                                          # refresh between updates (exercises the carried residual)
        a = model._kernel_cache
        ok = ok and ((a["WtW"].stencil - b["WtW"].stencil).abs().max().item() < 1e-4 * sc and
                     (a["interpolation_cache"] - b["interpolation_cache"]).abs().max().item() < 1e-4 * float(b["interpolation_cache"].abs().max()) and
                     torch.allclose(a["_stats"], b["_stats"], rtol=1e-6) and torch.allclose(a["_cnt"], b["_cnt"], rtol=1e-4, atol=1e-4) and
                     model.num_data == ref.num_data == 40 + 3 * 2048 and abs(model._wsum[0] - ref._wsum[0]) < 1e-6 and
                     upd.last_exchange == mode)
        ok = ok and torch.allclose(model(Xs).mean, ref(Xs).mean, rtol=1e-3, atol=1e-4)
    open(os.path.join(tmpdir, f"ok_{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_updater_on_gpu_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ok_{r}").read() == "1"
