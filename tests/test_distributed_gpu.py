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


def _join(rank, world, port):
    """Join the test's process group and return this rank's device.  On a box with a device for every rank: nccl (= RCCL over
    xGMI), one device per rank -- the arrangement the product runs in; on the 1-GPU boxes: gloo with every rank on cuda:0
    (online_gp_amd.distributed.pick_backend; WISKI_TEST_BACKEND forces one)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from online_gp_amd.distributed import pick_backend

    backend, dev_of = pick_backend(world, torch.cuda.device_count(), os.environ.get("WISKI_TEST_BACKEND"))
    dev = torch.device("cuda", dev_of(rank))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _transport(monkeypatch, transport):
    """torch: the per-product all-reduce through torch.distributed's communicator (RCCL on a multi-GPU box, gloo through the host on
    a 1-GPU box).  rccl: the in-C transport (wiski_allreduce_stats on libwiski's own ncclComm_t, grouped launch on the solve's
    stream) -- it needs one device per rank, so it runs the first time a test box shows two devices."""
    if transport == "rccl" and torch.cuda.device_count() < 2:
        pytest.skip("the in-C RCCL transport needs one device per rank (RCCL refuses two ranks on one device)")
    monkeypatch.setenv("WISKI_SHARD_TRANSPORT", transport)        # (inherited by the spawned ranks)


def _worker(rank, world, port, tmpdir):
    dev = _join(rank, world, port)
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

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
    # predictions shard over the query points: each rank solves the variance columns of its slice, one all-gather
    from online_gp_amd.distributed import sharded_posterior_moments

    Xq = X[100:137]                                                  # 37 queries: uneven split 19 + 18
    mean_s, var_s = sharded_posterior_moments(model, Xq)
    mvn = ref(Xq)
    ok = ok and mean_s.shape == (37,) and torch.allclose(mean_s, mvn.mean, rtol=1e-3, atol=1e-4) and torch.allclose(var_s, mvn.variance, rtol=1e-2, atol=1e-6)
    open(os.path.join(tmpdir, f"ok_{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def _worker_stream_step(rank, world, port, tmpdir):
    """ShardedStatsUpdater.stream_step (point exchange -> the model's one-call step on the gathered batch, deferred poll) on a
    grid past the dense regime: every rank's replica equals plain conditioning on the concatenated shards, and the returned
    means are the rank's slice of the pre-update predictive means."""
    dev = _join(rank, world, port)
    from online_gp_amd import settings
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(3)
    q, steps = 4608, 4            # 9216 gathered points per step: the owner-computes absorb (batches >= 8192) inside the one-call step
    X = torch.as_tensor(rng.uniform(-1, 1, (500 + steps * world * q, 3)), device=dev, dtype=torch.float32)
    y = torch.sin(2 * X[:, :1]) * torch.cos(X[:, 1:2]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=torch.float32)
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    ok = True
    with settings.cg_tolerance(1e-6), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
        ref = FixedNoiseOnlineSKIGP(X[:500], y[:500], torch.ones_like(y[:500]), grid_bounds=gb, grid_size=16, learn_additional_noise=True).eval()
        model = FixedNoiseOnlineSKIGP(X[:500], y[:500], torch.ones_like(y[:500]), grid_bounds=gb, grid_size=16, learn_additional_noise=True).eval()
        ref.prediction_cache; model.prediction_cache
        upd = ShardedStatsUpdater(model, equal_shards=True, exchange="points")
        for s in range(steps):
            lo = 500 + s * world * q
            mine = slice(lo + rank * q, lo + (rank + 1) * q)
            want = ref(X[mine]).mean                               # predictive mean before the update
            got = upd.stream_step(X[mine], y[mine])
            ok = ok and got.shape == (q,) and torch.allclose(got, want, rtol=1e-3, atol=2e-4)
            ref.condition_on_observations(X[lo:lo + world * q], y[lo:lo + world * q], inplace=True)
            ref.prediction_cache
        model._finish_pending()
        ok = ok and upd.last_exchange == "points" and model.num_data == ref.num_data == 500 + steps * world * q
        Xs = X[:64]
        ok = ok and torch.allclose(model(Xs).mean, ref(Xs).mean, rtol=1e-3, atol=2e-4)
    open(os.path.join(tmpdir, f"ss_{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def _worker_stencil_shard(rank, world, port, tmpdir):
    """The work-dividing exchange (wiski_shard): after the all-gather of the shards every rank scatters and
    multiplies only ITS groups of the half stencil, one m-vector all-reduce per CG iteration completes A p (here through gloo:
    both ranks share the one GPU).  Checked on every rank: identical CG iteration counts to a single-process model fed the
    concatenated batches, the same means, and -- after leave_stencil_shard() sums the disjoint shards -- the same statistics."""
    dev = _join(rank, world, port)
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    rng = np.random.default_rng(7)
    q, steps, n0 = 1024, 6, 3000
    X = torch.as_tensor(rng.uniform(-1, 1, (n0 + steps * world * q, 3)), device=dev, dtype=torch.float32)
    y = torch.sin(2 * X[:, :1]) * torch.cos(X[:, 1:2]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=torch.float32)
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    ok = True
    msgs = []
    with settings.cg_tolerance(1e-5), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
        ref = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=24, learn_additional_noise=True).eval()
        model = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=24, learn_additional_noise=True).eval()
        ref.prediction_cache; model.prediction_cache
        upd = ShardedStatsUpdater(model, equal_shards=True, exchange="stencil")
        its_ref, its_got = [], []
        for s in range(steps):
            lo = n0 + s * world * q
            mine = slice(lo + rank * q, lo + (rank + 1) * q)
            want = ref.stream_step(X[lo:lo + world * q], y[lo:lo + world * q])
            got = upd.stream_step(X[mine], y[mine])
            ok = ok and got.shape == (q,) and torch.allclose(got, want[rank * q:(rank + 1) * q], rtol=1e-3, atol=2e-4)
        ref._finish_pending(); model._finish_pending()
        its_ref, its_got = ref._last_iters, model._last_iters
        ok = ok and upd.last_exchange == "stencil" and model.__dict__.get("_stencil_shard") is not None
        ok = ok and model.num_data == ref.num_data == n0 + steps * world * q
        ok = ok and its_ref == its_got
        msgs.append(f"iters ref {its_ref} got {its_got}")
        # this rank's stencil holds its groups only ...
        lo_g, hi_g = grid_ops.shard_groups(3, rank, world)
        flat = model._kernel_cache["WtW"].stencil.reshape(-1)
        reff = ref._kernel_cache["WtW"].stencil.reshape(-1)
        for a, b in grid_ops.half_stencil_group_slices(model._grid, 0, lo_g) + grid_ops.half_stencil_group_slices(model._grid, hi_g, 25):
            ok = ok and float(flat[a:b].abs().max()) == 0.0
        for a, b in grid_ops.half_stencil_group_slices(model._grid, lo_g, hi_g):
            ok = ok and (flat[a:b] - reff[a:b]).abs().max().item() < 1e-5 * float(reff.abs().max())
        # ... and the posterior mean is that of the single-process model
        Xs = X[:64]
        m1 = grid_ops.gather(model._grid, Xs, model._mean_state["U"], model._err)[:, 0]
        m2 = grid_ops.gather(ref._grid, Xs, ref._mean_state["U"], ref._err)[:, 0]
        ok = ok and torch.allclose(m1, m2, rtol=1e-3, atol=2e-4)
        msgs.append(f"mean dev {(m1 - m2).abs().max().item():.2e}")
        # a consumer that needs the whole stencil (predictive variances) sums the shards back first (collective)
        with settings.skip_posterior_variances(False), settings.spectral_factor(False), settings.variance_cg_tolerance(1e-4):
            v1 = model(Xs[:8]).variance
            v2 = ref(Xs[:8]).variance
        ok = ok and model.__dict__.get("_stencil_shard") is None
        ok = ok and (model._kernel_cache["WtW"].stencil - ref._kernel_cache["WtW"].stencil).abs().max().item() < 1e-5 * float(reff.abs().max())
        ok = ok and torch.allclose(v1, v2, rtol=1e-2, atol=1e-7)
        # and the sharded step can be re-entered afterwards
        lo = n0
        got = upd.stream_step(X[lo + rank * q:lo + (rank + 1) * q], y[lo + rank * q:lo + (rank + 1) * q])
        want = ref.stream_step(X[lo:lo + world * q], y[lo:lo + world * q])
        ok = ok and torch.allclose(got, want[rank * q:(rank + 1) * q], rtol=1e-3, atol=2e-4) and model.__dict__.get("_stencil_shard") is not None
    # the two-level preconditioner inside the sharded step: replicas must switch a refreshed block in at the SAME step (lock-step
    # activation) or their iteration counts -- and with them their collectives -- diverge.  Forced on (min_iters 0) on a
    # road-like stream; the single-process reference runs the same lock-step schedule and the same subsampling of the Gram
    # accumulation (a sharded replica takes every world-th point of the gathered batch).
    sys.path.insert(0, ROOT)
    import bench

    Xc, yc = bench.synth_stream(n0 + 10 * world * q, 3, 3, dev, torch.float32, "clustered")
    with settings.cg_tolerance(1e-5), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), \
            settings.two_level_min_iters(0.0), settings.two_level_rank(96), settings.two_level_lockstep(True), settings.two_level_subsample(2), torch.no_grad():
        ref = FixedNoiseOnlineSKIGP(Xc[:n0], yc[:n0], torch.ones_like(yc[:n0]), grid_bounds=gb, grid_size=24, learn_additional_noise=True).eval()
        model = FixedNoiseOnlineSKIGP(Xc[:n0], yc[:n0], torch.ones_like(yc[:n0]), grid_bounds=gb, grid_size=24, learn_additional_noise=True).eval()
        ref.prediction_cache; model.prediction_cache
        upd = ShardedStatsUpdater(model, equal_shards=True, exchange="stencil")
        hist_ref, hist_got = [], []
        for s in range(10):
            lo = n0 + s * world * q
            want = ref.stream_step(Xc[lo:lo + world * q], yc[lo:lo + world * q])
            got = upd.stream_step(Xc[lo + rank * q:lo + (rank + 1) * q], yc[lo + rank * q:lo + (rank + 1) * q])
            ok = ok and torch.allclose(got, want[rank * q:(rank + 1) * q], rtol=1e-3, atol=3e-4)
            hist_ref.append(ref._last_iters[0]); hist_got.append(model._last_iters[0])
        ref._finish_pending(); model._finish_pending()
        trm, trr = model.__dict__["_two_level"], ref.__dict__["_two_level"]
        ok = ok and trm.block is not None and trm.block.active >= 0 and trm.block.refreshes == trr.block.refreshes
        ok = ok and hist_ref == hist_got and model.__dict__.get("_stencil_shard") is not None
        msgs.append(f"two-level iters ref {hist_ref} got {hist_got} refreshes {trm.block.refreshes if trm.block else None}")
    msgs.append(f"transport {upd.shard_transport} backend {dist.get_backend()} device {dev}")
    open(os.path.join(tmpdir, f"st_{rank}"), "w").write(("1" if ok else "0") + " " + "; ".join(msgs))
    dist.barrier()
    dist.destroy_process_group()


def _worker_stencil_shard_any_d(rank, world, port, tmpdir):
    """The stencil-sharded step outside d = 3 / fp32 (round 4): the LDS-window SpMV restricted to the replica's group range.
    BASELINE config 2's geometry scaled down (d = 4, fp64, 12^4) and a 2-D fp32 grid beyond the dense regime (64^2): same means,
    iteration counts and (re-summed) statistics as a single-process model fed the concatenated batches."""
    dev = _join(rank, world, port)
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    ok, msgs = True, []

    def chk(ok_, tag, val):
        if not val:
            msgs.append(f'check {tag} failed')
        return ok_ and bool(val)

    for d, g, dtype, n0, q, tol, cgt in ((4, 12, torch.float64, 6000, 256, 1e-7, 1e-9), (2, 64, torch.float32, 6000, 256, 2e-4, 1e-5)):
        rng = np.random.default_rng(11 + d)
        steps = 4
        X = torch.as_tensor(rng.uniform(-1, 1, (n0 + steps * world * q, d)), device=dev, dtype=dtype)
        y = torch.sin(2 * X[:, :1]) * torch.cos(X[:, 1:2]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=dtype)
        gb = torch.tensor([[-1.1, 1.1]] * d)
        with settings.cg_tolerance(cgt), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
            ref = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
            model = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
            ref.prediction_cache; model.prediction_cache
            upd = ShardedStatsUpdater(model, equal_shards=True, exchange="stencil")
            for s in range(steps):
                lo = n0 + s * world * q
                want = ref.stream_step(X[lo:lo + world * q], y[lo:lo + world * q])
                got = upd.stream_step(X[lo + rank * q:lo + (rank + 1) * q], y[lo + rank * q:lo + (rank + 1) * q])
                ok = chk(ok, 1, torch.allclose(got, want[rank * q:(rank + 1) * q], rtol=10 * tol, atol=10 * tol))
            ref._finish_pending(); model._finish_pending()
            ok = chk(ok, 2, upd.last_exchange == "stencil" and model.__dict__.get("_stencil_shard") is not None)
            ok = chk(ok, 3, ref._last_iters == model._last_iters)
            msgs.append(f"d={d}: iters ref {ref._last_iters} got {model._last_iters}")
            ng = (model._grid.R // 7 + 1) // 2
            lo_g, hi_g = grid_ops.shard_groups(d, rank, world)
            flat = model._kernel_cache["WtW"].stencil.reshape(-1)
            reff = ref._kernel_cache["WtW"].stencil.reshape(-1)
            for a, b in grid_ops.half_stencil_group_slices(model._grid, 0, lo_g) + grid_ops.half_stencil_group_slices(model._grid, hi_g, ng):
                ok = chk(ok, 4, float(flat[a:b].abs().max()) == 0.0)
            for a, b in grid_ops.half_stencil_group_slices(model._grid, lo_g, hi_g):
                ok = chk(ok, 5, (flat[a:b] - reff[a:b]).abs().max().item() <= tol * float(reff.abs().max()))
            m1 = grid_ops.gather(model._grid, X[:64], model._mean_state["U"], model._err)[:, 0]
            m2 = grid_ops.gather(ref._grid, X[:64], ref._mean_state["U"], ref._err)[:, 0]
            ok = chk(ok, 6, torch.allclose(m1, m2, rtol=10 * tol, atol=10 * tol))
            msgs.append(f"d={d}: mean dev {(m1 - m2).abs().max().item():.2e}")
            model.leave_stencil_shard()
            ok = chk(ok, 7, (model._kernel_cache["WtW"].stencil - ref._kernel_cache["WtW"].stencil).abs().max().item() <= tol * float(reff.abs().max()))
    open(os.path.join(tmpdir, f"sd_{rank}"), "w").write(("1" if ok else "0") + " " + "; ".join(msgs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_stencil_sharded_stream_step_any_dimension_world2(tmp_path, monkeypatch, transport):
    _transport(monkeypatch, transport)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_stencil_shard_any_d, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = open(tmp_path / f"sd_{r}").read()
        assert res.startswith("1"), res


@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_stencil_sharded_stream_step_world2(tmp_path, monkeypatch, transport):
    _transport(monkeypatch, transport)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_stencil_shard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = open(tmp_path / f"st_{r}").read()
        assert res.startswith("1"), res
        if transport == "rccl":
            assert "transport rccl" in res, res


def test_sharded_stream_step_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_stream_step, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ss_{r}").read() == "1"


def test_sharded_updater_on_gpu_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"ok_{r}").read() == "1"


def test_cabi_rccl_allreduce_stats_single_rank():
    """`wiski_allreduce_stats` + `wiski_comm_*` (the C-ABI collective of SURVEY 8b): a 1-rank RCCL communicator (RCCL refuses two
    ranks on one device, and the test boxes have one) drives the statistics-delta exchange of ShardedStatsUpdater; the result
    must equal plain in-place conditioning.  Multi-rank correctness of the same code path is covered with gloo above."""
    sys.path.insert(0, ROOT)
    from online_gp_amd.distributed import RcclCommunicator, ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    X = torch.as_tensor(rng.uniform(-1, 1, (40 + 2 * 512, 3)), device=dev, dtype=torch.float32)
    y = torch.sin(2 * X[:, :1]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=torch.float32)
    nz = torch.as_tensor(rng.uniform(0.5, 1.5, (X.shape[0], 1)), device=dev, dtype=torch.float32)
    gb = torch.tensor([[-1.1, 1.1]] * 3)
    ref = FixedNoiseOnlineSKIGP(X[:40], y[:40], nz[:40], grid_bounds=gb, grid_size=12, learn_additional_noise=True).eval()
    model = FixedNoiseOnlineSKIGP(X[:40], y[:40], nz[:40], grid_bounds=gb, grid_size=12, learn_additional_noise=True).eval()
    comm = RcclCommunicator()
    assert comm.world == 1 and comm.handle
    upd = ShardedStatsUpdater(model, exchange="stats", comm=comm)
    for s in range(2):
        sl = slice(40 + s * 512, 40 + (s + 1) * 512)
        ref.condition_on_observations(X[sl], y[sl], nz[sl], inplace=True)
        upd.update(X[sl], y[sl], nz[sl])
        assert upd.last_exchange == "stats"
    a, b = model._kernel_cache, ref._kernel_cache
    sc = float(b["WtW"].stencil.abs().max())
    assert (a["WtW"].stencil - b["WtW"].stencil).abs().max().item() < 1e-5 * sc
    assert torch.allclose(a["interpolation_cache"], b["interpolation_cache"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(a["_stats"], b["_stats"], rtol=1e-7) and torch.allclose(a["_cnt"], b["_cnt"], rtol=1e-4, atol=1e-5)
    assert model.num_data == ref.num_data and abs(model._wsum[0] - ref._wsum[0]) < 1e-3
    assert torch.allclose(model(X[:16]).mean, ref(X[:16]).mean, rtol=1e-3, atol=1e-4)
    comm.close()


@pytest.mark.parametrize("d,g,dtype", [(3, 24, torch.float32), (3, 24, torch.float64), (4, 12, torch.float64)])
def test_sharded_solver_in_c_rccl_branch_single_rank(d, g, dtype):
    """The in-C transport of the stencil-sharded step (include/wiski.h: wiski_shard with `comm` set): `wiski_stream_step` ->
    `wiski_scatter_stats_step_sharded` + `wiski_pcg_sharded`, whose per-product all-reduce is ONE grouped RCCL launch
    (`wiski_allreduce_stats` on an ncclComm_t) on the solve's stream -- the default transport on the nccl backend.  A 1-rank
    communicator owns all 25 groups (RCCL refuses two ranks on one device), so every kernel and collective of that branch
    executes on the hardware: part-table SpMV (fp32, d = 3) or the group-range LDS-window kernel (fp64, d = 4: BASELINE config 2's
    geometry scaled down), k_shard_reduce, ncclAllReduce of the m-vector (ncclFloat32 / ncclFloat64) and the p.Ap slots.  Must
    equal the unsharded step: identical iteration counts, means, statistics."""
    sys.path.insert(0, ROOT)
    from online_gp_amd import grid_ops, settings
    from online_gp_amd.distributed import RcclCommunicator
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    q, steps, n0 = 1024, 6, 3000
    X = torch.as_tensor(rng.uniform(-1, 1, (n0 + steps * q, d)), device=dev, dtype=dtype)
    y = torch.sin(2 * X[:, :1]) * torch.cos(X[:, 1:2]) + 0.1 * torch.as_tensor(rng.standard_normal((X.shape[0], 1)), device=dev, dtype=dtype)
    gb = torch.tensor([[-1.1, 1.1]] * d)
    comm = RcclCommunicator()
    calls = []
    f64 = dtype == torch.float64
    with settings.cg_tolerance(1e-9 if f64 else 1e-5), settings.skip_posterior_variances(True), settings.deferred_refresh(True), settings.deferred_bounds_check(True), torch.no_grad():
        ref = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        model = FixedNoiseOnlineSKIGP(X[:n0], y[:n0], torch.ones_like(y[:n0]), grid_bounds=gb, grid_size=g, learn_additional_noise=True).eval()
        ref.prediction_cache; model.prediction_cache
        # the Python callback must never be needed on this route: it records if it is
        assert model.enter_stencil_shard(0, 1, lambda vec, dots: calls.append(1), allreduce_full=lambda t: None, comm=comm.handle)
        for s in range(steps):
            sl = slice(n0 + s * q, n0 + (s + 1) * q)
            want = ref.stream_step(X[sl], y[sl])
            got = model.stream_step(X[sl], y[sl])
            assert torch.allclose(got, want, rtol=1e-7 if f64 else 1e-3, atol=1e-8 if f64 else 2e-4), s
            step = model.__dict__["_stream_step_cache"][1]
            assert step.args.shard and step._shard.comm == comm.handle.value and step._shard.nranks == 1
        ref._finish_pending(); model._finish_pending()
        assert model.__dict__.get("_stencil_shard") is not None and not calls
        assert ref._last_iters == model._last_iters and model.num_data == ref.num_data
        a, b = model._kernel_cache["WtW"].stencil, ref._kernel_cache["WtW"].stencil
        assert (a - b).abs().max().item() < (1e-12 if f64 else 1e-5) * float(b.abs().max())
        m1 = grid_ops.gather(model._grid, X[:64], model._mean_state["U"], model._err)[:, 0]
        m2 = grid_ops.gather(ref._grid, X[:64], ref._mean_state["U"], ref._err)[:, 0]
        assert torch.allclose(m1, m2, rtol=1e-7 if f64 else 1e-3, atol=1e-8 if f64 else 2e-4)
        model.leave_stencil_shard()
    comm.close()


def _worker_c5(rank, world, port, tmpdir):
    """BASELINE config 5's data-parallel leg on its geometry: d = 2, 30^2 grid, Matern-1/2, heteroscedastic noise, batch 6 per
    step, points dealt round-robin to the ranks, exchange = all-reduce of the statistics (the north-star form); every rank's
    replica must equal the data-space oracle on the whole stream."""
    dev = _join(rank, world, port)
    from oracle import dataspace
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.kernels import GridInterpolationKernel, MaternKernel, ScaleKernel
    from online_gp_amd.models import OnlineSKIBotorchModel

    rng = np.random.default_rng(5)
    n0, q, steps = 10, 6, 20
    X = rng.uniform(0, 1, (n0 + q * steps, 2)); y = np.sin(5 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.1 * rng.standard_normal(X.shape[0])
    nz = rng.uniform(1e-6, 0.05, X.shape[0])
    Xt, yt, nt = torch.as_tensor(X, device=dev), torch.as_tensor(y, device=dev)[:, None], torch.as_tensor(nz, device=dev)[:, None]
    gb = torch.tensor([[0.0, 1.0]] * 2, dtype=torch.float64)
    cov = GridInterpolationKernel(ScaleKernel(MaternKernel(nu=0.5, ard_num_dims=2)), grid_size=30, num_dims=2, grid_bounds=gb)
    model = OnlineSKIBotorchModel(Xt[:n0], yt[:n0], nt[:n0], covar_module=cov, learn_additional_noise=True)
    model.eval()
    from online_gp_amd.mlls import BatchedWoodburyMarginalLogLikelihood

    mll = BatchedWoodburyMarginalLogLikelihood(model.likelihood, model)
    with torch.no_grad():
        mll(None, None)                                             # (a BO / AL loop scores the MLL: this builds the spectral factor)
    upd = ShardedStatsUpdater(model, exchange="stats")
    for s in range(steps):
        lo = n0 + s * q
        mine = slice(lo + rank, lo + q, world)                      # round-robin shard of the batch (3 points per rank)
        upd.update(Xt[mine], yt[mine], nt[mine])
        if s % 5 == 4:
            model.posterior(Xt[:4]).mean                            # posterior and MLL requests between synchronisation points
            with torch.no_grad():
                mll(None, None)
    ok = upd.last_exchange == "stats" and model.num_data == n0 + q * steps
    # (round 5) the spectral factor follows the all-reduced increments through the points gathered beside the all-reduce: built from the
    # stencil ONCE, not once per synchronisation point -- and stays exact (MLL against the data-space oracle below)
    fac = model.__dict__.get("_spectral", {}).get(0)
    ok = ok and fac is not None and fac.rebuilds == 1
    with torch.no_grad():
        mll_val = float(mll(None, None))
    s2 = float(model._sigma2(0))
    ell = model.covar_module.base_kernel.base_kernel.lengthscale.detach().double().cpu().numpy().reshape(-1)
    osc = float(model.covar_module.base_kernel.outputscale.detach().double())
    O = dataspace.DataSpaceGP(gb.numpy(), 30, "matern12", ell, osc, s2).fit(X, y, nz)
    ok = ok and abs(mll_val - O.mll()) < 1e-6 * abs(O.mll())
    Xq = rng.uniform(0, 1, (12, 2))                                  # same draw on both ranks
    mo, vo = O.predict(Xq)
    post = model.posterior(torch.as_tensor(Xq, device=dev))
    ok = ok and np.abs(post.mean[:, 0].cpu().numpy() - mo).max() < 1e-4 * np.abs(mo).max()
    ok = ok and np.abs(post.variance[:, 0].cpu().numpy() - vo).max() < 1e-4 * vo.max()
    open(os.path.join(tmpdir, f"c5_{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_c5_malaria_geometry_stats_allreduce_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_c5, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"c5_{r}").read() == "1"
