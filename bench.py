#!/usr/bin/env python
"""bench.py -- WISKI streaming updates/s on the 50^3 inducing grid (BASELINE.json
configs[2]: "3droad UCI stream (d=3), 50^3 grid, 1xMI355X fp32, CG solve path").

A *step* is one pass of the hot path over one batch of q streamed points, in the
evaluate-then-update order of the reference's driver
(experiments/regression.py:48-54) at batch granularity:

  1. predictive mean of the incoming batch        (fused gather, wiski_gather)
  2. absorb the batch into the statistics          (wiski_scatter_stats; with
     N > 1 GPUs each rank absorbs its own q points and the deltas are
     all-reduced over RCCL -- weak scaling)
  3. refresh the inducing posterior mean           (wiski_pcg, warm-started)

Hyper-parameters are fixed ("no hyper steps"); predictive
variances are not part of the timed step (their latency per 64-query chunk is
reported in `extra`).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the stencil
SpMV of the CG, HBM-bound, timed live with HIP events on its launch stream)
and `cpu_baseline` (the C oracle port on a bounded sample, rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_stream(n, d, seed, device, dtype, kind="uniform"):
    """S3 of SURVEY.md 8d: X ~ U(-1,1)^3, y = sin(2 pi x0) cos(pi x1) + 0.5 x2 + 0.1 N(0,1), standardised.
    kind="clustered": the road-like variant of 8d -- points along 64 random poly-lines (8 segments each, the same
    lines for every seed) with sigma = 0.02 jitter, so that many points of a batch share grid cells."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind == "clustered":
        gl = torch.Generator(device="cpu").manual_seed(12345)
        knots = torch.rand(64, 9, d, generator=gl, dtype=torch.float64) * 1.8 - 0.9        # 64 lines x 9 knots
        line = torch.randint(0, 64, (n,), generator=g)
        t = torch.rand(n, generator=g, dtype=torch.float64) * 8
        seg = t.floor().clamp(max=7).long()
        fr = (t - seg)[:, None]
        X = (1 - fr) * knots[line, seg] + fr * knots[line, seg + 1] + 0.02 * torch.randn(n, d, generator=g, dtype=torch.float64)
        X = X.clamp(-1.0, 1.0)
    else:
        X = torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1
    y = torch.sin(2 * np.pi * X[:, 0]) * torch.cos(np.pi * X[:, 1 % d]) + 0.5 * X[:, (2 % d)] + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
    y = (y - y.mean()) / y.std()
    return X.to(device, dtype), y.to(device, dtype)[:, None]


def cpu_baseline(args, tol):
    """Time the C oracle (scalar port, 1 core) on a bounded sample of the same workload."""
    from oracle import cport, spec

    cport.build()
    ndt = np.float32 if args.dtype == "f32" else np.float64
    n_init, q, steps = 8192, 2048, 5
    Xt, yt = synth_stream(n_init + q * (steps + 1), args.dim, 0, "cpu", torch.float64)
    X, y = Xt.numpy(), yt.numpy()[:, 0]
    B2 = cport.MatrixFreeWISKI([[-1.1, 1.1]] * args.dim, args.grid, sigma2=spec.SOFTPLUS0 + 1e-4, dtype=ndt)
    B2.absorb(X[:n_init], y[:n_init], init=True)
    B2.refresh(tol, 2000)
    t0 = time.perf_counter()
    for s in range(steps):
        lo = n_init + s * q
        B2.predict_mean(X[lo:lo + q], tol)
        B2.absorb(X[lo:lo + q], y[lo:lo + q])
        B2.refresh(tol, 2000)   # the C port has no warm start: cold solve per step
    dt = time.perf_counter() - t0
    return {"value": steps * q / dt, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": f"{steps} steps of q={q} after a {n_init}-point init, same grid/dtype/tolerance, cold CG per step, "
                      f"scalar C oracle (oracle/wiski_oracle.c), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="streamed points per step per GPU (q)")
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--n-init", type=int, default=21743, help="5%% of 434874 (init_ratio of the reference config)")
    ap.add_argument("--tol", type=float, default=None, help="CG relative-residual tolerance")
    ap.add_argument("--stream", default="uniform", choices=["uniform", "clustered"], help="synthetic stream of SURVEY.md 8d")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-every", type=int, default=None, help="CG iterations between host convergence checks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    # self-test hook for 1-GPU boxes: WISKI_BENCH_BACKEND=gloo puts every rank on cuda:0 and uses gloo, so that the
    # N > 1 code path can be exercised without a second device (never used for reported numbers)
    selftest = os.environ.get("WISKI_BENCH_BACKEND") == "gloo"
    if selftest:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if selftest:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from online_gp_amd import _hip, settings
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.kernels import GridInterpolationKernel, RBFKernel, ScaleKernel
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    tol = args.tol if args.tol is not None else (1e-4 if dtype == torch.float32 else 1e-8)
    K, Wm, q, d = args.steps, args.warmup, args.batch, args.dim

    # identical init on every rank (replicated statistics), rank-local stream shards
    X0, y0 = synth_stream(args.n_init, d, 0, dev, dtype, args.stream)
    Xs, ys = synth_stream((K + Wm + 1) * q, d, 1000 + rank, dev, dtype, args.stream)
    gb = torch.tensor([[-1.1, 1.1]] * d)
    model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=args.grid, learn_additional_noise=True)
    model.eval()
    upd = ShardedStatsUpdater(model, equal_shards=True)   # every rank streams q points per step
    lib = _hip.lib()

    def step(t):
        xb, yb = Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q]
        mean = model(xb).mean                      # 1. evaluate
        upd.update(xb, yb)                         # 2. absorb (+ all-reduce of the deltas when N > 1)
        pc = model.prediction_cache                # 3. refresh
        return mean, pc["cg_iters"][0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.check_every:
        settings.cg_check_every._set_value(args.check_every)
    with settings.skip_posterior_variances(True), settings.cg_tolerance(tol), torch.no_grad():
        model.prediction_cache                     # cold solve on the init data (not timed)
        for t in range(Wm):
            step(t)
        barrier()
        # roofline leg: HIP events bracket every stencil-SpMV launch of every 4th timed step (each bracket costs
        # ~6 us of stream time on both sides of the kernel, so sampling keeps the timed region honest)
        iters = []
        tot_ms_sum, launches_sum = 0.0, 0
        t0 = time.perf_counter()
        for t in range(Wm, Wm + K):
            sampled = (t - Wm) % 4 == 0
            if sampled:
                lib.wiski_prof_start(ctypes.c_int32(4096))
            _, it = step(t)
            iters.append(it)
            if sampled:
                # no synchronisation needed: step() returned from the solver's convergence poll, which is ordered
                # after every bracketed SpMV on the same stream
                tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
                lib.wiski_prof_stop(ctypes.byref(tms), ctypes.byref(nl))
                tot_ms_sum += tms.value
                launches_sum += int(nl.value)
        barrier()
        elapsed = time.perf_counter() - t0
        tot_ms, launches = ctypes.c_double(tot_ms_sum), ctypes.c_int64(launches_sum)

        # un-timed extras: absorb-only rate, variance latency, parity of the streamed model vs the oracle
        xb, yb = Xs[(Wm + K) * q:(Wm + K + 1) * q], ys[(Wm + K) * q:(Wm + K + 1) * q]
        torch.cuda.synchronize(); ta = time.perf_counter()
        model.condition_on_observations(xb, yb, inplace=True)
        torch.cuda.synchronize(); ta = time.perf_counter() - ta
        # small-batch latencies (the reference driver streams with batch_size 1, config/regression.yaml:22)
        small = {}
        for qs in (1, 64):
            torch.cuda.synchronize(); tq = time.perf_counter()
            for i in range(10):
                xq, yq = Xs[i * qs:(i + 1) * qs], ys[i * qs:(i + 1) * qs]
                model(xq).mean
                model.condition_on_observations(xq, yq, inplace=True)
                model.prediction_cache
            torch.cuda.synchronize()
            small[qs] = (time.perf_counter() - tq) / 10 * 1e3
        # large-batch throughput (SURVEY.md 8d lists q = 16384): the same full step, 6 steps of fresh points
        qL = 16384
        XL, yL = synth_stream(7 * qL, d, 5000 + rank, dev, dtype, args.stream)
        for i in range(7):
            if i == 1:
                torch.cuda.synchronize(); tL = time.perf_counter()
            xq, yq = XL[i * qL:(i + 1) * qL], yL[i * qL:(i + 1) * qL]
            model(xq).mean
            model.condition_on_observations(xq, yq, inplace=True)
            model.prediction_cache
        torch.cuda.synchronize()
        large_rate = 6 * qL / (time.perf_counter() - tL)
    with settings.cg_tolerance(tol), torch.no_grad():
        xv = Xs[:64]
        model(Xs[64:128]).variance                 # warm the 64-column PCG workspace (first call allocates ~0.8 GB)
        torch.cuda.synchronize(); tv = time.perf_counter()
        var = model(xv).variance
        torch.cuda.synchronize(); tv = time.perf_counter() - tv

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        grid = model._grid
        es = 4 if dtype == torch.float32 else 8
        # algorithmic bytes of one k=1 stencil SpMV launch: the symmetric half stencil A_h once
        # (the model's native storage; every entry serves A[i,j] and A[j,i]) + v in + out (+ add)
        spmv_bytes = (grid.R + 1) // 2 * grid.m * es + 3 * grid.m * es
        n_l = int(launches.value)
        avg_ms = tot_ms.value / max(n_l, 1)
        achieved = spmv_bytes / (avg_ms * 1e-3) / 1e9 if n_l else 0.0
        traffic = None   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json), if recorded
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            key = "k_stencil_spmv4_sym<%s, 1, true>" % ("float" if args.dtype == "f32" else "double")
            if args.grid == 50 and d == 3 and key in pmc["kernels"]:
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
        except Exception:  # noqa: BLE001
            traffic = None
        res = {
            "metric": "streaming updates/sec (WISKI, 50^3 inducing grid)",
            "value": world * K * q / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": ("clustered (64 poly-lines, sigma 0.02) " if args.stream == "clustered" else "") + f"3droad-like synthetic stream d={d}, {args.grid}^{d} inducing grid (m={grid.m}), RBF-ARD fixed hypers, "
                                   f"CG solve path, q={q} points/step/GPU, init {args.n_init} points, cg_tol={tol:g}",
                       "batch_per_gpu": q, "global_batch": q * world, "parallelism": (f"dp{world} (" + ("shard all-gather + replicated scatter" if upd.last_exchange == "points"
                                                          else "all-reduce of the half-stencil statistics") + ")") if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": "k_stencil_spmv4_sym (symmetric half-stencil A_h . p inside wiski_pcg)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches": n_l, "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": spmv_bytes,
                         # context only: SURVEY.md 8(d) prices this product at the FULL stencil (R m s + 2 m s); the kernel
                         # computes the same A.p from the symmetric half, so `frac` above uses the bytes it really needs
                         "survey_8d_full_stencil_bytes": grid.R * grid.m * es + 2 * grid.m * es,
                         "full_stencil_equivalent_frac": ((grid.R * grid.m * es + 2 * grid.m * es) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if n_l else 0.0},
            "extra": {"cg_iters_per_step_mean": float(np.mean(iters)), "absorb_only_updates_per_s": q / ta,
                      "variance_ms_per_64_queries": tv * 1e3, "step_ms_q1": small[1], "step_ms_q64": small[64], "updates_per_s_q16384": large_rate, "spmv_time_share": tot_ms.value * 1e-3 / elapsed},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, tol)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)      # the ONE JSON line, last thing written to stdout


if __name__ == "__main__":
    main()
