#!/usr/bin/env python
"""bench.py -- WISKI streaming updates/s on the 50^3 inducing grid (BASELINE.json
configs[2]: "3droad UCI stream (d=3), 50^3 grid, 1xMI355X fp32, CG solve path").

A *step* is one pass of the hot path over one batch of q streamed points, in the
evaluate-then-update order of the reference's driver
(experiments/regression.py:48-54) at batch granularity:

  1. predictive mean of the incoming batch        (fused gather, wiski_gather)
  2. absorb the batch into the statistics          (wiski_scatter_stats; with
     N > 1 GPUs each rank absorbs its own q points and the ranks exchange --
     weak scaling)
  3. refresh the inducing posterior mean           (wiski_pcg, warm-started)

Hyper-parameters are fixed ("no hyper steps") and predictive
variances are not part of the headline step; the reference-fidelity step
(evaluate mean AND variance -> Adam step on the MLL -> condition) is timed
separately and reported in `extra.reference_step_ms_q*`.  Inputs are resident
in HBM before the timed region.

Timing: W warm-up steps, then R *blocks* of exactly K steps, every block
bracketed by a barrier + torch.cuda.synchronize() on both sides and reduced with
MAX over ranks; `ms_per_step` / `value` = ALL timed steps / the summed time of ALL
timed blocks -- no block is left out (what a mean without the blocks slower than
1.5x the median would read is in `extra`, with those blocks listed).  R is chosen
so that the timed region lasts >= ~0.3 s (a single 20-step block is 5 ms).  The
blocks of a pass are NOT alike -- at 50^3 the first ~45 steps after the init data
need 3 CG iterations, the later ones 2 -- so the block times are bimodal; their
median is reported beside the mean in `extra`.  The stream of a pass never exceeds
UCI 3droad's 434 874 points: when it is used up the model is rebuilt from the init
data (un-timed) and the next pass streams fresh points.

N > 1: `python bench.py --gpus N` spawns its own N ranks (re-executes itself under
torch.distributed.run on 127.0.0.1) when no launcher set WORLD_SIZE; under
`python -m torch.distributed.run ... bench.py --gpus N` it is a rank.  One device
per rank over nccl (= RCCL); a box with fewer devices than ranks is refused unless
WISKI_BENCH_BACKEND=gloo asks for the self-test arrangement (every rank on cuda:0,
gloo; exercises the N > 1 code path, never a reported number).  The line names the
exchange that ran (`config.parallelism`) and what carried it (`collective`).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the half-stencil
SpMV of the CG, HBM-bound; per-dispatch HIP events on its launch stream),
`roofline_secondary` (statistics scatter, ELL gather) and `cpu_baseline` (the
OpenMP port of the same step on all host cores, rank 0, N=1 only).
"""
import argparse
import ctypes
import gc
import json
import math
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


def usable_cpus():
    """CPUs this process may really use: logical CPUs capped by the scheduler affinity and the cgroup CPU quota (cgroup v2
    cpu.max = "<quota> <period>").  The GPU test boxes show 256 logical CPUs under a quota of 16: thread pools sized for 256
    get the whole container throttled for tens of milliseconds at a time, and the OpenMP baseline runs 5x slower on 128
    threads than on 16."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(args, tol, budget_s=20.0, max_steps=100):
    """The same step (predictive mean of the batch -> absorb -> warm-started refresh; same grid, q, init, dtype,
    tolerance) by the OpenMP port on all host cores: oracle/baseline.py.  Bounded sample: steps until `budget_s`."""
    from oracle import baseline, spec

    baseline.build()
    baseline.set_num_threads(usable_cpus())         # 16 on the test boxes, not the 256 logical CPUs they show
    ndt = np.float32 if args.dtype == "f32" else np.float64
    q = args.batch
    Xt, yt = synth_stream(args.n_init, args.dim, 0, "cpu", torch.float64, args.stream)
    Xs, ys = synth_stream(q * max_steps, args.dim, 1000, "cpu", torch.float64, args.stream)
    B = baseline.StreamingBaseline([[-1.1, 1.1]] * args.dim, args.grid, sigma2=spec.SOFTPLUS0 + 1e-4, dtype=ndt)
    B.absorb(Xt.numpy(), yt.numpy()[:, 0])
    B.refresh_profile(tol)                          # cold solve on the init data (not timed), as on the GPU leg
    X, y = Xs.numpy(), ys.numpy()[:, 0]
    steps, iters = 0, []
    t0 = time.perf_counter()
    while steps < max_steps and (steps < 3 or time.perf_counter() - t0 < budget_s):
        lo = steps * q
        B.predict_mean(X[lo:lo + q])
        B.absorb(X[lo:lo + q], y[lo:lo + q])
        it, _ = B.refresh_profile(tol)          # the GPU library's separable density-profile preconditioner, ported (wb_pcg_profile)
        iters.append(it)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps * q / dt, "unit": "updates/s", "cores": baseline.num_threads(), "kind": "port",
            "sample": f"{steps} steps of q={q} after the {args.n_init}-point init (same grid / dtype / tolerance / warm starts as the GPU leg), "
                      f"OpenMP C port on {baseline.num_threads()} threads (= the CPUs the container's quota allows; {os.cpu_count()} logical CPUs are "
                      f"visible, and 128 threads run 5x slower under that quota), {args.stream} stream, CG with the separable density-profile "
                      f"preconditioner of the GPU library ({np.mean(iters):.1f} iterations per step; the GPU leg adds the two-level block on "
                      f"road-like streams, which the port does not have; with the plain Kt preconditioner the port needed 155), {dt:.1f} s"}


def dense_reference_timings(dev):
    """SURVEY.md 8(d) last bullet: the faithful dense restatement of the reference algorithm (B1, oracle/dense_reference.py,
    numpy/LAPACK on the host) beside the GPU's dense regime, at the reference's own grid sizes: one streamed point
    (condition_on_observations) followed by mean + variance at 16 queries."""
    from oracle import dense_reference
    from online_gp_amd.models import FixedNoiseOnlineSKIGP

    out = {}
    for name, d, g in (("m256_16x16", 2, 16), ("m1000_10x10x10", 3, 10)):
        rng = np.random.default_rng(0)
        n0, nst = 64, 12
        X = rng.uniform(-1, 1, (n0 + nst, d)); y = np.sin(2 * X.sum(1)) + 0.1 * rng.standard_normal(n0 + nst)
        Xq = rng.uniform(-1, 1, (16, d))
        gb = [[-1.1, 1.1]] * d
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=usable_cpus()):       # LAPACK threads within the container's CPU quota (see usable_cpus)
            B1 = dense_reference.DenseWISKI(gb, g, sigma2=0.6932)
            B1.set_train_data(X[:n0], y[:n0], np.ones(n0))
            B1.predict(Xq)
            tc = []
            for i in range(nst):
                t0 = time.perf_counter()
                B1.condition_on_observations(X[n0 + i:n0 + i + 1], y[n0 + i:n0 + i + 1])
                B1.predict(Xq)
                tc.append(time.perf_counter() - t0)
        cpu_ms = float(np.median(tc)) * 1e3
        Xt = torch.as_tensor(X, device=dev); yt = torch.as_tensor(y, device=dev)[:, None]; Xqt = torch.as_tensor(Xq, device=dev)
        m = FixedNoiseOnlineSKIGP(Xt[:n0], yt[:n0], torch.ones_like(yt[:n0]), grid_bounds=torch.tensor(gb), grid_size=g, learn_additional_noise=True)
        m.eval()
        with torch.no_grad():
            mv = m(Xqt); mv.mean, mv.variance
            tg = []
            for i in range(nst):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m.condition_on_observations(Xt[n0 + i:n0 + i + 1], yt[n0 + i:n0 + i + 1], inplace=True)
                mv = m(Xqt); mv.mean, mv.variance
                torch.cuda.synchronize(); tg.append(time.perf_counter() - t0)
        out[name] = {"gpu_ms_per_point": float(np.median(tg)) * 1e3, "cpu_dense_reference_ms_per_point": cpu_ms}
    out["note"] = "fp64; CPU = oracle/dense_reference.py (dense W^T, m x m WtW, SVD root update, Cholesky of Q; numpy/LAPACK on the CPUs the container quota allows)"
    return out


def dense_regime_reference_steps(dev):
    """The reference's per-batch loop (evaluate -> Adam step on the MLL -> condition; experiments/regression.py:48-54, OSR:113-146) on the
    SMALL inducing grids every shipped reference configuration uses (BASELINE configs 1 / 4 / 5: 64 nodes, 10^3 Matern-5/2, 30^2 Matern-1/2),
    fp64, q = 1 and 8: ms per step through the device pipeline (settings.spectral_dense_regime), and the same loop with the
    pipeline off (the nodal dense factor, one framework op at a time: round 4's path) for the 64-node case."""
    from online_gp_amd import settings
    from online_gp_amd.kernels import MaternKernel, ScaleKernel
    from online_gp_amd.models import Identity, OnlineSKIRegression

    dt = torch.float64
    out = {}

    def leg(d, g, kind, nsteps=(40, 30)):
        cov = None if kind == "rbf" else ScaleKernel(MaternKernel(nu={"matern12": 0.5, "matern52": 2.5}[kind], ard_num_dims=d)).to(dev)
        X0, y0 = synth_stream(200, d, 0, dev, dt, "uniform")
        Xr, yr = synth_stream(1024, d, 31337, dev, dt, "uniform")
        reg = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, g, 1.0, covar_module=cov)
        res, lo = {}, 0
        for qs, nst in zip((1, 8), nsteps):
            ts = []
            for _ in range(nst):
                xb, yb = Xr[lo:lo + qs], yr[lo:lo + qs]; lo += qs
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reg.evaluate(xb, yb)
                reg.update(xb, yb)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[f"ms_per_step_q{qs}"] = float(np.median(ts[8:])) * 1e3
        fac = reg.gp.__dict__.get("_spectral", {}).get(0)
        gs = reg.__dict__.get("_graphed")
        res["path"] = ("device pipeline: spectral factor of rank %d of m = %d, %d eigenvector refreshes on the device, hyper step replayed %d times as a captured graph%s"
                       % (fac.cur["basis"].r, g ** d, fac.device_refreshes, 0 if gs is None else gs.replays, " (recorded without autograd)" if gs is not None and gs.fused else "")
                       if fac is not None and fac.cur is not None else "nodal dense factor, op by op")
        return res

    for d, g, kind in ((1, 64, "rbf"), (2, 30, "rbf"), (3, 10, "rbf"), (2, 30, "matern12"), (3, 10, "matern52")):
        out[f"{g}^{d}_{kind}"] = leg(d, g, kind)
    with settings.spectral_dense_regime(False):
        out["64^1_rbf_pipeline_off"] = leg(1, 64, "rbf", nsteps=(20, 16))
    out["note"] = "fp64 data, default (fp32) hyper-parameters, lr 1e-3; median over the steady steps; matern12 / matern52 have no spectral gap: full rank"
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command under torch.distributed.run with N ranks on this
    node (rendezvous on 127.0.0.1, a free port) -- exactly the driver's own launch line -- and return its exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher (WORLD_SIZE unset), spawning %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="streamed points per step per GPU (q)")
    ap.add_argument("--grid", type=int, default=50)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--n-init", type=int, default=21743, help="5%% of 434874 (init_ratio of the reference config)")
    ap.add_argument("--tol", type=float, default=None, help="CG relative-residual tolerance")
    ap.add_argument("--stream", default="clustered", choices=["uniform", "clustered"],
                    help="synthetic stream of SURVEY.md 8d; default: the road-like variant (BASELINE config 3 is 3droad), the uniform one goes to extra")
    ap.add_argument("--blocks", type=int, default=0, help="number of timed K-step blocks (0 = enough for ~0.3 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the un-timed extras (profiling runs)")
    ap.add_argument("--sample-every", type=int, default=0, help="n > 0: the SpMV dispatches of every n-th TIMED step carry start/stop events too "
                    "(they cost ~0.8 us per dispatch and skew the waves' start by XCD: default 0 = events only in one extra un-timed block)")
    ap.add_argument("--stamp-dump", default=None, help="write the raw per-wave stamps of every sampled SpMV dispatch here (.npz; tools/stamp_report.py)")
    ap.add_argument("--check-every", type=int, default=None, help="CG iterations between host convergence checks")
    args = ap.parse_args()

    # The host interpreter's cyclic garbage collector pauses for ~80 ms when it runs a FULL collection over torch's and numpy's
    # object graph (seen as one 10 ms "launch" in an event bracket and as 9 ms blocks of ten 0.15 ms steps).  Switching it off is
    # worse: the per-step temporaries that sit in reference cycles then pile up and the caching allocator has to hipMalloc in
    # the middle of a block (also ~80 ms).  So: collect + freeze the long-lived heap at every leg boundary (gc_settle), which
    # leaves the collector only the young objects of the leg to look at.
    # host-side tensor work (stream generation) must not over-subscribe the CPU quota -- which all ranks of a node share
    torch.set_num_threads(max(1, usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))))
    def gc_settle():
        gc.unfreeze()
        gc.collect()
        gc.freeze()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the launcher of its own N ranks (one per GPU) and relays their exit code; rank 0's
        # JSON line goes to the inherited stdout
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (they must agree: the line's n_gpus is the world size)")
    from online_gp_amd.distributed import pick_backend, rccl_info

    # WISKI_BENCH_BACKEND=gloo: self-test arrangement for 1-GPU boxes (every rank on cuda:0, gloo), so that the N > 1 code path
    # can be exercised without a second device -- never used for reported numbers; without it a rank per device over RCCL
    force = os.environ.get("WISKI_BENCH_BACKEND") or ("nccl" if world > 1 else None)
    backend, dev_of = pick_backend(world, torch.cuda.device_count(), force)
    selftest = world > 1 and backend == "gloo"
    local_rank = dev_of(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = None
    if world > 1:
        if selftest:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        collective = rccl_info(device=dev)           # backend, ranks an all-reduce really summed over, RCCL versions
        collective["one_device_per_rank"] = not selftest
        if collective["ranks_seen"] != world:
            raise SystemExit(f"bench.py: the communicator summed over {collective['ranks_seen']} ranks, expected {world}")

    from online_gp_amd import _hip, grid_ops, settings
    from online_gp_amd.distributed import ShardedStatsUpdater
    from online_gp_amd.models import FixedNoiseOnlineSKIGP, Identity, OnlineSKIRegression

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    tol = args.tol if args.tol is not None else (1e-4 if dtype == torch.float32 else 1e-8)
    K, Wm, q, d = args.steps, args.warmup, args.batch, args.dim
    lib = _hip.lib()
    gb = torch.tensor([[-1.1, 1.1]] * d)
    if args.check_every:
        settings.cg_check_every._set_value(args.check_every)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def block_seconds(bs):
        """(seconds per block over the kept blocks, blocks dropped): whole-region mean without host hiccups (> 1.5x the median;
        the two populations of regular blocks are 1.2x apart)."""
        m = float(np.median(bs))
        kept = [b for b in bs if b <= 1.5 * m]
        return float(np.sum(kept)) / len(kept), len(bs) - len(kept)

    N_STREAM = 434874            # UCI 3droad size: a pass never streams more points than the dataset holds

    def run_stream(kind, exchange, blocks, seed0, profile):
        """`blocks` timed blocks of K steps.  A *pass* = fresh model on the init data, cold solve, W un-timed warm-up steps,
        then as many timed K-step blocks as fit into a 3droad-sized stream (N_STREAM points over all ranks); passes are
        repeated on fresh points until `blocks` blocks have been timed (blocks <= 0: enough for ~0.3 s).
        Returns (model, updater, per-block seconds [MAX over ranks], CG iterations per step, SpMV event ms, SpMV launches)."""
        per_pass = max(1, ((N_STREAM - args.n_init) // (q * world) - Wm) // K)
        X0, y0 = synth_stream(args.n_init, d, seed0, dev, dtype, kind)        # identical init on every rank
        block_s, iters = [], []
        ms_sum, n_launch = 0.0, 0
        st_sum, st_n, st_each = 0.0, 0, []
        R, p = blocks, 0
        model = upd = None
        while R <= 0 or len(block_s) < R:
            model = upd = None
            gc_settle()                                    # the previous pass's model (reference cycles) goes before the new one allocates
            model = FixedNoiseOnlineSKIGP(X0, y0, torch.ones_like(y0), grid_bounds=gb, grid_size=args.grid, learn_additional_noise=True)
            model.eval()
            # N > 1, "auto": the stencil-sharded step (the one exchange that divides the step's work; distributed.py; any d, both
            # precisions since round 4); where it does not apply the updater falls back to the point exchange
            ex = exchange
            if ex == "auto" and world > 1 and os.environ.get("WISKI_BENCH_NO_STENCIL_SHARD") != "1":
                ex = "stencil"
            upd = ShardedStatsUpdater(model, equal_shards=True, exchange=ex)   # every rank streams q points per step

            def step(xb, yb):
                if world == 1:                             # evaluate -> absorb -> refresh behind one C-ABI call (wiski_stream_step)
                    mean = model.stream_step(xb, yb)
                    return mean, model._last_iters[0]
                # N > 1: evaluate -> exchange between the ranks -> absorb -> refresh; with the point exchange the gathered batch
                # goes through the same one-call step, with the statistics all-reduce through the three generic calls
                mean = upd.stream_step(xb, yb)
                return mean, (getattr(model, "_last_iters", None) or [0])[0]

            model.prediction_cache                         # cold solve on the init data (not timed)
            gc_settle()
            nb = per_pass if R <= 0 else min(per_pass, R - len(block_s))
            ev_block = profile and p == 0 and args.sample_every <= 0     # one extra, UN-TIMED block whose SpMV dispatches carry events
            Xs, ys = synth_stream((Wm + per_pass * K) * q, d, seed0 + 1000 + 97 * p + rank, dev, dtype, kind)
            lib.wiski_prof_start(ctypes.c_int32(256))      # creates the event pool outside the timed region
            lib.wiski_prof_stop(None, None)
            tw = float("inf")
            for t in range(Wm):                            # un-timed warm-up; the fastest step sizes the number of blocks
                torch.cuda.synchronize(); t0 = time.perf_counter()
                step(Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q])
                torch.cuda.synchronize(); tw = min(tw, time.perf_counter() - t0)
            if R <= 0:                                      # same R on every rank: >= ~0.3 s of timed blocks, <= 2400 steps
                est = torch.tensor([tw if Wm > 0 else 4e-4], dtype=torch.float64, device=dev)
                if world > 1:
                    dist.all_reduce(est, op=dist.ReduceOp.MAX)
                R = int(max(5, math.ceil(0.3 / max(float(est.item()) * K, 1e-6))))
                R = max(3, min(R, 2400 // max(K, 1)))
                nb = min(per_pass, R)
            for r in range(nb):
                barrier()
                if profile:
                    lib.wiski_prof_start(ctypes.c_int32(256))
                    lib.wiski_prof_enable(ctypes.c_int32(0))
                t0 = time.perf_counter()
                for k in range(K):
                    t = Wm + r * K + k
                    sampled = profile and args.sample_every > 0 and t % args.sample_every == 0   # (option) events inside the timed region
                    if sampled:
                        lib.wiski_prof_enable(ctypes.c_int32(1))
                    _, it = step(Xs[t * q:(t + 1) * q], ys[t * q:(t + 1) * q])
                    iters.append(it)
                    if sampled:
                        lib.wiski_prof_enable(ctypes.c_int32(0))
                    if k == K - 1:
                        model._finish_pending()             # a block ends with its last refresh converged, inside the timed region
                barrier()
                block_s.append(time.perf_counter() - t0)
                if profile:
                    # the loop is pipelined (a step returns with its refresh in flight): the events are read here, outside the
                    # timed region, once the block has drained -- reading them after each step would wait for the GPU
                    tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
                    each = (ctypes.c_double * 1024)()
                    if lib.wiski_prof_stamps(ctypes.byref(tms), ctypes.byref(nl), each, ctypes.c_int64(1024)) == 0:      # every SpMV dispatch of the block by its in-kernel stamps
                        st_sum += tms.value
                        st_n += int(nl.value)
                        st_each.extend(each[:min(int(nl.value), 1024)])
                        if args.stamp_dump:                 # (diagnostic: which waves, on which CUs, made a slow dispatch slow)
                            for i in range(1024):
                                nw = ctypes.c_int64(0)
                                lib.wiski_prof_stamps_raw(ctypes.c_int64(i), None, ctypes.c_int64(0), ctypes.byref(nw))
                                if nw.value <= 0:
                                    continue
                                raw = np.zeros(2 * nw.value, dtype=np.uint64)
                                lib.wiski_prof_stamps_raw(ctypes.c_int64(i), raw.ctypes.data_as(ctypes.c_void_p), nw, ctypes.byref(nw))
                                run_stream.raw.append(raw)
                    if lib.wiski_prof_stop(ctypes.byref(tms), ctypes.byref(nl)) == 0:
                        ms_sum += tms.value
                        n_launch += int(nl.value)
            if ev_block:
                # the second clock, outside the timed region: the next K steps of the same stream with start/stop events on every SpMV
                # dispatch (in-kernel stamps show what the events do to the dispatch they ride on: +0.8 us, the waves of six XCDs start
                # 1.2 us after the other two -- profiles/r06_spmv_tail.txt section 5)
                Xe, ye = synth_stream(K * q, d, seed0 + 5000 + rank, dev, dtype, kind)
                barrier()
                lib.wiski_prof_start(ctypes.c_int32(256))
                for k in range(K):
                    step(Xe[k * q:(k + 1) * q], ye[k * q:(k + 1) * q])
                model._finish_pending()
                barrier()
                tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
                if lib.wiski_prof_stop(ctypes.byref(tms), ctypes.byref(nl)) == 0:
                    ms_sum += tms.value
                    n_launch += int(nl.value)
            p += 1
        bt = torch.tensor(block_s, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
        run_stream.stamps = (st_sum, st_n, st_each)
        if args.stamp_dump and profile and rank == 0 and run_stream.raw:
            np.savez_compressed(args.stamp_dump, **{f"d{i:04d}": r for i, r in enumerate(run_stream.raw)})
        return model, upd, bt.tolist(), iters, ms_sum, n_launch

    run_stream.raw = []
    with settings.skip_posterior_variances(True), settings.cg_tolerance(tol), settings.deferred_bounds_check(True), settings.deferred_refresh(True), \
            torch.no_grad():
        # headline: the configured stream; N > 1: the exchange the cost model picks ("auto")
        headline_ex, calib = "auto", None
        if world > 1 and os.environ.get("WISKI_BENCH_EXCHANGE") in ("stencil", "points", "stats"):
            headline_ex = os.environ["WISKI_BENCH_EXCHANGE"]
        elif world > 1 and os.environ.get("WISKI_BENCH_NO_STENCIL_SHARD") != "1":
            # N > 1: two exchanges can carry the step -- the stencil-sharded one (divides the absorb and every A p by N, pays one m-vector
            # all-reduce per CG iteration) and the point exchange (no collective inside the solve; every rank absorbs all N q points
            # with the owner-computes kernel).  Which is faster depends on the node's collective latency: three un-timed blocks of each
            # (block times are MAX over ranks, so every rank sees the same numbers) decide which one the timed headline runs; both
            # are timed again, the same way, in extra.updates_per_s_exchange_*
            calib = {}
            for ex in ("stencil", "points"):
                try:
                    _, _, bs, _, _, _ = run_stream(args.stream, ex, 3, 7, profile=False)
                    calib[ex] = float(np.median(bs)) / K * 1e3
                except Exception as exc:  # noqa: BLE001
                    calib[ex] = None
                    calib[ex + "_error"] = repr(exc)[:200]
            ok = {k_: v_ for k_, v_ in calib.items() if isinstance(v_, float)}
            headline_ex = min(ok, key=ok.get) if ok else "auto"
            if calib.get("stencil") is None:
                os.environ["WISKI_BENCH_NO_STENCIL_SHARD"] = "1"       # (the fallback below and the extras must not try it again)
        try:
            model, upd, block_s, iters, spmv_ms, spmv_n = run_stream(args.stream, headline_ex, args.blocks, 0, profile=os.environ.get("WISKI_BENCH_NOSAMPLE") != "1")
            stamp_ms, stamp_n, stamp_each = run_stream.stamps
            headline_note = None
        except Exception as exc:  # noqa: BLE001
            if world == 1:
                raise
            # N > 1: the stencil-sharded step has only ever run with two ranks on one GPU; if it fails on a real multi-GPU node the
            # line falls back to the point / statistics exchange and says so (every rank takes this branch: the failure is collective)
            headline_note = ("stencil-sharded step failed, fell back: " + repr(exc))[:300]
            os.environ["WISKI_BENCH_NO_STENCIL_SHARD"] = "1"
            model, upd, block_s, iters, spmv_ms, spmv_n = run_stream(args.stream, "auto", args.blocks, 0, profile=False)
            stamp_ms, stamp_n, stamp_each = 0.0, 0, []
        R = len(block_s)
        med = float(np.median(block_s))
        # the headline counts EVERY timed block (round 5; rounds 1-4 left blocks > 1.5x the median out as host hiccups and said so): what a
        # hiccup-free mean would read is in extra, with the blocks it would have left out
        sec = float(np.sum(block_s)) / R
        sec_kept, dropped = block_seconds(block_s)
        extra = {"blocks": R, "blocks_dropped": 0, "block_ms_first_median_last_min": [block_s[0] * 1e3, med * 1e3, block_s[-1] * 1e3, min(block_s) * 1e3],
                 "updates_per_s_median_block": world * K * q / med, "updates_per_s_all_blocks": world * K * q * R / float(np.sum(block_s)),
                 "updates_per_s_without_blocks_over_1.5x_median": world * K * q / sec_kept,
                 "blocks_over_1.5x_median_ms": [b * 1e3 for b in block_s if b > 1.5 * med],
                 "timed_region_s": float(np.sum(block_s)), "cg_iters_per_step_mean": float(np.mean(iters)),
                 "stream_points_per_pass": int(model.num_data), "note": "each pass re-starts from the init data and streams at most a 3droad-sized "
                 "stream (434 874 points) of fresh synthetic points"}
        exchange_used = upd.last_exchange
        tr = model.__dict__.get("_two_level")
        extra["two_level_preconditioner"] = ({"rank": tr.block.r, "refreshes_in_last_pass": tr.block.refreshes, "growth": settings.two_level_growth.value(),
                                              "slabs_exchanging": tr.block.nslab} if tr is not None and tr.block is not None
                                             else "not built (the separable density model converges in < %g iterations on this stream)" % settings.two_level_min_iters.value())
        if headline_note:
            extra["headline_note"] = headline_note
        if calib is not None:
            extra["exchange_calibration_ms_per_step"] = calib
            extra["exchange_chosen"] = headline_ex

        if world > 1 and not args.no_extras:
            try:                                   # same on every rank (deterministic legs): a failure is recorded, the headline line survives
                # both exchanges, timed the same way: the point exchange divides no work (every rank scatters
                # all N q points and solves), the statistics all-reduce is the north-star form
                for ex in ("stencil", "points", "stats"):
                    _, _, bs, its, _, _ = run_stream(args.stream, ex, max(3, R // 4), 0, profile=False)
                    extra[f"updates_per_s_exchange_{ex}"] = world * K * q / (float(np.sum(bs)) / len(bs))
                    extra[f"cg_iters_exchange_{ex}"] = float(np.mean(its))
                    extra[f"cg_iters_exchange_{ex}_last_block"] = float(np.mean(its[-K:]))
                # the part of the path that divides work across ranks: predictive variances shard over the query points
                from online_gp_amd.distributed import sharded_posterior_moments

                Xv, _ = synth_stream(1024, d, 99, dev, dtype, args.stream)            # same queries on every rank
                with settings.skip_posterior_variances(False), settings.variance_cg_tolerance(3e-3):
                    sharded_posterior_moments(model, Xv[:64 * world])                # warm the workspaces
                    barrier(); t0 = time.perf_counter()
                    _, v_sh = sharded_posterior_moments(model, Xv)
                    barrier(); t_sh = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    v_loc = model(Xv).variance
                    barrier(); t_loc = time.perf_counter() - t0
                extra["variance_1024_queries_ms_sharded_over_ranks"] = t_sh * 1e3
                extra["variance_1024_queries_ms_every_rank_alone"] = t_loc * 1e3
                extra["variance_sharded_max_rel_dev"] = float(((v_sh - v_loc).abs() / v_loc).max())
            except Exception as exc:  # noqa: BLE001
                extra.setdefault("errors", []).append(("extras (N > 1 legs): " + repr(exc))[:400])

        roofline_secondary = None
        if world == 1 and not args.no_extras:
            try:                                   # an extra that fails must not cost the JSON line: the error is recorded instead
                # second value: the road-like clustered stream of SURVEY 8d (3droad IS road-like)
                other = "clustered" if args.stream == "uniform" else "uniform"
                _, _, bs, its, _, _ = run_stream(other, "auto", R, 0, profile=False)          # as many blocks as the headline, same seeds
                extra[f"{other}_stream_updates_per_s"] = K * q / (float(np.sum(bs)) / len(bs))
                extra[f"{other}_stream_cg_iters_per_step_mean"] = float(np.mean(its))

                # the reference's own CG tolerance (config/regression.yaml:24-27: cg_tolerance 1e-2; the headline uses 1e-4), and what
                # each tolerance costs in accuracy: predictive mean after the same 24 streamed steps against a 1e-7 solve
                with settings.cg_tolerance(1e-2):
                    _, _, bs, its, _, _ = run_stream(args.stream, "auto", max(3, R // 3), 0, profile=False)
                extra["updates_per_s_at_reference_cg_tolerance_1e-2"] = K * q / (float(np.sum(bs)) / len(bs))
                extra["cg_iters_per_step_mean_at_1e-2"] = float(np.mean(its))
                Xa, ya = synth_stream(args.n_init, d, 0, dev, dtype, args.stream)
                Xb, yb = synth_stream(24 * q, d, 555, dev, dtype, args.stream)
                Xt, _ = synth_stream(4096, d, 556, dev, dtype, args.stream)
                means = {}
                for tl in (1e-7, tol, 1e-2):
                    with settings.cg_tolerance(tl):
                        mt = FixedNoiseOnlineSKIGP(Xa, ya, torch.ones_like(ya), grid_bounds=gb, grid_size=args.grid, learn_additional_noise=True).eval()
                        mt.prediction_cache
                        for i in range(24):
                            mt.stream_step(Xb[i * q:(i + 1) * q], yb[i * q:(i + 1) * q], want_mean=False)
                        mt._finish_pending()
                        means[tl] = mt(Xt).mean.double()
                    del mt
                sc_m = float(means[1e-7].abs().max())
                extra["mean_max_err_vs_1e-7_solve_rel_to_max_abs"] = {f"cg_tol_{tol:g}": float((means[tol] - means[1e-7]).abs().max()) / sc_m,
                                                                       "cg_tol_0.01": float((means[1e-2] - means[1e-7]).abs().max()) / sc_m}
                del means
                torch.cuda.empty_cache()

                gc_settle()
                # absorb-only rate and the scatter kernel by itself (torch events on the launch stream)
                Xe, ye = synth_stream(12 * q, d, 4242, dev, dtype, args.stream)
                torch.cuda.synchronize(); ta = time.perf_counter()
                for i in range(4):
                    model.condition_on_observations(Xe[i * q:(i + 1) * q], ye[i * q:(i + 1) * q], inplace=True)
                torch.cuda.synchronize(); ta = (time.perf_counter() - ta) / 4
                extra["absorb_only_updates_per_s"] = q / ta
                grid = model._grid
                half = torch.zeros(((grid.R + 1) // 2, grid.m), device=dev, dtype=dtype)
                bvec = torch.zeros(grid.m, device=dev, dtype=dtype)
                st = torch.zeros(2, device=dev, dtype=torch.float64)
                ones = torch.ones(q, device=dev, dtype=dtype)
                errf = grid_ops.new_err_flag(dev)
                grid_ops.scatter_stats_sym(grid, Xe[:q], ye[:q, 0].contiguous(), ones, ones, ones, bvec, half, st, errf)
                def event_us(launch, n, reps=5):
                    """median over `reps` event brackets of n launches each (us per launch).  Single brackets are not robust: every
                    20-30 s something stalls the device for ~80 ms (seen as one 10 ms 'launch' in a bracket of 8) on these boxes."""
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ts = []
                    for _ in range(reps):
                        e0.record()
                        for i in range(n):
                            launch(i)
                        e1.record(); torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) / n * 1e3)
                    return float(np.median(ts))

                ycols = [ye[(i + 4) * q:(i + 5) * q, 0].contiguous() for i in range(8)]
                sc_us = event_us(lambda i: grid_ops.scatter_stats_sym(grid, Xe[(i + 4) * q:(i + 5) * q], ycols[i], ones, ones, ones, bvec, half, st, errf), 8)
                es = 4 if dtype == torch.float32 else 8
                T = grid.T
                sc_bytes = ((d + 2) + 2 * T + T * (T + 1)) * es       # read (d + 2 out) s, RMW 2 T s (W^T y), RMW T (T + 1) s (symmetric WtW)
                # the ELL form of the predictive interpolated MVM (north_star's "CSR/COO sparse interpolation SpMM"): idx/val streamed
                nq = 1 << 20
                Xq = torch.rand((nq, d), device=dev, dtype=dtype) * 2 - 1
                idx, val = grid_ops.interp(grid, Xq, errf)
                mu = model.prediction_cache["pred_mean"][0, :, 0].contiguous()
                grid_ops.gather_ell(idx, val, mu)
                ell_us = event_us(lambda i: grid_ops.gather_ell(idx, val, mu), 6)
                grid_ops.gather_ell(idx, val, mu, grid=grid)
                ellg_us = event_us(lambda i: grid_ops.gather_ell(idx, val, mu, grid=grid), 6)
                ell_bytes = nq * (T * (4 + es) + es)
                fused_us = event_us(lambda i: grid_ops.gather(grid, Xq, mu, errf), 6)
                # the owner-computes absorb (scatter_owner.h): what a rank pays for the N q points of a point exchange -- 8 q points here
                own_us = None
                if d == 3:
                    import ctypes as _ct

                    nb_pts = 8 * q
                    Xo, yo = synth_stream(nb_pts * 4, d, 31, dev, dtype, args.stream)
                    yo1 = yo[:, 0].contiguous()
                    oneso = torch.ones(nb_pts, dtype=dtype, device=dev)
                    fbytes = lib.wiski_scatter_bin_bytes
                    fbytes.restype = _ct.c_int64
                    nbin = int(fbytes(grid.ref, _ct.c_int64(nb_pts), _ct.c_int32(es)))
                    binw = torch.zeros(max(nbin, 8), dtype=torch.uint8, device=dev)
                    cntv, resv, uvec, meanv = torch.zeros_like(bvec), torch.zeros_like(bvec), torch.zeros_like(bvec), torch.empty(nb_pts, dtype=dtype, device=dev)
                    fstep = _hip.fn("wiski_scatter_stats_step", dtype)

                    def own(i):
                        sl = slice((i % 4) * nb_pts, (i % 4 + 1) * nb_pts)
                        rc = fstep(grid.ref, _hip.dptr(Xo[sl]), _hip.dptr(yo1[sl]), _hip.dptr(oneso), _hip.dptr(oneso), _hip.dptr(oneso), _ct.c_int64(nb_pts),
                                   _hip.dptr(bvec), _hip.dptr(half), _hip.dptr(cntv), _hip.dptr(uvec), _hip.dptr(resv), _hip.dptr(meanv), _hip.dptr(st),
                                   _hip.dptr(errf), None, _ct.c_int64(0), None, _ct.c_int64(0), None, _ct.c_int64(0), _hip.dptr(binw), _ct.c_int64(nbin),
                                   _hip.stream_ptr(dev))
                        if rc:
                            raise RuntimeError(f"wiski_scatter_stats_step: {rc}")
                    own(0)
                    own_us = event_us(own, 4)
                # the multi-column half-stencil product of the PCG variance / probe solves: A_h once + V in + out
                spmm = {}
                for kc in (16, 64):
                    Vm = torch.randn((kc, grid.m), device=dev, dtype=dtype)
                    Ah = model._kernel_cache["WtW"].stencil
                    grid_ops.stencil_spmv(grid, Ah, Vm)
                    spmm[kc] = event_us(lambda i: grid_ops.stencil_spmv(grid, Ah, Vm), 4)
                    if kc == 64:      # the product kernel alone, by its own dispatch timestamps (the call above adds the input transpose,
                        torch.cuda.synchronize(dev)     # the final beta * add reduction and a stream-ordered 256 MB scratch allocation)
                        lib.wiski_prof_start(ctypes.c_int32(64))
                        for _ in range(8):
                            grid_ops.stencil_spmv(grid, Ah, Vm)
                        torch.cuda.synchronize(dev)
                        tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
                        spmm_kernel_us = tms.value * 1e3 / nl.value if lib.wiski_prof_stop(ctypes.byref(tms), ctypes.byref(nl)) == 0 and nl.value else None
                    del Vm
                spmm_bytes = lambda kc: ((grid.R + 1) // 2 * grid.m + 2 * kc * grid.m) * es
                roofline_secondary = [
                    {"kernel": "k_spmm_sym_bcast (+ k_transpose_cm_rm, k_spmv_reduce): half-stencil A_h . V for 64 right-hand sides (PCG variance / probe solves; coefficients by vector loads + DPP row broadcast, csrc/spmm_sym_bcast.h)",
                     "bound": "hbm", "achieved": spmm_bytes(64) / (spmm[64] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": spmm_bytes(64) / (spmm[64] * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": spmm[64], "algorithmic_bytes_per_launch": spmm_bytes(64),
                     "flop_per_launch": 2 * grid.R * grid.m * 64, "vector_fma_frac_of_157_TFLOPs": 2 * grid.R * grid.m * 64 / (spmm[64] * 1e-6) / 157e12,
                     "k16_us": spmm[16], "k16_frac": spmm_bytes(16) / (spmm[16] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "kernel_only_us": spmm_kernel_us,
                     "kernel_only_frac_of_hbm": None if not spmm_kernel_us else spmm_bytes(64) / (spmm_kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "kernel_only_frac_of_unpacked_vector_fma_78.6_TFLOPs": None if not spmm_kernel_us else 2 * grid.R * grid.m * 64 / (spmm_kernel_us * 1e-6) / 78.6e12,
                     "bound_note": "2.74 G lane-FMAs through v_fmac_f32_dpp (no packed DPP form: 70 us at the unpacked issue rate) and 2.2 GB of V-window rows "
                                   "L2 -> L1; the 150 MB of HBM traffic (19 us) are not what bounds it (DESIGN 3.4)",
                     "timing": "median of 5 torch.cuda.Event brackets of 4 products (wiski_stencil_spmv_sym, back to back: transpose + product + reduction); "
                               "kernel_only_us: start/stop events on the product kernel's own dispatch, 8 launches"},
                    {"kernel": "k_scatter_stats_sym (statistics scatter of q points; memory-side atomics)", "bound": "hbm",
                     "achieved": q * sc_bytes / (sc_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": q * sc_bytes / (sc_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "avg_launch_us": sc_us, "algorithmic_bytes_per_point": sc_bytes, "points_per_s": q / (sc_us * 1e-6),
                     "lane_atomics_per_s": q * (T * (T + 1) // 2 + 2 * T) / (sc_us * 1e-6), "timing": "median of 5 torch.cuda.Event brackets of 8 launches"},
                    {"kernel": "k_gather_ell_dma<float, 16, 8, true> (+ k_ell_pack_v4s): predictive interpolated MVM from stored idx/val rows, 2^20 query rows, "
                               "rows staged through LDS by LDS-DMA, v gathered from its blocked copy (wiski_gather_ell_grid)", "bound": "hbm",
                     "achieved": ell_bytes / (ellg_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ell_bytes / (ellg_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "avg_launch_us": ellg_us, "algorithmic_bytes_per_row": T * (4 + es) + es, "infinity_cache_resident": False,
                     "plain_form_us": ell_us, "plain_form_frac": ell_bytes / (ell_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "plain_form": "k_gather_ell_dma<float, 16, 8, false> (wiski_gather_ell: any idx, v row-major)",
                     "fused_form_rows_per_s": nq / (fused_us * 1e-6), "note": "the product path uses the fused form (weights recomputed from x, 16 B/row); "
                     "what bounds the ELL form is the gather of v behind the stream, not the stream (profiles/r06_gather_ell_probe.txt: 6.4 TB/s with the gathers served by L1)",
                     "timing": "median of 5 torch.cuda.Event brackets of 6 calls (pack + gather)"},
                ]
                # the SAME product where HBM really serves the operand: BASELINE config 2's geometry (d = 4, 30^4, fp64) -- the symmetric half
                # stencil is 1 201 x 810 000 doubles = 7.8 GB, 30x the Infinity Cache, streamed once per launch (the 50^3 fp32 operand
                # of the headline kernel is 86 MB and stays cache-resident)
                try:
                    g4 = grid_ops.GridSpec(torch.tensor([[-1.1, 1.1]] * 4), 30)
                    H4 = (g4.R + 1) // 2
                    A4 = torch.empty((H4, g4.m), device=dev, dtype=torch.float64).uniform_(-1.0, 1.0)
                    V4 = torch.randn((1, g4.m), device=dev, dtype=torch.float64)
                    grid_ops.stencil_spmv(g4, A4, V4)
                    us4 = event_us(lambda i: grid_ops.stencil_spmv(g4, A4, V4), 3)
                    b4 = (H4 * g4.m + 3 * g4.m) * 8
                    torch.cuda.synchronize(dev)              # the product kernel alone (events on its own dispatch: the call adds the reduction of
                    lib.wiski_prof_start(ctypes.c_int32(16))   # the partial vectors and a stream-ordered scratch allocation)
                    for _ in range(6):
                        grid_ops.stencil_spmv(g4, A4, V4)
                    torch.cuda.synchronize(dev)
                    tms4, nl4 = ctypes.c_double(0), ctypes.c_int64(0)
                    k4_us = tms4.value * 1e3 / nl4.value if lib.wiski_prof_stop(ctypes.byref(tms4), ctypes.byref(nl4)) == 0 and nl4.value else None
                    roofline_secondary.append(
                        {"kernel": "k_stencil_spmv4_sym<double> (the symmetric half-stencil A_h . p at BASELINE config 2's geometry: d = 4, 30^4, fp64; A_h = %.2f GB, HBM-served)" % (H4 * g4.m * 8 / 1e9),
                         "bound": "hbm", "achieved": b4 / (us4 * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b4 / (us4 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_us": us4, "algorithmic_bytes_per_launch": b4, "infinity_cache_resident": False,
                         "frac_of_measured_copy_ceiling_6.29_TBs": b4 / (us4 * 1e-6) / 1e9 / 6290.0,
                         "kernel_only_us": k4_us, "kernel_only_frac": None if not k4_us else b4 / (k4_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "kernel_only_frac_of_measured_copy_ceiling_6.29_TBs": None if not k4_us else b4 / (k4_us * 1e-6) / 1e9 / 6290.0,
                         "timing": "median of 5 torch.cuda.Event brackets of 3 products (wiski_stencil_spmv_sym: product + partial-vector reduction), ~1.5 ms each; kernel_only_*: start/stop events on the product kernel's own dispatch, 6 launches"})
                    del A4, V4
                    torch.cuda.empty_cache()
                except Exception as exc:  # noqa: BLE001
                    extra.setdefault("errors", []).append(("extras (30^4 fp64 SpMV leg): " + repr(exc))[:300])
                if own_us is not None:
                    roofline_secondary.append(
                        {"kernel": "k_bin_points + k_owner_lines (owner-computes absorb of 8 q points: the batch a rank absorbs after a point exchange at N = 8)",
                         "bound": "hbm", "achieved": 8 * q * sc_bytes / (own_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": 8 * q * sc_bytes / (own_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": own_us, "algorithmic_bytes_per_point": sc_bytes,
                         "points_per_s": 8 * q / (own_us * 1e-6), "atomic_form_us_for_the_same_points": 8 * sc_us,
                         "timing": "median of 5 torch.cuda.Event brackets of 4 absorbs (two kernels each)"})
            except Exception as exc:  # noqa: BLE001
                extra.setdefault("errors", []).append(("extras (stream / roofline legs): " + repr(exc))[:400])

    if world == 1 and not args.no_extras:
        try:
            # predictive variances of 64 queries, two ways: the product's default path (on this grid and kernel the spectral
            # Woodbury factor, lazy/spectral_woodbury.py: one projection kernel + one fp64 GEMM against the cached factor) and
            # the PCG path (one 64-column solve), each the median of 3 -- a single 2-4 ms measurement can swallow a device stall
            def med3(fn):
                fn()
                ts = []
                for rep in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    out = fn()
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                return float(np.median(ts)), out

            with settings.cg_tolerance(tol), torch.no_grad():
                Xv, _ = synth_stream(1152, d, 99, dev, dtype, args.stream)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                model(Xv[64:128]).variance                 # first request: builds the factor's reference statistics from the stencil
                torch.cuda.synchronize(); t_first = time.perf_counter() - t0
                tv, v_def = med3(lambda: model(Xv[:64]).variance)
                tv1k, _ = med3(lambda: model(Xv[128:1152]).variance)
                fac = model.__dict__.get("_spectral", {}).get(0)
                spectral_on = fac is not None and fac.cur is not None
                with settings.spectral_factor(False):
                    model(Xv[64:128]).variance             # warm the 64-column PCG workspace (first call allocates ~0.8 GB)
                    tp, v_same = med3(lambda: model(Xv[:64]).variance)
                    # quadratic forms converge with the square of the residual: variance solves stopped at 3e-3 (settings.variance_cg_tolerance)
                    with settings.variance_cg_tolerance(3e-3):
                        tq, v_loose = med3(lambda: model(Xv[:64]).variance)
                    with settings.cg_tolerance(1e-7):
                        v_tight = model(Xv[:64]).variance
            extra["variance_ms_per_64_queries"] = tv * 1e3
            extra["variance_ms_per_1024_queries"] = tv1k * 1e3
            extra["variance_path"] = ("spectral Woodbury factor (rank %d of m = %d, trace tail %.0e, left-out prior variance <= %.1e of the variance; "
                                      "first request incl. the factor build: %.1f ms)" % (fac.cur["basis"].r, model._grid.m, fac.cur["tail"], fac.rel_bound(), t_first * 1e3)
                                      if spectral_on else "wiski_pcg, 64 columns per solve")
            extra["variance_ms_per_64_queries_pcg"] = tp * 1e3
            extra["variance_ms_per_64_queries_pcg_tol3e-3"] = tq * 1e3
            extra["variance_rel_err_vs_tight_solve"] = {"default_path": float(((v_def - v_tight).abs() / v_tight).max()),
                                                        "pcg_cg_tol": float(((v_same - v_tight).abs() / v_tight).max()),
                                                        "pcg_variance_cg_tolerance_3e-3": float(((v_loose - v_tight).abs() / v_tight).max())}
            del model
            torch.cuda.empty_cache()
            gc_settle()
            # the reference's timed step at full fidelity (experiments/regression.py:48-54, OSR:56-146): evaluate = predictive
            # mean AND variance (rmse, nll) of the incoming batch, update = one Adam step on the Woodbury MLL + condition
            X0, y0 = synth_stream(args.n_init, d, 0, dev, dtype, args.stream)
            Xr, yr = synth_stream(49152, d, 31337, dev, dtype, args.stream)
            with settings.cg_tolerance(tol), settings.variance_cg_tolerance(3e-3):
                reg = OnlineSKIRegression(Identity(d), X0, y0, 1e-3, args.grid, 1.0)

                def ref_steps(tag, lo0):
                    # (the first steps of a fresh wrapper run eagerly and then record the hyper step into a graph,
                    # models/_graphed_step.py: the median below is over the steady steps after them)
                    for qs, nst in ((1, 24), (64, 16), (1024, 12)):
                        ts = []
                        for i in range(nst):
                            lo = lo0 + i * qs
                            if qs > 1:
                                lo += 64 if qs == 64 else 2048         # past the points the smaller batch sizes used
                            xb, yb = Xr[lo:lo + qs], yr[lo:lo + qs]
                            torch.cuda.synchronize(); t0 = time.perf_counter()
                            reg.evaluate(xb, yb)
                            reg.update(xb, yb)
                            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                        extra[f"reference_step_ms_q{qs}{tag}"] = float(np.median(ts[5:])) * 1e3
                        extra[f"reference_step_updates_per_s_q{qs}{tag}"] = qs / float(np.median(ts[5:]))

                # default path: mean, variance and the MLL with its exact gradient from the spectral factor where it applies
                ref_steps("", 0)
                fac = reg.gp.__dict__.get("_spectral", {}).get(0)
                gs_ = reg.__dict__.get("_graphed")
                extra["reference_step_path"] = ("spectral Woodbury factor for mean / variance / MLL (rank %d, %d reference builds, %d eigenvector refreshes on the device); "
                                                "Cholesky + inverse of the factor as one launch of cooperating workgroups, evaluate() of <= 64 points as one projection + one launch; "
                                                "Adam step on the MLL as a captured HIP graph%s: %d captures, %d replays%s"
                                                % (fac.cur["basis"].r, fac.rebuilds, fac.device_refreshes,
                                                   " recorded without autograd (11 nodes)" if gs_ is not None and gs_.fused else "",
                                                   0 if gs_ is None else gs_.captures,
                                                   0 if gs_ is None else gs_.replays, "" if gs_ is None or gs_.disabled is None else " (disabled: %s)" % gs_.disabled)
                                                if fac is not None and fac.cur is not None else "wiski_pcg (mean, 64-column variance solves, Hutchinson MLL gradient)")
                # the PCG path of the same step (what every kernel / grid falls back to): 64-column variance solves, 10 Hutchinson probes
                with settings.spectral_factor(False):
                    ref_steps("_pcg", 16384)
                    # ... and under the reference's own solver setting (config/regression.yaml:24-27: cg_tolerance 1e-2 for every solve)
                    with settings.cg_tolerance(1e-2):
                        ref_steps("_pcg_at_cg_tolerance_1e-2", 32768)
                gc_settle()
                # the two-output counterpart (online_ski_classifier.py: Dirichlet classifier, per-point noise, predict -> update per batch)
                try:
                    from online_gp_amd.models import OnlineSKIClassifier

                    lab0, labr = (y0 > 0).long().reshape(-1), (yr > 0).long().reshape(-1)
                    clf = OnlineSKIClassifier(Identity(d), X0, lab0, 0.01, 1e-3, args.grid, 1.1)
                    tsc = []
                    for i in range(24):
                        xb, lb = Xr[40000 + 8 * i:40000 + 8 * (i + 1)], labr[40000 + 8 * i:40000 + 8 * (i + 1)]
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        clf.predict(xb)
                        clf.update(xb, lb)
                        torch.cuda.synchronize(); tsc.append(time.perf_counter() - t0)
                    extra["classifier_step_ms_q8"] = float(np.median(tsc[5:])) * 1e3
                    gsc = clf.__dict__.get("_graphed")
                    extra["classifier_step_path"] = "two outputs: %d graph captures, %d replays%s" % (
                        0 if gsc is None else gsc.captures, 0 if gsc is None else gsc.replays,
                        "" if gsc is None or gsc.disabled is None else " (disabled: %s)" % gsc.disabled)
                    del clf
                except Exception as exc:                                     # an extra, never the bench line
                    extra["classifier_step_error"] = repr(exc)[:200]
                gc_settle()
                # small-batch latencies of the headline step (the reference driver streams with batch_size 1, config/regression.yaml:22)
                gp = reg.gp
                with settings.skip_posterior_variances(True), settings.deferred_bounds_check(True), torch.no_grad():
                    for qs in (1, 64, 1024):
                        # the three model calls of the reference surface, one after the other
                        torch.cuda.synchronize(); tq = time.perf_counter()
                        for i in range(10):
                            lo = (2048 + i * qs) % (Xr.shape[0] - qs)
                            xq, yq = Xr[lo:lo + qs], yr[lo:lo + qs]
                            gp(xq).mean
                            gp.condition_on_observations(xq, yq, inplace=True)
                            gp.prediction_cache
                        torch.cuda.synchronize()
                        extra[f"step_ms_q{qs}_three_calls"] = (time.perf_counter() - tq) / 10 * 1e3
                        # the same step behind ONE C-ABI call with the deferred poll (what the headline loop uses)
                        with settings.deferred_refresh(True):
                            tb = []
                            for rep in range(8):                                  # rep 0 warms the path; median of 7 blocks of 10 steps
                                torch.cuda.synchronize(); tq = time.perf_counter()  # (a preconditioner-profile refresh lands in a block now and then)
                                for i in range(10):
                                    lo = (2048 + 640 + (rep * 10 + i) * qs) % (Xr.shape[0] - qs)
                                    gp.stream_step(Xr[lo:lo + qs], yr[lo:lo + qs])
                                gp._finish_pending()
                                torch.cuda.synchronize()
                                tb.append((time.perf_counter() - tq) / 10 * 1e3)
                            extra[f"step_ms_q{qs}"] = float(np.median(tb[1:]))
                            extra[f"step_ms_q{qs}_max_block"] = float(max(tb[1:]))
                    # large-batch throughput (SURVEY.md 8d lists q = 16384): the same full step, 6 steps of fresh points
                    qL = 16384
                    XL, yL = synth_stream(7 * qL, d, 5000, dev, dtype, args.stream)
                    with settings.deferred_refresh(True):             # the one-call step of the headline (owner-computes absorb at this size)
                        gp.stream_step(XL[:qL], yL[:qL])
                        gp._finish_pending()
                        torch.cuda.synchronize(); tL = time.perf_counter()
                        for i in range(1, 7):
                            gp.stream_step(XL[i * qL:(i + 1) * qL], yL[i * qL:(i + 1) * qL])
                        gp._finish_pending()
                        torch.cuda.synchronize()
                    extra["updates_per_s_q16384"] = 6 * qL / (time.perf_counter() - tL)
                del reg, gp
                torch.cuda.empty_cache()
            extra["dense_regime"] = dense_reference_timings(dev)
            gc_settle()
            extra["dense_regime"]["reference_step"] = dense_regime_reference_steps(dev)
        except Exception as exc:  # noqa: BLE001
            extra.setdefault("errors", []).append(("extras (variance / reference step / dense legs): " + repr(exc))[:400])

    if rank == 0:
        from online_gp_amd.grid_ops import GridSpec

        grid = GridSpec(gb, args.grid)
        es = 4 if dtype == torch.float32 else 8
        # algorithmic bytes of one k=1 stencil SpMV launch: the symmetric half stencil A_h once
        # (the model's native storage; every entry serves A[i,j] and A[j,i]) + v in + out (+ add)
        spmv_bytes = (grid.R + 1) // 2 * grid.m * es + 3 * grid.m * es
        if world > 1 and exchange_used == "stencil":
            # stencil-sharded step: this rank's launch streams only ITS groups of the half stencil (+ the vectors)
            from online_gp_amd.grid_ops import half_stencil_group_slices, shard_groups

            g_lo, g_hi = shard_groups(d, 0, world)
            spmv_bytes = sum(b - a for a, b in half_stencil_group_slices(grid, g_lo, g_hi)) * es + 3 * grid.m * es
        # two clocks: (a) stamps taken inside the kernel -- earliest wave start to latest wave end of a dispatch (100 MHz wall clock,
        # wiski_prof_stamps): the kernel alone, what rocprofv3's begin / end timestamps measure too.  They cost nothing measurable, so
        # EVERY SpMV dispatch of the timed blocks carries them (a stride-4 sample of the steps, the round-6 first cut, read 0.55-0.60
        # from one run to the next with ~115 dispatches: profiles/r06_spmv_tail.txt section 4);
        # (b) the start / stop HIP events attached to the dispatch packet, which bracket [predecessor complete -> this kernel complete]
        # and so contain the ~2.4 us of dispatch latency in front of the first wave.  The events are not free -- the stamps of the
        # dispatches they ride on read 0.8 us more, the waves of six XCDs start 1.2 us after those of the other two, the step is 5 %
        # slower with all of them on (profiles/r06_spmv_tail.txt section 5) -- so they are taken in ONE extra block of K steps of the
        # same stream, outside the timed region.  `achieved` / `frac` use (a) where the kernel is stamped (k_spmv_sym_dma); (b) is
        # reported beside it (event_*).
        ev_ms = spmv_ms / max(spmv_n, 1)
        ev_achieved = spmv_bytes / (ev_ms * 1e-3) / 1e9 if spmv_n else 0.0
        stamped = stamp_n > 0
        avg_ms = stamp_ms / stamp_n if stamped else ev_ms
        achieved = spmv_bytes / (avg_ms * 1e-3) / 1e9 if (spmv_n or stamped) else 0.0
        dma = dtype == torch.float32 and d == 3 and grid.m % 4 == 0 and os.environ.get("WISKI_SYM_DMA", "1") != "0"
        kname = "k_spmv_sym_dma<2, true>" if dma else "k_stencil_spmv4_sym<%s, 1, true>" % ("float" if args.dtype == "f32" else "double")
        # HBM bytes per launch by the PMC counters: collected in separate rocprofv3 --pmc passes (tools/pmc_traffic.py),
        # NOT in this run -- read back from the committed profile and labelled with its source
        traffic, traffic_source = None, None
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if args.grid == 50 and d == 3 and kname in pmc["kernels"]:
                    traffic, traffic_source = pmc["kernels"][kname]["hbm_bytes_per_launch"], "profiles/" + fn + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed with the round: NOT measured in this run)"
                    break
            except Exception:  # noqa: BLE001
                pass
        # the same kernel's average duration in the committed rocprofv3 kernel trace of this command (dispatch timestamps: they
        # exclude the ~1 us of packet processing that the start / stop events above include) -- read back, labelled with its source
        rp_us, rp_src = None, None
        try:
            import csv

            for fn in ("r06_bench_kernel_stats.csv", "r05_bench_kernel_stats.csv", "r04_bench_kernel_stats.csv", "r03_bench_kernel_stats.csv", "r02_bench_kernel_stats.csv"):
                if not os.path.exists(os.path.join(ROOT, "profiles", fn)):
                    continue
                with open(os.path.join(ROOT, "profiles", fn)) as fh:
                    for row in csv.DictReader(fh):
                        if args.grid == 50 and d == 3 and args.dtype == "f32" and row["Name"].replace("void ", "").startswith(kname):
                            rp_us, rp_src = float(row["AverageNs"]) / 1e3, f"profiles/{fn} (rocprofv3 --kernel-trace --stats of this command, the take committed with the round: NOT measured in this run)"
                            break
                break
        except Exception:  # noqa: BLE001
            pass
        par = "single"
        if world > 1:
            par = f"dp{world} (" + {"points": "shard all-gather + replicated scatter: divides no work, every rank scatters all N q points and solves",
                                    "stencil": "shard all-gather, then every rank scatters and multiplies only its 1/N of the half-stencil groups; one m-vector "
                                               "all-reduce per CG iteration (wiski_shard)",
                                    "stats": "all-reduce of the half-stencil statistics"}.get(exchange_used, str(exchange_used)) + ")"
        # what an empty dispatch costs by the clock of `roofline.avg_launch_us` (events attached to the dispatch packet)
        empty_us = ctypes.c_double(0)
        if lib.wiski_prof_empty(ctypes.c_int32(64), ctypes.byref(empty_us), _hip.stream_ptr(dev)) != 0:
            empty_us.value = float("nan")
        net_us = ev_ms * 1e3 - empty_us.value
        res = {
            "metric": "streaming updates/sec (WISKI, 50^3 inducing grid)",
            "timed_region_s": float(np.sum(block_s)),
            "timed_steps": R * K,
            "value": world * K * q / sec,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": sec / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "collective": collective,
            "config": {"workload": ("road-like CLUSTERED (points along 64 poly-lines, sigma 0.02: SURVEY 8d's stand-in for UCI 3droad; the uniform variant: extra.uniform_stream_*) " if args.stream == "clustered" else "UNIFORM (the road-like clustered variant of SURVEY 8d: extra.clustered_stream_*) ") + f"3droad-sized synthetic stream d={d}, {args.grid}^{d} inducing grid (m={grid.m}), RBF-ARD fixed hypers, "
                                   f"CG solve path, q={q} points/step/GPU, init {args.n_init} points, cg_tol={tol:g}; {R} timed blocks of {K} steps (all steps / summed block time)",
                       "batch_per_gpu": q, "global_batch": q * world, "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": kname + " (symmetric half-stencil A_h . p inside wiski_pcg)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "launches": stamp_n if stamped else spmv_n, "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": spmv_bytes,
                         "clock": ("in-kernel stamps: earliest wave start -> latest wave end of EVERY dispatch of the timed blocks (s_memrealtime, 100 MHz), wiski_prof_stamps"
                                   if stamped else "HIP events attached to each dispatch (this kernel carries no in-kernel stamps)"),
                         "median_launch_us": float(np.median(stamp_each)) if (stamped and stamp_each) else None,
                         "frac_at_median_launch": (spmv_bytes / (float(np.median(stamp_each)) * 1e-6) / 1e9 / HBM_PEAK_GBS) if (stamped and stamp_each) else None,
                         "launches_over_1.25x_median": int(np.sum(np.asarray(stamp_each) > 1.25 * np.median(stamp_each))) if (stamped and stamp_each) else None,
                         "launch_us_percentiles_10_25_50_75_90_95_99": [float(v) for v in np.percentile(stamp_each, [10, 25, 50, 75, 90, 95, 99])] if (stamped and stamp_each) else None,
                         "event_launches": spmv_n, "event_avg_launch_us": ev_ms * 1e3, "event_achieved": ev_achieved, "event_frac": ev_achieved / HBM_PEAK_GBS,
                         # everything in `committed_take` (and `traffic` above) was read back from files under profiles/: a separate run of
                         # this command under rocprofv3, committed with the round -- evidence beside this run's own events, not part of them
                         "committed_take": {"rocprofv3_avg_launch_us": rp_us, "rocprofv3_frac": (spmv_bytes / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp_us else None,
                                            "source": rp_src},
                         # how to read `frac`: the 86 MB operand is re-read every launch and is served by the 256 MB Infinity
                         # Cache, not by HBM (a bare LDS-DMA read of it runs at 7.2-8 TB/s, profiles/r02_stream_ubench.txt);
                         # and every per-dispatch time contains what an empty dispatch costs by the same clock
                         "infinity_cache_resident": bool(spmv_bytes < 200e6),
                         "empty_dispatch_us": empty_us.value,
                         "net_of_empty_dispatch_frac": (spmv_bytes / (net_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if net_us > 0 else None,
                         "timing": "avg_launch_us / achieved / frac: the kernel's own begin / end stamps (first wave started -> last wave finished) of every SpMV dispatch of the timed blocks, read once per block after it has drained; event_*: the start/stop HIP events attached to every SpMV dispatch of one extra, un-timed block of the same stream (hipExtLaunchKernel; inside the timed region they would perturb what they measure: +0.8 us per dispatch by the stamps), whose pair brackets [predecessor complete -> this kernel complete] and so contains the dispatch latency in front of the first wave (an empty kernel reads ~4 us by it)",
                         # context only: SURVEY.md 8(d) prices this product at the FULL stencil (R m s + 2 m s); the kernel
                         # computes the same A.p from the symmetric half, so `frac` above uses the bytes it really needs
                         "survey_8d_full_stencil_bytes": grid.R * grid.m * es + 2 * grid.m * es},
            "extra": extra,
        }
        if roofline_secondary:
            res["roofline_secondary"] = roofline_secondary
            # which kernel the timed step spends most of its time in, from this run's own measurements: the absorb (one launch per
            # step) against the SpMVs of the solve (iterations x per-dispatch time); the roofline kernel is the latter (the metric's MVM)
            sc = next((r_ for r_ in roofline_secondary if r_["kernel"].startswith("k_scatter_stats_sym")), None)
            if sc is not None and spmv_n:
                step_us = sec / K * 1e6
                it_mean = float(np.mean(iters))
                res["dominant_by_time"] = {"kernel": "k_scatter_stats_sym (absorb of q points)", "us_per_step": sc["avg_launch_us"],
                                           "share_of_step": sc["avg_launch_us"] / step_us,
                                           "roofline_kernel_us_per_step": it_mean * avg_ms * 1e3, "roofline_kernel_share_of_step": it_mean * avg_ms * 1e3 / step_us,
                                           "note": "kernel-only times of this run (event brackets / per-dispatch events) over the measured ms_per_step"}
        # SURVEY 8(d)'s batch sizes for the plain step (evaluate mean -> absorb -> refresh, fixed hyper-parameters), beside the headline's q
        if all(k_ in extra for k_ in ("step_ms_q1", "step_ms_q64", "step_ms_q1024", "updates_per_s_q16384")):
            extra["plain_step_updates_per_s_by_q"] = {"1": 1e3 / extra["step_ms_q1"], "64": 64e3 / extra["step_ms_q64"], "1024": 1024e3 / extra["step_ms_q1024"],
                                                      str(q): world * K * q / sec, "16384": extra["updates_per_s_q16384"]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, tol)
            except Exception as exc:  # noqa: BLE001
                extra.setdefault("errors", []).append(("cpu_baseline: " + repr(exc))[:400])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res), flush=True)      # the ONE JSON line, last thing written to stdout


if __name__ == "__main__":
    main()
