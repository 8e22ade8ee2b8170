"""CPU baseline (oracle/baseline.py, OpenMP) at different thread counts: the GPU boxes expose 256 logical CPUs but run under a
cgroup CPU quota (cpu.max), so "all threads" can be slower than "as many threads as the quota"."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    import bench
    from oracle import baseline, spec
    q = 4096
    Xt, yt = bench.synth_stream(21743, 3, 0, "cpu", torch.float64, "uniform")
    Xs, ys = bench.synth_stream(q * 8, 3, 1000, "cpu", torch.float64, "uniform")
    B = baseline.StreamingBaseline([[-1.1, 1.1]] * 3, 50, sigma2=spec.SOFTPLUS0 + 1e-4, dtype=np.float32)
    B.absorb(Xt.numpy(), yt.numpy()[:, 0]); B.refresh(1e-4)
    X, y = Xs.numpy(), ys.numpy()[:, 0]
    t0 = time.perf_counter()
    for s in range(6):
        B.predict_mean(X[s * q:(s + 1) * q]); B.absorb(X[s * q:(s + 1) * q], y[s * q:(s + 1) * q]); it, _ = B.refresh(1e-4)
    dt = time.perf_counter() - t0
    print(f"threads {baseline.num_threads():4d}: {6 * q / dt:9.1f} updates/s ({it} iterations in the last step)")
else:
    try: print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except Exception as e: print("no cpu.max", e)
    for t in (8, 16, 32, 64, 128, 256):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, OMP_NUM_THREADS=str(t)))
