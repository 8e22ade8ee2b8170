"""GPU parity of every C-ABI entry point against the CPU oracle (oracle/)."""
import numpy as np
import pytest
import torch

from oracle import cport, spec

pytestmark = pytest.mark.gpu

CASES = [(1, 20), (2, 12), (3, 9), (3, 12), (4, 6)]  # (3, 12) takes the fused-spectral and wide-SpMV paths
DTYPES = [(torch.float64, np.float64, 1e-11), (torch.float32, np.float32, 2e-4)]


def _setup(d, g, tdt, ndt, n=200, seed=0):
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(seed)
    gb = [[-1.1, 1.1]] * d
    grid = grid_ops.GridSpec(gb, g)
    X = rng.uniform(-1.1, 1.1, (n, d)).astype(ndt)  # includes boundary cells
    y = rng.standard_normal(n).astype(ndt)
    noise = rng.uniform(0.5, 2.0, n).astype(ndt)
    B2 = cport.MatrixFreeWISKI(gb, g, sigma2=0.5, dtype=np.float64)
    return grid, X, y, noise, B2, rng


def _t(a, tdt):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", tdt)


@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_interp(d, g, tdt, ndt, tol):
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    err = grid_ops.new_err_flag("cuda")
    idx, val = grid_ops.interp(grid, _t(X, tdt), err)
    ridx, rval = B2.interp(X.astype(np.float64))
    assert int(err.item()) == 0
    # compare as dense rows (boundary ties / fp32 cell flips move taps, not the row)
    Wg = np.zeros((X.shape[0], grid.m)); Wr = np.zeros_like(Wg)
    for p in range(X.shape[0]):
        np.add.at(Wg[p], idx[p].cpu().numpy(), val[p].double().cpu().numpy())
        np.add.at(Wr[p], ridx[p], rval[p])
    assert np.abs(Wg - Wr).max() < max(tol, 1e-12) * 10
    if tdt == torch.float64:
        assert np.array_equal(idx.cpu().numpy(), ridx.astype(np.int32))


def test_interp_out_of_bounds_flag():
    from online_gp_amd import grid_ops

    grid = grid_ops.GridSpec([[-1.0, 1.0]] * 2, 8)
    err = grid_ops.new_err_flag("cuda")
    x = torch.tensor([[0.0, 0.0], [5.0, 0.0]], device="cuda", dtype=torch.float64)
    grid_ops.interp(grid, x, err)
    assert int(err.item()) != 0


@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_gather_and_ell(d, g, tdt, ndt, tol):
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    V = rng.standard_normal((3, grid.m)).astype(ndt)
    err = grid_ops.new_err_flag("cuda")
    out = grid_ops.gather(grid, _t(X, tdt), _t(V, tdt), err)
    ref = B2.gather(X.astype(np.float64), V.astype(np.float64))
    scale = np.abs(ref).max()
    assert np.abs(out.double().cpu().numpy() - ref).max() < tol * scale * 10
    idx, val = grid_ops.interp(grid, _t(X, tdt), err)
    out1 = grid_ops.gather_ell(idx, val, _t(V[0], tdt))
    assert np.abs(out1.double().cpu().numpy() - ref[:, 0]).max() < tol * scale * 10
    # diag form: column p for query p
    Vd = rng.standard_normal((X.shape[0], grid.m)).astype(ndt)
    outd = grid_ops.gather(grid, _t(X, tdt), _t(Vd, tdt), err, diag=True)
    refd = np.einsum("pk,pk->p", B2.gather(X.astype(np.float64), Vd.astype(np.float64)), np.eye(X.shape[0]))
    assert np.abs(outd.double().cpu().numpy() - refd).max() < tol * np.abs(refd).max() * 10
    assert int(err.item()) == 0


@pytest.mark.parametrize("d,g", [(1, 300), (2, 23), (3, 14), (3, 50), (4, 9)])
@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float32, np.float32, 2e-5), (torch.float64, np.float64, 1e-12)])
def test_ell_gather_streaming_kernels_against_the_oracle(d, g, tdt, ndt, tol):
    """The large-row-count forms of the predictive interpolated MVM (BFN:206-210,235; north_star's sparse interpolation SpMM with
    LDS-staged row tiles): k_gather_ell_dma (idx / val rows by LDS-DMA, any idx) and its grid-aware form on the blocked copy of v
    (wiski_gather_ell_grid), against the C oracle's gather on the same points -- a ragged last tile, points in the one-hot boundary
    cells of the grid, non-cubic grids, arbitrary (unstructured) indices through the plain entry."""
    from online_gp_amd import grid_ops
    from oracle import cport

    rng = np.random.default_rng(100 * d + g)
    gs = [g + (q % 2) * 3 for q in range(d)]                      # (non-cubic: the last two dims differ)
    gb = [[-1.1, 1.1]] * d
    grid = grid_ops.GridSpec(gb, gs)
    T = 4 ** d
    n = (1 << 20) // T * 2 + 37                                     # past the streaming kernels' threshold, ragged last tile
    X = rng.uniform(-1.1, 1.1, (n, d))                             # the whole extent: first / last cells take the one-hot branch
    X[:7] = -1.1; X[7:13] = 1.1
    B2 = cport.MatrixFreeWISKI(gb, gs)
    v = rng.standard_normal(grid.m)
    ref = B2.gather(X.astype(ndt).astype(np.float64), v.astype(ndt).astype(np.float64)[None])[:, 0]
    err = grid_ops.new_err_flag("cuda")
    Xt, vt = _t(X.astype(ndt), tdt), _t(v.astype(ndt), tdt)
    idx, val = grid_ops.interp(grid, Xt, err)
    assert int(err.item()) == 0
    scale = np.abs(ref).max()
    same = (val.double() * vt.double()[idx.long()]).sum(1)         # the stored rows' own product in fp64: what both kernels must reproduce
    for name, out in (("plain", grid_ops.gather_ell(idx, val, vt)), ("grid", grid_ops.gather_ell(idx, val, vt, grid=grid))):
        assert np.abs(out.double().cpu().numpy() - ref).max() < 5 * tol * scale, name
        assert float((out.double() - same).abs().max()) < 0.1 * tol * scale, name
    # unstructured rows (every third row's second tap re-pointed): the plain entry's slow path gives the exact sparse product
    idx2 = idx.clone()
    idx2[::3, 1] = idx2[::3, 0]
    want = (val.double() * vt.double()[idx2.long()]).sum(1)
    got = grid_ops.gather_ell(idx2, val, vt)
    assert float((got.double() - want).abs().max()) < tol * scale


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 37, 5000])
def test_gather_zero_and_pcg_zero_regions(tdt, n):
    """wiski_gather_zero (the streaming step's batch-mean gather that also zeroes the following solve's scalar block and
    accumulated partial vector) and wiski_pcg_zero_regions, straight through the C ABI: same means as wiski_gather for k = 1..3
    columns, both regions zeroed and nothing around them touched, `zeroed` reported; 2-D grids fall back to the plain gather."""
    import ctypes

    from online_gp_amd import _hip, grid_ops

    rng = np.random.default_rng(11)
    for d, g in ((3, 12), (2, 16)):
        grid = grid_ops.GridSpec([[-1.1, 1.1]] * d, g)
        X = torch.as_tensor(rng.uniform(-1, 1, (n, d)), device="cuda", dtype=tdt)
        err = grid_ops.new_err_flag("cuda")
        for k in (1, 3):
            V = torch.as_tensor(rng.standard_normal((k, grid.m)), device="cuda", dtype=tdt)
            ref = grid_ops.gather(grid, X, V, err)
            work, _ = grid_ops.PCGWorkspace().get(grid, 1, 50, tdt, torch.device("cuda", torch.cuda.current_device()))
            work.fill_(1)
            p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
            n1, n2 = ctypes.c_int64(), ctypes.c_int64()
            rc = _hip.fn("wiski_pcg_zero_regions", tdt)(grid.ref, ctypes.c_int32(1), ctypes.c_int32(50), _hip.dptr(work), ctypes.c_int32(1),
                                                      ctypes.byref(p1), ctypes.byref(n1), ctypes.byref(p2), ctypes.byref(n2))
            assert rc == 0 and n1.value > 0 and n1.value % 8 == 0
            base = work.data_ptr()
            o1, o2 = p1.value - base, p2.value - base
            nbytes = work.numel() * work.element_size()
            assert 0 <= o2 and o2 + n2.value <= o1 and o1 + n1.value <= nbytes
            out = torch.empty((n, k), device="cuda", dtype=tdt)
            zeroed = ctypes.c_int32(-1)
            rc = _hip.fn("wiski_gather_zero", tdt)(grid.ref, _hip.dptr(X), ctypes.c_int64(n), _hip.dptr(V), ctypes.c_int32(k), _hip.dptr(out),
                                                 _hip.dptr(err), p1, n1, p2, n2, ctypes.byref(zeroed), _hip.stream_ptr(X.device))
            assert rc == 0
            assert torch.allclose(out, ref, rtol=1e-5 if tdt == torch.float32 else 1e-12, atol=1e-6 if tdt == torch.float32 else 1e-13)
            raw = work.view(torch.uint8).reshape(-1)
            assert zeroed.value == (1 if d == 3 else 0)
            if zeroed.value:
                assert int(raw[o1:o1 + n1.value].max()) == 0 and (n2.value == 0 or int(raw[o2:o2 + n2.value].max()) == 0)
                keep = torch.ones(nbytes, dtype=torch.bool, device="cuda")
                keep[o1:o1 + n1.value] = False
                keep[o2:o2 + n2.value] = False
                assert bool((raw[keep] != 0).all())
            else:
                assert bool((raw != 0).all())
        assert int(err.item()) == 0


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
@pytest.mark.parametrize("d,g,n", [(3, 12, 1), (3, 12, 700), (2, 16, 333), (4, 6, 50)])
def test_scatter_stats_step_emits_mean_and_zeroes(tdt, d, g, n):
    """wiski_scatter_stats_step (the absorb of a streaming step) through the C ABI: the same (b, A_half, cnt, res, stats) as
    wiski_scatter_stats_cnt with the residual carry, mean_out = W u as wiski_gather gives it, both regions of
    wiski_pcg_zero_regions zeroed and nothing else in the workspace touched; res = NULL leaves the residual alone."""
    import ctypes

    from online_gp_amd import _hip, grid_ops

    rng = np.random.default_rng(5)
    grid = grid_ops.GridSpec([[-1.1, 1.1]] * d, g)
    dev = torch.device("cuda", torch.cuda.current_device())
    mk = lambda a: torch.as_tensor(a, device="cuda", dtype=tdt)
    X, y = mk(rng.uniform(-1, 1, (n, d))), mk(rng.standard_normal(n))
    noise = mk(rng.uniform(0.5, 2.0, n))
    wa, wb = 1.0 / noise, 1.0 / noise
    u = mk(rng.standard_normal(grid.m))
    H = (grid.R + 1) // 2
    err = grid_ops.new_err_flag("cuda")
    ref = dict(b=torch.zeros(grid.m, device="cuda", dtype=tdt), A=torch.zeros((H, grid.m), device="cuda", dtype=tdt),
               cnt=torch.zeros(grid.m, device="cuda", dtype=tdt), res=torch.zeros(grid.m, device="cuda", dtype=tdt),
               stats=torch.zeros(2, device="cuda", dtype=torch.float64))
    grid_ops.scatter_stats_cnt(grid, X, y, wa, wb, noise, ref["b"], ref["A"], True, ref["cnt"], ref["stats"], err, u=u, res=ref["res"])
    mean_ref = grid_ops.gather(grid, X, u[None], err)[:, 0]
    rtol, atol = (2e-5, 2e-5) if tdt == torch.float32 else (1e-12, 1e-12)
    for with_res in (True, False):
        got = {k: torch.zeros_like(v) for k, v in ref.items()}
        work, _ = grid_ops.PCGWorkspace().get(grid, 1, 50, tdt, dev)
        work.fill_(1)
        p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
        n1, n2 = ctypes.c_int64(), ctypes.c_int64()
        rc = _hip.fn("wiski_pcg_zero_regions", tdt)(grid.ref, ctypes.c_int32(1), ctypes.c_int32(50), _hip.dptr(work), ctypes.c_int32(1),
                                                  ctypes.byref(p1), ctypes.byref(n1), ctypes.byref(p2), ctypes.byref(n2))
        assert rc == 0
        mean = torch.full((n,), float("nan"), device="cuda", dtype=tdt)
        rc = _hip.fn("wiski_scatter_stats_step", tdt)(grid.ref, _hip.dptr(X), _hip.dptr(y), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise), ctypes.c_int64(n),
                                                    _hip.dptr(got["b"]), _hip.dptr(got["A"]), _hip.dptr(got["cnt"]), _hip.dptr(u),
                                                    _hip.dptr(got["res"]) if with_res else None, _hip.dptr(mean), _hip.dptr(got["stats"]),
                                                    _hip.dptr(err), p1, n1, p2, n2, None, ctypes.c_int64(0), None, ctypes.c_int64(0), _hip.stream_ptr(dev))
        assert rc == 0
        assert torch.allclose(mean, mean_ref, rtol=rtol, atol=atol)
        for k in ("b", "A", "cnt", "stats"):
            assert torch.allclose(got[k], ref[k], rtol=rtol, atol=atol), k
        if with_res:
            assert torch.allclose(got["res"], ref["res"], rtol=rtol * 10, atol=atol * 10)
        else:
            assert float(got["res"].abs().max()) == 0.0
        raw = work.view(torch.uint8).reshape(-1)
        o1, o2 = p1.value - work.data_ptr(), p2.value - work.data_ptr()
        assert int(raw[o1:o1 + n1.value].max()) == 0 and (n2.value == 0 or int(raw[o2:o2 + n2.value].max()) == 0)
        keep = torch.ones(raw.numel(), dtype=torch.bool, device="cuda")
        keep[o1:o1 + n1.value] = False
        keep[o2:o2 + n2.value] = False
        assert bool((raw[keep] != 0).all())
    assert int(err.item()) == 0
    # the mean needs u; the residual carry needs u
    rc = _hip.fn("wiski_scatter_stats_step", tdt)(grid.ref, _hip.dptr(X), _hip.dptr(y), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise), ctypes.c_int64(n),
                                                _hip.dptr(got["b"]), _hip.dptr(got["A"]), _hip.dptr(got["cnt"]), None, None, _hip.dptr(mean),
                                                _hip.dptr(got["stats"]), _hip.dptr(err), None, ctypes.c_int64(0), None, ctypes.c_int64(0), None,
                                                ctypes.c_int64(0), None, ctypes.c_int64(0), _hip.stream_ptr(dev))
    assert rc != 0
    # guarded launch (speculation behind a pending solve, wiski_pcg_async_guard): a no-op unless the device word holds the expected value
    guard = torch.tensor([-7], device="cuda", dtype=torch.int64)
    for val, runs in ((-7, False), (7, True)):
        guard.fill_(val)
        got = {k: torch.zeros_like(v) for k, v in ref.items()}
        work.fill_(1)
        mean = torch.full((n,), float("nan"), device="cuda", dtype=tdt)
        rc = _hip.fn("wiski_scatter_stats_step", tdt)(grid.ref, _hip.dptr(X), _hip.dptr(y), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise), ctypes.c_int64(n),
                                                    _hip.dptr(got["b"]), _hip.dptr(got["A"]), _hip.dptr(got["cnt"]), _hip.dptr(u), _hip.dptr(got["res"]),
                                                    _hip.dptr(mean), _hip.dptr(got["stats"]), _hip.dptr(err), p1, n1, p2, n2, _hip.dptr(guard),
                                                    ctypes.c_int64(7), None, ctypes.c_int64(0), _hip.stream_ptr(dev))
        assert rc == 0
        raw = work.view(torch.uint8).reshape(-1)
        if runs:
            assert torch.allclose(mean, mean_ref, rtol=rtol, atol=atol) and torch.allclose(got["A"], ref["A"], rtol=rtol, atol=atol)
            assert int(raw[o1:o1 + n1.value].max()) == 0
        else:
            assert bool(torch.isnan(mean).all()) and all(float(v.abs().max()) == 0.0 for v in got.values())
            assert bool((raw != 0).all())


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
@pytest.mark.parametrize("gs,n", [((12, 12, 12), 9000), ((20, 9, 33), 8192), ((50, 50, 50), 10000), ((8, 64, 8), 12000)])
def test_owner_computes_absorb_equals_the_atomic_form(tdt, gs, n):
    """wiski_scatter_stats_step with a binning workspace (batches >= 8192 points, d = 3): points binned by cell, one block per
    grid line adds every contribution to its rows without memory-side atomics.  Same (b, A_half, cnt, res, stats), mean and
    zeroing as the atomic form up to the order of the additions -- dense cells (long lists), anisotropic grids, the clamped
    boundary cells, points outside the grid (flagged, skipped), and a guarded no-op."""
    import ctypes

    from online_gp_amd import _hip, grid_ops

    rng = np.random.default_rng(31)
    grid = grid_ops.GridSpec([[-1.1, 1.1]] * 3, list(gs))
    dev = torch.device("cuda", torch.cuda.current_device())
    mk = lambda a: torch.as_tensor(a, device="cuda", dtype=tdt)
    Xn = rng.uniform(-1.0999, 1.0999, (n, 3))                   # out to the box edge: clamped boundary stencils
    Xn[: n // 3] = 0.15 * rng.standard_normal((n // 3, 3))      # a dense clump: many points per cell
    Xn = np.clip(Xn, -1.0999, 1.0999)
    Xn[5] = [3.0, 0.0, 0.0]                                      # outside the grid
    X, y = mk(Xn), mk(rng.standard_normal(n))
    noise = mk(rng.uniform(0.5, 2.0, n))
    wa, wb = 1.0 / noise, 1.0 / noise
    u = mk(rng.standard_normal(grid.m))
    H = (grid.R + 1) // 2
    f = _hip.lib().wiski_scatter_bin_bytes
    f.restype = ctypes.c_int64
    nbytes = int(f(grid.ref, ctypes.c_int64(n), ctypes.c_int32(4 if tdt == torch.float32 else 8)))
    assert nbytes > 0
    binw = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

    def run(use_bin, guard=None, expect=0):
        out = dict(b=torch.zeros(grid.m, device="cuda", dtype=tdt), A=torch.zeros((H, grid.m), device="cuda", dtype=tdt),
                   cnt=torch.zeros(grid.m, device="cuda", dtype=tdt), res=torch.zeros(grid.m, device="cuda", dtype=tdt),
                   stats=torch.zeros(2, device="cuda", dtype=torch.float64), mean=torch.full((n,), float("nan"), device="cuda", dtype=tdt),
                   err=grid_ops.new_err_flag("cuda"))
        rc = _hip.fn("wiski_scatter_stats_step", tdt)(grid.ref, _hip.dptr(X), _hip.dptr(y), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise), ctypes.c_int64(n),
                                                    _hip.dptr(out["b"]), _hip.dptr(out["A"]), _hip.dptr(out["cnt"]), _hip.dptr(u), _hip.dptr(out["res"]),
                                                    _hip.dptr(out["mean"]), _hip.dptr(out["stats"]), _hip.dptr(out["err"]), None, ctypes.c_int64(0), None,
                                                    ctypes.c_int64(0), _hip.dptr(guard), ctypes.c_int64(expect), _hip.dptr(binw) if use_bin else None,
                                                    ctypes.c_int64(nbytes if use_bin else 0), _hip.stream_ptr(dev))
        assert rc == 0
        return out

    ref = run(False)
    for rep in range(2):                                         # twice: the cell heads of the first call are stale, not reset
        got = run(True)
        assert int(got["err"].item()) == int(ref["err"].item()) == 3          # one point dropped (2) | any outside (1)
        rtol = 3e-5 if tdt == torch.float32 else 1e-12
        for k in ("b", "A", "cnt", "res"):
            scale = float(ref[k].abs().max())
            assert float((got[k] - ref[k]).abs().max()) <= rtol * scale, (k, rep)
        ok = torch.ones(n, dtype=torch.bool, device="cuda"); ok[5] = False
        assert torch.allclose(got["mean"][ok], ref["mean"][ok], rtol=rtol, atol=rtol)
        assert torch.allclose(got["stats"], ref["stats"], rtol=1e-12)
    guard = torch.tensor([-9], device="cuda", dtype=torch.int64)
    noop = run(True, guard, 9)
    assert all(float(noop[k].abs().max()) == 0.0 for k in ("b", "A", "cnt", "res", "stats")) and bool(torch.isnan(noop["mean"]).all())


@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_scatter_stats(d, g, tdt, ndt, tol):
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    B2.absorb(X.astype(np.float64), y.astype(np.float64), noise.astype(np.float64), init=True)
    err = grid_ops.new_err_flag("cuda")
    b = torch.zeros(grid.m, device="cuda", dtype=tdt)
    A = torch.zeros((grid.R, grid.m), device="cuda", dtype=tdt)
    stats = torch.zeros(2, device="cuda", dtype=torch.float64)
    w = _t(1.0 / noise.astype(np.float64), tdt)
    grid_ops.scatter_stats(grid, _t(X, tdt), _t(y, tdt), w, w, _t(noise, tdt), b, A, stats, err)
    assert int(err.item()) == 0
    assert np.abs(b.double().cpu().numpy() - B2.b).max() < tol * np.abs(B2.b).max() * 10
    assert np.abs(A.double().cpu().numpy() - B2.A).max() < tol * np.abs(B2.A).max() * 10
    assert np.allclose(stats.cpu().numpy(), B2.c_ld, rtol=max(tol, 1e-12) * 10)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("d,g", CASES + [(2, 9), (3, 16)])
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_spmv_and_kron(d, g, tdt, ndt, tol, half):
    """half: the symmetric half-stencil layout (offsets >= centre), the model's native WtW storage."""
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    B2.absorb(X.astype(np.float64), y.astype(np.float64), noise.astype(np.float64), init=True)
    A = _t(B2.A, tdt)
    if half:
        A = grid_ops.half_stencil_from_offset_major(grid, A[(grid.R - 1) // 2:].contiguous())
    for k in (1, 2, 3, 5, 9, 17, 33, 50, 64):       # >= 16 columns, half: the DPP-broadcast SpMM (csrc/spmm_sym_bcast.h)
        V = rng.standard_normal((k, grid.m)).astype(ndt)
        add = rng.standard_normal((k, grid.m)).astype(ndt)
        out = grid_ops.stencil_spmv(grid, A, _t(V, tdt), _t(add, tdt), 0.7)
        ref = B2.stencil_mv(V.astype(np.float64)) + 0.7 * add.astype(np.float64)
        assert np.abs(out.double().cpu().numpy() - ref).max() < tol * np.abs(ref).max() * 10
        outk = grid_ops.kron_toeplitz_mm(grid, _t(B2.tcol, tdt), _t(V, tdt), 1.3)
        refk = B2.kuu_mv(V.astype(np.float64), 1.3)
        assert np.abs(outk.double().cpu().numpy() - refk).max() < tol * np.abs(refk).max() * 10


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("spectral", [False, True, "profile"])
@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float64, np.float64, 1e-8), (torch.float32, np.float32, 2e-3)])
def test_pcg_against_oracle_and_warm_start(d, g, tdt, ndt, tol, spectral, half):
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    B2.absorb(X.astype(np.float64), y.astype(np.float64), noise.astype(np.float64), init=True)
    RHS = np.stack([B2.b, rng.standard_normal(grid.m)])
    Uref, _, _ = B2.solve(RHS, tol=1e-13)
    cg_tol = 1e-10 if tdt == torch.float64 else 1e-6
    A = _t(B2.A, tdt); tc = _t(B2.tcol, tdt)
    if half:
        A = grid_ops.half_stencil_from_offset_major(grid, A[(grid.R - 1) // 2:].contiguous())
    kw = dict(eigen=grid_ops.kron_eigen(grid, tc), shift=X.shape[0] / grid.m) if spectral else {}
    if spectral == "profile":      # separable density-profile preconditioner (generalized eigenbasis X, Z)
        prof = [np.clip(0.2 + np.sin(np.linspace(0.1, 3.0, gq)) ** 2, 1e-2, None) for gq in grid.g]
        kw = dict(eigen=grid_ops.kron_eigen(grid, tc, profiles=prof), shift=X.shape[0] / grid.m)
    U, Z, it, res = grid_ops.pcg(grid, A, tc, 1.0 / B2.sigma2, _t(RHS, tdt), tol=cg_tol, max_iter=500, check_every=5, **kw)
    assert max(res) < cg_tol * 1.01, (it, res)
    assert np.abs(U.double().cpu().numpy() - Uref).max() < tol * np.abs(Uref).max()
    # U = Kt Z
    KZ = grid_ops.kron_toeplitz_mm(grid, tc, Z, 1.0 / B2.sigma2)
    assert (KZ - U).abs().max().item() < 50 * tol * U.abs().max().item()
    # warm start from the solution converges immediately
    U2, Z2, it2, res2 = grid_ops.pcg(grid, A, tc, 1.0 / B2.sigma2, _t(RHS, tdt), U=U.clone(), Z=Z.clone(), warm=True, tol=cg_tol * 10,
                                     max_iter=500, check_every=5, **kw)
    assert it2 <= 5
    assert np.abs(U2.double().cpu().numpy() - Uref).max() < tol * np.abs(Uref).max()


def test_wt_columns():
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(3, 9, torch.float64, np.float64, n=17)
    err = grid_ops.new_err_flag("cuda")
    Wt = grid_ops.wt_columns(grid, _t(X, torch.float64), err)
    idx, val = B2.interp(X)
    ref = np.zeros((17, grid.m))
    for p in range(17):
        np.add.at(ref[p], idx[p], val[p])
    assert np.abs(Wt.cpu().numpy() - ref).max() < 1e-13


@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_symmetric_half_scatter_and_expand(d, g, tdt, ndt, tol):
    """symmetric half-stencil scatter (row-interleaved layout) + unpack == direct full-stencil scatter == oracle."""
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt)
    B2.absorb(X.astype(np.float64), y.astype(np.float64), noise.astype(np.float64), init=True)
    err = grid_ops.new_err_flag("cuda")
    b = torch.zeros(grid.m, device="cuda", dtype=tdt)
    H = (grid.R + 1) // 2
    half = torch.zeros((H, grid.m), device="cuda", dtype=tdt)
    full = torch.zeros((grid.R, grid.m), device="cuda", dtype=tdt)
    stats = torch.zeros(2, device="cuda", dtype=torch.float64)
    w = _t(1.0 / noise.astype(np.float64), tdt)
    grid_ops.scatter_stats_sym(grid, _t(X, tdt), _t(y, tdt), w, w, _t(noise, tdt), b, half, stats, err)
    c = (grid.R - 1) // 2
    half_om = grid_ops.half_stencil_to_offset_major(grid, half)
    assert np.abs(half_om.double().cpu().numpy() - B2.A[c:]).max() < tol * np.abs(B2.A).max() * 10
    assert torch.equal(grid_ops.half_stencil_from_offset_major(grid, half_om), half)      # layout converters are inverses
    diag = half.reshape(-1)[0:4 * grid.m:4]
    assert np.abs(diag.double().cpu().numpy() - B2.A[c]).max() < tol * np.abs(B2.A).max() * 10
    grid_ops.stencil_expand_add(grid, half, full)
    assert float(half.abs().max()) == 0.0
    assert np.abs(full.double().cpu().numpy() - B2.A).max() < tol * np.abs(B2.A).max() * 10
    assert np.abs(b.double().cpu().numpy() - B2.b).max() < tol * np.abs(B2.b).max() * 10
    assert int(err.item()) == 0


@pytest.mark.parametrize("d,g", CASES)
@pytest.mark.parametrize("tdt,ndt,tol", DTYPES)
def test_gather_rows_row_major_operand(d, g, tdt, ndt, tol):
    from online_gp_amd import grid_ops

    grid, X, y, noise, B2, rng = _setup(d, g, tdt, ndt, n=37)
    Vr = rng.standard_normal((grid.m, 11)).astype(ndt)          # asymmetric, row-major [m, ncols]
    err = grid_ops.new_err_flag("cuda")
    out = grid_ops.gather_rows(grid, _t(X, tdt), _t(Vr, tdt), err)
    ref = B2.gather(X.astype(np.float64), Vr.T.astype(np.float64))
    assert out.shape == (37, 11) and int(err.item()) == 0
    assert np.abs(out.double().cpu().numpy() - ref).max() < tol * np.abs(ref).max() * 10


@pytest.mark.parametrize("gs", [(9, 8, 12), (8, 9, 12), (20, 17, 12), (33, 8, 8), (12, 50, 6), (8, 8, 53)])
@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float32, np.float32, 2e-3), (torch.float64, np.float64, 1e-8)])
def test_fused_spectral_pcg_on_anisotropic_grids(gs, tdt, ndt, tol):
    """d = 3 grids with unequal, odd and > 48-node axes: every inner-dimension padding (16/32/48/52/64) and both tile-load
    widths of the fp32 MFMA preconditioner kernels, and the fp64 register-tile ones, against the oracle solve."""
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(11)
    gb = [[-1.1, 1.1]] * 3
    grid = grid_ops.GridSpec(gb, list(gs))
    assert grid.m % 4 == 0                                   # fused path precondition (16-byte fibre loads)
    n = 400
    X = rng.uniform(-1.0, 1.0, (n, 3)); y = rng.standard_normal(n); noise = rng.uniform(0.5, 2.0, n)
    B2 = cport.MatrixFreeWISKI(gb, list(gs), sigma2=0.5, dtype=np.float64)
    B2.absorb(X, y, noise, init=True)
    RHS = np.stack([B2.b, rng.standard_normal(grid.m)])
    Uref, _, _ = B2.solve(RHS, tol=1e-13)
    A = grid_ops.half_stencil_from_offset_major(grid, _t(B2.A, tdt)[(grid.R - 1) // 2:].contiguous())
    tc = _t(B2.tcol, tdt)
    prof = [np.clip(0.3 + np.cos(np.linspace(0.0, 2.5, gq)) ** 2, 1e-2, None) for gq in grid.g]
    cg_tol = 1e-10 if tdt == torch.float64 else 1e-6
    for kw in (dict(eigen=grid_ops.kron_eigen(grid, tc), shift=n / grid.m),
               dict(eigen=grid_ops.kron_eigen(grid, tc, profiles=prof), shift=n / grid.m)):
        U, Z, it, res = grid_ops.pcg(grid, A, tc, 1.0 / B2.sigma2, _t(RHS, tdt), tol=cg_tol, max_iter=500, check_every=5, **kw)
        assert max(res) < cg_tol * 1.01, (it, res)
        assert np.abs(U.double().cpu().numpy() - Uref).max() < tol * np.abs(Uref).max()


@pytest.mark.parametrize("gs,k", [(gs, k) for k in (3, 7, 12) for gs in [(9, 8, 12), (20, 17, 12), (12, 50, 6), (8, 8, 53), (50, 20, 20)]] +
                         [((9, 8, 12), 56), ((12, 50, 6), 64), ((8, 8, 53), 20)])     # >= 16 columns: the broadcast SpMM with the p . Ap epilogue
def test_fused_spectral_pcg_many_columns(gs, k):
    """fp32 solves with >= 3 right-hand sides take the multi-column slab kernel of the preconditioner (k_spec_slab_mfma_mc: a
    block owns a slab for a strided set of columns, both output halves from one forward transform): columns per block 1 and > 1
    (g0 = 50 leaves 5 column groups), plain and generalized eigenbasis, against the oracle solve column by column."""
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(17)
    gb = [[-1.1, 1.1]] * 3
    grid = grid_ops.GridSpec(gb, list(gs))
    n = 400
    X = rng.uniform(-1.0, 1.0, (n, 3)); y = rng.standard_normal(n); noise = rng.uniform(0.5, 2.0, n)
    B2 = cport.MatrixFreeWISKI(gb, list(gs), sigma2=0.5, dtype=np.float64)
    B2.absorb(X, y, noise, init=True)
    RHS = np.concatenate([B2.b[None], rng.standard_normal((k - 1, grid.m))])
    Uref, _, _ = B2.solve(RHS, tol=1e-13)
    tdt = torch.float32
    A = grid_ops.half_stencil_from_offset_major(grid, _t(B2.A, tdt)[(grid.R - 1) // 2:].contiguous())
    tc = _t(B2.tcol, tdt)
    prof = [np.clip(0.3 + np.cos(np.linspace(0.0, 2.5, gq)) ** 2, 1e-2, None) for gq in grid.g]
    for kw in (dict(eigen=grid_ops.kron_eigen(grid, tc), shift=n / grid.m),
               dict(eigen=grid_ops.kron_eigen(grid, tc, profiles=prof), shift=n / grid.m)):
        U, Z, it, res = grid_ops.pcg(grid, A, tc, 1.0 / B2.sigma2, _t(RHS, tdt), tol=1e-6, max_iter=500, check_every=5, **kw)
        assert max(res) < 1e-6 * 1.01, (it, res)
        err = np.abs(U.double().cpu().numpy() - Uref).max(axis=1) / np.abs(Uref).max(axis=1)
        assert err.max() < 2e-3, err


@pytest.mark.parametrize("gs", [(6, 10, 14), (8, 5, 9), (4, 4, 4), (20, 24, 50), (5, 7, 64)])
def test_half_stencil_spmv_dma_kernel_on_3d_grids(gs):
    """fp32, d = 3, one right-hand side takes the LDS-DMA pipelined kernel (csrc/spmv_sym_dma.h): ragged
    last row block, odd innermost sizes, windows wider than a row block; against the C oracle's full-stencil product."""
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(3)
    gb = [[-1.1, 1.1]] * 3
    grid = grid_ops.GridSpec(gb, list(gs))
    assert grid.m % 4 == 0
    n = 500
    X = rng.uniform(-1.1, 1.1, (n, 3))
    y = rng.standard_normal(n)
    noise = rng.uniform(0.5, 2.0, n)
    B2 = cport.MatrixFreeWISKI(gb, list(gs), sigma2=0.5, dtype=np.float64)
    B2.absorb(X, y, noise, init=True)
    A = grid_ops.half_stencil_from_offset_major(grid, _t(B2.A, torch.float32)[(grid.R - 1) // 2:].contiguous())
    for rep in range(2):
        V = rng.standard_normal((1, grid.m))
        add = rng.standard_normal((1, grid.m))
        out = grid_ops.stencil_spmv(grid, A, _t(V, torch.float32), _t(add, torch.float32), -0.3)
        ref = B2.stencil_mv(V) - 0.3 * add
        assert np.abs(out.double().cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()


def test_in_kernel_stamps_of_the_dma_spmv_cover_every_dispatch():
    """Measurement hook (wiski_prof_start / enable / stamps / stamps_raw / stop, include/wiski.h): between start and stop EVERY
    dispatch of k_spmv_sym_dma is stamped by its own waves -- with the events attached (enable 1) or not (enable 0) -- each live wave
    leaves start < end and the compute unit it ran on, padding workgroups leave nothing, and the product is what it is without the hook."""
    import ctypes

    from online_gp_amd import _hip, grid_ops

    lib = _hip.lib()
    rng = np.random.default_rng(5)
    gb = [[-1.1, 1.1]] * 3
    gs = [20, 24, 50]
    grid = grid_ops.GridSpec(gb, gs)
    n = 400
    B2 = cport.MatrixFreeWISKI(gb, gs, sigma2=0.5, dtype=np.float64)
    B2.absorb(rng.uniform(-1.1, 1.1, (n, 3)), rng.standard_normal(n), rng.uniform(0.5, 2.0, n), init=True)
    A = grid_ops.half_stencil_from_offset_major(grid, _t(B2.A, torch.float32)[(grid.R - 1) // 2:].contiguous())
    V = _t(rng.standard_normal((1, grid.m)), torch.float32)
    plain = grid_ops.stencil_spmv(grid, A, V)
    torch.cuda.synchronize()
    assert lib.wiski_prof_start(ctypes.c_int32(8)) == 0
    outs = []
    for on in (1, 0, 1):
        lib.wiski_prof_enable(ctypes.c_int32(on))
        outs.append(grid_ops.stencil_spmv(grid, A, V))
    torch.cuda.synchronize()
    tms, nl = ctypes.c_double(0), ctypes.c_int64(0)
    each = (ctypes.c_double * 8)()
    assert lib.wiski_prof_stamps(ctypes.byref(tms), ctypes.byref(nl), each, ctypes.c_int64(8)) == 0
    assert nl.value == 3 and all(0.5 < each[i] < 1e4 for i in range(3)), (nl.value, list(each)[:3])     # microseconds
    assert abs(sum(each[:3]) * 1e-3 - tms.value) < 1e-9
    nrb = (grid.m + 255) // 256
    for i in range(3):
        nw = ctypes.c_int64(0)
        assert lib.wiski_prof_stamps_raw(ctypes.c_int64(i), None, ctypes.c_int64(0), ctypes.byref(nw)) == 0
        assert nw.value >= 4 * nrb and nw.value % 4 == 0, nw.value                    # 4 parts x (row blocks, padded to the XCD map)
        raw = np.zeros(2 * nw.value, dtype=np.uint64)
        assert lib.wiski_prof_stamps_raw(ctypes.c_int64(i), raw.ctypes.data_as(ctypes.c_void_p), nw, ctypes.byref(nw)) == 0
        start, end = raw[0::2] & np.uint64((1 << 48) - 1), raw[1::2]
        live = end != 0
        assert live.sum() == 4 * nrb, (int(live.sum()), nrb)                           # padding workgroups return before the stamp
        assert (end[live] > start[live]).all()
        span = (int(end[live].max()) - int(start[live].min())) * 1e-2
        assert abs(span - each[i]) < 1e-6, (span, each[i])
        xcc = (raw[0::2][live] >> np.uint64(60)) & np.uint64(15)
        assert xcc.max() <= 7 and len(np.unique(xcc)) >= 2                             # an MI355X has 8 XCDs; workgroups are dealt round-robin
    nw = ctypes.c_int64(-1)
    assert lib.wiski_prof_stamps_raw(ctypes.c_int64(3), None, ctypes.c_int64(0), ctypes.byref(nw)) == 0 and nw.value == 0
    ems, enl = ctypes.c_double(0), ctypes.c_int64(0)
    assert lib.wiski_prof_stop(ctypes.byref(ems), ctypes.byref(enl)) == 0
    assert enl.value == 2 and ems.value > 0                                            # events only where they were switched on
    for o in outs:
        assert torch.equal(o, plain) or (o - plain).abs().max() < 1e-6 * plain.abs().max()   # (the transposed term is accumulated with atomics)
    after = grid_ops.stencil_spmv(grid, A, V)                                          # after stop: no stamps, nothing recorded
    torch.cuda.synchronize()
    assert lib.wiski_prof_stamps(ctypes.byref(tms), ctypes.byref(nl), None, ctypes.c_int64(0)) == 0 and nl.value == 3
    assert (after - plain).abs().max() < 1e-6 * plain.abs().max()


@pytest.mark.parametrize("tdt,tol", [(torch.float32, 2e-5), (torch.float64, 1e-13)])
@pytest.mark.parametrize("k", [16, 49, 64, 100])
@pytest.mark.parametrize("gs", [(6, 10, 14), (8, 5, 9), (4, 4, 4), (20, 24, 50), (5, 7, 64), (4, 5, 7), (30, 10)])
def test_half_stencil_spmm_broadcast_kernel(gs, k, tdt, tol):
    """Products with >= 16 right-hand sides take k_spmm_sym_bcast (csrc/spmm_sym_bcast.h: coefficients by vector loads +
    DPP row broadcast, v_fmac_f32_dpp / v_fmac_f64_dpp; operands in 64-column slices, k = 100: a full slice and one padded
    from 36 columns): ragged last tile (m % 16 != 0), grids smaller than a window (every window clamped / zeroed, coefficient
    spans clamped at both ends of A_h), d = 2 and 3, both precisions; against the C oracle's full-stencil product."""
    from online_gp_amd import grid_ops

    rng = np.random.default_rng(5)
    d = len(gs)
    gb = [[-1.1, 1.1]] * d
    grid = grid_ops.GridSpec(gb, list(gs))
    n = 700
    X = rng.uniform(-1.1, 1.1, (n, d))
    y = rng.standard_normal(n)
    noise = rng.uniform(0.5, 2.0, n)
    B2 = cport.MatrixFreeWISKI(gb, list(gs), sigma2=0.5, dtype=np.float64)
    B2.absorb(X, y, noise, init=True)
    A = grid_ops.half_stencil_from_offset_major(grid, _t(B2.A, tdt)[(grid.R - 1) // 2:].contiguous())
    V = rng.standard_normal((k, grid.m))
    add = rng.standard_normal((k, grid.m))
    out = grid_ops.stencil_spmv(grid, A, _t(V, tdt), _t(add, tdt), 0.4)
    ref = B2.stencil_mv(V) + 0.4 * add
    assert np.abs(out.double().cpu().numpy() - ref).max() < tol * np.abs(ref).max()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-11)])
@pytest.mark.parametrize("gs,k", [((9,), 3), ((6, 11), 4), ((5, 8, 70), 2), ((7, 4, 9), 11), ((4, 5, 4, 6), 3)])
def test_kron_toeplitz_grad_against_autograd(dtype, tol, gs, k):
    """d/d tcol of sum_c x_c^T (kron_q T(tcol_q)) y_c (the MLL backward's contraction, BWM:19-51) against torch autograd on the
    dense Kronecker matrix: tile-Gram form (g <= 64; innermost and strided modes) and the per-element form (g = 70)."""
    from online_gp_amd import grid_ops

    d = len(gs)
    grid = grid_ops.GridSpec(torch.tensor([[-1.0, 1.0]] * d), list(gs))
    g = torch.Generator(device="cpu").manual_seed(sum(gs) + k)
    cols = [torch.rand(n, generator=g, dtype=torch.float64).add(0.1).requires_grad_(True) for n in gs]
    X = torch.randn(k, grid.m, generator=g, dtype=torch.float64)
    Y = torch.randn(k, grid.m, generator=g, dtype=torch.float64)
    K = torch.ones(1, 1, dtype=torch.float64)
    for c in cols:
        n = c.numel()
        idx = (torch.arange(n)[:, None] - torch.arange(n)[None, :]).abs()
        K = torch.kron(K, c[idx])
    ((X @ K) * Y).sum().backward()
    want = torch.cat([c.grad for c in cols])
    tcol = torch.cat([c.detach() for c in cols]).to("cuda", dtype)
    got = grid_ops.kron_toeplitz_grad(grid, tcol, X.to("cuda", dtype), Y.to("cuda", dtype)).cpu()
    assert (got - want).abs().max() < tol * want.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", ["rbf", "matern0.5", "matern1.5", "matern2.5"])
@pytest.mark.parametrize("ard,scaled,d", [(True, True, 3), (False, True, 2), (True, False, 4), (False, False, 1)])
def test_fused_hyper_columns_match_the_op_graph_and_the_oracle(dtype, kind, ard, scaled, d):
    """wiski_stationary_columns / _grad (one launch each) against the broadcasting-op graph they replace (values and the
    gradients autograd derives for lengthscales and outputscale) and against the oracle's closed-form columns (spec.py)."""
    from online_gp_amd import kernels, settings

    torch.manual_seed(d)
    g = [9, 12, 7, 6][:d]
    base = kernels.RBFKernel(ard_num_dims=d if ard else None) if kind == "rbf" else kernels.MaternKernel(nu=float(kind[6:]), ard_num_dims=d if ard else None)
    cov = kernels.ScaleKernel(base) if scaled else base
    gk = kernels.GridInterpolationKernel(cov, grid_size=g, num_dims=d, grid_bounds=[[-1.0, 1.0 + 0.3 * q] for q in range(d)]).to("cuda", dtype)
    with torch.no_grad():
        base.raw_lengthscale.add_(torch.randn_like(base.raw_lengthscale) * 0.3)
        if scaled:
            cov.raw_outputscale.add_(0.4)
    w = torch.randn(sum(gk.grid_spec.g), dtype=torch.float64, device="cuda")
    res = {}
    for fused in (True, False):
        for p in gk.parameters():
            p.grad = None
        with settings.fused_hyper_columns(fused):
            cols = gk.toeplitz_columns(device="cuda")
        assert cols.dtype == torch.float64
        (cols * w).sum().backward()
        res[fused] = (cols.detach().clone(), [p.grad.clone() for p in gk.parameters()])
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert (res[True][0] - res[False][0]).abs().max().item() < tol * max(1.0, res[False][0].abs().max().item())
    for a, b in zip(res[True][1], res[False][1]):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert (a - b).abs().max().item() < (1e-10 if dtype == torch.float64 else 1e-4) * max(1.0, b.abs().max().item())
    # oracle: closed-form columns at the same hyper-parameters
    ell = base.lengthscale.detach().double().cpu().numpy().reshape(-1)
    ell = np.broadcast_to(ell, (d,)) if ell.size == 1 else ell
    s = float(cov.outputscale.detach()) if scaled else 1.0
    g0, h, gs = spec.make_grid([[-1.0, 1.0 + 0.3 * q] for q in range(d)], g)
    okind = {"rbf": "rbf", "matern0.5": "matern12", "matern1.5": "matern32", "matern2.5": "matern52"}[kind]
    ref = np.concatenate(spec.toeplitz_columns(okind, h, gs, ell, s))
    assert np.abs(res[True][0].cpu().numpy() - ref).max() < tol * 10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("n", [1, 7, 256, 1024])
def test_gaussian_metrics_kernel(dtype, tol, n):
    from online_gp_amd import grid_ops

    g = torch.Generator(device="cpu").manual_seed(n)
    mu = torch.randn(n, generator=g, dtype=torch.float64).to("cuda", dtype)
    y = torch.randn(n, generator=g, dtype=torch.float64).to("cuda", dtype)
    var = (torch.rand(n, generator=g, dtype=torch.float64) + 0.05).to("cuda", dtype)
    s2 = torch.tensor([0.3], device="cuda", dtype=dtype)
    out = grid_ops.gaussian_metrics(mu, var, y, s2)
    sq = (mu.double() - y.double()) ** 2
    v = var.double() + 0.3
    ref = torch.stack([sq.mean().sqrt(), (0.5 * (sq / v + v.log() + 1.8378770664093453)).mean()])
    assert (out.double() - ref).abs().max().item() < tol * 10 * max(1.0, ref.abs().max().item())
