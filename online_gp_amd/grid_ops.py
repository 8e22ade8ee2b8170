"""Torch-tensor front-end of the C ABI (include/wiski.h): one function per
entry point, device pointers borrowed from contiguous ROCm tensors, launches on
torch's current stream.  Host-side plumbing only -- all arithmetic happens in
the HIP kernels under csrc/.
"""
import ctypes
import math
import warnings

import torch

from . import _hip


class GridSpec:
    """Inducing-grid geometry.  Mirrors gpytorch's GridInterpolationKernel grid
    (reference call site online_gp/models/batched_fixed_noise_online_gp.py:114-120):
    ``delta=(hi-lo)/(g-2)``, ``grid=linspace(lo-delta, hi+delta, g)``."""

    def __init__(self, grid_bounds, grid_size):
        gb = [[float(lo), float(hi)] for lo, hi in (torch.as_tensor(grid_bounds, dtype=torch.float64).reshape(-1, 2).tolist())]
        d = len(gb)
        if not 1 <= d <= _hip.MAX_DIM:
            raise ValueError(f"grid dimension must be 1..{_hip.MAX_DIM}, got {d}")
        if isinstance(grid_size, int):
            g = [grid_size] * d
        else:
            g = [int(v) for v in grid_size]
        if len(g) != d:
            raise ValueError("grid_size / grid_bounds mismatch")
        if min(g) < 4:
            raise ValueError("cubic interpolation needs at least 4 grid points per dim")
        self.d = d
        self.g = g
        self.grid_bounds = gb
        delta = [(hi - lo) / (gi - 2) for (lo, hi), gi in zip(gb, g)]
        self.g0 = [lo - dl for (lo, hi), dl in zip(gb, delta)]
        self.h = [((hi + dl) - g0) / (gi - 1) for (lo, hi), dl, g0, gi in zip(gb, delta, self.g0, g)]
        from . import settings

        if settings.float32_grid.on():          # gpytorch's float32 grid, promoted
            for q, ((lo, hi), gi) in enumerate(zip(gb, g)):
                dl = torch.tensor(hi - lo, dtype=torch.float32) / (gi - 2)
                pts = torch.linspace(float(torch.tensor(lo, dtype=torch.float32) - dl), float(torch.tensor(hi, dtype=torch.float32) + dl), gi,
                                     dtype=torch.float32)
                self.g0[q] = float(pts[0])
                self.h[q] = float(pts[1] - pts[0])
        self.m = int(math.prod(g))
        self.T = 4 ** d
        self.R = 7 ** d
        c = _hip.wiski_grid()
        c.d = d
        for q in range(d):
            c.g[q] = g[q]
            c.g0[q] = self.g0[q]
            c.h[q] = self.h[q]
        self.c = c

    @property
    def ref(self):
        return ctypes.byref(self.c)

    def grid_points(self, dtype=torch.float64, device="cpu"):
        """Per-dim 1-D grids (list of d tensors), as gpytorch's ``covar_module.grid``."""
        return [self.g0[q] + self.h[q] * torch.arange(self.g[q], dtype=dtype, device=device) for q in range(self.d)]

    def stencil_offsets(self, device="cpu"):
        offs = torch.zeros(1, dtype=torch.int64)
        stride = [int(math.prod(self.g[q + 1:])) for q in range(self.d)]
        for q in range(self.d):
            offs = (offs[:, None] + (torch.arange(7) - 3)[None, :] * stride[q]).reshape(-1)
        return offs.to(device)


def new_err_flag(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def read_flag(flag):
    """Host value of a device int32 flag after everything queued so far on the current stream (no stream sync)."""
    val = ctypes.c_int32(0)
    rc = _hip.lib().wiski_read_flag(_hip.dptr(flag), ctypes.byref(val), _hip.stream_ptr(flag.device))
    _hip.check(rc, "wiski_read_flag")
    return int(val.value)


def _x2d(x, grid):
    if x.dim() != 2 or x.shape[1] != grid.d:
        raise ValueError(f"expected inputs of shape [n, {grid.d}], got {tuple(x.shape)}")
    return x.contiguous()


def interp(grid, x, err):
    """(idx int32 [n,T], val [n,T]) -- a1."""
    x = _x2d(x, grid)
    n = x.shape[0]
    idx = torch.empty((n, grid.T), dtype=torch.int32, device=x.device)
    val = torch.empty((n, grid.T), dtype=x.dtype, device=x.device)
    rc = _hip.fn("wiski_interp", x.dtype)(grid.ref, _hip.dptr(x), ctypes.c_int64(n), _hip.dptr(idx), _hip.dptr(val), _hip.dptr(err),
                                          _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_interp")
    return idx, val


def gather(grid, x, V, err, diag=False):
    """W(x) @ V for V given as [k, m]; returns [n, k] (or [n] when diag) -- a14."""
    x = _x2d(x, grid)
    V = V.contiguous()
    if V.dim() == 1:
        V = V[None]
    k = V.shape[0]
    n = x.shape[0]
    assert V.shape[1] == grid.m and V.dtype == x.dtype
    out = torch.empty((n,) if diag else (n, k), dtype=x.dtype, device=x.device)
    rc = _hip.fn("wiski_gather", x.dtype)(grid.ref, _hip.dptr(x), ctypes.c_int64(n), _hip.dptr(V), ctypes.c_int32(k), ctypes.c_int32(int(diag)),
                                          _hip.dptr(out), _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_gather")
    return out


def gather_grad(grid, x, V, diag=False):
    """d/dx of W(x_p) . V_c  (c = 0, or c = p when diag): [n, d]."""
    x = _x2d(x, grid)
    V = V.contiguous()
    out = torch.empty((x.shape[0], grid.d), dtype=x.dtype, device=x.device)
    rc = _hip.fn("wiski_gather_grad", x.dtype)(grid.ref, _hip.dptr(x), ctypes.c_int64(x.shape[0]), _hip.dptr(V), ctypes.c_int32(int(diag)),
                                               _hip.dptr(out), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_gather_grad")
    return out


class InterpDot(torch.autograd.Function):
    """f(x)_p = W(x_p) . V_c (c = 0 or c = p), differentiable w.r.t. the inputs x only."""

    @staticmethod
    def forward(ctx, grid, x, V, diag, err):
        xd = x.detach().contiguous()
        Vd = V.detach().contiguous().reshape(-1, grid.m)
        ctx.grid, ctx.diag = grid, diag
        ctx.save_for_backward(xd, Vd)
        out = gather(grid, xd, Vd, err, diag=diag)
        return out if diag else out[:, 0]

    @staticmethod
    def backward(ctx, g):
        xd, Vd = ctx.saved_tensors
        gx = gather_grad(ctx.grid, xd, Vd, diag=ctx.diag) * g[:, None]
        return None, gx, None, None, None


def gather_rows(grid, x, Vr, err):
    """W(x) @ Vr for a row-major dense operand Vr [m, ncols]; returns [n, ncols]."""
    x = _x2d(x, grid)
    Vr = Vr.contiguous()
    assert Vr.shape[0] == grid.m and Vr.dtype == x.dtype
    out = torch.empty((x.shape[0], Vr.shape[1]), dtype=x.dtype, device=x.device)
    rc = _hip.fn("wiski_gather_rows", x.dtype)(grid.ref, _hip.dptr(x), ctypes.c_int64(x.shape[0]), _hip.dptr(Vr), ctypes.c_int32(Vr.shape[1]),
                                               _hip.dptr(out), _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_gather_rows")
    return out


_ELL_PACK = {}


def gather_ell(idx, val, v, grid=None):
    """out[r] = sum_t val[r, t] * v[idx[r, t]] from stored interpolation rows (a14, ELL form; BFN:206-210).  With ``grid`` -- the
    rows were written by :func:`interp` on that grid -- large row counts go through ``wiski_gather_ell_grid``: v is first copied
    into a blocked layout in which a row's taps touch about half as many cache lines (csrc/gather_ell_dma.h)."""
    n, T = idx.shape
    out = torch.empty((n,), dtype=val.dtype, device=val.device)
    v = v.contiguous()
    if grid is not None and grid.d >= 2 and T == grid.T:
        f = _hip.lib().wiski_gather_ell_pack_elems
        f.restype = ctypes.c_int64
        ne = int(f(grid.ref))
        key = (val.device, val.dtype)
        pack = _ELL_PACK.get(key)
        if ne and (pack is None or pack.numel() < ne):
            pack = _ELL_PACK[key] = torch.empty((ne,), dtype=val.dtype, device=val.device)
        rc = _hip.fn("wiski_gather_ell_grid", val.dtype)(grid.ref, _hip.dptr(idx), _hip.dptr(val), ctypes.c_int64(n), _hip.dptr(v),
                                                         _hip.dptr(pack) if ne else None, _hip.dptr(out), _hip.stream_ptr(val.device))
        _hip.check(rc, "wiski_gather_ell_grid")
        return out
    rc = _hip.fn("wiski_gather_ell", val.dtype)(_hip.dptr(idx), _hip.dptr(val), ctypes.c_int64(n), ctypes.c_int32(T), _hip.dptr(v),
                                                _hip.dptr(out), _hip.stream_ptr(val.device))
    _hip.check(rc, "wiski_gather_ell")
    return out


def scatter_stats(grid, x, y, wa, wb, noise, b, A_st, stats, err):
    """In-place accumulation of (b, A_st, stats) -- a3/a4/a5."""
    x = _x2d(x, grid)
    n = x.shape[0]
    rc = _hip.fn("wiski_scatter_stats", x.dtype)(grid.ref, _hip.dptr(x), _hip.dptr(y.contiguous()), _hip.dptr(wa.contiguous()),
                                                 _hip.dptr(wb.contiguous()), _hip.dptr(noise.contiguous()), ctypes.c_int64(n), _hip.dptr(b),
                                                 _hip.dptr(A_st), _hip.dptr(stats), _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_scatter_stats")


def scatter_stats_sym(grid, x, y, wa, wb, noise, b, A_half, stats, err):
    """As scatter_stats, but W^T diag(wa) W goes into a symmetric half-stencil delta [(R+1)/2, m]."""
    x = _x2d(x, grid)
    rc = _hip.fn("wiski_scatter_stats_sym", x.dtype)(grid.ref, _hip.dptr(x), _hip.dptr(y.contiguous()), _hip.dptr(wa.contiguous()),
                                                     _hip.dptr(wb.contiguous()), _hip.dptr(noise.contiguous()), ctypes.c_int64(x.shape[0]),
                                                     _hip.dptr(b), _hip.dptr(A_half), _hip.dptr(stats), _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_scatter_stats_sym")


def scatter_stats_cnt(grid, x, y, wa, wb, noise, b, A, half, cnt, stats, err, u=None, res=None):
    """One launch: (b, A full or symmetric half, stats) as scatter_stats[_sym] plus cnt += W^T wa and, with
    (u, res) given, the residual carry-over res += W^T (wb y - wa (W u))."""
    x = _x2d(x, grid)
    rc = _hip.fn("wiski_scatter_stats_cnt", x.dtype)(grid.ref, _hip.dptr(x), _hip.dptr(y.contiguous()), _hip.dptr(wa.contiguous()),
                                                     _hip.dptr(wb.contiguous()), _hip.dptr(noise.contiguous()), ctypes.c_int64(x.shape[0]),
                                                     _hip.dptr(b), _hip.dptr(A), ctypes.c_int32(int(half)), _hip.dptr(cnt), _hip.dptr(u), _hip.dptr(res),
                                                     _hip.dptr(stats), _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_scatter_stats_cnt")


def scatter_stats_multi(grid, x, Yt, wa, wb, noise, b, A_pack, cnt, stats, err, u=None, res=None):
    """ONE absorb launch for all outputs (``wiski_scatter_stats_multi``): Yt [out, n]; wa / wb / noise [out, n] or [n] (shared by
    the outputs); b / cnt / u / res [out, m]; A_pack [out, H, m] half stencils; stats [out, 2]."""
    x = _x2d(x, grid)
    out, n = Yt.shape
    shared = wa.dim() == 1
    rc = _hip.fn("wiski_scatter_stats_multi", x.dtype)(grid.ref, _hip.dptr(x), _hip.dptr(Yt), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise), ctypes.c_int64(n),
                                                       ctypes.c_int32(out), ctypes.c_int64(n), ctypes.c_int64(0 if shared else n), _hip.dptr(b), _hip.dptr(A_pack),
                                                       ctypes.c_int64(A_pack.shape[1] * A_pack.shape[2]), _hip.dptr(cnt), _hip.dptr(u), _hip.dptr(res), _hip.dptr(stats),
                                                       _hip.dptr(err), _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_scatter_stats_multi")


def stencil_expand_add(grid, A_half, A_st):
    """A_st += expand(A_half) (delta and its mirror image); A_half is zeroed."""
    rc = _hip.fn("wiski_stencil_expand_add", A_st.dtype)(grid.ref, _hip.dptr(A_half), _hip.dptr(A_st), _hip.stream_ptr(A_st.device))
    _hip.check(rc, "wiski_stencil_expand_add")


def wt_columns(grid, x, err):
    """Dense columns of W(x)^T: [n, m]."""
    x = _x2d(x, grid)
    out = torch.zeros((x.shape[0], grid.m), dtype=x.dtype, device=x.device)
    rc = _hip.fn("wiski_wt_columns", x.dtype)(grid.ref, _hip.dptr(x), ctypes.c_int64(x.shape[0]), _hip.dptr(out), _hip.dptr(err),
                                              _hip.stream_ptr(x.device))
    _hip.check(rc, "wiski_wt_columns")
    return out


def is_half_stencil(grid, A_st):
    """True for the symmetric half-stencil layout [(R+1)/2, m], False for the full [R, m] one."""
    rows = A_st.shape[-2]
    if rows == (grid.R + 1) // 2:
        return True
    if rows == grid.R:
        return False
    raise ValueError(f"stencil with {rows} rows matches neither the full ({grid.R}) nor the half ({(grid.R + 1) // 2}) layout")


def half_stencil_from_offset_major(grid, A_om):
    """Offset-major half stencil ``A_om[oh, i] = A[i, i + off(c + oh)]`` ([(R+1)/2, m], e.g. the upper half of
    the oracle's full stencil) -> the native row-interleaved layout of wiski_scatter_stats_sym (same shape,
    different memory order: see include/wiski.h)."""
    H, m = A_om.shape
    flat = torch.empty(H * m, dtype=A_om.dtype, device=A_om.device)
    flat[:4 * m] = A_om[:4].t().reshape(-1)
    if H > 4:
        flat[4 * m:] = A_om[4:].reshape(-1, 7, m).transpose(1, 2).reshape(-1)
    return flat.view(H, m)


def half_stencil_to_offset_major(grid, A_h):
    """Inverse of :func:`half_stencil_from_offset_major`."""
    H, m = A_h.shape
    flat = A_h.reshape(-1)
    out = torch.empty((H, m), dtype=A_h.dtype, device=A_h.device)
    out[:4] = flat[:4 * m].view(m, 4).t()
    if H > 4:
        out[4:] = flat[4 * m:].view(-1, m, 7).transpose(1, 2).reshape(H - 4, m)
    return out


def stencil_spmv(grid, A_st, V, add=None, beta=1.0):
    """A @ V on the block stencil: A_st is either the full offset-major [R, m] stencil or the symmetric half
    ([(R+1)/2, m] reals in the native row-interleaved layout)."""
    V2 = V.contiguous().reshape(-1, grid.m)
    out = torch.empty_like(V2)
    cr = _hip.creal(V2.dtype)
    name = "wiski_stencil_spmv_sym" if is_half_stencil(grid, A_st) else "wiski_stencil_spmv"
    rc = _hip.fn(name, V2.dtype)(grid.ref, _hip.dptr(A_st), _hip.dptr(V2), ctypes.c_int32(V2.shape[0]),
                                 _hip.dptr(add.contiguous().reshape(-1, grid.m)) if add is not None else None, cr(beta),
                                 _hip.dptr(out), _hip.stream_ptr(V2.device))
    _hip.check(rc, name)
    return out.reshape(V.shape)


def kron_toeplitz_mm(grid, tcol, V, scale=1.0):
    V2 = V.contiguous().reshape(-1, grid.m)
    out = torch.empty_like(V2)
    tmp = torch.empty_like(V2) if grid.d > 1 else None
    cr = _hip.creal(V2.dtype)
    rc = _hip.fn("wiski_kron_toeplitz_mm", V2.dtype)(grid.ref, _hip.dptr(tcol.contiguous()), _hip.dptr(V2), ctypes.c_int32(V2.shape[0]), cr(scale),
                                                     _hip.dptr(tmp), _hip.dptr(out), _hip.stream_ptr(V2.device))
    _hip.check(rc, "wiski_kron_toeplitz_mm")
    return out.reshape(V.shape)


class PCGWorkspace:
    """Device scratch for wiski_pcg, cached per (k, max_iter)."""

    def __init__(self):
        self.buf = None

    def get(self, grid, k, max_iter, dtype, device):
        es = 4 if dtype == torch.float32 else 8
        need = int(_hip.lib().wiski_pcg_workspace_bytes(grid.ref, ctypes.c_int32(k), ctypes.c_int32(max_iter), ctypes.c_int32(es)))
        if need < 0:
            raise _hip.WiskiError("wiski_pcg_workspace_bytes: bad arguments")
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.empty(need, dtype=torch.uint8, device=device)
        return self.buf, need


class _StreamArgs32(ctypes.Structure):
    _fields_ = [("d_A_half", ctypes.c_void_p), ("d_b", ctypes.c_void_p), ("d_cnt", ctypes.c_void_p), ("d_stats", ctypes.c_void_p),
                ("d_err", ctypes.c_void_p), ("d_U", ctypes.c_void_p), ("d_Z", ctypes.c_void_p), ("d_R", ctypes.c_void_p),
                ("d_tcol", ctypes.c_void_p), ("kscale", ctypes.c_float), ("d_evec", ctypes.c_void_p), ("d_evec2", ctypes.c_void_p),
                ("d_eval", ctypes.c_void_p), ("shift", ctypes.c_float), ("tol", ctypes.c_double), ("max_iter", ctypes.c_int32),
                ("check_every", ctypes.c_int32), ("d_work", ctypes.c_void_p), ("work_bytes", ctypes.c_int64),
                ("d_bin", ctypes.c_void_p), ("bin_bytes", ctypes.c_int64), ("shard", ctypes.c_void_p), ("two_level", ctypes.c_void_p)]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                ctypes.c_void_p)


class _Shard(ctypes.Structure):
    """include/wiski.h: wiski_shard"""
    _fields_ = [("rank", ctypes.c_int32), ("nranks", ctypes.c_int32), ("comm", ctypes.c_void_p), ("allreduce", ALLREDUCE_FN), ("ctx", ctypes.c_void_p)]


class TwoLevelStruct(ctypes.Structure):
    """include/wiski.h: wiski_twolevel"""
    _fields_ = [("r", ctypes.c_int32), ("nslab", ctypes.c_int32), ("d_mask", ctypes.c_void_p), ("d_off", ctypes.c_void_p), ("d_pos", ctypes.c_void_p),
                ("d_N", ctypes.c_void_p), ("d_cs", ctypes.c_void_p), ("d_mc", ctypes.c_void_p), ("mc_cols", ctypes.c_int32)]


def shard_groups(d, rank, nranks):
    """[g_lo, g_hi): the stencil groups rank `rank` of `nranks` owns (wiski_shard_groups; pure host arithmetic)."""
    lo, hi = ctypes.c_int32(0), ctypes.c_int32(0)
    _hip.check(_hip.lib().wiski_shard_groups(ctypes.c_int32(d), ctypes.c_int32(rank), ctypes.c_int32(nranks), ctypes.byref(lo), ctypes.byref(hi)),
               "wiski_shard_groups")
    return int(lo.value), int(hi.value)


def half_stencil_group_slices(grid, g_lo, g_hi):
    """Element ranges [(start, stop), ...] of the flat row-interleaved half stencil that hold the groups [g_lo, g_hi):
    group 0 is A_h[0 : 4 m], group g >= 1 is A_h[(7 g - 3) m : (7 g + 4) m] (include/wiski.h)."""
    m = grid.m
    out = []
    for g in range(g_lo, g_hi):
        out.append((0, 4 * m) if g == 0 else ((7 * g - 3) * m, (7 * g + 4) * m))
    return out


class _StreamArgs64(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double if t is ctypes.c_float else t) for n, t in _StreamArgs32._fields_]



class _PcgAsync(ctypes.Structure):
    _fields_ = [("state", ctypes.c_int32), ("it", ctypes.c_int32), ("seq", ctypes.c_int64), ("poll", ctypes.c_void_p),
                ("prezeroed", ctypes.c_int32), ("guard_ok", ctypes.c_int32), ("shift", ctypes.c_double)]


class StreamStep:
    """Prepared call of ``wiski_stream_step`` (include/wiski.h): the model-resident pointers are marshalled once, a step only
    passes the batch.  Returns (iterations, relative residual, raw out-of-grid flag, converged)."""

    def __init__(self, grid, dtype, device, A_half, b, cnt, stats, err, U, Z, R, tcol, workspace, max_iter):
        self.grid, self.dtype, self.device = grid, dtype, device
        self.keep = (A_half, b, cnt, stats, err, U, Z, R, tcol)             # the tensors behind the raw pointers
        self.args = (_StreamArgs32 if dtype == torch.float32 else _StreamArgs64)()
        a = self.args
        a.d_A_half, a.d_b, a.d_cnt, a.d_stats, a.d_err = A_half.data_ptr(), b.data_ptr(), cnt.data_ptr(), stats.data_ptr(), err.data_ptr()
        a.d_U, a.d_Z, a.d_R, a.d_tcol = U.data_ptr(), Z.data_ptr(), R.data_ptr(), tcol.data_ptr()
        buf, need = workspace.get(grid, 1, max_iter, dtype, device)
        self.keep += (buf,)
        a.d_work, a.work_bytes, a.max_iter = buf.data_ptr(), need, max_iter
        self.bin_buf = None                    # binning workspace of the owner-computes absorb: grown on demand, zeroed once
        self.fn = _hip.fn("wiski_stream_step", dtype)
        self.it, self.herr, self.rr = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0)
        self.eig_keep = None
        self.handle = _PcgAsync()              # deferred-poll state (include/wiski.h: wiski_pcg_async)
        self.resumed = ctypes.c_int32(0)

    @property
    def pending(self):
        return self.handle.state == 1

    def __del__(self):
        try:
            if self.handle.poll:
                _hip.lib().wiski_pcg_async_free(ctypes.byref(self.handle))
        except Exception:  # noqa: BLE001
            pass

    def set_shard(self, rank, nranks, comm=None, allreduce=None):
        """Stencil-sharded step (wiski_shard): this replica owns the groups shard_groups(d, rank, nranks) of the half stencil.
        comm: an ncclComm_t handle (int) for the RCCL route; allreduce(vec, dots): a Python callable that all-reduces (SUM, in
        place) the two torch views it is given -- the views alias the solver's workspace."""
        if nranks < 1 or (nranks == 1 and not comm):   # (a single rank WITH a communicator runs the sharded path: it owns every group)
            self.args.shard = None
            self._shard = None
            return
        buf = self.keep[-1]                      # the PCG workspace tensor the device pointers of the callback point into
        base = buf.data_ptr()

        def cb(ctx, d_vec, n_vec, elem_bytes, d_dots, n_dots, stream):
            try:
                dt = torch.float32 if elem_bytes == 4 else torch.float64
                vec = buf[d_vec - base:d_vec - base + n_vec * elem_bytes].view(dt)
                dots = buf[d_dots - base:d_dots - base + n_dots * 8].view(torch.float64) if (d_dots and n_dots) else None
                allreduce(vec, dots)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return -2

        self._shard_cb = ALLREDUCE_FN(cb) if allreduce is not None else ALLREDUCE_FN()
        self._shard = _Shard(rank, nranks, comm, self._shard_cb, None)
        self.args.shard = ctypes.addressof(self._shard)

    def set_two_level(self, tl):
        """tl: a TwoLevelStruct kept alive by the caller (its d_N may be re-pointed between steps), or None."""
        self._two_level = tl
        self.args.two_level = ctypes.addressof(tl) if tl is not None else None

    def set_solver(self, kscale, eig, shift, tol, check_every):
        a = self.args
        evec, evals, evec2 = (tuple(eig) + (None,))[:3]
        self.eig_keep = (evec, evals, evec2)
        a.kscale, a.shift, a.tol, a.check_every = kscale, shift, tol, check_every
        a.d_evec, a.d_eval = evec.data_ptr(), evals.data_ptr()
        a.d_evec2 = evec2.data_ptr() if evec2 is not None else None

    def __call__(self, x, y, wa, wb, noise, mean_out, carry, first_check, defer=False):
        """Without deferral: (iterations, relres, out-of-grid flag, converged) of THIS step's solve.  With `defer` (or a pending
        solve): the same four values describe the solve a previous call started (None when there was none) and the fifth
        element says whether this step's solve is now pending."""
        q = x.shape[0] if x is not None else 0
        if q >= 4096 and self.grid.d == 3 and (self.bin_buf is None or self.bin_q < q):
            f = _hip.lib().wiski_scatter_bin_bytes
            f.restype = ctypes.c_int64
            need = int(f(self.grid.ref, ctypes.c_int64(q), ctypes.c_int32(4 if self.dtype == torch.float32 else 8)))
            if need > 0:
                self.bin_buf, self.bin_q = torch.zeros(need, dtype=torch.uint8, device=self.device), q
                self.args.d_bin, self.args.bin_bytes = self.bin_buf.data_ptr(), need
        use_handle = defer or self.pending
        self.herr.value = 0
        rc = self.fn(self.grid.ref, ctypes.byref(self.args), _hip.dptr(x), _hip.dptr(y), _hip.dptr(wa), _hip.dptr(wb), _hip.dptr(noise),
                     ctypes.c_int64(q), _hip.dptr(mean_out), ctypes.c_int32(int(carry)), ctypes.c_int32(int(first_check)), ctypes.byref(self.it),
                     ctypes.byref(self.rr), ctypes.byref(self.herr), _hip.stream_ptr(self.device),
                     ctypes.byref(self.handle) if use_handle else None, ctypes.c_int32(int(defer)), ctypes.byref(self.resumed))
        if rc == -4:
            warnings.warn(f"wiski_pcg stopped at max_iter={self.args.max_iter} with relative residual {self.rr.value:.3e}", RuntimeWarning)
        elif rc != 1:
            _hip.check(rc, "wiski_stream_step")
        res = (int(self.it.value), float(self.rr.value), int(self.herr.value), rc != -4)
        if not use_handle:
            return res
        # resumed == 2: a pending solve was finished AND this step's solve ran synchronously -- `res` describes the latter
        return (res if self.resumed.value == 1 else None), self.pending


def precond_apply(grid, eig, kscale, shift, r, two_level=None):
    """One application of the fused fp32 preconditioner to the grid vector r (``wiski_precond_apply``): (y = P r, t = Kt^-1 y, r . y)."""
    evec, evals, evec2 = (tuple(eig) + (None,))[:3]
    r = r.contiguous()
    m = grid.m
    w0 = torch.empty(m, dtype=r.dtype, device=r.device)
    w1 = torch.empty(2 * m, dtype=r.dtype, device=r.device)
    y, t = torch.empty_like(r), torch.empty_like(r)
    rho = torch.zeros(2, dtype=torch.float64, device=r.device)
    rc = _hip.lib().wiski_precond_apply_f32(grid.ref, _hip.dptr(evec), _hip.dptr(evec2), _hip.dptr(evals), ctypes.c_float(kscale), ctypes.c_float(shift),
                                            _hip.dptr(r), _hip.dptr(w0), _hip.dptr(w1), _hip.dptr(y), _hip.dptr(t), ctypes.c_void_p(rho.data_ptr() + 8),
                                            ctypes.byref(two_level) if two_level is not None else None, _hip.stream_ptr(r.device))
    _hip.check(rc, "wiski_precond_apply")
    return y, t, rho[1]


def precond_apply_cols(grid, eig, kscale, shift, R, two_level=None):
    """`precond_apply` for the k rows of R [k, m] through the multi-column kernels (``wiski_precond_apply_cols``): (Y, T, rho [k])."""
    evec, evals, evec2 = (tuple(eig) + (None,))[:3]
    R = R.contiguous()
    k, m = R.shape
    w0 = torch.empty(k * m, dtype=R.dtype, device=R.device)
    w1 = torch.empty(2 * k * m, dtype=R.dtype, device=R.device)
    Y, T = torch.empty_like(R), torch.empty_like(R)
    rho = torch.zeros(k, dtype=torch.float64, device=R.device)
    rc = _hip.lib().wiski_precond_apply_cols_f32(grid.ref, _hip.dptr(evec), _hip.dptr(evec2), _hip.dptr(evals), ctypes.c_float(kscale), ctypes.c_float(shift),
                                                 _hip.dptr(R), ctypes.c_int32(k), _hip.dptr(w0), _hip.dptr(w1), _hip.dptr(Y), _hip.dptr(T), _hip.dptr(rho),
                                                 ctypes.byref(two_level) if two_level is not None else None, _hip.stream_ptr(R.device))
    _hip.check(rc, "wiski_precond_apply_cols")
    return Y, T, rho


def kron_eigen(grid, tcol, profiles=None, host_out=None):
    """Per-dim (generalized) eigen-decomposition of the d small symmetric-Toeplitz Kronecker
    factors (host side, fp64, O(d g^3) -- done when the hyper-parameters or the data-density
    profile change, not per streaming update).

    profiles=None:  K_q = V_q diag(lam_q) V_q^T                         -> (evec, evals)
    profiles=[t_q]: K_q = X_q D_q X_q^T with X_q^T diag(t_q) X_q = I     -> (evec = X, evals = D, evec2 = Z = diag(t) X)
    Eigenvalues are clamped >= 0.  Feeds wiski_pcg's preconditioner (K^-1 + a kron diag(t_q))^-1."""
    import numpy as np

    tc = tcol.detach().to("cpu", torch.float64).numpy()
    X, Z, vals, off = [], [], [], 0
    for q, g in enumerate(grid.g):
        c = tc[off:off + g]
        idx = np.abs(np.arange(g)[:, None] - np.arange(g)[None, :])
        K = c[idx]
        if profiles is None:
            w, V = np.linalg.eigh(K)
            X.append(V.reshape(-1))
        else:
            t = np.asarray(profiles[q], dtype=np.float64)
            rt = np.sqrt(t)
            w, U = np.linalg.eigh(rt[:, None] * K * rt[None, :])
            X.append((U / rt[:, None]).reshape(-1))
            Z.append((U * rt[:, None]).reshape(-1))
        vals.append(np.clip(w, 0.0, None))
        off += g
    mk = lambda parts: torch.as_tensor(np.concatenate(parts)).to(tcol.device, tcol.dtype)
    if host_out is not None:              # fp64 host copies (per dim: eigenvectors [g, g] column = mode, eigenvalues ascending)
        host_out["X"] = [x.reshape(g, g) for x, g in zip(X, grid.g)]
        host_out["D"] = [v.copy() for v in vals]
    if profiles is None:
        return mk(X), mk(vals)
    return mk(X), mk(vals), mk(Z)


def pcg(grid, A_st, tcol, kscale, RHS, U=None, Z=None, warm=False, tol=1e-6, max_iter=1000, check_every=10, workspace=None,
        raise_on_fail=False, eigen=None, shift=0.0, first_check=0, err=None, inplace=False, R=None, two_level=None):
    """Solve (Kt^-1 + A) U = RHS, Kt = kscale*Kuu.  Returns (U, Z, iters, relres).
    two_level: a TwoLevelStruct (include/wiski.h: wiski_twolevel) -- the exact block on the dominant modes inside the fused
    preconditioner (``wiski_pcg_twolevel_f32``: fp32, d = 3, eigen tables given; one column, or up to ``two_level.mc_cols``
    columns when the block carries the multi-column scratch ``d_mc``; otherwise the argument is ignored).
    eigen = (evec, evals) from :func:`kron_eigen` selects the spectral
    preconditioner (Kt^-1 + shift I)^-1; otherwise Kt itself preconditions.
    R [k, m] (optional, contiguous): caller-owned residual buffer, left holding RHS - Z - A U;
    warm=2 starts from the residual already in R (see wiski.h) instead of forming A U."""
    RHS2 = RHS.contiguous().reshape(-1, grid.m)
    k = RHS2.shape[0]
    if U is None or Z is None or (not warm and not inplace):
        U = torch.empty_like(RHS2)
        Z = torch.empty_like(RHS2)
        warm = False
    if int(warm) == 2 and R is None:
        raise ValueError("warm=2 needs the carried-over residual R")
    ws = workspace if workspace is not None else PCGWorkspace()
    buf, need = ws.get(grid, k, max_iter, RHS2.dtype, RHS2.device)
    iters = ctypes.c_int32(0)
    h_err = ctypes.c_int32(0)
    relres = (ctypes.c_double * k)()
    cr = _hip.creal(RHS2.dtype)
    evec, evals, evec2 = (tuple(eigen) + (None,))[:3] if eigen is not None else (None, None, None)
    common = (grid.ref, _hip.dptr(A_st), _hip.dptr(tcol.contiguous()), cr(kscale), _hip.dptr(evec), _hip.dptr(evec2), _hip.dptr(evals),
              cr(shift), _hip.dptr(RHS2), ctypes.c_int32(k),
              _hip.dptr(U), _hip.dptr(Z), ctypes.c_int32(int(warm)), ctypes.c_double(tol), ctypes.c_int32(max_iter),
              ctypes.c_int32(check_every), ctypes.c_int32(first_check), _hip.dptr(buf), ctypes.c_int64(need), ctypes.byref(iters), relres, _hip.dptr(err), ctypes.byref(h_err),
              ctypes.c_int32(1 if is_half_stencil(grid, A_st) else 0), _hip.dptr(R), _hip.stream_ptr(RHS2.device))
    if two_level is not None and (k == 1 or (two_level.d_mc and k <= two_level.mc_cols)) and RHS2.dtype == torch.float32 and evec is not None:
        rc = _hip.lib().wiski_pcg_twolevel_f32(*common, None, ctypes.c_int32(0), None, ctypes.byref(two_level))
    else:
        rc = _hip.fn("wiski_pcg", RHS2.dtype)(*common)
    if rc == -4 and not raise_on_fail:
        # gpytorch emits a NumericalWarning when CG stops at max_cg_iterations; callers also see it in `relres`
        warnings.warn(f"wiski_pcg stopped at max_iter={max_iter} with relative residual {max(relres):.3e} (tolerance {tol:.1e})",
                      RuntimeWarning)
        pcg.last_converged = False
    else:
        _hip.check(rc, "wiski_pcg")
        pcg.last_converged = True
    pcg.last_err = int(h_err.value)
    return U, Z, int(iters.value), list(relres)


pcg.last_err = 0
pcg.last_converged = True


def kron_toeplitz_grad(grid, tcol, X, Y):
    """d/d tcol of sum_c X[c]^T (kron SymToeplitz(tcol)) Y[c]  -> float64 [sum g]."""
    X2 = X.contiguous().reshape(-1, grid.m)
    Y2 = Y.contiguous().reshape(-1, grid.m)
    k = X2.shape[0]
    tmp = torch.empty((2 * k, grid.m), dtype=X2.dtype, device=X2.device)
    grad = torch.zeros(sum(grid.g), dtype=torch.float64, device=X2.device)
    rc = _hip.fn("wiski_kron_toeplitz_grad", X2.dtype)(grid.ref, _hip.dptr(tcol.contiguous()), _hip.dptr(X2), _hip.dptr(Y2), ctypes.c_int32(k),
                                                       _hip.dptr(tmp), _hip.dptr(grad), _hip.stream_ptr(X2.device))
    _hip.check(rc, "wiski_kron_toeplitz_grad")
    return grad


def kron_spectral_mm(grid, eigen, V, kscale=1.0, shift=0.0, power=0.5, rpower=0.0):
    """V diag(lam^power / (1 + shift lam)^rpower) V^T applied to the columns of V ([k, m])."""
    evec, evals = eigen[0], eigen[1]
    assert len(eigen) == 2, "kron_spectral_mm needs the orthogonal eigenbasis (kron_eigen without profiles)"
    V2 = V.contiguous().reshape(-1, grid.m)
    out = torch.empty_like(V2)
    tmp = torch.empty_like(V2)
    cr = _hip.creal(V2.dtype)
    rc = _hip.fn("wiski_kron_spectral_mm", V2.dtype)(grid.ref, _hip.dptr(evec), _hip.dptr(evals), cr(kscale), cr(shift), cr(power), cr(rpower),
                                                     _hip.dptr(V2), ctypes.c_int32(V2.shape[0]), _hip.dptr(tmp), _hip.dptr(out),
                                                     _hip.stream_ptr(V2.device))
    _hip.check(rc, "wiski_kron_spectral_mm")
    return out.reshape(V.shape)


# ------------------------------------------------------------ dense (small-m) ops --
def gemm(A, B, ta=False, tb=False, alpha=1.0, beta=0.0, C=None):
    """C = alpha op(A) op(B) + beta C on the matrix cores (row-major 2-D tensors)."""
    A, B = A.contiguous(), B.contiguous()
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    assert (B.shape[1] if tb else B.shape[0]) == K and A.dtype == B.dtype
    if C is None:
        C = torch.empty((M, N), dtype=A.dtype, device=A.device)
        beta = 0.0
    cr = _hip.creal(A.dtype)
    rc = _hip.fn("wiski_gemm", A.dtype)(ctypes.c_int32(int(ta)), ctypes.c_int32(int(tb)), ctypes.c_int32(M), ctypes.c_int32(N), ctypes.c_int32(K),
                                        cr(alpha), _hip.dptr(A), ctypes.c_int32(A.shape[1]), _hip.dptr(B), ctypes.c_int32(B.shape[1]), cr(beta),
                                        _hip.dptr(C), ctypes.c_int32(C.shape[1]), _hip.stream_ptr(A.device))
    _hip.check(rc, "wiski_gemm")
    return C


def potrf_(A, info=None):
    """In-place lower Cholesky; returns the device info flag (non-zero: not positive definite)."""
    assert A.dim() == 2 and A.shape[0] == A.shape[1] and A.is_contiguous()
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
    rc = _hip.fn("wiski_potrf", A.dtype)(ctypes.c_int32(A.shape[0]), _hip.dptr(A), ctypes.c_int32(A.shape[1]), _hip.dptr(info),
                                         _hip.stream_ptr(A.device))
    _hip.check(rc, "wiski_potrf")
    return info


def potrf_inverse_(A, info=None):
    """In-place lower Cholesky AND the explicit inverse of the factor (``wiski_potrf_inverse``: two launches for n <= 480).
    Returns (Linv, info)."""
    assert A.dim() == 2 and A.shape[0] == A.shape[1] and A.is_contiguous()
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
    X = torch.empty_like(A)
    rc = _hip.fn("wiski_potrf_inverse", A.dtype)(ctypes.c_int32(A.shape[0]), _hip.dptr(A), ctypes.c_int32(A.shape[1]), _hip.dptr(X),
                                                 ctypes.c_int32(X.shape[1]), _hip.dptr(info), _hip.stream_ptr(A.device))
    _hip.check(rc, "wiski_potrf_inverse")
    return X, info


def psd_safe_cholesky(A, jitter=None, max_tries=6):
    """Cholesky with gpytorch's jitter escalation (psd_safe_cholesky; imported by the
    reference at updated_root_lazy_tensor.py:5): try plain, then add jitter*10^i."""
    if jitter is None:
        jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    for i in range(max_tries + 1):
        L = A.clone()
        if i > 0:
            L.diagonal().add_(jitter * (10 ** (i - 1)))
        if int(potrf_(L).item()) == 0 and bool(torch.isfinite(L.diagonal()).all()):
            return L
    raise RuntimeError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter * 10 ** (max_tries - 1):.1e}.")


def trsm_(L, B, trans=False):
    """In-place B <- L^-1 B (trans=False) or L^-T B (trans=True); B is [n, nrhs]."""
    assert B.dim() == 2 and B.is_contiguous() and L.is_contiguous() and L.shape[0] == B.shape[0]
    rc = _hip.fn("wiski_trsm", L.dtype)(ctypes.c_int32(int(trans)), ctypes.c_int32(L.shape[0]), ctypes.c_int32(B.shape[1]), _hip.dptr(L),
                                        ctypes.c_int32(L.shape[1]), _hip.dptr(B), ctypes.c_int32(B.shape[1]), _hip.stream_ptr(L.device))
    _hip.check(rc, "wiski_trsm")
    return B


def root_update_(L, R, V):
    """In-place rank-q update of a root / inverse-root pair (a6, URLT:69-119; wiski_root_update): on entry L L^T = A and
    R^T L = I, on return L L^T = A + V V^T and R^T L = I.  L, R [m, r], V [m, q], all contiguous."""
    assert L.dim() == 2 and R.shape == L.shape and V.dim() == 2 and V.shape[0] == L.shape[0]
    assert L.is_contiguous() and R.is_contiguous()
    V = V.contiguous()
    m, r = L.shape
    q = V.shape[1]
    n = int(_hip.lib().wiski_root_update_workspace_elems(ctypes.c_int32(m), ctypes.c_int32(r), ctypes.c_int32(q)))
    ws = torch.empty(n, dtype=L.dtype, device=L.device)
    rc = _hip.fn("wiski_root_update", L.dtype)(ctypes.c_int32(m), ctypes.c_int32(r), ctypes.c_int32(q), _hip.dptr(L), ctypes.c_int32(r), _hip.dptr(R),
                                               ctypes.c_int32(r), _hip.dptr(V), ctypes.c_int32(q), _hip.dptr(ws), ctypes.c_int64(n),
                                               _hip.stream_ptr(L.device))
    _hip.check(rc, "wiski_root_update")
    return L, R


def dense_factor(grid, A_half, eigen, kscale):
    """The dense regime's posterior factor in one C call (``wiski_dense_factor``): returns (M = (Kt^-1 + A)^-1 [m, m], C with
    I + sym(G A G) = C C^T, logdet(I + Kt A) as a 0-dim fp64 tensor, info int32[1] -- non-zero on a non-positive pivot)."""
    evec, evals = eigen[0], eigen[1]
    assert len(eigen) == 2 and is_half_stencil(grid, A_half)
    m, dt, dev = grid.m, A_half.dtype, A_half.device
    work = torch.empty(4 * m * m, dtype=dt, device=dev)
    chol = torch.empty((m, m), dtype=dt, device=dev)
    M = torch.empty((m, m), dtype=dt, device=dev)
    ld = torch.zeros(1, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _hip.fn("wiski_dense_factor", dt)(grid.ref, _hip.dptr(A_half), _hip.dptr(evec.contiguous()), _hip.dptr(evals.contiguous()), _hip.creal(dt)(kscale),
                                           _hip.dptr(work), ctypes.c_int64(work.numel()), _hip.dptr(chol), _hip.dptr(M), _hip.dptr(ld), _hip.dptr(info),
                                           _hip.stream_ptr(dev))
    _hip.check(rc, "wiski_dense_factor")
    return M, chol, 2.0 * ld[0], info


def chol_logdet(L):
    out = torch.zeros(1, dtype=torch.float64, device=L.device)
    rc = _hip.fn("wiski_logdiag", L.dtype)(ctypes.c_int32(L.shape[0]), _hip.dptr(L), ctypes.c_int32(L.shape[1]), _hip.dptr(out),
                                           _hip.stream_ptr(L.device))
    _hip.check(rc, "wiski_logdiag")
    return 2.0 * out[0]


# ------------------------------------------------- reduced Kronecker-eigenbasis factor --
def basis_project(grid, x, V, kmax, S, scale=None, colscale=None, tcol=None, want_prior=False, err=None):
    """F = diag(scale) (W(x) B) diag(colscale), fp64 [n, r], for the tensor-product basis b_j = kron_q V_q[:, S[q, j]]
    (``wiski_basis_project``).  V: fp64 per-dim tables [g_q, kmax] concatenated; S int32 [d, r].  With `want_prior`
    also returns the prior variances w_p^T Kuu w_p (needs the fp64 Toeplitz columns `tcol`)."""
    x2 = _x2d(x, grid)
    n, r = x2.shape[0], S.shape[1]
    F = torch.empty((n, r), dtype=torch.float64, device=x2.device)
    prior = torch.empty(n, dtype=torch.float64, device=x2.device) if want_prior else None
    rc = _hip.fn("wiski_basis_project", x2.dtype)(grid.ref, _hip.dptr(x2), ctypes.c_int64(n), _hip.dptr(V), ctypes.c_int32(kmax), _hip.dptr(S),
                                                  ctypes.c_int32(r), _hip.dptr(scale), _hip.dptr(colscale), _hip.dptr(tcol), _hip.dptr(F),
                                                  ctypes.c_int64(r), _hip.dptr(prior), _hip.dptr(err), _hip.stream_ptr(x2.device))
    _hip.check(rc, "wiski_basis_project")
    return (F, prior) if want_prior else F


def stationary_columns(grid, kind, ell, scale):
    """``wiski_stationary_columns``: fp64 Toeplitz columns [sum g] of S * k(lag / ell) (kind 0 RBF, 1-3 Matern 1/2, 3/2, 5/2);
    ell [1] or [d] and scale [1] / None are device tensors of one dtype (fp32 / fp64) -- no host read."""
    out = torch.empty(sum(grid.g), dtype=torch.float64, device=ell.device)
    rc = _hip.fn("wiski_stationary_columns", ell.dtype)(grid.ref, ctypes.c_int32(kind), _hip.dptr(ell), ctypes.c_int32(ell.numel()),
                                                        None if scale is None else _hip.dptr(scale), _hip.dptr(out), _hip.stream_ptr(ell.device))
    _hip.check(rc, "wiski_stationary_columns")
    return out


def stationary_columns_grad(grid, kind, ell, scale, gout):
    g_ell = torch.empty_like(ell)
    g_scale = None if scale is None else torch.empty_like(scale)
    rc = _hip.fn("wiski_stationary_columns_grad", ell.dtype)(grid.ref, ctypes.c_int32(kind), _hip.dptr(ell), ctypes.c_int32(ell.numel()),
                                                             None if scale is None else _hip.dptr(scale), _hip.dptr(gout),
                                                             _hip.dptr(g_ell), None if g_scale is None else _hip.dptr(g_scale),
                                                             _hip.stream_ptr(ell.device))
    _hip.check(rc, "wiski_stationary_columns_grad")
    return g_ell, g_scale


def hyper_columns(plan, grid, kind, ell, scale, s2, s2_f64=None, tcol64=None, tcol=None):
    """``wiski_hyper_columns``: constrained values (ell, scale, s2: pre-allocated, parameter dtype) of the plan's raw parameters and, with
    `tcol64`, the Toeplitz columns (fp64 and, with `tcol`, the parameter dtype) -- one launch, no host read."""
    rc = _hip.fn("wiski_hyper_columns", ell.dtype)(ctypes.byref(plan), grid.ref, ctypes.c_int32(kind), _hip.dptr(ell), _hip.dptr(scale), _hip.dptr(s2),
                                                   _hip.dptr(s2_f64), _hip.dptr(tcol64), _hip.dptr(tcol), _hip.stream_ptr(ell.device))
    _hip.check(rc, "wiski_hyper_columns")


def hyper_mid(bMb, logdet, s2, c, ld, n_dev, out, loss_out=None):
    """``wiski_hyper_mid``: out [9] fp64 = {val, coef0..2, g = -1/n, g coef0, g coef1, loss, 1/s2}."""
    rc = _hip.fn("wiski_hyper_mid", s2.dtype)(_hip.dptr(bMb), None if logdet is None else _hip.dptr(logdet), _hip.dptr(s2), _hip.dptr(c), _hip.dptr(ld),
                                              _hip.dptr(n_dev), _hip.dptr(out), _hip.dptr(loss_out), _hip.stream_ptr(s2.device))
    _hip.check(rc, "wiski_hyper_mid")


def hyper_adam(plan, scale, s2, g_ell, g_scale, mid, g_kap, n_dev, lr, beta1, beta2, eps):
    """``wiski_hyper_adam``: chain rule to the raw parameters + torch.optim.Adam's update of the plan's parameters, in place."""
    rc = _hip.fn("wiski_hyper_adam", s2.dtype)(ctypes.byref(plan), _hip.dptr(scale), _hip.dptr(s2), _hip.dptr(g_ell), _hip.dptr(g_scale), _hip.dptr(mid),
                                               _hip.dptr(g_kap), _hip.dptr(n_dev), ctypes.c_double(lr), ctypes.c_double(beta1), ctypes.c_double(beta2),
                                               ctypes.c_double(eps), _hip.stream_ptr(s2.device))
    _hip.check(rc, "wiski_hyper_adam")


def multi_copy(pairs, scalar=None, scalar_dst=None):
    """``wiski_multi_copy_f64``: dst.copy_(src) for up to 12 (dst, src) pairs of contiguous fp64 tensors (and one scalar store) in ONE launch."""
    plan = _hip.wiski_copy_plan()
    plan.count = len(pairs)
    dev = None
    for i, (dst, src) in enumerate(pairs):
        if dst.dtype != torch.float64 or src.dtype != torch.float64 or dst.numel() != src.numel() or not dst.is_contiguous() or not src.is_contiguous():
            raise _hip.WiskiError("multi_copy: contiguous fp64 tensors of equal size")
        plan.src[i], plan.dst[i], plan.n[i] = src.data_ptr(), dst.data_ptr(), dst.numel()
        dev = dst.device
    if scalar_dst is not None:
        plan.scalar, plan.scalar_dst = float(scalar), scalar_dst.data_ptr()
        dev = scalar_dst.device
    rc = _hip.lib().wiski_multi_copy_f64(ctypes.byref(plan), _hip.stream_ptr(dev))
    _hip.check(rc, "wiski_multi_copy_f64")


def mll_value(bMb, logdet, s2, c, ld, n):
    """``wiski_mll_value``: (val, coef [3]) fp64 device scalars of one output's Woodbury MLL tail; s2 a 1-element tensor (fp32 / fp64)."""
    val = torch.empty((), dtype=torch.float64, device=bMb.device)
    coef = torch.empty(3, dtype=torch.float64, device=bMb.device)
    nd = torch.is_tensor(n)                          # the count as a device scalar (fp64): calls recorded into a captured graph
    rc = _hip.fn("wiski_mll_value", s2.dtype)(_hip.dptr(bMb), None if logdet is None else _hip.dptr(logdet), _hip.dptr(s2), _hip.dptr(c), _hip.dptr(ld),
                                              ctypes.c_double(0.0 if nd else float(n)), _hip.dptr(n) if nd else None, _hip.dptr(val), _hip.dptr(coef),
                                              _hip.stream_ptr(bMb.device))
    _hip.check(rc, "wiski_mll_value")
    return val, coef


def mll_s2_grad(g, coef, s2, n, g_kap):
    out = torch.empty_like(s2)
    nd = torch.is_tensor(n)
    rc = _hip.fn("wiski_mll_s2_grad", s2.dtype)(_hip.dptr(g), _hip.dptr(coef), _hip.dptr(s2), ctypes.c_double(0.0 if nd else float(n)),
                                                _hip.dptr(n) if nd else None, None if g_kap is None else _hip.dptr(g_kap), _hip.dptr(out),
                                                _hip.stream_ptr(s2.device))
    _hip.check(rc, "wiski_mll_s2_grad")
    return out


def gaussian_metrics(mu, var, y, add_var=None):
    """``wiski_gaussian_metrics``: device tensor [rmse, mean nll] of one batch (all arguments contiguous, one dtype)."""
    out = torch.empty(2, dtype=mu.dtype, device=mu.device)
    rc = _hip.fn("wiski_gaussian_metrics", mu.dtype)(ctypes.c_int64(mu.numel()), _hip.dptr(mu), _hip.dptr(var), _hip.dptr(y),
                                                     None if add_var is None else _hip.dptr(add_var), _hip.dptr(out), _hip.stream_ptr(mu.device))
    _hip.check(rc, "wiski_gaussian_metrics")
    return out


def basis_eig_update(g_dev, tcol64, Vin, kw, kuse, ref_Vtab=None, kref=0, resid_ok=None, niter=0):
    """``wiski_basis_eig_update``: the per-dim eigenvector tables Vin ([sum g_q * kw] fp64, row-major [g_q, kw] blocks) refined for the
    Toeplitz columns tcol64, all on the device.  Returns (Vout, ev [d, kw] descending, resid [d]).  resid_ok: the adaptive form
    (``wiski_basis_eig_update_adaptive``): Rayleigh-Ritz in the previous span first, subspace iteration only if the residual exceeds it."""
    d = g_dev.shape[0]
    Vout = torch.empty_like(Vin)
    ev = torch.empty((d, kw), dtype=torch.float64, device=Vin.device)
    resid = torch.empty(d, dtype=torch.float64, device=Vin.device)
    Tq = None if ref_Vtab is None else torch.empty((d, 32, 32), dtype=torch.float64, device=Vin.device)
    args = [ctypes.c_int32(d), _hip.dptr(g_dev), _hip.dptr(tcol64), _hip.dptr(Vin), ctypes.c_int32(kw), ctypes.c_int32(kuse), _hip.dptr(Vout), _hip.dptr(ev),
            _hip.dptr(resid), None if ref_Vtab is None else _hip.dptr(ref_Vtab), ctypes.c_int32(kref), None if Tq is None else _hip.dptr(Tq)]
    if resid_ok is None:
        rc = _hip.lib().wiski_basis_eig_update(*args, _hip.stream_ptr(Vin.device))
    else:
        rc = _hip.lib().wiski_basis_eig_update_adaptive(*args, ctypes.c_int32(int(niter)), ctypes.c_double(float(resid_ok)), _hip.stream_ptr(Vin.device))
    _hip.check(rc, "wiski_basis_eig_update")
    return (Vout, ev, resid) if ref_Vtab is None else (Vout, ev, resid, Tq)


def basis_change(g_dev, Tq, kref, ref_S, kw, S, ev, tcol64, resid, work, verdict_pinned=None):
    """``wiski_basis_change``: (TS [r_ref, r], lam [r], verdict [3]) for a device-refreshed basis; work: r + 1 zeroed doubles.
    verdict_pinned: a pinned host tensor (fp64 [3]) the kernel writes the verdict to directly (host-mapped memory: no copy launch; read it
    after an event recorded behind this call); the returned verdict is then that tensor."""
    d, r_ref = ref_S.shape
    r = S.shape[1]
    TS = torch.empty((r_ref, r), dtype=torch.float64, device=Tq.device)
    lam = torch.empty(r, dtype=torch.float64, device=Tq.device)
    if verdict_pinned is not None:
        if not verdict_pinned.is_pinned() or verdict_pinned.dtype != torch.float64 or verdict_pinned.numel() < 3:
            raise _hip.WiskiError("basis_change: verdict_pinned must be a pinned fp64 tensor of >= 3 elements")
        verdict, vptr = verdict_pinned, ctypes.c_void_p(verdict_pinned.data_ptr())
    else:
        verdict = torch.empty(3, dtype=torch.float64, device=Tq.device)
        vptr = _hip.dptr(verdict)
    rc = _hip.lib().wiski_basis_change(ctypes.c_int32(d), _hip.dptr(g_dev), ctypes.c_int32(kref), ctypes.c_int32(kw), ctypes.c_int32(r_ref),
                                       ctypes.c_int32(r), _hip.dptr(Tq), _hip.dptr(ref_S), _hip.dptr(S), _hip.dptr(ev),
                                       _hip.dptr(tcol64), _hip.dptr(resid), _hip.dptr(TS), _hip.dptr(lam), _hip.dptr(work), vptr,
                                       _hip.stream_ptr(Tq.device))
    _hip.check(rc, "wiski_basis_change")
    return TS, lam, verdict


_REFRESH_LAYOUT = {}


def factor_refresh(g_dev, tcol64, Vin, kw, kuse, ref_Vtab, kref, ref_S, S, work, verdict_pinned, G_ref, h_ref, kscale, info, resid_ok=None, verdict_event=None):
    """``wiski_factor_refresh``: the spectral factor's whole refresh after a hyper-parameter step -- eigenvector update, change of basis
    (+ verdict into pinned memory), G = T^T G_ref T, C, Cholesky + inverse, the tail products -- queued by ONE call into ONE packed buffer.
    verdict_event: a torch.cuda.Event that has been recorded at least once (its handle exists); the call records it behind the change of basis.
    Returns a dict of views: Vtab, ev [d, kw], TS, lam_kuu, G, chol, sqG, lam, sq, Linv and factor_tail's seven outputs."""
    d, r_ref = ref_S.shape
    r = S.shape[1]
    nV = Vin.numel()
    key = (d, nV, kw, r_ref, r)
    off = _REFRESH_LAYOUT.get(key)
    if off is None:
        buf = (ctypes.c_int64 * 15)()
        _hip.check(_hip.lib().wiski_factor_refresh_layout(ctypes.c_int32(d), ctypes.c_int64(nV), ctypes.c_int32(kw), ctypes.c_int32(r_ref), ctypes.c_int32(r), buf),
                   "wiski_factor_refresh_layout")
        off = _REFRESH_LAYOUT[key] = [int(v) for v in buf]
    out = torch.empty(off[14], dtype=torch.float64, device=Vin.device)
    niter, rok = (0, float(resid_ok)) if resid_ok is not None else (2, 1e300)
    rc = _hip.lib().wiski_factor_refresh(ctypes.c_int32(d), _hip.dptr(g_dev), ctypes.c_int64(nV), _hip.dptr(tcol64), _hip.dptr(Vin), ctypes.c_int32(kw),
                                         ctypes.c_int32(kuse), _hip.dptr(ref_Vtab), ctypes.c_int32(kref), ctypes.c_int32(niter), ctypes.c_double(rok),
                                         _hip.dptr(ref_S), _hip.dptr(S), ctypes.c_int32(r_ref), ctypes.c_int32(r), _hip.dptr(work),
                                         ctypes.c_void_p(verdict_pinned.data_ptr()), _hip.dptr(G_ref), _hip.dptr(h_ref), ctypes.c_double(float(kscale)),
                                         _hip.dptr(info), _hip.dptr(out), None if verdict_event is None else ctypes.c_void_p(verdict_event.cuda_event),
                                         _hip.stream_ptr(Vin.device))
    _hip.check(rc, "wiski_factor_refresh")
    t = out[off[13]:off[13] + 6 * r + 2]
    return {"Vtab": out[off[0]:off[0] + nV], "ev": out[off[1]:off[1] + d * kw].view(d, kw), "TS": out[off[4]:off[4] + r_ref * r].view(r_ref, r),
            "lam_kuu": out[off[5]:off[5] + r], "G": out[off[7]:off[7] + r * r].view(r, r), "chol": out[off[8]:off[8] + r * r].view(r, r),
            "sqG": out[off[9]:off[9] + r * r].view(r, r), "lam": out[off[10]:off[10] + r], "sq": out[off[11]:off[11] + r],
            "Linv": out[off[12]:off[12] + r * r].view(r, r),
            "tail": (t[:r], t[r:2 * r], t[2 * r:3 * r], t[3 * r:4 * r], t[4 * r:5 * r], t[5 * r], t[5 * r + 1])}


def factor_tail(TS, h_ref, sq, Linv, chol):
    """``wiski_factor_tail``: (hr, c_half, t, coef, zeta, bMb, logdet) -- views of one packed fp64 tensor -- in three launches."""
    r_ref, r = TS.shape
    out = torch.empty(6 * r + 2, dtype=torch.float64, device=TS.device)
    rc = _hip.lib().wiski_factor_tail(ctypes.c_int32(r_ref), ctypes.c_int32(r), _hip.dptr(TS), _hip.dptr(h_ref), _hip.dptr(sq), _hip.dptr(Linv),
                                      _hip.dptr(chol), _hip.dptr(out), _hip.stream_ptr(TS.device))
    _hip.check(rc, "wiski_factor_tail")
    return out[:r], out[r:2 * r], out[2 * r:3 * r], out[3 * r:4 * r], out[4 * r:5 * r], out[5 * r], out[5 * r + 1]


def woodbury_c(G, lam_kuu, kscale):
    """``wiski_woodbury_c``: (C = I + Lam^1/2 G Lam^1/2, lam = lam_kuu * kscale, sqrt(lam), Lam^1/2 G) in one launch."""
    r = G.shape[0]
    C = torch.empty_like(G)
    sqG = torch.empty_like(G)
    lam = torch.empty(r, dtype=torch.float64, device=G.device)
    sq = torch.empty(r, dtype=torch.float64, device=G.device)
    rc = _hip.lib().wiski_woodbury_c(ctypes.c_int32(r), _hip.dptr(G), _hip.dptr(lam_kuu), ctypes.c_double(float(kscale)), _hip.dptr(C), _hip.dptr(lam),
                                     _hip.dptr(sq), _hip.dptr(sqG), _hip.stream_ptr(G.device))
    _hip.check(rc, "wiski_woodbury_c")
    return C, lam, sq, sqG


def mll_weights(G, P, zeta, lam_kuu, g_b, g_ld):
    """``wiski_mll_weights``: (Wt = g_ld (G - P) + g_b zeta zeta^T, g_kap = sum_i Wt_ii lam_kuu_i); g_b, g_ld fp64 device scalars."""
    r = G.shape[0]
    Wt = torch.empty_like(G)
    g_kap = torch.empty((), dtype=torch.float64, device=G.device)
    rc = _hip.lib().wiski_mll_weights(ctypes.c_int32(r), _hip.dptr(G), _hip.dptr(P), _hip.dptr(zeta), _hip.dptr(lam_kuu), _hip.dptr(g_b), _hip.dptr(g_ld),
                                      _hip.dptr(Wt), _hip.dptr(g_kap), _hip.stream_ptr(G.device))
    _hip.check(rc, "wiski_mll_weights")
    return Wt, g_kap


def basis_lag_grad(g_dev, Vtab, kw, D, scale):
    """``wiski_basis_lag_grad``: [sum g] fp64 gradient w.r.t. the Toeplitz columns from the pair-reduced weights D [d, kw, kw]."""
    out = torch.empty(Vtab.shape[0] // kw, dtype=torch.float64, device=Vtab.device)
    on_dev = torch.is_tensor(scale)                  # a device scalar (fp64): calls recorded into a captured graph
    rc = _hip.lib().wiski_basis_lag_grad(ctypes.c_int32(g_dev.shape[0]), _hip.dptr(g_dev), ctypes.c_int32(kw), _hip.dptr(Vtab), _hip.dptr(D.contiguous()),
                                         ctypes.c_double(0.0 if on_dev else float(scale)), _hip.dptr(scale) if on_dev else None, _hip.dptr(out),
                                         _hip.stream_ptr(Vtab.device))
    _hip.check(rc, "wiski_basis_lag_grad")
    return out


def spectral_var(Y, F, prior, kscale):
    """``wiski_spectral_var``: (|Y[:, j]|^2, max(prior_j * kscale - |F[j]|^2, 0)) for Y [r, n], F [n, r] (fp64, contiguous)."""
    r, n = Y.shape
    diag = torch.empty(n, dtype=torch.float64, device=Y.device)
    tail = torch.empty(n, dtype=torch.float64, device=Y.device)
    if n == 0:
        return diag, tail
    rc = _hip.lib().wiski_spectral_var(ctypes.c_int32(n), ctypes.c_int32(r), _hip.dptr(Y), _hip.dptr(F), _hip.dptr(prior), ctypes.c_double(float(kscale)),
                                       _hip.dptr(diag), _hip.dptr(tail), _hip.stream_ptr(Y.device))
    _hip.check(rc, "wiski_spectral_var")
    return diag, tail


def spectral_evaluate(F, prior, Linv, t, kscale, s2, y, err, ws, want_moments=False):
    """``wiski_spectral_evaluate``: rmse / nll / out-of-grid flag / max |mean| (fp64 [4], on the device) of a query batch from the spectral
    factor: ONE launch for n <= 64 (and r <= 1024), else the MFMA GEMM chol^-1 F^T + one launch (``wiski_spectral_evaluate_y``); with
    `want_moments` also (mean, latent variance) in y's dtype.  ws: 200 zeroed doubles, reused."""
    n, r = F.shape
    out = torch.empty(4, dtype=torch.float64, device=F.device)
    mean = torch.empty(n, dtype=y.dtype, device=F.device) if want_moments else None
    var = torch.empty(n, dtype=y.dtype, device=F.device) if want_moments else None
    if n <= 64 and r <= 1024:
        rc = _hip.fn("wiski_spectral_evaluate", y.dtype)(ctypes.c_int32(n), ctypes.c_int32(r), _hip.dptr(F), _hip.dptr(prior), _hip.dptr(Linv),
                                                         ctypes.c_int32(Linv.shape[1]), _hip.dptr(t), ctypes.c_double(float(kscale)), _hip.dptr(s2), _hip.dptr(y),
                                                         _hip.dptr(err), _hip.dptr(ws), _hip.dptr(out), _hip.dptr(mean), _hip.dptr(var),
                                                         _hip.stream_ptr(F.device))
    else:
        Y = gemm(Linv, F, tb=True)                                                   # chol^-1 F^T  [r, n]
        rc = _hip.fn("wiski_spectral_evaluate_y", y.dtype)(ctypes.c_int32(n), ctypes.c_int32(r), _hip.dptr(Y), _hip.dptr(F), _hip.dptr(prior), _hip.dptr(t),
                                                           ctypes.c_double(float(kscale)), _hip.dptr(s2), _hip.dptr(y), _hip.dptr(err), _hip.dptr(ws),
                                                           _hip.dptr(out), _hip.dptr(mean), _hip.dptr(var), _hip.stream_ptr(F.device))
    _hip.check(rc, "wiski_spectral_evaluate")
    return (out, mean, var) if want_moments else out


def basis_pair_reduce(Wt, S, ev, kmax):
    """D [d, kmax, kmax] of ``wiski_basis_pair_reduce`` (Wt [r, r] fp64, S int32 [d, r], ev fp64 [d, kmax])."""
    d, r = S.shape
    D = torch.zeros((d, kmax, kmax), dtype=torch.float64, device=Wt.device)
    rc = _hip.lib().wiski_basis_pair_reduce(ctypes.c_int32(d), ctypes.c_int32(r), ctypes.c_int32(kmax), _hip.dptr(Wt.contiguous()), _hip.dptr(S),
                                            _hip.dptr(ev), _hip.dptr(D), _hip.stream_ptr(Wt.device))
    _hip.check(rc, "wiski_basis_pair_reduce")
    return D
