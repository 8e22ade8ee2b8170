"""ctypes binding of libwiski_hip.so (C ABI declared in include/wiski.h).

The library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``);
there is no CPU fallback: every op raises if the extension is missing or if a
tensor is not a contiguous ROCm tensor.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libwiski_hip.so")
_SOURCES = ["interp_gather.hip", "scatter_stats.hip", "solve.hip", "spectral.hip", "dense.hip", "collective.hip", "stream_step.hip", "spectral_basis.hip", "hyper_columns.hip", "two_level.hip", "hyper_step.hip"]
_HEADERS = ["wiski_common.h", "spmv_sym_dma.h", "spmv_sym_dma_mc.h", "spmm_sym_cols.h", "spmm_sym_bcast.h", "scatter_owner.h", "dense_small.h", "dense_coop.h",
            os.path.join("..", "..", "include", "wiski.h")]
MAX_DIM = 4

_lib = None


class WiskiError(RuntimeError):
    pass


_ERRS = {-1: "WISKI_E_BADARG", -2: "WISKI_E_LAUNCH", -3: "WISKI_E_WORKSPACE", -4: "WISKI_E_NOTCONV"}


class wiski_grid(ctypes.Structure):
    _fields_ = [
        ("d", ctypes.c_int32),
        ("g", ctypes.c_int32 * MAX_DIM),
        ("g0", ctypes.c_double * MAX_DIM),
        ("h", ctypes.c_double * MAX_DIM),
    ]


HYPER_MAX_PARAMS = 6


class wiski_hyper_param(ctypes.Structure):
    _fields_ = [
        ("raw", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("step", ctypes.c_void_p),
        ("numel", ctypes.c_int32), ("step_numel", ctypes.c_int32), ("role", ctypes.c_int32), ("kind", ctypes.c_int32),
        ("lower", ctypes.c_double), ("upper", ctypes.c_double),
    ]


class wiski_hyper_plan(ctypes.Structure):
    _fields_ = [("count", ctypes.c_int32), ("reserved", ctypes.c_int32), ("p", wiski_hyper_param * HYPER_MAX_PARAMS)]


COPY_MAX_SEGMENTS = 12


class wiski_copy_plan(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p * COPY_MAX_SEGMENTS), ("dst", ctypes.c_void_p * COPY_MAX_SEGMENTS), ("n", ctypes.c_int64 * COPY_MAX_SEGMENTS),
                ("count", ctypes.c_int32), ("reserved", ctypes.c_int32), ("scalar", ctypes.c_double), ("scalar_dst", ctypes.c_void_p)]


def sources():
    return [os.path.join(_CSRC, s) for s in _SOURCES if os.path.exists(os.path.join(_CSRC, s))]


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into csrc/libwiski_hip.so: one object per source (built concurrently, only when
    the source or a header is newer than its object), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = sources()
    import glob

    hdrs = sorted(set([os.path.join(_CSRC, h) for h in _HEADERS] + glob.glob(os.path.join(_CSRC, "*.h"))))   # every header in csrc/ counts
    hdr_time = max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))
    if not force and os.path.exists(_SO):
        deps = srcs + [h for h in hdrs if os.path.exists(h)]
        if all(os.path.getmtime(d_) <= os.path.getmtime(_SO) for d_ in deps):
            return _SO                               # the shipped library is current (the object cache need not exist)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj])
    if not todo and os.path.exists(_SO) and all(os.path.getmtime(o) <= os.path.getmtime(_SO) for o in objs):
        return _SO

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", _SO] + objs + ["-ldl"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get("WISKI_HIP_SO")          # A/B testing of kernel variants (tools/ubench)
        if alt:
            _lib = ctypes.CDLL(alt)
            _lib.wiski_pcg_workspace_bytes.restype = ctypes.c_int64
            if hasattr(_lib, "wiski_root_update_workspace_elems"):
                _lib.wiski_root_update_workspace_elems.restype = ctypes.c_int64
            return _lib
        if not os.path.exists(_SO):
            raise WiskiError(
                f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). online_gp_amd has no CPU fallback."
            )
        _lib = ctypes.CDLL(_SO)
        _lib.wiski_pcg_workspace_bytes.restype = ctypes.c_int64
        _lib.wiski_twolevel_refresh_workspace_bytes.restype = ctypes.c_int64
    if hasattr(_lib, "wiski_root_update_workspace_elems"):
        _lib.wiski_root_update_workspace_elems.restype = ctypes.c_int64
    return _lib


def check(rc, what):
    if rc != 0:
        raise WiskiError(f"{what} failed: {_ERRS.get(rc, rc)}")


def dptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise WiskiError("online_gp_amd ops need ROCm device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise WiskiError("online_gp_amd ops need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """The caller's current HIP stream on `device` (a torch.device, index or None = current device)."""
    cur = torch.cuda.current_device()
    if device is None:
        idx = cur
    elif isinstance(device, int):
        idx = device
    else:
        idx = torch.device(device).index
        if idx is None:
            idx = cur
    if idx != cur:
        # the library launches on the CURRENT HIP device; a stream of another device would make every launch fail (or hang a
        # convergence poll).  Fail loudly instead of guarding silently: the caller owns the device selection.
        raise RuntimeError(f"online_gp_amd: tensors live on cuda:{idx} but the current device is cuda:{cur}; "
                           f"call torch.cuda.set_device({idx}) (or use `with torch.cuda.device({idx}):`) around the model calls")
    if _raw_stream is not None:                      # ~0.3 us instead of ~3 us through torch.cuda.current_stream
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


_fn_cache = {}


def suffix(dtype):
    if dtype == torch.float32:
        return "_f32"
    if dtype == torch.float64:
        return "_f64"
    raise WiskiError(f"unsupported dtype {dtype}")


def creal(dtype):
    return ctypes.c_float if dtype == torch.float32 else ctypes.c_double


def fn(name, dtype):
    key = (name, dtype)
    f = _fn_cache.get(key)
    if f is None:
        f = _fn_cache[key] = getattr(lib(), name + suffix(dtype))
    return f
