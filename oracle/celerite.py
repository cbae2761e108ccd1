"""oracle/celerite.py -- TEST INFRASTRUCTURE: ctypes view of celerite_ref.c.

See oracle/celerite_ref.c for the reference lines each function follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_celerite.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, "celerite_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_celerite.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _declare(handle):
    for name in ("oracle_logprob_batch", "oracle_logprob_batch_fused"):
        fn = getattr(handle, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [
            ctypes.c_long, ctypes.c_long, _dp, _dp, _dp, ctypes.c_int, _ip, _dp, ctypes.c_int,
            ctypes.c_int, _dp, ctypes.c_long, _dp, _ip, ctypes.c_int, ctypes.c_int, _dp, _ip]
    handle.oracle_max_threads.restype = ctypes.c_int
    return handle


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _declare(ctypes.CDLL(_SO))
    return _lib


def build_native(out_dir):
    """A second copy of the same source built for THIS host (-O3 -march=native): the CPU baseline of
    bench.py.  The copy that travels with the repository is built -O2 without
    -march on another machine.  Returns a ctypes handle, or None when there is no compiler."""
    so = os.path.join(out_dir, "liboracle_celerite_native.so")
    src = os.path.join(_HERE, "celerite_ref.c")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-fvisibility=hidden", "-shared",
                               "-o", so, src, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return _declare(ctypes.CDLL(so))
    except (OSError, subprocess.CalledProcessError):
        return None


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def max_threads():
    return int(lib().oracle_max_threads())


def logprob_batch(t, y, dy, kinds, params_full, bounds=None, lc_index=None, mean_kind=0,
                  extra=None, add_prior=False, nthreads=1, fused=False, handle=None):
    """params_full: [B][PF] full vectors (kernel params then mean params).
    Returns (lnP[B], status[B]).  fused: the one-sweep recurrence instead of celerite's two sweeps
    over stored generators; handle: another build of the library (build_native)."""
    t_, tp = _d(t)
    y_ = np.atleast_2d(np.ascontiguousarray(y, dtype=np.float64))
    dy_ = np.atleast_2d(np.ascontiguousarray(dy, dtype=np.float64))
    L, N = y_.shape
    assert t_.shape == (N,) and dy_.shape == (L, N)
    params_, pp = _d(np.atleast_2d(params_full))
    B, PF = params_.shape
    kinds_, kp = _i(kinds)
    if extra is None:
        extra = np.full(len(kinds_), 0.01)
    extra_, ep = _d(extra)
    if bounds is None:
        bounds = np.tile([-np.inf, np.inf], (PF, 1))
    bounds_, bp = _d(bounds)
    assert bounds_.shape == (PF, 2)
    if lc_index is None:
        lc_, lp = None, None
    else:
        lc_, lp = _i(lc_index)
        assert lc_.shape == (B,) and lc_.min() >= 0 and lc_.max() < L
    out = np.empty(B, dtype=np.float64)
    status = np.zeros(B, dtype=np.int32)
    h = handle if handle is not None else lib()
    fn = h.oracle_logprob_batch_fused if fused else h.oracle_logprob_batch
    rc = fn(N, L, tp, y_.ctypes.data_as(_dp), dy_.ctypes.data_as(_dp),
            len(kinds_), kp, ep, int(mean_kind), PF, bp, B, pp, lp,
            int(bool(add_prior)), int(nthreads),
            out.ctypes.data_as(_dp), status.ctypes.data_as(_ip))
    if rc != 0:
        raise RuntimeError("oracle_logprob_batch failed (unknown term kind?)")
    return out, status
