"""ctypes binding of libmtg_hip.so (C-ABI: include/mtg.h).

This is the only door between the Python host code and the MI355X kernels.
There is no CPU fallback: if the shared library or a GPU is missing, every
entry point raises (``EngineUnavailable``) -- a likelihood is never computed
on the host.
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ACF_PLAN_SLOTS = 4     # plan pairs mtg_chain_autocorr keeps (csrc/mtg_capi.hip: acf_slots)
# MTG_HIP_LIB selects an alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("MTG_HIP_LIB") or os.path.join(_HERE, "libmtg_hip.so")

# term kinds / mean kinds / status codes: numerically identical to include/mtg.h
TERM_REAL, TERM_COMPLEX3, TERM_COMPLEX4, TERM_SHO, TERM_MATERN32, TERM_JITTER, \
    TERM_DRW, TERM_LORENTZIAN, TERM_COSINUS, TERM_BPL = range(10)
MEAN_CONSTANT, MEAN_LINEAR = 0, 1
ST_OK, ST_PRIOR, ST_NOTPD, ST_NONFINITE = 0, 1, 2, 3
E_ARG, E_NODEVICE, E_HIP, E_STATE, E_UNSUPPORTED = -1, -2, -3, -4, -5

EXPORTS = (
    "mtg_device_count", "mtg_version", "mtg_term_nparams", "mtg_create", "mtg_create_on_slice", "mtg_destroy",
    "mtg_last_error", "mtg_set_lightcurves", "mtg_set_lightcurves_device", "mtg_set_model",
    "mtg_loglike_batch", "mtg_loglike_batch_device", "mtg_loglike_coeffs", "mtg_synchronize",
    "mtg_last_kernel_ms", "mtg_structure_supported", "mtg_profile_begin", "mtg_profile_read",
    "mtg_math_probe", "mtg_ensemble_init", "mtg_ensemble_run", "mtg_ensemble_get",
    "mtg_predict", "mtg_simulate_tk95", "mtg_set_time_parallel", "mtg_set_window_bytes",
    "mtg_apply_inverse", "mtg_set_tp_direct", "mtg_tk95_observe_series", "mtg_rccl_load",
    "mtg_rccl_unique_id", "mtg_ensemble_shard_rccl", "mtg_ensemble_shard_host", "mtg_ensemble_unshard",
    "mtg_chain_autocorr", "mtg_fft_warmup", "mtg_simulate_plan", "mtg_ensemble_restore", "mtg_set_sort", "mtg_set_pipeline", "mtg_set_stream_base", "mtg_set_speculation", "mtg_last_solver", "mtg_pair_contexts", "mtg_unpair_contexts", "mtg_pair_stats", "mtg_set_simulate_pairs", "mtg_set_simulate_transform", "mtg_set_simulate_pdf", "mtg_set_simulate_kraft", "mtg_set_simulate_pdf_draws", "mtg_simulate_pdf_report", "mtg_set_pair_patience", "mtg_chain_autocorr_plans_built",
    "mtg_set_simulate_draws",
    "mtg_ensemble_shard_info", "mtg_ensemble_shard_profile", "mtg_ensemble_shard_profile_read",
)

# the exchange of a walker-sharded ensemble as a callback (include/mtg.h, mtg_exchange_fn)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_int32), ctypes.c_int64, ctypes.c_int64, ctypes.c_int64)


class EngineUnavailable(RuntimeError):
    """libmtg_hip.so is not built, or no MI355X is visible."""


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mtg error %d: %s" % (code, message))
        self.code = code


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lib = None


def _adopt_pytorch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 /
    libhsa-runtime64 (RPATH $ORIGIN, same SONAMEs as /opt/rocm's); if libmtg_hip.so pulled in
    the system copies first, a later `import torch` would bring a second runtime that finds the
    GPUs taken ("No HIP GPUs are available").  So when PyTorch is installed, its copies are
    loaded first -- located without importing torch -- and the dynamic linker resolves
    libmtg_hip.so's libamdhip64.so.7 / libhipfft.so.0 to them by SONAME."""
    if "torch" in sys.modules:
        return  # already resident: the linker will reuse torch's copies
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libamdhip64.so", "libhipfft.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return  # a broken wheel is torch's problem; fall back to /opt/rocm


ROCFFT_CACHE_SEED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocfft_cache_gfx950.db")
ROCFFT_CACHE_STAMP = ROCFFT_CACHE_SEED + ".version"   # the librocfft the seed was made with (make_rocfft_cache.py)
rocfft_cache_seeded = False
_rocfft_cache_copy = None


def rocfft_library_version():
    """Fingerprint of the librocfft this process will load -- PyTorch's bundled copy when PyTorch is installed
    (``_adopt_pytorch_hip_runtime`` loads its libhipfft first), else /opt/rocm's: ``<file name>:<size in bytes>``.
    rocFFT's run-time-compiled code objects are only valid for the build that made them."""
    dirs = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            dirs.append(os.path.join(list(spec.submodule_search_locations)[0], "lib"))
    except (ImportError, ValueError):
        pass
    dirs.append(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"))
    for libdir in dirs:
        for name in ("librocfft.so", "librocfft.so.0"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                real = os.path.realpath(path)
                return "%s:%d" % (os.path.basename(real), os.path.getsize(real))
    return "unknown"


def _seed_rocfft_cache():
    """rocFFT compiles the kernels of every new transform length at run time (~1.2 s each on this ROCm build) and
    keeps them in the file ROCFFT_RTC_CACHE_PATH names -- by default under ~/.cache, which a fresh machine or an
    ephemeral box does not have.  A SEED of that file with the power-of-two lengths mtg_chain_autocorr uses may lie
    next to the library (generated on a GPU by scripts/make_rocfft_cache.py -- __graft_entry__.build() does it where a
    GPU is visible --, git-ignored: 1.4 MB of code objects are not source).  Without it everything works: the
    convergence check's transforms run on the host until a device plan has paid for itself
    (device_sampler._autocorr_time_where_it_is_cheapest).  The seed itself is never opened by rocFFT: every
    process works on its own copy in the temporary directory (rocFFT writes to the file it is given -- eight ranks
    of a node must not share one, and the tracked file must not change under a run), removed at exit; the seed is
    skipped when it was made with another rocFFT build (version stamp next to it) or when the user chose a file."""
    global rocfft_cache_seeded, _rocfft_cache_copy
    rocfft_cache_seeded = False
    if "ROCFFT_RTC_CACHE_PATH" in os.environ:
        return
    if not os.path.exists(ROCFFT_CACHE_SEED):
        return
    try:
        stamp = open(ROCFFT_CACHE_STAMP).read().strip()
    except OSError:
        stamp = None
    if stamp != rocfft_library_version():
        return
    import atexit
    import shutil
    import tempfile
    try:
        fd, copy = tempfile.mkstemp(prefix="mtg_rocfft_%d_" % os.getpid(), suffix=".db")
        os.close(fd)
        shutil.copyfile(ROCFFT_CACHE_SEED, copy)
    except OSError:
        return
    _rocfft_cache_copy = copy

    def _remove(path=copy):
        try:
            os.remove(path)
        except OSError:
            pass
    atexit.register(_remove)
    os.environ["ROCFFT_RTC_CACHE_PATH"] = copy
    rocfft_cache_seeded = True


def _ask_for_hardware_queues():
    """More hardware queues than HIP's default of four: contexts that work side by side (the two models of the Protassov
    test, hipFFT's own streams, the fan-out streams of the structures) should not share one -- streams on one hardware
    queue run strictly one after the other.  The HIP runtime reads GPU_MAX_HW_QUEUES once, when it initialises: the
    variable is set here only if nobody set it (the user's choice wins), and a process that is known to have initialised
    HIP already -- PyTorch says so -- is told once what it keeps instead (a side-by-side refit of both models of BASELINE
    configs[3] on a rank's share of 8 GPUs: 4.07 s on shared queues against 3.55 s)."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    torch = sys.modules.get("torch")
    try:
        late = torch is not None and torch.cuda.is_initialized()
    except Exception:
        late = False
    if late:
        import warnings
        warnings.warn("mind_the_gaps_amd: HIP was initialised before this library was loaded, so GPU_MAX_HW_QUEUES=8 cannot "
                      "take effect any more; refits that run side by side (ppp.protassov_test) may share a hardware queue and "
                      "serialise (about 15 % slower).  Set GPU_MAX_HW_QUEUES=8 in the environment before the first GPU call.",
                      RuntimeWarning, stacklevel=3)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


def load_library():
    """dlopen libmtg_hip.so and declare the prototypes (no GPU needed).  The first call also asks the HIP runtime for
    eight hardware queues (``_ask_for_hardware_queues``) and points rocFFT at a private copy of the kernel-cache seed
    (``_seed_rocfft_cache``): two environment variables, set only when the user left them unset, and only here -- importing
    the package touches nothing."""
    global _lib
    if _lib is not None:
        return _lib
    _ask_for_hardware_queues()
    _adopt_pytorch_hip_runtime()
    _seed_rocfft_cache()
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mind_the_gaps_amd/csrc`" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise EngineUnavailable("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    c_i64, c_int, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    lib.mtg_device_count.restype = c_int
    lib.mtg_version.restype = ctypes.c_char_p
    lib.mtg_term_nparams.restype = c_int
    lib.mtg_term_nparams.argtypes = [c_int]
    lib.mtg_create.restype = c_vp
    lib.mtg_create.argtypes = [c_int]
    lib.mtg_create_on_slice.restype = c_vp
    lib.mtg_create_on_slice.argtypes = [c_int, c_int, c_int]
    lib.mtg_destroy.restype = None
    lib.mtg_destroy.argtypes = [c_vp]
    lib.mtg_last_error.restype = ctypes.c_char_p
    lib.mtg_last_error.argtypes = [c_vp]
    lib.mtg_set_lightcurves.restype = c_int
    lib.mtg_set_lightcurves.argtypes = [c_vp, c_i64, c_i64, _dp, c_int, _dp, _dp, _dp]
    lib.mtg_set_lightcurves_device.restype = c_int
    lib.mtg_set_lightcurves_device.argtypes = [c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_vp, c_vp]
    lib.mtg_set_model.restype = c_int
    lib.mtg_set_model.argtypes = [c_vp, c_int, _ip, _dp, c_int, c_int, _dp, c_int, _ip, _dp]
    lib.mtg_loglike_batch.restype = c_int
    lib.mtg_loglike_batch.argtypes = [c_vp, c_i64, _dp, _ip, c_int, _dp, _ip]
    lib.mtg_loglike_batch_device.restype = c_int
    lib.mtg_loglike_batch_device.argtypes = [c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]
    lib.mtg_loglike_coeffs.restype = c_int
    lib.mtg_loglike_coeffs.argtypes = [c_vp, c_i64, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                       _dp, c_int, _dp, _ip, _dp, _ip]
    lib.mtg_synchronize.restype = c_int
    lib.mtg_synchronize.argtypes = [c_vp]
    lib.mtg_last_kernel_ms.restype = ctypes.c_double
    lib.mtg_last_kernel_ms.argtypes = [c_vp]
    lib.mtg_profile_begin.restype = c_int
    lib.mtg_profile_begin.argtypes = [c_vp, c_int]
    lib.mtg_profile_read.restype = c_int
    lib.mtg_profile_read.argtypes = [c_vp, c_int, _dp, _dp]
    lib.mtg_ensemble_init.restype = c_int
    lib.mtg_ensemble_init.argtypes = [c_vp, c_i64, c_int, ctypes.c_uint64, _dp, _ip]
    lib.mtg_ensemble_run.restype = c_int
    lib.mtg_ensemble_run.argtypes = [c_vp, c_int, _dp, _dp]
    lib.mtg_rccl_load.restype = c_int
    lib.mtg_rccl_load.argtypes = [ctypes.c_char_p]
    lib.mtg_rccl_unique_id.restype = c_int
    lib.mtg_rccl_unique_id.argtypes = [c_vp]
    lib.mtg_ensemble_shard_rccl.restype = c_int
    lib.mtg_ensemble_shard_rccl.argtypes = [c_vp, c_vp, c_int, c_int]
    lib.mtg_ensemble_shard_host.restype = c_int
    lib.mtg_ensemble_shard_host.argtypes = [c_vp, c_int, c_int, EXCHANGE_FN, c_vp]
    lib.mtg_ensemble_unshard.restype = c_int
    lib.mtg_ensemble_unshard.argtypes = [c_vp]
    lib.mtg_fft_warmup.restype = c_int
    lib.mtg_fft_warmup.argtypes = [c_vp]
    lib.mtg_simulate_plan.restype = c_int
    lib.mtg_simulate_plan.argtypes = [c_vp, c_i64]
    lib.mtg_set_simulate_pairs.restype = c_int
    lib.mtg_set_simulate_pairs.argtypes = [c_vp, c_int]
    lib.mtg_set_simulate_transform.restype = c_int
    lib.mtg_set_simulate_transform.argtypes = [c_vp, c_int]
    lib.mtg_set_pair_patience.restype = c_int
    lib.mtg_set_pair_patience.argtypes = [c_vp, c_int]
    lib.mtg_set_simulate_pdf.restype = c_int
    lib.mtg_set_simulate_pdf.argtypes = [c_vp, c_int, c_int]
    lib.mtg_set_simulate_kraft.restype = c_int
    lib.mtg_set_simulate_kraft.argtypes = [c_vp, c_i64, c_int, ctypes.c_double, _dp, _dp, _dp, _dp]
    lib.mtg_set_simulate_pdf_draws.restype = c_int
    lib.mtg_set_simulate_pdf_draws.argtypes = [c_vp, c_i64, c_i64, _dp]
    lib.mtg_simulate_pdf_report.restype = c_int
    lib.mtg_simulate_pdf_report.argtypes = [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]
    lib.mtg_chain_autocorr_plans_built.restype = c_i64
    lib.mtg_chain_autocorr_plans_built.argtypes = [c_vp]
    lib.mtg_set_simulate_draws.restype = c_int
    lib.mtg_set_simulate_draws.argtypes = [c_vp, c_i64, c_i64, _dp, ctypes.POINTER(c_i64)]
    lib.mtg_pair_contexts.restype = c_int
    lib.mtg_pair_contexts.argtypes = [c_vp, c_vp]
    lib.mtg_unpair_contexts.restype = c_int
    lib.mtg_unpair_contexts.argtypes = [c_vp]
    lib.mtg_pair_stats.restype = c_int
    lib.mtg_pair_stats.argtypes = [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_int)]
    lib.mtg_set_sort.restype = c_int
    lib.mtg_set_sort.argtypes = [c_vp, c_int]
    lib.mtg_set_stream_base.restype = c_int
    lib.mtg_set_stream_base.argtypes = [c_vp, ctypes.c_int64]
    lib.mtg_set_pipeline.restype = c_int
    lib.mtg_set_pipeline.argtypes = [c_vp, c_int]
    lib.mtg_set_speculation.restype = c_int
    lib.mtg_set_speculation.argtypes = [c_vp, c_int]
    lib.mtg_last_solver.restype = ctypes.c_char_p
    lib.mtg_last_solver.argtypes = [c_vp]
    lib.mtg_ensemble_shard_info.restype = c_int
    lib.mtg_ensemble_shard_info.argtypes = [c_vp, _ip, _ip, _ip, _ip]
    lib.mtg_ensemble_shard_profile.restype = c_int
    lib.mtg_ensemble_shard_profile.argtypes = [c_vp, c_int]
    lib.mtg_ensemble_shard_profile_read.restype = c_int
    lib.mtg_ensemble_shard_profile_read.argtypes = [c_vp, c_int, _dp]
    lib.mtg_chain_autocorr.restype = c_int
    lib.mtg_chain_autocorr.argtypes = [c_vp, c_i64, c_i64, c_int, c_int, _dp, _dp]
    lib.mtg_ensemble_restore.restype = c_int
    lib.mtg_ensemble_restore.argtypes = [c_vp, c_i64, _dp, _ip, _dp, _dp]
    lib.mtg_ensemble_get.restype = c_int
    lib.mtg_ensemble_get.argtypes = [c_vp, _dp, _dp, _dp, _dp, _ip, ctypes.POINTER(c_i64), _ip]
    lib.mtg_simulate_tk95.restype = c_int
    lib.mtg_simulate_tk95.argtypes = [c_vp, c_i64, _dp, _dp, c_i64, ctypes.c_uint64, c_i64, ctypes.c_double, ctypes.c_double,
                                      c_i64, _ip, _ip, c_int, ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp, c_int]
    lib.mtg_tk95_observe_series.restype = c_int
    lib.mtg_tk95_observe_series.argtypes = [c_vp, c_i64, c_i64, c_i64, c_i64, _dp, _ip, _ip, _dp]
    lib.mtg_set_time_parallel.restype = c_int
    lib.mtg_set_time_parallel.argtypes = [c_vp, c_int]
    lib.mtg_set_tp_direct.restype = c_int
    lib.mtg_set_tp_direct.argtypes = [c_vp, c_int]
    lib.mtg_apply_inverse.restype = c_int
    lib.mtg_apply_inverse.argtypes = [c_vp, _dp, ctypes.c_int32, c_i64, _dp, _ip]
    lib.mtg_set_window_bytes.restype = c_int
    lib.mtg_set_window_bytes.argtypes = [c_vp, ctypes.c_uint64]
    lib.mtg_predict.restype = c_int
    lib.mtg_predict.argtypes = [c_vp, c_i64, _dp, _ip, _dp, _dp, _ip]
    lib.mtg_math_probe.restype = c_int
    lib.mtg_math_probe.argtypes = [c_vp, c_i64, _dp, _dp, _dp, _dp, _dp]
    lib.mtg_structure_supported.restype = c_int
    lib.mtg_structure_supported.argtypes = [c_int, c_int]
    _lib = lib
    return lib


def device_count():
    return int(load_library().mtg_device_count())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


_live_engines = None


def _register_for_exit(engine):
    """Helper threads (hipFFT start-up, the simulator's plan) are daemons: at interpreter exit one of them may still be
    inside plan creation, and being killed there takes the process down with it.  They are joined first."""
    global _live_engines
    if _live_engines is None:
        import atexit
        import weakref
        _live_engines = weakref.WeakSet()

        def join_all():
            for eng in list(_live_engines):
                try:
                    eng._join_fft_warmup()
                except Exception:
                    pass
        atexit.register(join_all)
    _live_engines.add(engine)


def _one_thread_at_a_time(method):
    """An ``mtg_ctx`` is not re-entrant: a second thread that re-uploads light curves while the first one's kernels are
    in flight ends in a GPU memory fault.  Calls from another thread while one is inside the library are refused."""
    import functools

    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        if not self._busy.acquire(blocking=False):
            raise EngineError(E_STATE, "this Engine is inside a call made by another thread; an mtg_ctx serves one "
                                       "thread at a time -- give every thread its own Engine")
        try:
            return method(self, *args, **kwargs)
        finally:
            self._busy.release()
    return guarded


class Engine:
    """One MI355X with resident light curves and a model (an ``mtg_ctx``)."""

    def __init__(self, device=0, cu_slice=None):
        """``cu_slice`` = (part, parts): the context's kernels keep to that slice of the GPU's compute units
        (include/mtg.h: mtg_create_on_slice), so that contexts on different slices run side by side."""
        import threading
        self._busy = threading.RLock()
        self._lib = load_library()
        if cu_slice is None:
            self._ctx = self._lib.mtg_create(int(device))
        else:
            self._ctx = self._lib.mtg_create_on_slice(int(device), int(cu_slice[0]), int(cu_slice[1]))
        if not self._ctx:
            raise EngineUnavailable(
                "mtg_create(%d) failed: %s" % (device, self._lib.mtg_last_error(None).decode()))
        self.device = int(device)
        self.N = 0
        self.L = 0
        self.P = None
        _register_for_exit(self)

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._join_fft_warmup()   # the helper thread works on this context's device
            self._lib.mtg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            pending, self._exchange_error = getattr(self, "_exchange_error", None), None
            if pending is not None:  # the exchange callback of a walker-sharded ensemble raised
                raise pending
            raise EngineError(rc, self._lib.mtg_last_error(self._ctx).decode())

    # -- data ---------------------------------------------------------------
    def set_lightcurves(self, t, y, yerr, y_offset=None):
        """t: [N] (shared sampling) or [L][N]; y, yerr: [N] or [L][N].  ``yerr`` is what
        celerite's ``compute`` receives, i.e. ``dy + 1e-12`` in the reference (gpmodelling.py:54).
        ``y_offset``: [L] frozen per-light-curve means (gpmodelling.py:83-87), subtracted at upload."""
        y = np.atleast_2d(_f64(y))
        dy = np.atleast_2d(_f64(yerr))
        t = _f64(t)
        L, N = y.shape
        if dy.shape != (L, N):
            raise ValueError("y and yerr must have the same shape")
        t_per_lc = 0
        if t.ndim == 2:
            if t.shape == (L, N) and L > 1:
                t_per_lc = 1
            elif t.shape[1] == N and t.shape[0] == 1:
                t = t[0]
            else:
                raise ValueError("t must be [N] or [L][N]")
        if t.ndim == 1 and t.shape[0] != N:
            raise ValueError("t and y lengths differ")
        off = None
        if y_offset is not None:
            off = _f64(np.broadcast_to(np.asarray(y_offset, dtype=np.float64), (L,)))
        rc = self._lib.mtg_set_lightcurves(self._ctx, N, L, _ptr(np.ascontiguousarray(t)), t_per_lc,
                                           _ptr(y), _ptr(dy), _ptr(off))
        if rc == E_ARG and b"sorted" in self._lib.mtg_last_error(self._ctx):
            raise ValueError("the input coordinates must be sorted")  # celerite GP.compute wording
        self._check(rc)
        self.N, self.L = N, L

    def set_lightcurves_device(self, N, L, t_ptr, y_ptr, dy_ptr, t_per_lc=False, y_offset_ptr=None):
        self._check(self._lib.mtg_set_lightcurves_device(self._ctx, N, L, t_ptr, int(bool(t_per_lc)),
                                                         y_ptr, dy_ptr, y_offset_ptr))
        self.N, self.L = int(N), int(L)

    def set_model(self, kinds, full_values, free_index, bounds, mean_kind=MEAN_CONSTANT, extra=None):
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        full_values = _f64(full_values)
        free_index = np.ascontiguousarray(free_index, dtype=np.int32)
        bounds = _f64(bounds).reshape(-1, 2)
        if bounds.shape[0] != full_values.shape[0]:
            raise ValueError("bounds must hold one (lo, hi) pair per full parameter")
        extra_ = _f64(extra) if extra is not None else None
        self._check(self._lib.mtg_set_model(self._ctx, len(kinds), _iptr(kinds), _ptr(extra_),
                                            int(mean_kind), len(full_values), _ptr(full_values),
                                            len(free_index), _iptr(free_index), _ptr(bounds)))
        self.P = len(free_index)

    # -- evaluation ---------------------------------------------------------
    def loglike(self, theta, lc_index=None, add_prior=True):
        """theta: [B][P] -> (lnP[B], status[B])."""
        theta = np.atleast_2d(_f64(theta))
        B = theta.shape[0]
        if self.P is None:
            raise EngineError(E_STATE, "set_model has not been called")
        if theta.shape[1] != self.P:
            raise ValueError("theta has %d columns, the model has %d free parameters"
                             % (theta.shape[1], self.P))
        lc = None if lc_index is None else np.ascontiguousarray(lc_index, dtype=np.int32)
        if lc is not None and lc.shape != (B,):
            raise ValueError("lc_index must have one entry per theta row")
        out = np.empty(B, dtype=np.float64)
        status = np.empty(B, dtype=np.int32)
        self._check(self._lib.mtg_loglike_batch(self._ctx, B, _ptr(theta), _iptr(lc),
                                                int(bool(add_prior)), _ptr(out), _iptr(status)))
        return out, status

    def loglike_device(self, B, theta_ptr, lc_ptr, out_ptr, status_ptr, add_prior=True, stream=None):
        """Raw device pointers (ints); asynchronous on ``stream``: a hipStream_t as an int -- 0 is
        HIP's default stream, e.g. ``torch.cuda.current_stream().cuda_stream`` -- or None for the
        context's own stream (``synchronize()`` waits for that one)."""
        handle = ctypes.c_void_p(-1 if stream is None else int(stream))   # MTG_STREAM_CONTEXT
        self._check(self._lib.mtg_loglike_batch_device(self._ctx, int(B), theta_ptr, lc_ptr,
                                                       int(bool(add_prior)), out_ptr, status_ptr,
                                                       handle))

    def loglike_coeffs(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter=None,
                       mean_kind=MEAN_CONSTANT, mean_params=None, lc_index=None):
        """Raw celerite coefficients [B][jr] / [B][jc] evaluated on the host."""
        a_real, c_real = np.atleast_2d(_f64(a_real)), np.atleast_2d(_f64(c_real))
        a_comp, b_comp = np.atleast_2d(_f64(a_comp)), np.atleast_2d(_f64(b_comp))
        c_comp, d_comp = np.atleast_2d(_f64(c_comp)), np.atleast_2d(_f64(d_comp))
        B = max(a_real.shape[0], a_comp.shape[0])
        jr = a_real.shape[1] if a_real.size else 0
        jc = a_comp.shape[1] if a_comp.size else 0
        jit = _f64(jitter) if jitter is not None else None
        mp = np.atleast_2d(_f64(mean_params)) if mean_params is not None else None
        lc = None if lc_index is None else np.ascontiguousarray(lc_index, dtype=np.int32)
        out = np.empty(B, dtype=np.float64)
        status = np.empty(B, dtype=np.int32)
        self._check(self._lib.mtg_loglike_coeffs(
            self._ctx, B, jr, jc, _ptr(a_real) if jr else None, _ptr(c_real) if jr else None,
            _ptr(a_comp) if jc else None, _ptr(b_comp) if jc else None,
            _ptr(c_comp) if jc else None, _ptr(d_comp) if jc else None, _ptr(jit),
            int(mean_kind), _ptr(mp), _iptr(lc), _ptr(out), _iptr(status)))
        return out, status

    def profile_begin(self, capacity):
        """Record HIP-event timings of the next ``capacity`` batch calls."""
        self._check(self._lib.mtg_profile_begin(self._ctx, int(capacity)))
        self._prof_cap = int(capacity)

    def profile_read(self):
        """-> (prepare_ms[n], solve_ms[n]) of the calls recorded since profile_begin."""
        cap = getattr(self, "_prof_cap", 0)
        prep = np.zeros(cap)
        solve = np.zeros(cap)
        n = self._lib.mtg_profile_read(self._ctx, cap, _ptr(prep), _ptr(solve))
        if n < 0:
            self._check(n)
        return prep[:n], solve[:n]

    # -- device-resident ensembles ------------------------------------------------------
    def ensemble_init(self, coords, seed=0, lc_of_ensemble=None):
        """coords [E][W][P]: start E lock-step ensembles of W walkers on the device."""
        coords = _f64(coords)
        if coords.ndim != 3 or coords.shape[2] != self.P:
            raise ValueError("coords must be [E][W][P] with P = %r free parameters" % (self.P,))
        lc = None if lc_of_ensemble is None else np.ascontiguousarray(lc_of_ensemble, dtype=np.int32)
        if lc is not None and lc.shape != (coords.shape[0],):
            raise ValueError("lc_of_ensemble must have one entry per ensemble")
        self._check(self._lib.mtg_ensemble_init(self._ctx, coords.shape[0], coords.shape[1],
                                                int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(coords), _iptr(lc)))
        self._ens_shape = coords.shape

    # -- walker sharding of the resident ensembles (include/mtg.h, mtg_ensemble_shard_*) ------
    def rccl_unique_id(self):
        """A fresh ncclUniqueId (128 bytes): made by ONE rank, handed to the others by the caller."""
        buf = ctypes.create_string_buffer(128)
        rc = self._lib.mtg_rccl_unique_id(buf)
        if rc:
            raise EngineUnavailable("librccl.so.1 could not be loaded (mtg_rccl_unique_id -> %d)" % rc)
        return buf.raw

    def ensemble_shard_rccl(self, unique_id, rank, world):
        """This process evaluates its block of every half-step's proposals; ncclAllGather on the
        engine's stream brings the others' log-probabilities.  Collective over the ``world`` ranks."""
        if len(unique_id) != 128:
            raise ValueError("a ncclUniqueId is 128 bytes")
        self._check(self._lib.mtg_ensemble_shard_rccl(self._ctx, ctypes.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def ensemble_shard_host(self, rank, world, exchange):
        """Same, with ``exchange(lnp[count], status[count], lo, hi)`` -- numpy views of the host staging
        arrays, rows [lo, hi) filled in -- responsible for filling in every other row (any transport)."""
        def trampoline(_user, lnp_p, st_p, count, lo, hi):
            try:
                exchange(np.ctypeslib.as_array(lnp_p, shape=(count,)), np.ctypeslib.as_array(st_p, shape=(count,)),
                         int(lo), int(hi))
                return 0
            except Exception as exc:  # noqa: BLE001 -- reported through the C return code, re-raised by _check
                self._exchange_error = exc
                return 1
        self._exchange_cb = EXCHANGE_FN(trampoline)  # keep the thunk alive as long as the engine uses it
        self._check(self._lib.mtg_ensemble_shard_host(self._ctx, int(rank), int(world), self._exchange_cb, None))

    def ensemble_unshard(self):
        self._check(self._lib.mtg_ensemble_unshard(self._ctx))
        self._exchange_cb = None

    def ensemble_shard_info(self):
        """dict(kind: "none" | "rccl" | "host", rank, world, comm_ranks: size of the library's RCCL communicator or 0)."""
        v = [np.zeros(1, dtype=np.int32) for _ in range(4)]
        self._check(self._lib.mtg_ensemble_shard_info(self._ctx, *[_iptr(a) for a in v]))
        return dict(kind=("none", "rccl", "host")[int(v[0][0])], rank=int(v[1][0]), world=int(v[2][0]), comm_ranks=int(v[3][0]))

    def shard_profile_begin(self, capacity):
        """HIP events around the first ``capacity`` RCCL exchanges of the following ``ensemble_run`` calls."""
        self._check(self._lib.mtg_ensemble_shard_profile(self._ctx, int(capacity)))
        self._shard_prof_cap = int(capacity)

    def shard_profile_read(self):
        cap = getattr(self, "_shard_prof_cap", 0)
        ms = np.zeros(cap)
        n = self._lib.mtg_ensemble_shard_profile_read(self._ctx, cap, _ptr(ms))
        if n < 0:
            self._check(n)
        return ms[:n]

    def ensemble_run(self, steps, store_chain=False):
        """Advance every ensemble ``steps`` iterations; optionally return
        (chain [steps][E][W][P], log_prob [steps][E][W])."""
        E, W, P = self._ens_shape
        chain = np.empty((steps, E, W, P)) if store_chain else None
        lnp = np.empty((steps, E, W)) if store_chain else None
        self._check(self._lib.mtg_ensemble_run(self._ctx, int(steps), _ptr(chain), _ptr(lnp)))
        return chain, lnp

    def start_fft_warmup(self):
        """hipFFT's one-time start-up (~1.4 s) on a helper thread, so that the first convergence check or simulation
        that wants the device does not wait for it; ``fft_ready`` turns True when it is done."""
        import threading
        if getattr(self, "fft_ready", False) or getattr(self, "_fft_thread", None) is not None:
            return

        ctx = self._ctx

        def work():
            if self._lib.mtg_fft_warmup(ctx) == 0:   # (selects the context's device: HIP's current device is per thread)
                self.fft_ready = True
        self._fft_thread = threading.Thread(target=work, name="mtg-fft-warmup", daemon=True)
        self._fft_thread.start()

    def _join_fft_warmup(self):
        for name in ("_fft_thread", "_sim_plan_thread"):
            thread = getattr(self, name, None)
            if thread is not None and thread.is_alive():
                thread.join()

    def start_simulate_warmup(self, nfft):
        """The plan ``simulate_tk95`` needs for series of ``nfft`` points, built on a helper thread (mtg_simulate_plan):
        0.9 s for the Bluestein plan of BASELINE configs[3], hidden behind whatever the context does meanwhile."""
        import threading
        thread = getattr(self, "_sim_plan_thread", None)
        if thread is not None and thread.is_alive():
            return
        ctx, nfft = self._ctx, int(nfft)
        self._sim_plan_thread = threading.Thread(target=lambda: self._lib.mtg_simulate_plan(ctx, nfft),
                                                 name="mtg-simulate-plan", daemon=True)
        self._sim_plan_thread.start()

    def chain_autocorr(self, chain):
        """chain [n_t][W][P] -> walker-averaged normalised autocorrelation function [n_t][P] (emcee's
        ``function_1d`` per walker and dimension, averaged), computed on the device; or, for E independent
        ensembles at once, [n_t][E][W][P] -> [n_t][E][P]."""
        chain = _f64(chain)
        if chain.ndim not in (3, 4):
            raise ValueError("chain must be [n_t][W][P] or [n_t][E][W][P]")
        n_t, (W, P) = chain.shape[0], chain.shape[-2:]
        E = chain.shape[1] if chain.ndim == 4 else 1
        rho = np.empty((n_t, E, P))
        self._join_fft_warmup()   # (two threads inside hipFFT's first plan is not something to find out about)
        self._check(self._lib.mtg_chain_autocorr(self._ctx, n_t, E, W, P, _ptr(chain), _ptr(rho)))
        self.fft_ready = True
        # the shapes the library now holds plans for, least recently used first: kept HERE, beside the call that changes
        # them, so that callers of this method and of device_sampler's placement rule see the same list
        key = (1 << max(n_t - 1, 1).bit_length(),) + tuple(chain.shape[1:])
        state = self.__dict__.setdefault("_acf_state", {"planned": [], "rented": {}})
        state["planned"] = [k for k in state["planned"] if k != key][-(ACF_PLAN_SLOTS - 1):] + [key]
        return rho if chain.ndim == 4 else rho[:, 0]

    def ensemble_restore(self, iteration, log_prob=None, naccept=None, best_log_prob=None, best_coords=None):
        """After ``ensemble_init`` with a saved state's coordinates and seed: continue from ``iteration`` with the
        saved ``log_prob`` [E][W] (None: keep what ``ensemble_init`` has just evaluated -- equal to rounding only)."""
        lp = None if log_prob is None else _f64(log_prob)
        if lp is not None and lp.shape != tuple(self._ens_shape[:2]):
            raise ValueError("log_prob must be [E][W]")
        na = None if naccept is None else np.ascontiguousarray(naccept, dtype=np.int32)
        bl = None if best_log_prob is None else _f64(best_log_prob)
        bc = None if best_coords is None else _f64(best_coords)
        self._check(self._lib.mtg_ensemble_restore(self._ctx, int(iteration), _ptr(lp), _iptr(na), _ptr(bl), _ptr(bc)))

    def ensemble_state(self):
        """dict(coords, log_prob, best_log_prob, best_coords, naccept, iteration, n_not_pd)."""
        E, W, P = self._ens_shape
        out = dict(coords=np.empty((E, W, P)), log_prob=np.empty((E, W)), best_log_prob=np.empty(E),
                   best_coords=np.empty((E, P)), naccept=np.empty((E, W), dtype=np.int32))
        it = ctypes.c_int64(0)
        bad = np.zeros(1, dtype=np.int32)
        self._check(self._lib.mtg_ensemble_get(self._ctx, _ptr(out["coords"]), _ptr(out["log_prob"]),
                                               _ptr(out["best_log_prob"]), _ptr(out["best_coords"]),
                                               _iptr(out["naccept"]), ctypes.byref(it), _iptr(bad)))
        out["iteration"] = int(it.value)
        out["n_not_pd"] = int(bad[0])
        return out

    def simulate_tk95(self, theta, seed, nfft, sim_dt, mean_rate, seg_len, win_lo, win_hi, noise_kind=0,
                      sigma_noise=0.0, exposures=None, want_clean=False, make_resident=False, psd_table=None,
                      want_segments=False):
        """TK95 light curves on the resident sampling for S posterior samples ``theta`` of the model, or --
        ``psd_table`` [1 or S][nfft/2 + 1], ``theta`` = S -- for a spectrum tabulated by the caller ->
        dict(rates[S][N], dy[S][N], means[S], clean[S][N] or None, segments[S][seg_len] or None)."""
        table = None
        if psd_table is not None:
            table = np.atleast_2d(_f64(psd_table))
            S = int(theta)
            if table.shape[1] != int(nfft) // 2 + 1 or table.shape[0] not in (1, S):
                raise ValueError("psd_table must be [1 or S][nfft / 2 + 1]")
            theta = None
        else:
            theta = np.atleast_2d(_f64(theta))
            S = theta.shape[0]
        lo = np.ascontiguousarray(win_lo, dtype=np.int32)
        hi = np.ascontiguousarray(win_hi, dtype=np.int32)
        if lo.shape != (self.N,) or hi.shape != (self.N,):
            raise ValueError("win_lo / win_hi must have one entry per epoch")
        expo = None if exposures is None else _f64(np.broadcast_to(exposures, (self.N,)))
        rates, dy, means = np.empty((S, self.N)), np.empty((S, self.N)), np.empty(S)
        clean = np.empty((S, self.N)) if want_clean else None
        segments = np.empty((S, int(seg_len))) if want_segments else None
        self._join_fft_warmup()
        self._check(self._lib.mtg_simulate_tk95(
            self._ctx, S, _ptr(theta), _ptr(table), 0 if table is None else table.shape[0],
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(nfft), float(sim_dt), float(mean_rate),
            int(seg_len), _iptr(lo), _iptr(hi), int(noise_kind), float(sigma_noise), _ptr(expo), _ptr(clean),
            _ptr(rates), _ptr(dy), _ptr(means), _ptr(segments), int(bool(make_resident))))
        self.fft_ready = True   # hipFFT is up from here on (device_sampler: the convergence check may use it at once)
        if make_resident:
            self.L = S
        return dict(rates=rates, dy=dy, means=means, clean=clean, segments=segments)

    def tk95_observe_series(self, series, seg_len, start, win_lo, win_hi):
        """Test entry: window averages of fine-grid ``series`` [S][nfft] -> rates [S][N]."""
        series = np.atleast_2d(_f64(series))
        lo = np.ascontiguousarray(win_lo, dtype=np.int32)
        hi = np.ascontiguousarray(win_hi, dtype=np.int32)
        rates = np.empty((series.shape[0], self.N))
        self._check(self._lib.mtg_tk95_observe_series(self._ctx, series.shape[0], series.shape[1], int(seg_len), int(start),
                                                      _ptr(series), _iptr(lo), _iptr(hi), _ptr(rates)))
        return rates

    def predict(self, theta, lc_index=None):
        """Conditional mean / variance at the training times -> (mu[B][N], var[B][N], status[B]);
        mu excludes the per-light-curve y_offset, var excludes the jitter."""
        theta = np.atleast_2d(_f64(theta))
        B = theta.shape[0]
        lc = None if lc_index is None else np.ascontiguousarray(lc_index, dtype=np.int32)
        mu, var = np.full((B, self.N), np.nan), np.full((B, self.N), np.nan)
        status = np.empty(B, dtype=np.int32)
        self._check(self._lib.mtg_predict(self._ctx, B, _ptr(theta), _iptr(lc), _ptr(mu), _ptr(var),
                                          _iptr(status)))
        return mu, var, status

    def apply_inverse(self, theta, rhs, lc_index=0):
        """K^-1 rhs for rhs[N] or rhs[N][M] at parameter vector ``theta`` -> (x, status)."""
        rhs = _f64(rhs)
        one = rhs.ndim == 1
        x = np.ascontiguousarray(rhs.reshape(self.N, -1)).copy()
        status = np.zeros(1, dtype=np.int32)
        self._check(self._lib.mtg_apply_inverse(self._ctx, _ptr(_f64(theta)), int(lc_index), x.shape[1], _ptr(x),
                                                _iptr(status)))
        return (x[:, 0] if one else x), int(status[0])

    def math_probe(self, x):
        """Device exp(-x), sin(x), cos(x), 1/x of the kernel's own math (accuracy tests)."""
        x = _f64(x).ravel()
        outs = [np.empty_like(x) for _ in range(4)]
        self._check(self._lib.mtg_math_probe(self._ctx, len(x), _ptr(x), *[_ptr(o) for o in outs]))
        return outs

    def set_time_parallel(self, mode):
        """0 = throughput kernel only, 1 = time-parallel kernel whenever available, 2 = auto (default), 3 = the one-wave
        time-parallel kernel whatever the batch (0 and 3: a row's bits do not depend on the batch it travels in)."""
        self._check(self._lib.mtg_set_time_parallel(self._ctx, int(mode)))

    def set_tp_direct(self, enabled):
        """J = 10 time-parallel path: likelihood without the filter pass (default on); see include/mtg.h."""
        self._check(self._lib.mtg_set_tp_direct(self._ctx, int(enabled)))

    def set_speculation(self, mode):
        """0: every half-step of the device sampler gets its own solve; 1 (default): both half-steps of an iteration
        in one batch where the GPU would otherwise idle; 2: as 1, with each iteration's split ranked inside its sampler
        launch (the form long runs fall back to; same chain) (include/mtg.h: mtg_set_speculation)."""
        self._check(self._lib.mtg_set_speculation(self._ctx, int(mode)))

    def set_sort(self, mode):
        """Order of the throughput kernel's sweep: 0 the caller's, 1 sorted by (structure, light curve), 2 auto."""
        self._check(self._lib.mtg_set_sort(self._ctx, int(mode)))

    def set_stream_base(self, first_index):
        """Global index of this context's first ensemble / simulated series: random counters only (include/mtg.h)."""
        self._check(self._lib.mtg_set_stream_base(self._ctx, int(first_index)))

    def set_simulate_pairs(self, on):
        """False: one series per transform in simulate_tk95's chirp-z path -- a series' values then do not depend on the
        other series of the call (include/mtg.h: mtg_set_simulate_pairs)."""
        self._check(self._lib.mtg_set_simulate_pairs(self._ctx, 1 if on else 0))

    def set_simulate_transform(self, mode):
        """"auto" / 0: hipFFT's plan for grid lengths it takes natively, chirp-z otherwise; "library" / 1; "chirp-z" / 2
        (include/mtg.h: mtg_set_simulate_transform)."""
        mode = {"auto": 0, "library": 1, "chirp-z": 2}.get(mode, mode)
        self._check(self._lib.mtg_set_simulate_transform(self._ctx, int(mode)))

    def set_simulate_pdf(self, kind, max_iter=400):
        """The flux PDF of simulate_tk95's light curves: "gaussian" / 0 (TK95 as it is), "lognormal" / 1, "uniform" / 2 (the
        E13 adjustment of every cut segment, on the device; include/mtg.h: mtg_set_simulate_pdf)."""
        kind = {"gaussian": 0, "lognormal": 1, "uniform": 2}.get(str(kind).lower(), kind)
        self._check(self._lib.mtg_set_simulate_pdf(self._ctx, int(kind), int(max_iter)))

    def set_simulate_kraft(self, bkg_counts, bkg_rate_err, median, half, threshold):
        """KraftNoise for simulate_tk95(noise_kind=3): background counts and rate errors per epoch [N], posterior median and
        half-width of the 68 % interval per epoch and total counts [N][K] (include/mtg.h: mtg_set_simulate_kraft)."""
        bkg, err, med, hw = _f64(bkg_counts), _f64(bkg_rate_err), _f64(median), _f64(half)
        if med.ndim != 2 or med.shape != hw.shape or bkg.shape != (med.shape[0],) or err.shape != bkg.shape:
            raise ValueError("median / half must be [N][K], bkg_counts / bkg_rate_err [N]")
        self._check(self._lib.mtg_set_simulate_kraft(self._ctx, med.shape[0], med.shape[1], float(threshold), _ptr(bkg), _ptr(err),
                                                     _ptr(med), _ptr(hw)))

    def set_simulate_pdf_draws(self, draws):
        """The white series the NEXT simulate_tk95's E13 adjustment starts from, [S][seg_len] (None: clear)."""
        if draws is None:
            self._check(self._lib.mtg_set_simulate_pdf_draws(self._ctx, 0, 0, None))
            return
        draws = _f64(draws)
        if draws.ndim != 2:
            raise ValueError("draws must be [S][seg_len]")
        self._check(self._lib.mtg_set_simulate_pdf_draws(self._ctx, draws.shape[0], draws.shape[1], _ptr(draws)))

    def simulate_pdf_report(self):
        """{"not_converged": segments of the last simulate_tk95 that used up their iterations, "iterations": the most any took}"""
        a, b = ctypes.c_int64(0), ctypes.c_int(0)
        self._check(self._lib.mtg_simulate_pdf_report(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return {"not_converged": int(a.value), "iterations": int(b.value)}

    def set_simulate_draws(self, normals, starts):
        """Hand the next simulate_tk95 its random numbers (include/mtg.h: mtg_set_simulate_draws): ``normals`` [S][2][nk]
        standard normals, ``starts`` [S] first fine-grid index of every cut.  ``None``: clear."""
        if normals is None:
            self._check(self._lib.mtg_set_simulate_draws(self._ctx, 0, 0, None, None))
            return
        normals = _f64(normals)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        if normals.ndim != 3 or normals.shape[1] != 2 or starts.shape != (normals.shape[0],):
            raise ValueError("normals must be [S][2][nk] and starts [S]")
        self._check(self._lib.mtg_set_simulate_draws(self._ctx, normals.shape[0], normals.shape[2], _ptr(normals),
                                                     starts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))

    @property
    def acf_plans_built(self):
        """hipFFT plan pairs chain_autocorr has built on this context (a cached shape does not add to it)"""
        return int(self._lib.mtg_chain_autocorr_plans_built(self._ctx))

    def set_pipeline(self, mode):
        """0: never the two-wave pipeline of the serial sweep, 1: whenever compiled, 2 (default): for batches of ~8e3 to
        128 rows per compute unit (include/mtg.h: mtg_set_pipeline)."""
        self._check(self._lib.mtg_set_pipeline(self._ctx, int(mode)))

    def pair_with(self, other):
        """This context's pipelined half-steps and ``other``'s go out in one launch from now on (include/mtg.h:
        mtg_pair_contexts): the two models of the Protassov test refitted side by side, each from a thread of its own."""
        self._check(self._lib.mtg_pair_contexts(self._ctx, other._ctx))

    def unpair(self):
        self._check(self._lib.mtg_unpair_contexts(self._ctx))

    def set_pair_patience(self, milliseconds):
        """Longest host-side wait of this (paired) context for its partner's half-step (include/mtg.h: mtg_set_pair_patience)."""
        self._check(self._lib.mtg_set_pair_patience(self._ctx, int(milliseconds)))

    def pair_stats(self):
        """{"paired": launches shared with the partner, "solo": pipelined launches made alone, "broken": bool} since pairing."""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
        self._check(self._lib.mtg_pair_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"paired": int(a.value), "solo": int(b.value), "broken": bool(c.value)}

    @property
    def last_solver(self):
        """Name of the kernel the last batch was dispatched to (include/mtg.h, mtg_last_solver)."""
        return self._lib.mtg_last_solver(self._ctx).decode()

    def set_window_bytes(self, nbytes):
        """Testing aid: reach of one buffer descriptor of the sweep (default 2^32 - 1); see include/mtg.h."""
        self._check(self._lib.mtg_set_window_bytes(self._ctx, int(nbytes)))

    def synchronize(self):
        self._check(self._lib.mtg_synchronize(self._ctx))

    @property
    def last_kernel_ms(self):
        return float(self._lib.mtg_last_kernel_ms(self._ctx))


for _name in ("set_lightcurves", "set_lightcurves_device", "set_model", "loglike", "loglike_device", "loglike_coeffs",
              "ensemble_init", "ensemble_run", "ensemble_restore", "ensemble_state", "chain_autocorr", "simulate_tk95",
              "tk95_observe_series", "predict", "apply_inverse", "math_probe"):
    setattr(Engine, _name, _one_thread_at_a_time(getattr(Engine, _name)))
del _name
