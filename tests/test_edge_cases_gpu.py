"""GPU edge cases of the hot path: degenerate sampling, extreme parameters, the OCML
(large-phase) sweep variant, empty batches, non-finite input."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c
from oracle import dense

pytestmark = pytest.mark.gpu


def hip_vs_dense(engine, kinds, t, y, dy, thetas, mean, tol=1e-8):
    full = np.concatenate([thetas[0], [mean]])
    bounds = np.tile([-np.inf, np.inf], (len(full), 1))
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, np.arange(len(thetas[0]), dtype=np.int32), bounds)
    out, st = engine.loglike(np.asarray(thetas), add_prior=False)
    worst = 0.0
    for th, o, s in zip(thetas, out, st):
        want = dense.dense_loglike(t, y, dy, dense.build_coeffs(kinds, th), 0, [mean])
        assert s == 0 and np.isfinite(want)
        worst = max(worst, abs(o - want) / abs(want))
    assert worst <= tol, worst
    return worst


def test_duplicate_timestamps_and_huge_gaps(engine):
    """dx = 0 (phi = 1, repeated epochs) and gaps long enough for exp(-c dx) to underflow."""
    rng = np.random.default_rng(1)
    t = np.sort(np.concatenate([np.cumsum(rng.exponential(1.0, 120)), [3.0, 3.0, 40.0]]))
    t[60:] += 5000.0                       # c dx ~ 1500 >> 745 for the DRW term
    t[100:] += 1.0e6
    y, dy = 10 + rng.standard_normal(len(t)), rng.uniform(0.5, 1.5, len(t))
    kinds = synth.ALT_MODEL
    thetas = synth.draw_thetas(kinds, 6, seed=2)
    hip_vs_dense(engine, kinds, t, y, dy, thetas, float(np.mean(y)))


def test_large_phase_increments(engine):
    """d * max(dx) = 3.6e5 rad per step: far beyond the 1e5 that used to send a wave to the OCML sincos variant of the
    sweep, inside what the table reduction takes exactly (MTG_TRIG_FAST_MAX = 1e12)."""
    rng = np.random.default_rng(2)
    t = np.cumsum(rng.exponential(0.5, 150))
    t[75:] += 40.0
    y, dy = rng.standard_normal(150), rng.uniform(0.2, 0.5, 150)
    kinds = [synth.K_COMPLEX3, synth.K_DRW]
    base = np.array([np.log(2.0), np.log(0.3), np.log(9000.0), np.log(1.5), np.log(0.2)])   # d = 9000 rad/day
    thetas = base + 0.01 * rng.standard_normal((5, 5))
    thetas[3:, 2] = np.log(3.0)
    hip_vs_dense(engine, kinds, t, y, dy, thetas, 0.0, tol=1e-8)


def test_phase_increments_beyond_the_table_range_take_the_ocml_sweep(engine):
    """One row with d * max(dx) > 1e12 sends its whole wave through the OCML sincos variant (phases at the elapsed
    time): its wave-mates, ordinary rows, must come out as they do on the table variant, to rounding."""
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.exponential(0.5, 150))
    t[75:] += 2.0e8                                            # a gap of 2e8 days: d dx = 1.8e12 for d = 9000
    y, dy = rng.standard_normal(150), rng.uniform(0.2, 0.5, 150)
    kinds = [synth.K_COMPLEX3, synth.K_DRW]
    base = np.array([np.log(2.0), np.log(0.3), np.log(3.0), np.log(1.5), np.log(0.2)])
    full, free = np.concatenate([base, [0.0]]), np.arange(5, dtype=np.int32)
    bounds = np.tile([-np.inf, np.inf], (6, 1))
    ordinary = base + 0.01 * rng.standard_normal((40, 5))
    wild = base.copy()
    wild[2] = np.log(9000.0)
    engine.set_time_parallel(0)
    try:
        engine.set_lightcurves(t, y, dy + 1e-12)
        engine.set_model(kinds, full, free, bounds)
        alone, st0 = engine.loglike(ordinary, add_prior=False)                       # table variant
        mixed, st1 = engine.loglike(np.vstack([ordinary, wild[None]]), add_prior=False)   # the wave takes OCML
    finally:
        engine.set_time_parallel(2)
    assert np.all(st0 == 0) and np.all(st1[:40] == 0)
    # the OCML variant evaluates d (t_n - t_0) at the elapsed time, ~6e8 rad behind the gap: ulp / 2 = 6e-8 rad of phase
    # noise per sample where the table variant has ~1e-16 -- which is what the difference shows
    assert np.max(np.abs(mixed[:40] - alone) / np.abs(alone)) < 1e-7
    assert np.any(mixed[:40] != alone)                       # (really another variant: not bit for bit)


def test_prior_box_corners_match_oracle(engine):
    """Parameters at the edges of the tutorial box (-10, 50) / (-10, 10)."""
    kinds = synth.NULL_MODEL
    N = 300
    t, y, dy = synth.make_lightcurves(N, 1, seed=4)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    th = synth.truth(kinds)
    corners = []
    for i in range(len(th)):
        for edge in bounds[i]:
            c = th.copy(); c[i] = edge; corners.append(c)
            c = th.copy(); c[i] = edge + np.sign(edge) * 1e-9; corners.append(c)     # just outside
    corners = np.array(corners)
    out, st = engine.loglike(corners, add_prior=True)
    ref, rst = oracle_c.logprob_batch(t, y[0], dy[0], kinds, np.hstack([corners, np.full((len(corners), 1), y.mean())]),
                                      bounds=bounds, add_prior=True)
    assert np.array_equal(st == 1, rst == 1)                 # same prior verdicts (bounds are inclusive)
    both = (st == 0) & (rst == 0)
    assert both.sum() >= len(th)
    assert np.max(np.abs(out[both] - ref[both]) / np.abs(ref[both])) < 1e-8
    assert np.all(np.isneginf(out[st != 0]))


def test_empty_batch_nan_theta_and_state_errors(engine):
    from mind_the_gaps_amd.engine import Engine, EngineError
    kinds = [synth.K_DRW]
    t, y, dy = synth.make_lightcurves(50, 1, seed=5)
    full, free, bounds = synth.model_spec(kinds, y)
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, free, bounds)
    out, st = engine.loglike(np.empty((0, 2)))
    assert out.shape == (0,) and st.shape == (0,)
    th = np.array([[np.nan, 0.0], [1.0, np.nan], synth.truth(kinds)])
    out, st = engine.loglike(th, add_prior=True)             # NaN fails every bound comparison
    assert list(st) == [1, 1, 0] and np.isneginf(out[0]) and np.isfinite(out[2])
    out, st = engine.loglike(th, add_prior=False)            # no prior: NaN propagates to a non-finite lnL
    assert st[0] != 0 and st[1] != 0 and np.isneginf(out[0]) and st[2] == 0
    with pytest.raises(ValueError):
        engine.loglike(np.zeros((2, 3)))                     # wrong number of columns
    with pytest.raises(ValueError):
        engine.set_lightcurves(t[::-1].copy(), y, dy)        # unsorted times (celerite ValueError)
    with pytest.raises(EngineError):
        engine.loglike(th[2:], lc_index=np.array([3]))       # light curve out of range
    fresh = Engine(0)
    with pytest.raises(EngineError):
        fresh.set_model(kinds, full, free, bounds) or fresh.loglike(th[2:])   # no light curves yet
    fresh.close()
    with pytest.raises(EngineError):                         # J = 12 > 10: no compiled kernel
        engine.set_model([synth.K_SHO] * 6, np.zeros(19), np.arange(18, dtype=np.int32),
                         np.tile([-np.inf, np.inf], (19, 1)))
    engine.set_lightcurves(t, y, dy + 1e-12)


def test_single_sample_and_two_samples(engine):
    for N in (1, 2):
        t, y, dy = synth.make_lightcurves(N, 1, seed=6 + N)
        thetas = synth.draw_thetas(synth.ALT_MODEL, 4, seed=1)
        hip_vs_dense(engine, synth.ALT_MODEL, t, y[0], dy[0], thetas, float(np.mean(y)))


class _Hip:
    """hipMalloc / hipMemcpy through ctypes on the HIP runtime the engine already loaded: the
    `_device` entry points take plain device pointers, whoever owns them."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.lib = ctypes.CDLL("libamdhip64.so.7")       # resolves to the copy already in the process
        self.lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.lib.hipFree.argtypes = [ctypes.c_void_p]
        self.owned = []

    def put(self, a):
        a = np.ascontiguousarray(a)
        p = self.c.c_void_p()
        assert self.lib.hipMalloc(self.c.byref(p), max(a.nbytes, 8)) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0           # hipMemcpyHostToDevice
        self.owned.append(p)
        return p.value

    def get(self, ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0    # hipMemcpyDeviceToHost
        return out

    def free(self):
        for p in self.owned:
            self.lib.hipFree(p)
        self.owned = []


def test_device_pointer_entry_points(engine):
    """mtg_set_lightcurves_device / mtg_loglike_batch_device on caller-owned device memory
    against the host-pointer calls; device-side argument checks."""
    from mind_the_gaps_amd.engine import EngineError
    kinds = synth.NULL_MODEL
    N, L, B = 300, 3, 40
    t, y, dy = synth.make_lightcurves(N, L, seed=31)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    off = y.mean(axis=1)
    theta = synth.draw_thetas(kinds, B, seed=2)
    lc = (np.arange(B) % L).astype(np.int32)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=off)
    engine.set_model(kinds, full, free, bounds)
    ref, rst = engine.loglike(theta, lc)

    hip = _Hip()
    try:
        d_t, d_y, d_dy, d_off = (hip.put(a) for a in (t, y, dy + 1e-12, off))
        engine.set_lightcurves_device(N, L, d_t, d_y, d_dy, t_per_lc=False, y_offset_ptr=d_off)
        d_theta, d_lc = hip.put(theta), hip.put(lc)
        d_out, d_st = hip.put(np.zeros(B)), hip.put(np.zeros(B, dtype=np.int32))
        for mode in (0, 1):                                  # throughput and time-parallel kernels
            try:
                engine.set_time_parallel(mode)
                engine.loglike_device(B, d_theta, d_lc, d_out, d_st)
                engine.synchronize()
            finally:
                engine.set_time_parallel(2)
            assert np.array_equal(hip.get(d_st, B, np.int32), rst)
            assert np.allclose(hip.get(d_out, B, np.float64), ref, rtol=1e-10, atol=0)

        # a light-curve index the host cannot see: no fault, a non-OK status
        d_bad = hip.put(np.full(B, 7, dtype=np.int32))
        for mode in (0, 1):
            try:
                engine.set_time_parallel(mode)
                engine.loglike_device(B, d_theta, d_bad, d_out, d_st)
                engine.synchronize()
            finally:
                engine.set_time_parallel(2)
            assert np.all(hip.get(d_st, B, np.int32) != 0) and np.all(np.isneginf(hip.get(d_out, B, np.float64)))

        # unsorted device-resident times are found by the set-up kernel
        d_bad_t = hip.put(t[::-1])
        with pytest.raises(EngineError, match="sorted"):
            engine.set_lightcurves_device(N, L, d_bad_t, d_y, d_dy)
    finally:
        engine.set_lightcurves(t, y, dy + 1e-12)
        hip.free()


def test_ensemble_guards(engine):
    from mind_the_gaps_amd.engine import EngineError
    kinds = [synth.K_DRW]
    t, y, dy = synth.make_lightcurves(100, 2, seed=8)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    p0 = np.stack([synth.draw_thetas(kinds, 8, seed=s) for s in (1, 2)])
    engine.ensemble_init(p0, seed=3)
    engine.ensemble_run(2)
    engine.set_lightcurves(t, y[:1], dy[:1] + 1e-12)          # the ensembles index light curve 1
    with pytest.raises(EngineError, match="changed shape"):
        engine.ensemble_run(1)
    with pytest.raises(EngineError, match="walkers"):
        engine.ensemble_init(np.zeros((1, 4098, 2)), seed=1)


def test_default_stream_ordering(engine):
    """mtg_loglike_batch_device(stream = 0) runs on HIP's default stream, which is PyTorch's default
    stream: a slow producer of d_theta queued there before the launch and a consumer of d_out queued
    after it need no synchronisation in between (a private non-blocking stream would read the stale
    theta)."""
    import torch
    kinds = synth.ALT_MODEL
    N, L, B = 2000, 2, 512
    t, y, dy = synth.make_lightcurves(N, L, seed=77)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=5)
    lc = (np.arange(B) % L).astype(np.int32)
    ref, rst = engine.loglike(theta, lc)
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(theta).to(dev)
    d_lc = torch.from_numpy(lc).to(dev)
    d_theta = torch.zeros_like(src)
    d_out = torch.zeros(B, dtype=torch.float64, device=dev)
    d_st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    x = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize(dev)
    assert torch.cuda.current_stream(dev).cuda_stream == 0
    for _ in range(30):                    # ~tens of ms of queued work in front of the producer
        x = (x @ x) * 1e-4
    d_theta.copy_(src)                     # producer, default stream
    engine.loglike_device(B, d_theta.data_ptr(), d_lc.data_ptr(), d_out.data_ptr(), d_st.data_ptr(),
                          add_prior=True, stream=torch.cuda.current_stream(dev).cuda_stream)
    got = d_out.clone()                    # consumer, default stream
    got_st = d_st.clone()
    torch.cuda.synchronize(dev)
    assert np.array_equal(got_st.cpu().numpy(), rst)
    assert np.array_equal(got.cpu().numpy(), ref)
