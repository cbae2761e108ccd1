"""Resident sets beyond the reach of one buffer descriptor (4 GiB): the sweep places every wave's
descriptors at the first light curve the wave needs, reaches the others within a window, and a
second launch sweeps the evaluations left over, one per wave (csrc/mtg_kernels.hip,
mtg_capi.hip::sweep_launch).  `mtg_set_window_bytes` shrinks the window so that a small set
exercises exactly that logic; results must not depend on it."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

FULL_WINDOW = 2 ** 32 - 1


@pytest.fixture
def windowed(engine):
    engine.set_time_parallel(0)          # the serial sweep is the kernel with 32-bit offsets
    yield engine
    engine.set_window_bytes(FULL_WINDOW)
    engine.set_time_parallel(2)


@pytest.mark.parametrize("kinds", [[synth.K_DRW], synth.ALT_MODEL, [synth.K_SHO, synth.K_SHO]], ids=["drw", "alt", "2sho"])
@pytest.mark.parametrize("order", ["random", "grouped", "descending"])
def test_results_do_not_depend_on_the_window(windowed, kinds, order):
    eng = windowed
    N, L, B = 200, 40, 3000                       # 3200 bytes per light curve, 128 KB resident
    t, y, dy = synth.make_lightcurves(N, L, seed=5)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_window_bytes(FULL_WINDOW)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    rng = np.random.default_rng(3)
    theta = synth.draw_thetas(kinds, B, seed=9, percent=0.6 if synth.K_SHO in kinds else 0.1)   # mixed SHO signatures
    lc = rng.integers(0, L, B).astype(np.int32)
    if order == "grouped":
        lc = np.sort(lc)
    elif order == "descending":
        lc = np.sort(lc)[::-1].copy()
    want, wst = eng.loglike(theta, lc, add_prior=True)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y.mean(axis=1)[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=4)
    ok = wst == 0
    assert np.array_equal(wst, rst) and ok.sum() > B // 3
    assert np.max(np.abs(want[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8
    for window in (3200, 3 * 3200 + 100, 20 * 3200):      # one, three and twenty light curves in reach
        eng.set_window_bytes(window)
        out, st = eng.loglike(theta, lc, add_prior=True)
        assert np.array_equal(st, wst), window
        assert np.array_equal(out[ok], want[ok]), window   # same arithmetic per evaluation: bit for bit


def test_window_with_per_lightcurve_times_and_raw_coefficients(windowed):
    eng = windowed
    N, L, B = 150, 12, 700
    rng = np.random.default_rng(11)
    t = np.cumsum(0.05 + rng.exponential(1.0, (L, N)), axis=1)
    _, y, dy = synth.make_lightcurves(N, L, seed=6)
    kinds = synth.NULL_MODEL
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_window_bytes(FULL_WINDOW)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=2)
    lc = rng.integers(0, L, B).astype(np.int32)
    want, wst = eng.loglike(theta, lc, add_prior=False)
    a_real = np.exp(rng.normal(3.0, 0.3, (B, 1)))
    c_real = np.exp(rng.normal(-1.0, 0.3, (B, 1)))
    e0 = np.empty((B, 0))
    want_c, wst_c = eng.loglike_coeffs(a_real, c_real, e0, e0, e0, e0, lc_index=lc)
    eng.set_window_bytes(2 * N * 16)
    out, st = eng.loglike(theta, lc, add_prior=False)
    out_c, st_c = eng.loglike_coeffs(a_real, c_real, e0, e0, e0, e0, lc_index=lc)
    assert np.array_equal(st, wst) and np.array_equal(out, want) and np.all(st == 0)
    assert np.array_equal(st_c, wst_c) and np.array_equal(out_c, want_c) and np.all(st_c == 0)


def test_window_argument_checks(windowed):
    from mind_the_gaps_amd.engine import EngineError
    eng = windowed
    t, y, dy = synth.make_lightcurves(100, 2, seed=1)
    eng.set_window_bytes(FULL_WINDOW)
    eng.set_lightcurves(t, y, dy + 1e-12)
    with pytest.raises(EngineError):
        eng.set_window_bytes(100 * 16 - 1)          # a resident light curve would not fit
    with pytest.raises(EngineError):
        eng.set_window_bytes(2 ** 32)
    eng.set_window_bytes(100 * 16)
    with pytest.raises(EngineError):
        eng.set_lightcurves(*[a for a in (np.arange(101.0), np.zeros((1, 101)), np.ones((1, 101)))])


def test_device_sampler_under_a_small_window(windowed):
    """The device-resident ensemble sampler goes through the same windowed sweep (its accept
    kernel clears the structure counters the left-over lists share a buffer with): chains must not
    depend on the window.  Ensembles are mapped to light curves in a scattered order so that waves
    do leave the window."""
    eng = windowed
    kinds = synth.ALT_MODEL
    N, L, E, W, steps = 120, 24, 48, 16, 12
    t, y, dy = synth.make_lightcurves(N, L, seed=8)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    rng = np.random.default_rng(4)
    lc_of = rng.integers(0, L, E).astype(np.int32)
    p0 = synth.truth(kinds) * (1 + 0.02 * rng.standard_normal((E, W, len(free))))
    chains = []
    for window in (FULL_WINDOW, 2 * N * 16, N * 16):
        eng.set_window_bytes(FULL_WINDOW)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, free, bounds)
        eng.set_time_parallel(0)
        eng.set_window_bytes(window)
        eng.ensemble_init(p0, seed=77, lc_of_ensemble=lc_of)
        chain, lnp = eng.ensemble_run(steps, store_chain=True)
        assert eng.ensemble_state()["n_not_pd"] == 0 and np.all(np.isfinite(lnp))
        chains.append((chain, lnp))
    for chain, lnp in chains[1:]:
        assert np.array_equal(chain, chains[0][0]) and np.array_equal(lnp, chains[0][1])
    assert len(np.unique(chains[0][0][:, 0, 0, 0])) > 2
