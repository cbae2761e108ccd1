"""GPU parity: the HIP path (through the C-ABI) against the oracle on the same
seeded inputs.  Tolerance: |dlnL| / |lnL| <= 1e-8 (BASELINE.json north_star);
observed ~1e-13."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

RTOL = 1e-8

MODELS = {
    "drw": [synth.K_DRW],
    "drw+sho": [synth.K_DRW, synth.K_SHO],
    "drw+sho+lor": [synth.K_DRW, synth.K_SHO, synth.K_LORENTZIAN],
    "sho": [synth.K_SHO],
    "real+complex3": [synth.K_REAL, synth.K_COMPLEX3],
    "complex4+jitter": [synth.K_COMPLEX4, synth.K_JITTER],
    "matern32": [synth.K_MATERN32],
    "cosinus+drw": [synth.K_COSINUS, synth.K_DRW],
    "bpl": [synth.K_BPL],
    "5sho": [synth.K_SHO] * 5,
}


def rel(a, b):
    return np.abs(a - b) / np.abs(b)


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("N", [1, 2, 100, 1000])
def test_hip_vs_oracle(engine, name, N):
    kinds = MODELS[name]
    L, B = 3, 96
    t, y, dy = synth.make_lightcurves(N, L, seed=100 + N)
    full, free, bounds = synth.model_spec(kinds, y)
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=7)
    lc = (np.arange(B) % L).astype(np.int32)
    out, st = engine.loglike(theta, lc, add_prior=True)
    full_b = np.hstack([theta, np.full((B, 1), full[-1])])
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, full_b, bounds=bounds, lc_index=lc,
                                      add_prior=True, nthreads=4)
    assert np.array_equal(st, rst)
    ok = st == 0
    assert ok.sum() > B // 2
    assert np.all(np.isneginf(out[~ok]))
    assert rel(out[ok], ref[ok]).max() <= RTOL


def test_hip_vs_dense_n10000(engine):
    """BASELINE size N = 1e4, J = 6 model, against the O(N J^2) oracle."""
    kinds = MODELS["drw+sho+lor"]
    N, L, B = 10000, 2, 128
    t, y, dy = synth.make_lightcurves(N, L, seed=20250704)
    full, free, bounds = synth.model_spec(kinds, y)
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=3)
    lc = (np.arange(B) % L).astype(np.int32)
    out, st = engine.loglike(theta, lc, add_prior=False)
    full_b = np.hstack([theta, np.full((B, 1), full[-1])])
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, full_b, lc_index=lc, nthreads=8)
    assert np.all(st == 0) and np.all(rst == 0)
    assert rel(out, ref).max() <= RTOL


def test_per_lightcurve_frozen_means(engine):
    """The reference freezes each light curve's mean at its own average
    (gpmodelling.py:83-87): y_offset carries those values, the model mean is 0."""
    kinds = MODELS["drw+sho+lor"]
    N, L, B = 700, 5, 160
    t, y, dy = synth.make_lightcurves(N, L, seed=77)
    y += np.arange(L)[:, None] * 13.0                      # clearly different means
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=5)
    lc = (np.arange(B) % L).astype(np.int32)
    out, st = engine.loglike(theta, lc, add_prior=True)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y_mean[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=4)
    assert np.array_equal(st, rst)
    ok = st == 0
    assert rel(out[ok], ref[ok]).max() <= RTOL


def test_per_lightcurve_times(engine):
    """t may differ per light curve ([L][N]); the kernel then reads (dx, t) per lane."""
    kinds = MODELS["drw+sho"]
    N, L, B = 300, 4, 64
    rng = np.random.default_rng(3)
    t = np.vstack([synth.make_times(N, rng, offset=100.0 * i) for i in range(L)])
    dy = rng.uniform(0.5, 2.0, (L, N))
    y = 100.0 + 10.0 * rng.standard_normal((L, N)) + 0.02 * (t - t[:, :1])
    kinds_full = synth.truth(kinds)
    full = np.concatenate([kinds_full, [0.02, 100.0]])           # fitted linear mean
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)] * 2])
    free = np.arange(len(full), dtype=np.int32)
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, free, bounds, mean_kind=1)
    theta = np.hstack([synth.draw_thetas(kinds, B, seed=9), np.tile([0.02, 100.0], (B, 1))])
    lc = (np.arange(B) % L).astype(np.int32)
    out, st = engine.loglike(theta, lc, add_prior=False)
    ref = np.empty(B)
    for l in range(L):
        sel = lc == l
        ref[sel] = oracle_c.logprob_batch(t[l], y[l], dy[l], kinds, theta[sel], mean_kind=1)[0]
    assert np.all(st == 0) and rel(out, ref).max() <= RTOL


@pytest.mark.parametrize("kinds", [[synth.K_DRW, synth.K_SHO, synth.K_LORENTZIAN], [synth.K_SHO] * 4])
def test_mixed_sho_signatures_in_one_batch(engine, kinds):
    """SHOTerm is one complex term for Q >= 1/2 and two real terms below: a batch whose
    walkers straddle Q = 1/2 is split by signature on the device and every row still
    matches the oracle."""
    N, L, B = 600, 2, 200
    t, y, dy = synth.make_lightcurves(N, L, seed=55)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds)
    rng = np.random.default_rng(8)
    theta = synth.draw_thetas(kinds, B, seed=6)
    off = 0
    n_over = np.zeros(B, dtype=int)
    for k in kinds:
        if k == synth.K_SHO:
            theta[:, off + 1] = np.log(0.5) + rng.uniform(-1.0, 1.0, B)      # Q in (0.18, 1.36)
            n_over += theta[:, off + 1] < np.log(0.5)
        off += synth.NPARAMS[k]
    assert len(np.unique(n_over)) >= 2                                     # several signatures present
    lc = (np.arange(B) % L).astype(np.int32)
    out, st = engine.loglike(theta, lc, add_prior=True)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y_mean[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=8)
    assert np.array_equal(st, rst) and np.all(st == 0)
    assert rel(out, ref).max() <= RTOL
