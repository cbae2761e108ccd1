"""GPU parity of the time-parallel kernel (one wave per evaluation; four waves per evaluation
for B <= 256, N >= 4096, J <= 5 -- mtg_timeparallel.hip) against the oracle and against the
throughput kernel."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

MODELS = {
    "drw": [synth.K_DRW],
    "drw+sho": [synth.K_DRW, synth.K_SHO],
    "drw+sho+lor": [synth.K_DRW, synth.K_SHO, synth.K_LORENTZIAN],
    "bpl+jitter": [synth.K_BPL, synth.K_JITTER],
    "cosinus+drw": [synth.K_COSINUS, synth.K_DRW],
    "matern32+real": [synth.K_MATERN32, synth.K_REAL],
    "3sho": [synth.K_SHO] * 3,
    "drw+lor+cosinus": [synth.K_DRW, synth.K_LORENTZIAN, synth.K_COSINUS],   # (1, 2) without SHO terms
}
# models with SHO terms run every signature in one launch (mtg_tp_fused_kernel): give them mixed ones
OVERDAMP = {"drw+sho": [3], "drw+sho+lor": [3], "3sho": [4]}


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("N", [1, 3, 70, 1000, 4095, 4096, 4097, 20011])
def test_time_parallel_vs_oracle(engine, name, N):
    kinds = MODELS[name]
    L, B = 2, 24
    t, y, dy = synth.make_lightcurves(N, L, seed=200 + N)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=17)
    for col in OVERDAMP.get(name, []):
        theta[::2, col] = np.log(0.3)                     # some over-damped: mixed signatures
    if name == "3sho":
        theta[::3, 1] = np.log(0.2)
    lc = (np.arange(B) % L).astype(np.int32)
    try:
        engine.set_time_parallel(1)
        out, st = engine.loglike(theta, lc, add_prior=True)
        engine.set_time_parallel(0)
        thr, st_thr = engine.loglike(theta, lc, add_prior=True)
    finally:
        engine.set_time_parallel(2)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y_mean[lc][:, None]]), bounds=bounds,
                                      lc_index=lc, add_prior=True, nthreads=4)
    assert np.array_equal(st, rst) and np.array_equal(st_thr, rst)
    ok = st == 0
    assert ok.sum() > B // 2
    assert np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8
    assert np.max(np.abs(out[ok] - thr[ok]) / np.abs(thr[ok])) <= 1e-9


def test_time_parallel_linear_mean_and_per_lc_times(engine):
    kinds = [synth.K_DRW, synth.K_SHO]
    N, L, B = 900, 3, 12
    rng = np.random.default_rng(4)
    t = np.vstack([synth.make_times(N, rng, offset=50.0 * i) for i in range(L)])
    dy = rng.uniform(0.5, 2.0, (L, N))
    y = 100.0 + 10.0 * rng.standard_normal((L, N)) + 0.02 * (t - t[:, :1])
    full = np.concatenate([synth.truth(kinds), [0.02, 100.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)] * 2])
    engine.set_lightcurves(t, y, dy + 1e-12)
    engine.set_model(kinds, full, np.arange(len(full), dtype=np.int32), bounds, mean_kind=1)
    theta = np.hstack([synth.draw_thetas(kinds, B, seed=9), np.tile([0.02, 100.0], (B, 1))])
    lc = (np.arange(B) % L).astype(np.int32)
    try:
        engine.set_time_parallel(1)
        out, st = engine.loglike(theta, lc, add_prior=False)
    finally:
        engine.set_time_parallel(2)
    ref = np.empty(B)
    for l in range(L):
        sel = lc == l
        ref[sel] = oracle_c.logprob_batch(t[l], y[l], dy[l], kinds, theta[sel], mean_kind=1)[0]
    assert np.all(st == 0) and np.max(np.abs(out - ref) / np.abs(ref)) <= 1e-8


@pytest.mark.parametrize("N", [3, 300, 5000, 8192, 9001])
def test_time_parallel_five_sho(engine, N):
    """The J = 10 structures (BASELINE configs[4]): 64 chunks per evaluation with the elements in
    LDS below 8192 samples, 256 chunks with the elements exchanged through global memory from
    there on; all six signatures in one batch."""
    kinds = [synth.K_SHO] * 5
    L, B = 2, 18
    t, y, dy = synth.make_lightcurves(N, L, seed=900 + N)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=23)
    for b in range(B):                                       # b % 6 over-damped oscillators
        for k in range(b % 6):
            theta[b, 3 * k + 1] = np.log(0.1 + 0.05 * k)
    lc = (np.arange(B) % L).astype(np.int32)
    try:
        engine.set_time_parallel(1)
        out, st = engine.loglike(theta, lc, add_prior=True)
        engine.set_time_parallel(0)
        thr, st_thr = engine.loglike(theta, lc, add_prior=True)
    finally:
        engine.set_time_parallel(2)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y_mean[lc][:, None]]), bounds=bounds,
                                      lc_index=lc, add_prior=True, nthreads=4)
    assert np.array_equal(st, rst) and np.array_equal(st_thr, rst)
    ok = st == 0
    assert ok.sum() >= B // 2
    assert np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8
    assert np.max(np.abs(out[ok] - thr[ok]) / np.abs(thr[ok])) <= 1e-9


@pytest.mark.parametrize("name", ["drw", "drw+sho", "drw+sho+lor", "3sho"])
@pytest.mark.parametrize("N,B", [(300, 5), (5000, 40), (5000, 300)])
def test_scanned_likelihood_against_the_filter_pass(engine, name, N, B):
    """The likelihood carried by the scan (mtg_set_tp_direct 1, the default) against the filter pass over every
    chunk (0) -- the path an evaluation takes when the scanned number is suspect: one wave (scan) and four waves
    (tree, then re-composition and scan), every rank up to 6, mixed SHO signatures; both against the oracle."""
    kinds = MODELS[name]
    t, y, dy = synth.make_lightcurves(N, 1, seed=77 + N)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=5)
    for col in OVERDAMP.get(name, []):
        theta[::2, col] = np.log(0.3)
    try:
        engine.set_time_parallel(1)
        engine.set_tp_direct(1)
        scanned, st_s = engine.loglike(theta, add_prior=False)
        engine.set_tp_direct(0)
        filtered, st_f = engine.loglike(theta, add_prior=False)
    finally:
        engine.set_tp_direct(1)
        engine.set_time_parallel(2)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, np.full((B, 1), y_mean[0])]), bounds=bounds,
                                      add_prior=False, nthreads=4)
    assert np.array_equal(st_s, rst) and np.array_equal(st_f, rst)
    ok = rst == 0
    assert ok.sum() > B // 2
    assert np.max(np.abs(scanned[ok] - filtered[ok]) / np.abs(filtered[ok])) <= 1e-11
    assert np.max(np.abs(scanned[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8
