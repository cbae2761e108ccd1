"""Seeded random sweep of the HIP path against the oracle: random sums of terms (every term
kind, J <= 10), sizes, batch shapes, light-curve maps, priors on/off, both kernels (throughput
and time-parallel).  Tolerance 1e-8 relative (BASELINE.json north_star); statuses must agree."""
import os

import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

# MTG_FUZZ_OFFSET=k shifts every seed: a soak run over other cases than the 120 the suite pins
OFFSET = int(os.environ.get("MTG_FUZZ_OFFSET", "0"))

RANK = {synth.K_REAL: 1, synth.K_DRW: 1, synth.K_JITTER: 0}   # everything else is one complex term (2)


def random_model(rng, jmax, ncmax):
    pool = [synth.K_REAL, synth.K_COMPLEX3, synth.K_COMPLEX4, synth.K_SHO, synth.K_MATERN32, synth.K_JITTER,
            synth.K_DRW, synth.K_LORENTZIAN, synth.K_COSINUS, synth.K_BPL]
    while True:
        kinds = list(rng.choice(pool, size=rng.integers(1, 7)))
        nr = sum(1 for k in kinds if RANK.get(k, 2) == 1)
        nc = sum(1 for k in kinds if RANK.get(k, 2) == 2)
        # compiled structures: J <= 10, at most 5 complex terms; the time-parallel kernel J <= 6
        if 0 < nr + nc and nr + 2 * nc <= jmax and nc <= ncmax and kinds.count(synth.K_JITTER) <= 1:
            return [int(k) for k in kinds]


@pytest.mark.parametrize("case", range(120))
def test_random_model_vs_oracle(engine, case):
    _random_model_vs_oracle(engine, case + OFFSET)


# cases a soak (MTG_FUZZ_OFFSET) once failed on, kept for good:
#   112070  ComplexTerm (4 parameters) + over-damped SHO, no prior, time-parallel kernels asked for: a row with b d > a c --
#           outside that term's own prior, its power spectrum negative in places -- came out 1e-7 off through the time-parallel
#           filter pass (round 5).  Batches expanded without the prior now keep the sweep for models with such terms.
@pytest.mark.parametrize("case", [112070])
def test_cases_a_soak_found(engine, case):
    _random_model_vs_oracle(engine, case)


def _random_model_vs_oracle(engine, case):
    rng = np.random.default_rng(9000 + case)
    tp_mode = int(rng.integers(0, 2))
    kinds = random_model(rng, *((6, 3) if tp_mode else (10, 5)))
    linear_mean = bool(rng.integers(0, 3) == 0)
    N = int(rng.choice([1, 2, 5, 37, 256, 257, 800, 2500, 4096, 5001]))
    L = int(rng.integers(1, 5))
    B = int(rng.choice([1, 3, 64, 65, 200, 700]))
    per_lc_t = bool(rng.integers(0, 2)) and L > 1
    add_prior = bool(rng.integers(0, 2))
    t, y, dy = synth.make_lightcurves(N, L, seed=1000 + case)
    if per_lc_t:
        t = np.vstack([synth.make_times(N, rng, offset=float(10 * i)) for i in range(L)])
    if linear_mean:   # fitted LinearModel mean: two more free parameters (slope, intercept), no y_offset
        y = y + 0.01 * (t - t.min())
        full, free, bounds = synth.model_spec(kinds, y, mean_kind=1, fit_mean=True)
        y_mean = None
    else:
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        y_mean = y.mean(axis=1)
    theta = synth.draw_thetas(kinds, B, seed=case, percent=0.25)
    # push a few rows across SHO's Q = 1/2 and outside the box
    off = 0
    for k in kinds:
        if k == synth.K_SHO:
            theta[rng.random(B) < 0.4, off + 1] = np.log(rng.uniform(0.05, 0.45))
        off += synth.NPARAMS[k]
    pushed = rng.random(B) < 0.1
    theta[pushed, 0] = 60.0
    lc = rng.integers(0, L, B).astype(np.int32)
    if linear_mean:
        theta = np.hstack([theta, 0.01 + 0.002 * rng.standard_normal((B, 1)), 100.0 + rng.standard_normal((B, 1))])
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds, mean_kind=1 if linear_mean else 0)
    try:
        engine.set_time_parallel(tp_mode)
        out, st = engine.loglike(theta, lc, add_prior=add_prior)
    finally:
        engine.set_time_parallel(2)
    full_b = theta if linear_mean else np.hstack([theta, y_mean[lc][:, None]])
    okw = dict(bounds=bounds, add_prior=add_prior, nthreads=4, mean_kind=1 if linear_mean else 0)
    if per_lc_t:
        ref, rst = np.empty(B), np.empty(B, dtype=np.int32)
        for l in range(L):
            sel = lc == l
            if sel.any():
                ref[sel], rst[sel] = oracle_c.logprob_batch(t[l], y[l], dy[l], kinds, full_b[sel], **okw)
    else:
        ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, full_b, lc_index=lc, **okw)
    label = "kinds=%s N=%d L=%d B=%d per_lc_t=%s prior=%s tp=%d linear_mean=%s" % (
        kinds, N, L, B, per_lc_t, add_prior, tp_mode, linear_mean)
    if not add_prior:
        # Without the prior the rows pushed to log-amplitude 60 are EVALUATED: an amplitude of e^60
        # against unit noise is a covariance of condition ~1e26, where the sign of a pivot -- the
        # status -- is rounding noise in celerite's recursion and in the kernels alike (soak runs,
        # MTG_FUZZ_OFFSET, show them disagree on a few such rows of undamped-cosine models).  They
        # are in the batch to sit next to the healthy rows, not to be compared.
        out, st, ref, rst = out[~pushed], st[~pushed], ref[~pushed], rst[~pushed]
    assert np.array_equal(st, rst), label
    ok = st == 0
    assert np.all(np.isneginf(out[~ok])), label
    if ok.any():
        err = np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok]))
        assert err <= 1e-8, "%s: %.3g" % (label, err)


@pytest.mark.parametrize("case", range(60))
def test_random_model_pipeline_against_the_one_lane_sweep(engine, case):
    """The same kind of sweep for the two-wave pipelined form of the serial sweep (mtg_set_pipeline(1): whenever the model
    has the kernel): random models of rank 3-6 with one or two complex terms, light-curve lengths on all sides of the
    four-sample hand-over and the twelve-sample trip, shared and per-light-curve sampling, fitted linear mean, prior on /
    off, rejected rows, both SHO regimes -- bit for bit the one-lane sweep's values and statuses."""
    case = case + OFFSET
    rng = np.random.default_rng(77000 + case)
    while True:
        kinds = random_model(rng, 6, 2)
        nr = sum(1 for k in kinds if RANK.get(k, 2) == 1)
        nc = sum(1 for k in kinds if RANK.get(k, 2) == 2)
        # the Lorentzian's null real term is dropped, so the rank of arithmetic is nr + 2 nc as counted here
        if nc >= 1 and 3 <= nr + 2 * nc <= 6 and kinds.count(synth.K_SHO) <= 2:
            break
    linear_mean = bool(rng.integers(0, 3) == 0)
    N = int(rng.choice([64, 65, 66, 67, 75, 76, 77, 100, 333, 1000, 1201, 4099]))
    L = int(rng.integers(1, 6))
    B = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 500, 1300]))
    per_lc_t = bool(rng.integers(0, 2)) and L > 1
    add_prior = bool(rng.integers(0, 2))
    t, y, dy = synth.make_lightcurves(N, L, seed=2000 + case)
    if per_lc_t:
        t = np.vstack([synth.make_times(N, rng, offset=float(10 * i)) for i in range(L)])
    if linear_mean:
        y = y + 0.01 * (t - t.min())
        full, free, bounds = synth.model_spec(kinds, y, mean_kind=1, fit_mean=True)
        y_mean = None
    else:
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        y_mean = y.mean(axis=1)
    theta = synth.draw_thetas(kinds, B, seed=case, percent=0.25)
    off = 0
    for k in kinds:
        if k == synth.K_SHO:
            theta[rng.random(B) < 0.4, off + 1] = np.log(rng.uniform(0.05, 0.45))
        off += synth.NPARAMS[k]
    theta[rng.random(B) < 0.1, 0] = 60.0
    lc = rng.integers(0, L, B).astype(np.int32)
    if linear_mean:
        theta = np.hstack([theta, 0.01 + 0.002 * rng.standard_normal((B, 1)), 100.0 + rng.standard_normal((B, 1))])
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)
    engine.set_model(kinds, full, free, bounds, mean_kind=1 if linear_mean else 0)
    label = "kinds=%s N=%d L=%d B=%d per_lc_t=%s prior=%s linear_mean=%s" % (kinds, N, L, B, per_lc_t, add_prior, linear_mean)
    try:
        engine.set_time_parallel(0)
        engine.set_pipeline(0)
        want, wst = engine.loglike(theta, lc, add_prior=add_prior)
        assert "mtg_pipe" not in engine.last_solver
        engine.set_pipeline(1)
        got, gst = engine.loglike(theta, lc, add_prior=add_prior)
        used = engine.last_solver
    finally:
        engine.set_time_parallel(2)
        engine.set_pipeline(2)
    if "mtg_pipe_kernel" not in used:        # (a model outside the compiled shapes -- three structures of rank > 6, ...: skipped)
        pytest.skip("no pipeline for %s (%s)" % (kinds, used))
    assert np.array_equal(gst, wst), label
    assert np.array_equal(got, want), label                  # NaN-free by construction: -inf where rejected
