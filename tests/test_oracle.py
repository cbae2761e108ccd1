"""CPU tests: the oracle itself is pinned before anything is compared with it.

* oracle/celerite_ref.c (celerite's recurrences in C) against the committed
  golden vectors (dense float64 / mpmath-50 / OU closed form),
* the coefficient builders against the closed-form PSD known answers of the
  reference's own tests (/root/reference/tests/models_test.py:14-102; closed
  forms from mind_the_gaps/models/psd_models.py:7-85, restated below),
* prior and status conventions.
"""
import os

import numpy as np
import pytest

import golden_util
from oracle import celerite as oc
from oracle import dense

K = dense


def rel(a, b):
    return abs(a - b) / abs(b)


def test_c_oracle_vs_golden():
    worst0, worst_off = 0.0, 0.0
    for c in golden_util.cases():
        out, st = oc.logprob_batch(c["t"], c["y"], c["dy"], c["kinds"], c["full"], mean_kind=c["mean_kind"])
        assert st[0] == 0
        e = rel(out[0], golden_util.best_truth(c))
        if c["t_offset"] == 0.0:
            worst0 = max(worst0, e)
        else:
            worst_off = max(worst_off, e)
    # celerite evaluates cos/sin(d * t_n) at ABSOLUTE times: with t ~ 5e8 s the phase
    # d*t carries ~1e-7 rad of rounding, so celerite's own lnL is only good to ~1e-7
    # there (DESIGN.md "Large time offsets").  Without an offset it is ~1e-13.
    assert worst0 < 1e-11
    assert worst_off < 1e-5


def test_dense_f64_vs_mpmath():
    n = 0
    for c in golden_util.cases():
        if not np.isnan(c["lnL_mpmath50"]):
            assert rel(c["lnL_dense_f64"], c["lnL_mpmath50"]) < 1e-12
            n += 1
    assert n >= 10


def test_ou_closed_form_agrees():
    n = 0
    for c in golden_util.cases():
        if not np.isnan(c["lnL_ou_closed_form"]):
            assert rel(c["lnL_dense_f64"], c["lnL_ou_closed_form"]) < 1e-11
            n += 1
    assert n >= 10


# ---- PSD known answers (reference tests/models_test.py) --------------------
def reference_psd(model, *params):
    """Output of the reference's own closed-form PSD (mind_the_gaps/models/psd_models.py) on
    omega = 1..999 for one of the cases of its tests/models_test.py -- committed as
    tests/golden/psd_golden.npz by tests/golden/make_psd_golden.py, which runs the reference's
    functions themselves."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psd_golden.npz"))
    want = np.array(list(params) + [np.nan] * (3 - len(params)))
    for m, p, psd in zip(g["model"], g["params"], g["psd"]):
        if m == model and np.allclose(p, want, rtol=1e-15, atol=0, equal_nan=True):
            return g["omega"], psd
    raise KeyError((model, params))


def test_psd_drw():                 # models_test.py:14-29
    co = dense.build_coeffs([K.K_DRW], [np.log(10.0), np.log(5.0)])
    w, ref = reference_psd("drw", 10.0, 5.0)
    np.testing.assert_array_almost_equal(ref, dense.psd(co, w))


@pytest.mark.parametrize("Q", [10, 1, 1 / np.sqrt(2), 0.1])
def test_psd_sho(Q):                # models_test.py:31-46
    co = dense.build_coeffs([K.K_SHO], [np.log(10.0), np.log(Q), np.log(5.0)])
    w, ref = reference_psd("sho", 10.0, Q, 5.0)
    np.testing.assert_array_almost_equal(ref, dense.psd(co, w))


@pytest.mark.parametrize("rho", [1, 10, 20])
def test_psd_matern32(rho):         # models_test.py:48-63
    co = dense.build_coeffs([K.K_MATERN32], [np.log(10.0), np.log(rho)], extra=[1e-15])
    w, ref = reference_psd("matern32", 10.0, rho)
    np.testing.assert_array_almost_equal(ref, dense.psd(co, w))


@pytest.mark.parametrize("Q", [10, 1, 1 / np.sqrt(2), 0.1])
@pytest.mark.parametrize("S0", [10, 5, 1])
def test_psd_lorentzian(Q, S0):     # models_test.py:86-102
    co = dense.build_coeffs([K.K_LORENTZIAN], [np.log(S0), np.log(Q), np.log(5.0)])
    w, ref = reference_psd("lorentzian", S0, Q, 5.0)
    np.testing.assert_array_almost_equal(ref, dense.psd(co, w))


def test_c_builders_match_python_builders():
    """celerite_ref.c expands theta exactly like oracle/dense.py (same lnL to rounding)."""
    rng = np.random.default_rng(5)
    t = np.cumsum(0.05 + rng.exponential(1.0, 50))
    y, dy = rng.standard_normal(50), rng.uniform(0.5, 2, 50)
    for kinds in ([K.K_REAL], [K.K_COMPLEX3], [K.K_COMPLEX4], [K.K_SHO], [K.K_MATERN32],
                  [K.K_DRW, K.K_JITTER], [K.K_LORENTZIAN], [K.K_COSINUS, K.K_DRW], [K.K_BPL]):
        nk = dense.n_kernel_params(kinds)
        th = rng.uniform(-1.5, 1.0, nk)
        if kinds == [K.K_BPL]:
            th[0] = th[1] + 0.5
        co = dense.build_coeffs(kinds, th)
        want = dense.dense_loglike(t, y, dy, co, 0, [0.1])
        got, st = oc.logprob_batch(t, y, dy, kinds, np.append(th, 0.1))
        assert st[0] == 0 and rel(got[0], want) < 1e-11, kinds


def test_prior_and_status():
    rng = np.random.default_rng(6)
    t = np.cumsum(0.05 + rng.exponential(1.0, 40))
    y, dy = rng.standard_normal(40), rng.uniform(0.5, 2, 40)
    kinds = [K.K_DRW, K.K_BPL]
    bounds = np.array([(-10, 50), (-10, 10), (-10, 50), (-10, 10), (-10, 10), (-np.inf, np.inf)], float)
    ok = np.array([1.0, -1.0, 2.0, 1.0, -0.5, 0.0])
    out_of_box = ok.copy(); out_of_box[1] = 10.5
    bpl_violation = ok.copy(); bpl_violation[2], bpl_violation[3] = 0.5, 1.0   # log_S0 < log_Q
    edge = ok.copy(); edge[1] = 10.0                                        # bounds are inclusive
    out, st = oc.logprob_batch(t, y, dy, kinds, np.vstack([ok, out_of_box, bpl_violation, edge]),
                               bounds=bounds, add_prior=True)
    assert list(st) == [0, 1, 1, 0]
    assert np.isfinite(out[0]) and np.isneginf(out[1]) and np.isneginf(out[2]) and np.isfinite(out[3])
    assert dense.log_prior(kinds, ok, bounds) == 0.0
    assert dense.log_prior(kinds, out_of_box, bounds) == -np.inf
    assert dense.log_prior(kinds, bpl_violation, bounds) == -np.inf
    # without the prior the same vectors are evaluated (gpmodelling.py:168-169)
    out2, st2 = oc.logprob_batch(t, y, dy, kinds, np.vstack([ok, out_of_box]), bounds=bounds, add_prior=False)
    assert list(st2) == [0, 0] and out2[0] == out[0]


def test_not_positive_definite_status():
    """A negative-amplitude real term makes K indefinite: celerite raises LinAlgError (status 2)."""
    t = np.arange(20.0)
    y, dy = np.zeros(20), np.full(20, 1e-3)
    # ComplexTerm with huge b: a cos + b sin goes strongly negative -> not a valid kernel
    full = np.array([np.log(1.0), np.log(50.0), np.log(0.01), np.log(1.0), 0.0])
    out, st = oc.logprob_batch(t, y, dy, [K.K_COMPLEX4], full)
    assert st[0] == 2 and np.isneginf(out[0])
    assert dense.dense_loglike(t, y, dy, dense.build_coeffs([K.K_COMPLEX4], full[:4]), 0, [0.0]) == -np.inf
