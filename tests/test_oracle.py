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
from mind_the_gaps_amd import synthetic as synth

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


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """ASan + UBSan build of the C restatement on the CPU (never on the GPU box's
    card): every term kind, prior on, several light curves, threads; clean exit and the same numbers as
    the plain library."""
    import shutil
    import struct
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    here = os.path.dirname(os.path.abspath(oc.__file__))
    exe = str(tmp_path / "oracle_sanitized")
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-fno-omit-frame-pointer", "-fopenmp", os.path.join(here, "celerite_ref.c"),
                            os.path.join(here, "sanitize_driver.c"), "-lm", "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    kinds = [synth.K_DRW, synth.K_SHO, synth.K_LORENTZIAN, synth.K_MATERN32, synth.K_JITTER, synth.K_BPL]
    N, L, B = 257, 3, 24
    t, y, dy = synth.make_lightcurves(N, L, seed=3)
    theta = synth.draw_thetas(kinds, B, seed=4, percent=0.2)
    theta[::5, 4] = np.log(0.2)                                   # over-damped SHO rows
    theta[3, 0] = 99.0                                            # outside the box
    full = np.hstack([theta, y.mean(axis=1)[np.arange(B) % L][:, None]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    lc = (np.arange(B) % L).astype(np.int32)
    extra = np.full(len(kinds), 0.01)
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(struct.pack("8q", N, L, B, len(kinds), full.shape[1], 0, 1, 3))
        for arr, dt in ((kinds, np.int32), (extra, np.float64), (t, np.float64), (y, np.float64), (dy, np.float64),
                        (bounds, np.float64), (full, np.float64), (lc, np.int32)):
            f.write(np.ascontiguousarray(arr, dtype=dt).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, str(case)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and run.stderr.strip() == "", run.stderr[-2000:]
    rows = np.array([line.split() for line in run.stdout.split("\n") if line], dtype=float)
    assert rows.shape == (2 * B, 2)
    ref, rst = oc.logprob_batch(t, y, dy, kinds, full, bounds=bounds, lc_index=lc, add_prior=True, nthreads=3)
    for block in (rows[:B], rows[B:]):                            # two-sweep, then fused
        assert np.array_equal(block[:, 1].astype(int), rst)
        ok = rst == 0
        assert np.all(np.isneginf(block[~ok, 0]))
        assert np.max(np.abs(block[ok, 0] - ref[ok]) / np.abs(ref[ok])) < 1e-12


def test_fused_sweep_equals_the_two_sweep_restatement():
    """The one-sweep variant bench.py times as the CPU baseline is the same arithmetic in another order of
    loops: identical results."""
    kinds = synth.ALT_MODEL
    t, y, dy = synth.make_lightcurves(3000, 2, seed=8)
    th = synth.draw_thetas(kinds, 16, seed=9)
    full = np.hstack([th, np.full((16, 1), y.mean())])
    a, sa = oc.logprob_batch(t, y, dy, kinds, full, nthreads=4)
    b, sb = oc.logprob_batch(t, y, dy, kinds, full, nthreads=4, fused=True)
    assert np.array_equal(sa, sb) and np.max(np.abs(a - b) / np.abs(a)) < 1e-13


def test_celerite_restatement_on_the_high_frequency_fixture():
    """tests/golden/highfreq_golden.json (50-digit dense values, phase steps up to 2e6 rad): over the fixture's 200 days
    celerite's algorithm with cos(d t_n) at the absolute times is still good to a few 1e-12."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "highfreq_golden.json")))
    t, y, dy = (np.array(fx[k]) for k in ("t", "y", "dy"))
    for case in fx["cases"]:
        full = np.concatenate([case["theta"], [fx["mean"]]])[None, :]
        out, st = oc.logprob_batch(t, y, dy, case["kinds"], full, bounds=None, add_prior=False)
        assert st[0] == 0
        assert abs(out[0] - case["lnL_mpmath50"]) / abs(case["lnL_mpmath50"]) < 1e-10, (case["name"], case["omega"])
