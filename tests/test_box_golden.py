"""Accuracy over the WHOLE prior box (tests/golden/box_golden.json: theta uniform over
amplitudes (-10, 50) / (-10, 10), mpmath truth at 80 digits, N = 50).

Far from the data's scale the covariance is ill conditioned and celerite's own recursion is off
by up to ~1e-3; a fixed 1e-8 tolerance is then meaningless for ANY double-precision
implementation.  What is pinned instead: the error DISTRIBUTION of the HIP kernels against the
truth is no worse than that of celerite's algorithm (oracle/celerite_ref.c) on the same cases.
(The first version of the throughput kernel's large-phase fallback failed exactly this: rotating
the (cos, sin) pair step by step instead of evaluating it at the elapsed time cost four orders of
magnitude on the cosinus + SHO cases.)"""
import json
import os

import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "box_golden.json")) as f:
        g = json.load(f)
    t, y, dy = synth.make_lightcurves(g["N"], 1, seed=g["seed"])
    by_model = {}
    for c in g["cases"]:
        m = by_model.setdefault(c["model"], {"kinds": c["kinds"], "theta": [], "truth": []})
        m["theta"].append(c["theta"]); m["truth"].append(c["lnL_mp80"])
    for m in by_model.values():
        m["theta"], m["truth"] = np.array(m["theta"]), np.array(m["truth"])
    return t, y, dy, g["mean"], by_model


def stats(values, status, truth):
    ok = np.asarray(status) == 0
    rel = np.abs(np.asarray(values)[ok] - truth[ok]) / np.abs(truth[ok])
    return dict(median=float(np.median(rel)), q90=float(np.quantile(rel, 0.9)), over=int((rel > 1e-8).sum()),
                not_ok=int((~ok).sum()))


def oracle_stats(t, y, dy, mean, m):
    B = len(m["theta"])
    ref, rst = oracle_c.logprob_batch(t, y, dy, m["kinds"], np.hstack([m["theta"], np.full((B, 1), mean)]), nthreads=4)
    return stats(ref, rst, m["truth"])


def test_oracle_on_the_whole_box():
    """celerite's algorithm itself: exact to rounding where the problem is well conditioned
    (median), far from 1e-8 in the corners."""
    t, y, dy, mean, models = load()
    assert y.mean() == pytest.approx(mean, abs=0)
    worst_q90 = 0.0
    for name, m in models.items():
        s = oracle_stats(t, y, dy, mean, m)
        assert s["median"] < 1e-13, (name, s)
        assert s["not_ok"] <= 6, (name, s)       # positive definite at 80 digits, not always in float64
        worst_q90 = max(worst_q90, s["q90"])
    assert worst_q90 > 1e-10                      # the corners are really ill conditioned


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["throughput-kernel", "time-parallel-kernel"])
def test_hip_no_worse_than_celerite_on_the_whole_box(engine, mode):
    t, y, dy, mean, models = load()
    try:
        engine.set_time_parallel(mode)
        for name, m in models.items():
            kinds, theta = m["kinds"], m["theta"]
            full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
            engine.set_lightcurves(t, y, dy + 1e-12, y_offset=np.array([mean]))
            engine.set_model(kinds, full, free, bounds)
            out, st = engine.loglike(theta, add_prior=False)
            hip, ref = stats(out, st, m["truth"]), oracle_stats(t, y, dy, mean, m)
            label = "%s: hip %s  celerite %s" % (name, hip, ref)
            assert hip["median"] < 1e-13, label
            assert hip["q90"] <= max(10.0 * ref["q90"], 1e-10), label
            assert hip["over"] <= ref["over"] + 3, label
            assert hip["not_ok"] <= ref["not_ok"] + 3, label
    finally:
        engine.set_time_parallel(2)
