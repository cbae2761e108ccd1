"""The device against 50-digit dense likelihoods at phase steps of 10^3 ... 2 10^6 rad (tests/golden/highfreq_golden.json):
the table reduction of csrc/mtg_math.h serves them all since MTG_TRIG_FAST_MAX went from 1e5 to 1e12."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("tp_mode", [0, 1], ids=["serial_sweep", "time_parallel"])
def test_highfreq_golden(engine, tp_mode):
    fx = json.load(open(os.path.join(HERE, "golden", "highfreq_golden.json")))
    t, y, dy = (np.array(fx[k]) for k in ("t", "y", "dy"))
    engine.set_lightcurves(t, y, dy + 1e-12)
    worst = 0.0
    try:
        engine.set_time_parallel(tp_mode)
        for case in fx["cases"]:
            theta = np.array(case["theta"])
            full = np.concatenate([theta, [fx["mean"]]])
            bounds = np.tile([-np.inf, np.inf], (len(full), 1))
            engine.set_model(case["kinds"], full, np.arange(len(theta), dtype=np.int32), bounds)
            # a whole wave of copies, and one lone row: the table path whatever the wave holds
            out, st = engine.loglike(np.tile(theta, (70, 1)), add_prior=False)
            assert np.all(st == 0), case["name"]
            assert np.all(out == out[0])
            err = abs(out[0] - case["lnL_mpmath50"]) / abs(case["lnL_mpmath50"])
            worst = max(worst, err)
            assert err < 1e-12, (case["name"], case["omega"], err)   # measured: 1.6e-14 (sweep), 2.9e-14 (time-parallel)
    finally:
        engine.set_time_parallel(2)
    print("worst relative error against the 50-digit values: %.2e" % worst)
