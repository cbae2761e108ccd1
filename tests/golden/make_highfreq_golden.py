#!/usr/bin/env python3
"""Generates tests/golden/highfreq_golden.json (committed fixture): log-likelihoods of models whose complex terms
advance by 10^3 ... 2*10^6 radians per sampling step -- walkers at the top of the tutorial's prior box, omega ~ e^10
per day over gaps of days -- from the definition-level oracle at 50 digits (oracle/dense.py::dense_loglike_mp: dense
covariance with cos(d tau) evaluated in mpmath on the exact doubles d and tau).  It pins the device's table reduction
of large phase steps (csrc/mtg_math.h: mtg_phase_step, MTG_TRIG_FAST_MAX) against the definition itself rather than
against another double-precision evaluation of the phases.  (Over this fixture's 200 days the C restatement of
celerite's algorithm, cos(d t_n) at the absolute times, still agrees with the 50-digit values to 2.5e-12:
tests/test_oracle.py checks that too.)

Run from the repo root:  python tests/golden/make_highfreq_golden.py   (~1 minute)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import dense  # noqa: E402
from mind_the_gaps_amd import synthetic as K  # noqa: E402


def main():
    rng = np.random.default_rng(20250704 + 77)
    cases = []
    N = 64
    t = np.cumsum(rng.exponential(0.5, N))
    t[20:] += 3.0
    t[40:] += 40.0
    t[55:] += 100.0                                         # max dx ~ 100 days
    y = rng.standard_normal(N)
    dy = rng.uniform(0.2, 0.5, N)
    models = {
        "complex3+drw": ([K.K_COMPLEX3, K.K_DRW], lambda w: [np.log(2.0), np.log(0.3), np.log(w), np.log(1.5), np.log(0.2)]),
        "lorentzian+drw": ([K.K_LORENTZIAN, K.K_DRW], lambda w: [np.log(2.0), np.log(50.0), np.log(w), np.log(1.5), np.log(0.2)]),
        "drw+sho+lor": (K.ALT_MODEL, lambda w: [np.log(1.5), np.log(0.2), np.log(1.0), np.log(3.0), np.log(0.5 * w),
                                                np.log(2.0), np.log(80.0), np.log(w)]),
    }
    for name, (kinds, theta_of) in models.items():
        for omega in (10.0, 1.0e3, 9.0e3, 2.2e4):           # 2.2e4 = e^10: the top of the box (-10, 10)
            theta = np.array(theta_of(omega)) + 0.01 * rng.standard_normal(len(theta_of(omega)))
            co = dense.build_coeffs(kinds, theta)
            ll = dense.dense_loglike_mp(t, y, dy, co, 0, [0.0])
            cases.append({"name": name, "kinds": [int(k) for k in kinds], "omega": omega, "theta": [float(v) for v in theta],
                          "max_phase_step_rad": float(omega * np.max(np.diff(t))), "lnL_mpmath50": float(ll)})
            print(name, omega, cases[-1]["max_phase_step_rad"], ll, flush=True)
    with open(os.path.join(HERE, "highfreq_golden.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_highfreq_golden.py", "oracle": "oracle/dense.py::dense_loglike_mp (50 digits)",
                   "t": [float(v) for v in t], "y": [float(v) for v in y], "dy": [float(v) for v in dy], "mean": 0.0,
                   "cases": cases}, fh, indent=1)
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
