#!/usr/bin/env python3
"""Generates tests/golden/box_golden.json: theta drawn uniformly over the WHOLE prior box of
the tutorials (amplitudes (-10, 50), everything else (-10, 10)) -- not a 10 % cloud around a
truth -- with the log-likelihood of the dense covariance at 80 digits (mpmath).

Far from the data's scale the covariance is ill conditioned (signal variance up to e^50 against
unit noise): celerite's recursion itself is then off by up to 1e-3, so these cases pin the
ACCURACY of the kernels against the truth relative to the accuracy of celerite's own algorithm
(oracle/celerite_ref.c), not a fixed tolerance.

Run from the repo root:  python tests/golden/make_box_golden.py   (~2 minutes)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dense  # noqa: E402
from mind_the_gaps_amd import synthetic as synth  # noqa: E402

K = synth
MODELS = {
    "null": K.NULL_MODEL,
    "alt": K.ALT_MODEL,
    "bpl+matern32": [K.K_BPL, K.K_MATERN32],
    "cosinus+jitter+sho": [K.K_COSINUS, K.K_JITTER, K.K_SHO],
    "complex4+real": [K.K_COMPLEX4, K.K_REAL],
}
N, PER_MODEL, SEED = 50, 40, 20250705


def main():
    rng = np.random.default_rng(SEED)
    t, y, dy = synth.make_lightcurves(N, 1, seed=SEED)
    cases = []
    for name, kinds in MODELS.items():
        bounds = synth.bounds_for(kinds)
        kept = 0
        while kept < PER_MODEL:
            th = rng.uniform(bounds[:, 0], bounds[:, 1])
            if name == "bpl+matern32" and th[0] < th[1]:
                continue                                   # BendingPowerlaw prior
            if name == "complex4+real" and th[0] + th[2] < th[1] + th[3]:
                continue                                   # ComplexTerm prior (positive definite)
            try:
                truth = dense.dense_loglike_mp(t, y[0], dy[0], dense.build_coeffs(kinds, th), 0, [float(y.mean())], dps=80)
            except Exception:
                continue                                   # not positive definite even at 80 digits
            cases.append({"model": name, "kinds": [int(k) for k in kinds], "theta": [float(v) for v in th],
                          "lnL_mp80": truth})
            kept += 1
        print(name, kept, flush=True)
    out = {"N": N, "seed": SEED, "mean": float(y.mean()), "cases": cases,
           "note": "light curve = synthetic.make_lightcurves(N, 1, seed); lnL of dense K at 80 digits"}
    with open(os.path.join(HERE, "box_golden.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
