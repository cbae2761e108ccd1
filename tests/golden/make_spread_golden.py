#!/usr/bin/env python3
"""Generates tests/golden/spread_golden.npz: seeded outputs of the REFERENCE's own
``GPModelling.spread_walkers`` (mind_the_gaps/gpmodelling.py:289-350).

The method is pure numpy and never reads ``self``; the module around it imports celerite,
emcee and astropy, none of which is installed.  So only that one FunctionDef is compiled here,
from the reference file where it lies -- read at generation time, never copied; what is
committed are the inputs (seed, centre, box, percent, max_attempts) and the arrays it returned,
plus whether it warned.

Cases: the reference's four test shapes (tests/gpmodelling_test.py:9-114), the tutorial's DRW
and DRW+SHO+Lorentzian centres inside their boxes (no redraw), a centre next to a bound (many
redraws: where the order the generator is consumed in shows), open sides, percent = 0 inside
an impossible box, and max_attempts used up (1, 2, 3, 5).

Run from the repo root (needs /root/reference):  python tests/golden/make_spread_golden.py
"""
import ast
import os
import warnings
from typing import List, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/mind_the_gaps/gpmodelling.py"

ns = {"np": np, "warnings": warnings, "List": List, "Tuple": Tuple, "ArrayLike": object}
for node in ast.walk(ast.parse(open(SRC).read())):
    if isinstance(node, ast.FunctionDef) and node.name == "spread_walkers":
        exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
reference_spread = ns["spread_walkers"]

NONE = np.nan      # an open side is stored as NaN and handed to the reference as None

TEST5 = [5.0, 10.0, 10.0, 5.0, -5.0]                      # gpmodelling_test.py:11-13
BOX5 = [(4.0, 6.0), (8.0, 12.0), (5, 15), (1, 6), (-7, -1)]
OPEN5 = [(None, None), (8.0, 12.0), (5, 15), (1, 6), (-7, -1)]
TIGHT5 = [(v - 0.01, v + 0.01) for v in TEST5]
# centres of the tutorial's fits (docs/notebooks/tutorial_ppp.ipynb: log variance, log bend, mean)
DRW3 = [np.log(100.0), np.log(2 * np.pi / 10), 100.0]
DRW3_BOX = [(-10, 50), (-10, 10), (0, 200)]
ALT6 = [np.log(100.0), np.log(2 * np.pi / 10), np.log(50.0), np.log(100.0), np.log(2 * np.pi / 3.0), 100.0]
ALT6_BOX = [(-10, 50), (-10, 10), (-10, 50), (np.log(1.5), np.log(1000.0)), (-10, 10), (0, 200)]

cases = []      # (name, walkers, centre, box, percent, max_attempts)
cases.append(("test_within_bounds_a", 100, TEST5, BOX5, 0.1, 100))
cases.append(("test_within_bounds_b", 100, TEST5, BOX5, 0.9, 2))
cases.append(("test_infinite_bounds_a", 100, TEST5, OPEN5, 0.1, 50))
cases.append(("test_infinite_bounds_b", 100, TEST5, OPEN5, 0.99, 5))
cases.append(("test_zero_percent", 100, TEST5, OPEN5, 0.0, 50))
cases.append(("test_max_attempts", 100, TEST5, TIGHT5, 0.0, 50))
cases.append(("tutorial_drw", 32, DRW3, DRW3_BOX, 0.1, 20))
cases.append(("tutorial_alt", 256, ALT6, ALT6_BOX, 0.1, 20))
cases.append(("near_bound_95", 32, [9.5, 9.8], [(-10, 10), (-10, 10)], 0.1, 20))
cases.append(("near_bound_open_low", 32, [9.5, -9.8], [(None, 10), (-10, None)], 0.1, 20))
cases.append(("near_bound_negative", 64, [-0.95, 9.9, -9.9], [(-1, 0), (0, 10), (-10, -5)], 0.2, 20))
cases.append(("all_open", 16, [1.0, -2.0, 0.0], [(None, None)] * 3, 0.5, 20))
cases.append(("zero_percent_outside_box", 8, [1.0, 1.0], [(100.0, 200.0), (-50.0, -40.0)], 0.0, 4))
cases.append(("attempts_1", 32, [9.5, 9.8], [(-10, 10), (-10, 10)], 0.1, 1))
cases.append(("attempts_2", 32, [9.5, 9.8], [(-10, 10), (-10, 10)], 0.3, 2))
cases.append(("attempts_3_clamp", 4, [1.0, 1.0], [(100.0, 200.0), (-50.0, -40.0)], 0.1, 3))
cases.append(("attempts_5_wide", 48, TEST5, BOX5, 0.99, 5))
cases.append(("zero_bound_sides", 24, [0.05, -0.05], [(0.0, 1.0), (-1.0, 0.0)], 1.0, 3))

out = {"names": np.array([c[0] for c in cases])}
n_redrawn = 0
for name, walkers, centre, box, percent, attempts in cases:
    for seed in (0, 1, 2):
        np.random.seed(seed)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = reference_spread(None, walkers, np.array(centre, dtype=np.float64), list(box),
                                   percent=percent, max_attempts=attempts)
        after = np.random.random_sample()          # where the reference left numpy's global stream
        key = "%s/%d" % (name, seed)
        out[key + "/p0"] = np.asarray(got, dtype=np.float64)
        out[key + "/warned"] = np.array(len(caught))
        out[key + "/next_uniform"] = np.array(after)
    out[name + "/walkers"] = np.array(walkers)
    out[name + "/centre"] = np.array(centre, dtype=np.float64)
    out[name + "/box"] = np.array([[NONE if lo is None else lo, NONE if hi is None else hi] for lo, hi in box],
                                  dtype=np.float64)
    out[name + "/percent"] = np.array(percent)
    out[name + "/max_attempts"] = np.array(attempts)
np.savez_compressed(os.path.join(HERE, "spread_golden.npz"), **out)
print(len(cases), "cases x 3 seeds")
