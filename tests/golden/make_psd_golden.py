#!/usr/bin/env python3
"""Generates tests/golden/psd_golden.npz: outputs of the REFERENCE's closed-form power spectra
(mind_the_gaps/models/psd_models.py: SHO :7-11, Lorentzian :14-32, BendingPowerlaw :35-46,
Matern32 :63-67) on the cases of the reference's own tests/models_test.py:14-102, which check
celerite's ``Term.get_psd`` of the coefficient builders against exactly these functions.

psd_models.py wraps the functions in astropy's ``custom_model`` (astropy is not installed); the
functions themselves are plain numpy.  They are compiled here from the reference file with the
decorator line dropped -- read from /root/reference at generation time, never copied; only inputs
and outputs are committed.

Run from the repo root (needs /root/reference):  python tests/golden/make_psd_golden.py
"""
import ast
import os
from math import pi, sqrt

import numpy as np
from scipy.special import gamma

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/mind_the_gaps/models/psd_models.py"

tree = ast.parse(open(SRC).read())
ns = {"np": np, "pi": pi, "sqrt": sqrt, "gamma": gamma}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in ("SHO", "Lorentzian", "BendingPowerlaw", "Matern32"):
        node.decorator_list = []
        exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)

omega = np.arange(1, 1000).astype(np.float64)         # models_test.py: frequencies = np.arange(1, 1000)
out = {"omega": omega}
rows = []
# test_DRW (models_test.py:14-29)
rows.append(("drw", (10.0, 5.0), ns["BendingPowerlaw"](omega, S0=10.0, omega0=5.0, Q=0.5)))
# test_SHO (:31-46)
for Q in (10.0, 1.0, 1.0 / np.sqrt(2.0), 0.1):
    rows.append(("sho", (10.0, Q, 5.0), ns["SHO"](omega, S0=10.0, omega0=5.0, Q=Q)))
# test_materns (:48-63)
for rho in (1.0, 10.0, 20.0):
    rows.append(("matern32", (10.0, rho), ns["Matern32"](omega, sigma=10.0, rho=rho)))
# test_Lorentzian (:86-102)
for Q in (10.0, 1.0, 1.0 / np.sqrt(2.0), 0.1):
    for S in (10.0, 5.0, 1.0):
        rows.append(("lorentzian", (S, Q, 5.0), ns["Lorentzian"](omega, S0=S, omega0=5.0, Q=Q)))
out["model"] = np.array([r[0] for r in rows])
out["params"] = np.array([list(r[1]) + [np.nan] * (3 - len(r[1])) for r in rows])
out["psd"] = np.array([r[2] for r in rows])
np.savez_compressed(os.path.join(HERE, "psd_golden.npz"), **out)
print(len(rows), "spectra")
