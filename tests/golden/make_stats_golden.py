#!/usr/bin/env python3
"""Generates tests/golden/stats_golden.json by running the REFERENCE's own
mind_the_gaps/stats.py (bic :155-167, aic :170-179, aicc :182-195) -- the one module of the
reference that imports in this image (numpy + scipy only).  The reference file is loaded from
/root/reference at generation time and never copied; only inputs and outputs are committed.

Run from the repo root (needs /root/reference):  python tests/golden/make_stats_golden.py
"""
import importlib.util
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_stats", "/root/reference/mind_the_gaps/stats.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20250706)
cases = []
for _ in range(60):
    lnl = float(-rng.uniform(1.0, 1e5))
    n = int(rng.integers(20, 200000))
    k = int(rng.integers(1, 16))
    cases.append({"lnL": lnl, "n": n, "k": k, "bic": float(ref.bic(lnl, n, k)), "aic": float(ref.aic(lnl, k)),
                  "aicc": float(ref.aicc(lnl, n, k))})
with open(os.path.join(HERE, "stats_golden.json"), "w") as f:
    json.dump({"source": "mind_the_gaps/stats.py of the reference, imported at generation time", "cases": cases}, f, indent=0)
print(len(cases), "cases")
