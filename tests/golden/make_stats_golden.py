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
# the distributions of stats.py:10-29, 116-146 (kraft_pdf._pdf calls np.math, which numpy 2 dropped: the reference's line
# is run with the alias it was written against)
if not hasattr(np, "math"):
    import math
    np.math = math
dists = []
for _ in range(20):
    mean, std = float(rng.uniform(0.5, 200.0)), float(rng.uniform(0.1, 50.0))
    x = np.sort(rng.uniform(max(1e-3, mean - 2 * std), mean + 3 * std, 7))
    ln, un = ref.create_log_normal(mean, std), ref.create_uniform_distribution(mean, std)
    N, B = int(rng.integers(0, 40)), float(rng.uniform(0.0, 8.0))
    s = np.sort(rng.uniform(0.0, N + 10.0, 7))
    center, sigma = float(rng.uniform(-1, 3)), float(rng.uniform(0.1, 1.5))
    dists.append({"mean": mean, "std": std, "x": x.tolist(), "lognormal_pdf": ln.pdf(x).tolist(), "lognormal_ppf": ln.ppf([0.1, 0.5, 0.9]).tolist(),
                  "lognormal_moments": [float(ln.mean()), float(ln.std())], "uniform_pdf": un.pdf(x).tolist(),
                  "uniform_support": [float(un.ppf(0.0)), float(un.ppf(1.0))], "uniform_moments": [float(un.mean()), float(un.std())],
                  "N": N, "B": B, "s": s.tolist(), "kraft_pdf": [float(ref.kraft_pdf(a=0)._pdf(v, N, B)) for v in s],
                  "center": center, "sigma": sigma, "lognormal_class_pdf": ref.lognormal(a=0)._pdf(x, center, sigma).tolist()})
with open(os.path.join(HERE, "stats_golden.json"), "w") as f:
    json.dump({"source": "mind_the_gaps/stats.py of the reference, imported at generation time", "cases": cases, "distributions": dists}, f, indent=0)
print(len(cases), "cases,", len(dists), "distribution cases")
