"""Loader for tests/golden/loglike_golden.{json,npz} (made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_cache = None


def cases():
    global _cache
    if _cache is None:
        man = json.load(open(os.path.join(HERE, "golden", "loglike_golden.json")))["cases"]
        arr = np.load(os.path.join(HERE, "golden", "loglike_golden.npz"))
        for c in man:
            c["t"], c["y"], c["dy"] = arr[c["id"] + "_t"], arr[c["id"] + "_y"], arr[c["id"] + "_dy"]
            c["full"] = np.array(c["theta"] + c["mean_params"])
        _cache = man
    return _cache


def best_truth(c):
    """Most trustworthy value a case carries: mpmath > OU closed form > dense float64."""
    for k in ("lnL_mpmath50", "lnL_ou_closed_form", "lnL_dense_f64"):
        if not np.isnan(c[k]):
            return c[k]
    raise ValueError(c["id"])
