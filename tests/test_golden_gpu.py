"""GPU parity against the committed golden vectors (dense float64 / mpmath / OU
closed form), through the C-ABI.  Tolerance 1e-8 relative (north_star)."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["throughput-kernel", "time-parallel-kernel", "auto-dispatch"])
def test_hip_vs_golden(engine, mode):
    """Every golden case through the serial sweep (one lane per evaluation), through the
    time-parallel kernel where one is compiled for the structure, and as dispatched by default."""
    worst = 0.0
    try:
        engine.set_time_parallel(mode)
        for c in golden_util.cases():
            nk = len(c["theta"])
            full = c["full"]
            bounds = np.tile([-np.inf, np.inf], (len(full), 1))
            engine.set_lightcurves(c["t"], c["y"], c["dy"] + 1e-12)
            engine.set_model(c["kinds"], full, np.arange(nk, dtype=np.int32), bounds, mean_kind=c["mean_kind"])
            out, st = engine.loglike(np.array([c["theta"]]), add_prior=False)
            assert st[0] == 0, c["id"]
            e = abs(out[0] - golden_util.best_truth(c)) / abs(golden_util.best_truth(c))
            worst = max(worst, e)
            assert e <= 1e-8, (c["id"], c["name"], c["N"], c["t_offset"], e)
    finally:
        engine.set_time_parallel(2)
    print("worst relative error vs golden (mode %d): %.2e" % (mode, worst))


def test_hip_fitted_mean_and_frozen_params(engine):
    """theta may address any subset of the full vector (celerite freeze_parameter /
    fit_mean=True): free the linear-mean parameters and freeze one kernel parameter."""
    c = [k for k in golden_util.cases() if k["name"] == "drw+sho_linear_mean" and k["N"] == 1000][0]
    full = c["full"].copy()
    PF = len(full)
    free = np.array([0, 1, 2, 4, 5, 6], dtype=np.int32)   # log_Q of the SHO frozen
    bounds = np.tile([-np.inf, np.inf], (PF, 1))
    engine.set_lightcurves(c["t"], c["y"], c["dy"] + 1e-12)
    engine.set_model(c["kinds"], full, free, bounds, mean_kind=1)
    out, st = engine.loglike(full[free][None, :])
    assert st[0] == 0 and abs(out[0] - c["lnL_dense_f64"]) / abs(c["lnL_dense_f64"]) < 1e-8
