"""Kernel time of the fused time-parallel launch for the alt model (dev aid; MTG_HIP_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
kinds = synth.ALT_MODEL
for N in (10000,):
    t, y, dy = synth.make_lightcurves(N, 1, seed=1)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1)); eng.set_model(kinds, full, free, bounds)
    for B in (64, 128):
        theta = synth.draw_thetas(kinds, B, seed=2)
        eng.set_time_parallel(2)
        ms = []
        for _ in range(5):
            eng.loglike(theta); ms.append(eng.last_kernel_ms)
        print("N=%d B=%d kernel ms (prepare + solve): %s" % (N, B, " ".join("%.3f" % m for m in ms)), flush=True)
