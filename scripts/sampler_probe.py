"""Raw device-sampler rate (mtg_ensemble_run only) for the single-light-curve configs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

eng = Engine()
for name, kinds, N, W, steps in (("config0 DRW N=1e3 W=32", [synth.K_DRW], 1000, 32, 2000),
                                 ("config1 DRW+SHO N=1e4 W=128", synth.NULL_MODEL, 10000, 128, 500),
                                 ("config2 alt N=1e4 W=256", synth.ALT_MODEL, 10000, 256, 500)):
    t, y, dy = synth.make_lightcurves(N, 1, seed=1)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    p0 = synth.draw_thetas(kinds, W, seed=3)[None]
    for store in (False, True):
        eng.ensemble_init(p0, seed=5)
        eng.ensemble_run(20, store_chain=store)
        t0 = time.perf_counter()
        eng.ensemble_run(steps, store_chain=store)
        el = time.perf_counter() - t0
        acc = eng.ensemble_state()["naccept"].mean() / (steps + 20)
        print("%-30s store_chain=%-5s %8.1f it/s  %7.1f us/it  acceptance %.2f" % (name, store, steps / el, 1e6 * el / steps, acc),
              flush=True)
