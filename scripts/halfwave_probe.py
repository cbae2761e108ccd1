"""Does a wave whose upper (or lower) 32 lanes are idle issue its FP64 instructions faster on gfx950?  The serial sweep
on 500 full waves against 1000 waves with 32 live rows each (the other 32 rejected by the prior, so their lanes leave
the kernel before the sweep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
N = 10000
eng = Engine(0)
rng = np.random.default_rng(5)
kinds = [synth.K_DRW, synth.K_COMPLEX4, synth.K_LORENTZIAN]      # (1, 2), one structure
th = synth.truth(kinds)
full = np.concatenate([th, [0.0]])
bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
t, y, dy = synth.make_lightcurves(N, 1, seed=1)
eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
eng.set_time_parallel(0); eng.set_pipeline(0); eng.set_sort(0)
for label, B, pattern in (("500 full waves", 32000, None), ("1000 waves, upper half idle", 64000, "upper"),
                          ("1000 waves, lower half idle", 64000, "lower"), ("1000 waves, odd lanes idle", 64000, "odd"),
                          ("1000 full waves", 64000, None)):
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    lane = np.arange(B) % 64
    dead = {"upper": lane >= 32, "lower": lane < 32, "odd": lane % 2 == 1, None: np.zeros(B, bool)}[pattern]
    theta[dead, 0] = 60.0
    best = 1e9
    for _ in range(5):
        out, st = eng.loglike(theta, None)
        best = min(best, eng.last_kernel_ms)
    print("%-32s live rows %6d  %.3f ms  [%s]" % (label, int((st == 0).sum()), best, eng.last_solver), flush=True)
