"""Launch time of the time-parallel kernel against N at a fixed small batch: separates the per-sample
cost of the composition pass from the fixed cost of the scan.  python scripts/small_nscan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
for name, kinds, B in (("configs[1]", synth.NULL_MODEL, 64), ("configs[2]", synth.ALT_MODEL, 128)):
    th = synth.truth(kinds)
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    for N in (4096, 8192, 16384, 32768, 65536):
        t, y, dy = synth.make_lightcurves(N, 1, seed=1)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
        eng.set_time_parallel(1)
        ms = []
        for _ in range(6):
            out, st = eng.loglike(theta); ms.append(eng.last_kernel_ms)
        print("%s B=%d N=%6d: %.1f us (prepare included), %d samples per lane" % (name, B, N, 1e3 * min(ms), N // 256), flush=True)
