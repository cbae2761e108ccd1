"""Where the time-parallel kernels stop paying against the serial sweep: device time of one batch, both ways.
python scripts/crossover_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
for name, kinds in (("J=3", synth.NULL_MODEL), ("J=5", synth.ALT_MODEL)):
    th = synth.truth(kinds)
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    for N in (1000, 10000, 100000):
        t, y, dy = synth.make_lightcurves(N, 1, seed=1)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
        row = []
        for B in (256, 512, 1024, 2048, 4096, 8192, 16384):
            theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
            ms = {}
            for mode in (0, 1):
                eng.set_time_parallel(mode)
                best = 1e9
                for _ in range(3):
                    eng.loglike(theta); best = min(best, eng.last_kernel_ms)
                ms[mode] = best
            row.append("B=%d: %.2f / %.2f" % (B, ms[0], ms[1]))
        eng.set_time_parallel(2)
        print("%s N=%d  serial sweep / time-parallel [ms]:  %s" % (name, N, "   ".join(row)), flush=True)
