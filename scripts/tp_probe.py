import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
for name, kinds in (("alt", synth.ALT_MODEL), ("null", synth.NULL_MODEL), ("drw", [synth.K_DRW])):
    for N in (1000, 10000, 100000):
        t, y, dy = synth.make_lightcurves(N, 1, seed=1)
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1)); eng.set_model(kinds, full, free, bounds)
        for B in (64, 512, 4096):
            theta = synth.draw_thetas(kinds, B, seed=2)
            res = []
            for mode in (0, 1):
                eng.set_time_parallel(mode)
                for _ in range(3):
                    eng.loglike(theta); ms = eng.last_kernel_ms
                res.append(ms)
            print("%-5s N=%-7d B=%-5d throughput-kernel %8.3f ms   time-parallel %8.3f ms   ratio %.1f" % (name, N, B, res[0], res[1], res[0] / res[1]), flush=True)
