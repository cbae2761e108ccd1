import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
kinds = [synth.K_SHO] * 5
N = 200000
t, y, dy = synth.make_lightcurves(N, 1, seed=20250709)
th = synth.truth(kinds)
for i in range(5):
    th[3 * i:3 * i + 3] = [np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
full = np.concatenate([th, [0.0]])
bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
eng.set_model(kinds, full, np.arange(15, dtype=np.int32), bounds)
rng = np.random.default_rng(5)
for B in (64, 256):
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    eng.set_time_parallel(1)
    ms = []
    for _ in range(3):
        out, st = eng.loglike(theta); ms.append(eng.last_kernel_ms)
    print("B=%d kernel ms %s" % (B, " ".join("%.2f" % m for m in ms)), flush=True)
