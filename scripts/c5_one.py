"""configs[4] likelihood half-steps at one batch size, nothing else: for kernel traces and counter passes.
    python scripts/c5_one.py B [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = Engine(0)
kinds = [synth.K_SHO] * 5
N = 200000
t, y, dy = synth.make_lightcurves(N, 1, seed=20250709)
th = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))] for i in range(5)])
full = np.concatenate([th, [0.0]])
bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
eng.set_model(kinds, full, np.arange(15, dtype=np.int32), bounds)
if os.environ.get('MTG_C5_DIRECT'):
    eng.set_tp_direct(int(os.environ['MTG_C5_DIRECT']))   # 2: never take the filter pass (timing of broken variants)
rng = np.random.default_rng(5)
theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
ms = []
for _ in range(reps):
    out, st = eng.loglike(theta); ms.append(eng.last_kernel_ms)
print("config5 B=%d: %.3f ms (min of %d), kernel %s, ok=%d" % (B, min(ms), reps, eng.last_solver, int((st == 0).sum())))
