import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for B in (32, 64, 256, 4096, 65536):
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    for mode in (0, 1) if B <= 4096 else (0,):
        eng.set_time_parallel(mode)
        for _ in range(2):
            out, st = eng.loglike(theta); ms = eng.last_kernel_ms
        print("config5 J=10 N=2e5 B=%-6d %-13s kernel %9.2f ms -> %.3e evals/s, HBM-equivalent %.1f GB/s  lnL[0]=%.6f" % (
            B, "time-parallel" if mode else "throughput", ms, B / ms * 1e3, B / ms * 1e3 * (24 * N + 8 * 15 + 12) / 1e9, out[0]), flush=True)
