"""configs[4] kernel sweep: half-step time of the rank-10 time-parallel path against the batch size
(N = 2e5, five SHO terms): python scripts/c5_sweep.py [B ...]"""
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
Bs = [int(b) for b in sys.argv[1:]] or [8, 32, 64, 128, 256, 512, 1024]
ref = None
for B in Bs:
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    theta[0] = th
    eng.set_time_parallel(1)
    ms = []
    for _ in range(4):
        out, st = eng.loglike(theta); ms.append(eng.last_kernel_ms)
    ms = min(ms)
    if ref is None:
        eng.set_time_parallel(0)
        ref = eng.loglike(theta[:1])[0][0]
    print("config5 J=10 N=2e5 B=%-5d time-parallel %8.3f ms -> %.3e evals/s, algorithmic HBM %.4f of 8 TB/s; lnL[0]=%.9f (serial sweep %.9f, rel %.1e) ok=%d" % (
        B, ms, B / ms * 1e3, B / ms * 1e3 * (24 * N + 8 * 15 + 12) / 8e12, out[0], ref, abs(out[0] - ref) / abs(ref), int((st == 0).sum())), flush=True)
