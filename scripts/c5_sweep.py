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
    eng.set_tp_direct(0)
    out_f, st_f = eng.loglike(theta); ms_f = eng.last_kernel_ms
    out_f, st_f = eng.loglike(theta); ms_f = min(ms_f, eng.last_kernel_ms)
    eng.set_tp_direct(6)
    mag = eng.loglike(theta)[0]
    eng.set_tp_direct(2)
    out_d, st_d = eng.loglike(theta)
    print("   direct, no fallback: max rel diff to the filter pass %.2e; largest terms / result %.2e" % (
        np.nanmax(np.abs(out_d - out_f) / np.abs(out_f)), np.max(mag / np.abs(out_f))))
    eng.set_tp_direct(1)
    ms = []
    for _ in range(4):
        out, st = eng.loglike(theta); ms.append(eng.last_kernel_ms)
    ms = min(ms)
    print("   direct vs filter pass: max rel diff %.2e, statuses equal %s; filter-pass mode %.3f ms" % (
        np.max(np.abs(out - out_f) / np.abs(out_f)), np.array_equal(st, st_f), ms_f))
    if ref is None:
        eng.set_time_parallel(0)
        ref = eng.loglike(theta[:1])[0][0]
    print("config5 J=10 N=2e5 B=%-5d time-parallel %8.3f ms -> %.3e evals/s, algorithmic HBM %.4f of 8 TB/s; lnL[0]=%.9f (serial sweep %.9f, rel %.1e) ok=%d" % (
        B, ms, B / ms * 1e3, B / ms * 1e3 * (24 * N + 8 * 15 + 12) / 8e12, out[0], ref, abs(out[0] - ref) / abs(ref), int((st == 0).sum())), flush=True)
