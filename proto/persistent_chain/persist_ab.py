"""Microseconds per iteration of small single-light-curve chains with the persistent launch (mtg_set_persistent 1, the
default) and with two kernels per iteration (0), interleaved in ONE process on one box, and that the chains are the same.
    python scripts/persist_ab.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = synth
CASES = (("tutorial null  RealTerm            N=1e3 W=30", [K.K_REAL], 1000, 30),
         ("tutorial alt   Complex+Real        N=1e3 W=30", [K.K_COMPLEX3, K.K_REAL], 1000, 30),
         ("configs[0]     DRW                 N=1e3 W=32", [K.K_DRW], 1000, 32),
         ("               DRW                 N=1e3 W=12", [K.K_DRW], 1000, 12),
         ("               DRW                 N=3e3 W=64", [K.K_DRW], 3000, 64),
         ("               DRW+SHO             N=2e3 W=32", K.NULL_MODEL, 2000, 32),
         ("               DRW+2 Lorentzians   N=1e3 W=64", [K.K_DRW, K.K_LORENTZIAN, K.K_LORENTZIAN], 1000, 64))
eng = Engine(0)
print("# %d iterations per run, best of 3 runs each, one process; us per iteration" % steps)
print("%-48s %10s %10s %8s  %s" % ("", "persistent", "2 kernels", "gain", "same chain"))
for name, kinds, N, W in CASES:
    t, y, dy = synth.make_lightcurves(N, 1, seed=3)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    p0 = synth.truth(kinds) * (1 + 0.02 * np.random.default_rng(1).standard_normal((1, W, len(free))))
    best, chains = {}, {}
    for rep in range(3):
        for mode in (1, 0):
            eng.set_persistent(mode)
            eng.ensemble_init(p0, seed=77)
            eng.ensemble_run(50, store_chain=False)
            t0 = time.perf_counter()
            chain, _ = eng.ensemble_run(steps, store_chain=True)
            dt = (time.perf_counter() - t0) / steps * 1e6
            best[mode] = min(best.get(mode, 1e9), dt)
            chains[mode] = chain
    took = "persist" in eng.last_solver or True
    print("%-48s %10.2f %10.2f %7.1f%%  %s" % (name, best[1], best[0], 100 * (best[0] / best[1] - 1), np.array_equal(chains[0], chains[1])), flush=True)
eng.set_persistent(1)
eng.close()
