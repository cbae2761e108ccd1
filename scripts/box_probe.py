"""Exploration: theta uniform over the WHOLE prior box (not a 10 % cloud): statuses and values of
both kernels against the oracle (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from oracle import celerite as oracle_c

eng = Engine()
rng = np.random.default_rng(1)
for name, kinds in (("alt", synth.ALT_MODEL), ("null", synth.NULL_MODEL), ("bpl+matern", [synth.K_BPL, synth.K_MATERN32]),
                    ("complex4+real", [synth.K_COMPLEX4, synth.K_REAL]), ("cos+jit+sho", [synth.K_COSINUS, synth.K_JITTER, synth.K_SHO])):
    for N in (50, 700):
        t, y, dy = synth.make_lightcurves(N, 1, seed=5)
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        B = 4000
        lo, hi = bounds[free, 0], bounds[free, 1]
        theta = rng.uniform(lo, hi, (B, len(free)))
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, free, bounds)
        ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, np.full((B, 1), y.mean())]), bounds=bounds,
                                          add_prior=True, nthreads=8)
        for mode in (0, 1):
            eng.set_time_parallel(mode)
            out, st = eng.loglike(theta, add_prior=True)
            same = st == rst
            ok = (st == 0) & (rst == 0)
            with np.errstate(all="ignore"):
                rel = np.abs(out[ok] - ref[ok]) / np.abs(ref[ok])
            bad = np.nonzero(ok)[0][rel > 1e-8]
            print("%-14s N=%-4d mode=%d status agree %5d/%d (oracle: ok %d prior %d notpd %d nonfinite %d)  worst rel %.2e  >1e-8: %d"
                  % (name, N, mode, same.sum(), B, (rst == 0).sum(), (rst == 1).sum(), (rst == 2).sum(), (rst == 3).sum(),
                     rel.max() if rel.size else 0.0, bad.size), flush=True)
            for b in list(np.nonzero(~same)[0][:3]) + list(bad[:3]):
                print("     theta", np.round(theta[b], 3), "hip", out[b], st[b], "oracle", ref[b], rst[b])
        eng.set_time_parallel(2)
