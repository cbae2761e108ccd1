"""Randomised soak of the rank-10 time-parallel path against the serial sweep of the same library: five-SHO models
with random parameters (every mix of over- and under-damped terms), sizes and batches.
python scripts/tp_big_soak.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
eng = Engine(0)
kinds = [synth.K_SHO] * 5
worst, bad = 0.0, 0
for case in range(cases):
    N = int(rng.choice([1024, 1500, 4097, 12000, 50000]))
    B = int(rng.choice([1, 2, 7, 33, 64, 130, 300]))
    L = int(rng.integers(1, 3))
    t, y, dy = synth.make_lightcurves(N, L, seed=int(rng.integers(1 << 30)))
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    theta = synth.draw_thetas(kinds, B, seed=int(rng.integers(1 << 30)), percent=float(rng.choice([0.05, 0.3, 1.0])))
    over = rng.random((B, 5)) < rng.random()
    theta[:, 1::3] = np.where(over, np.log(rng.uniform(0.05, 0.45, (B, 5))), np.log(rng.uniform(0.55, 30.0, (B, 5))))
    lc = rng.integers(0, L, B).astype(np.int32)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    eng.set_time_parallel(1)
    out, st = eng.loglike(theta, lc, add_prior=False)
    eng.set_time_parallel(0)
    ref, rst = eng.loglike(theta, lc, add_prior=False)
    eng.set_time_parallel(2)
    ok = rst == 0
    same = np.array_equal(st, rst)
    err = float(np.max(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok]))) if ok.any() else 0.0
    worst = max(worst, err)
    if not same or err > 1e-9 or not np.all(np.isneginf(out[~ok])):
        bad += 1
        print("case %d N=%d B=%d L=%d: statuses equal %s, max rel diff %.2e, status counts %s / %s" % (
            case, N, B, L, same, err, np.bincount(st, minlength=4), np.bincount(rst, minlength=4)), flush=True)
print("%d cases, %d bad, worst relative difference %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
