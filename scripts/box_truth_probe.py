"""Who is right where the HIP kernels and the C port of celerite's recursion disagree over the whole
prior box?  mpmath (80 digits, dense covariance) decides (development aid; N = 50)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from oracle import celerite as oracle_c, dense

eng = Engine()
rng = np.random.default_rng(1)
N, B = 50, 4000
for name, kinds in (("null", synth.NULL_MODEL), ("alt", synth.ALT_MODEL), ("bpl+matern", [synth.K_BPL, synth.K_MATERN32]),
                    ("cos+jit+sho", [synth.K_COSINUS, synth.K_JITTER, synth.K_SHO])):
    t, y, dy = synth.make_lightcurves(N, 1, seed=5)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    theta = rng.uniform(bounds[free, 0], bounds[free, 1], (B, len(free)))
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, np.full((B, 1), y.mean())]), bounds=bounds,
                                      add_prior=True, nthreads=8)
    res = {}
    for mode in (0, 1):
        eng.set_time_parallel(mode)
        res[mode] = eng.loglike(theta, add_prior=True)
    eng.set_time_parallel(2)
    with np.errstate(all="ignore"):
        disagree = np.zeros(B, dtype=bool)
        for mode in (0, 1):
            out, st = res[mode]
            disagree |= (st != rst) | ((st == 0) & (rst == 0) & (np.abs(out - ref) > 1e-8 * np.abs(ref)))
    disagree &= rst != 1
    idx = np.nonzero(disagree)[0][:14]
    print("== %s: %d of %d evaluations disagree somewhere; mpmath on %d of them" % (name, disagree.sum(), B, len(idx)), flush=True)
    stats = {"oracle": [], "thr": [], "tp": []}
    for b in idx:
        co = dense.build_coeffs(kinds, theta[b])
        try:
            truth = dense.dense_loglike_mp(t, y[0], dy[0], co, 0, [y.mean()], dps=80)
        except Exception as ex:  # not positive definite even at 80 digits
            truth = float("nan")
        def err(v, s):
            if s != 0:
                return "st%d" % s
            return "%.1e" % (abs(v - truth) / abs(truth))
        print("  truth %-22.14g oracle %-8s throughput %-8s time-parallel %-8s  theta %s"
              % (truth, err(ref[b], rst[b]), err(res[0][0][b], res[0][1][b]), err(res[1][0][b], res[1][1][b]),
                 np.round(theta[b], 2)), flush=True)
