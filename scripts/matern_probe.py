"""Matern32Term with the tiny eps the reference's tutorial uses (tutorial_model_selection.ipynb:
eps = 1e-8): accuracy of both kernels against the dense definition (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from oracle import celerite as oracle_c, dense

eng = Engine()
kinds = [synth.K_MATERN32]
for eps in (1e-2, 1e-5, 1e-8):
    for N in (64, 1000, 4000):
        t, y, dy = synth.make_lightcurves(N, 1, seed=11)
        full = np.array([np.log(10.0), np.log(10.0), y.mean()])
        theta = np.array([[np.log(10.0), np.log(10.0)], [3.06, 2.30], [1.0, 4.0]])
        bounds = np.tile([-np.inf, np.inf], (3, 1))
        eng.set_lightcurves(t, y, dy + 1e-12)
        eng.set_model(kinds, full, np.arange(2, dtype=np.int32), bounds, extra=[eps])
        fullb = np.hstack([theta, np.full((3, 1), y.mean())])
        ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, fullb, extra=np.array([eps]))
        truth = []
        for th in theta:
            co = dense.build_coeffs(kinds, th, extra=[eps])
            truth.append(dense.dense_loglike_mp(t, y[0], dy[0], co, 0, [y.mean()], dps=60) if N <= 64
                         else dense.dense_loglike(t, y[0], dy[0], co, 0, [y.mean()]))
        truth = np.array(truth)
        row = "eps=%-6g N=%-5d oracle %s" % (eps, N, np.array2string(np.abs(ref - truth) / np.abs(truth), precision=1))
        for mode in (0, 1):
            eng.set_time_parallel(mode)
            out, st = eng.loglike(theta, add_prior=False)
            row += "  %s %s st%s" % (("throughput", "time-parallel")[mode], np.array2string(np.abs(out - truth) / np.abs(truth), precision=1), st)
        eng.set_time_parallel(2)
        print(row, flush=True)
