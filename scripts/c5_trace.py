"""configs[4] chain (5 x SHO, N = 2e5, 512 walkers), a few iterations: for a kernel trace (dev aid)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth, terms
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
amp, other = (-10, 50), (-10, 10)
k = None
for i in range(5):
    term = terms.SHOTerm(np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i)), bounds=[amp, other, other])
    k = term if k is None else k + term
t, y, dy = synth.make_lightcurves(200000, 1, seed=20250704 + 2)
g = GPModelling(GappyLightcurve(t, y[0], dy[0]), k)
np.random.seed(1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    g.derive_posteriors(fit=False, max_steps=2, convergence_steps=2, walkers=512, progress=False, device_sampler=True)
    t0 = time.perf_counter()
    g.derive_posteriors(fit=False, max_steps=6, convergence_steps=6, walkers=512, progress=False, device_sampler=True)
    print("6 iterations: %.1f ms each" % ((time.perf_counter() - t0) / 6 * 1e3))
