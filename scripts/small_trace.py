"""configs[1] / configs[2] chains through the device sampler, few iterations: for a kernel trace."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth, terms
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
AMP, OTHER = (-10, 50), (-10, 10)
th = synth.truth(synth.ALT_MODEL)
null = lambda: DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])
alt = lambda: null() + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])
which = sys.argv[1] if len(sys.argv) > 1 else "2"
make, W = (null, 128) if which == "1" else (alt, 256)
t, y, dy = synth.make_lightcurves(10000, 1, seed=1)
g = GPModelling(GappyLightcurve(t, y[0], dy[0]), make())
np.random.seed(1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    g.derive_posteriors(fit=False, max_steps=10, convergence_steps=10, walkers=W, progress=False, device_sampler=True)
    t0 = time.perf_counter()
    g.derive_posteriors(fit=False, max_steps=200, convergence_steps=200, walkers=W, progress=False, device_sampler=True)
    el = time.perf_counter() - t0
print("configs[%s]: %.1f it/s, %.0f evals/s" % (which, 200 / el, 200 * W / el))
