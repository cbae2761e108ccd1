"""End-to-end lock-step sweep: L light curves x W walkers on the device sampler."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth, terms
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.ppp import derive_posteriors_batch

N, L, W, steps = (int(a) for a in (sys.argv[1:5] + [10000, 2000, 256, 20][len(sys.argv) - 1:]))
AMP, OTHER = (-10, 50), (-10, 10)
th = synth.truth(synth.ALT_MODEL)
kernel = (DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])
          + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER]))
t, y, dy = synth.make_lightcurves(N, L, seed=1)
for dev in (True, False):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        res = derive_posteriors_batch(t, y, dy, kernel, walkers=W, max_steps=steps, fit=False, seed=3,
                                      store_chain=False, device_sampler=dev)
        el = time.perf_counter() - t0
    evals = L * W * (steps + 1)
    print("device_sampler=%s: %d evals in %.2f s -> %.3e evals/s end to end; max lnL[0] = %.3f"
          % (dev, evals, el, evals / el, res.max_loglikelihood[0]), flush=True)
