"""Where the starting fit of a block of light curves goes (ppp.batched_minimize through derive_posteriors_batch, 250 light
curves, alternative model, N = 1e4): launches by batch size, wall and kernel time.  python scripts/fit_probe.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth, terms, ppp
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.gp import get_engine
AMP, OTHER = (-10, 50), (-10, 10)
th = synth.truth(synth.ALT_MODEL)
def alt_kernel():
    return DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER]) + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])
N, L = 10000, 250
t, y, dy = synth.make_lightcurves(N, L, seed=3)
calls = []
orig = ppp.batched_minimize
def traced(fun, x0, lower, upper, **kw):
    def f2(x, lc):
        t0 = time.perf_counter(); r = fun(x, lc); calls.append((len(x), time.perf_counter() - t0, get_engine(0).last_kernel_ms, get_engine(0).last_solver)); return r
    t0 = time.perf_counter(); out = orig(f2, x0, lower, upper, **kw); print("batched_minimize: %.3f s, %d iterations, %d calls" % (time.perf_counter() - t0, out[2], len(calls))); return out
ppp.batched_minimize = traced
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(2):
        calls.clear()
        r = ppp.derive_posteriors_batch(t, y, dy, alt_kernel(), walkers=256, max_steps=2, fit=True, seed=5, store_chain=False, quiet=True, index_base=0)
        print({k: round(v, 3) for k, v in r.seconds.items()})
rows = np.array([c[0] for c in calls]); wall = np.array([c[1] for c in calls]); ker = np.array([c[2] for c in calls])
print("calls by size:", {int(s): (int((rows == s).sum()), round(float(wall[rows == s].sum()), 4), round(float(ker[rows == s].sum()) / 1e3, 4)) for s in sorted(set(rows))}, "(count, wall s, kernel s)")
print("total wall in calls %.3f s, kernel %.3f s; last solver %s" % (wall.sum(), ker.sum() / 1e3, calls[-1][3]))
