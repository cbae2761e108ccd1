"""One small chain on the device sampler (the cases of scripts/small_chain_ab.py), for rocprofv3 --kernel-trace --stats:
    python scripts/chain_case.py {null|alt|c0|c1|c2} [steps]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mind_the_gaps_amd import terms, synthetic as synth
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
B = dict(log_a=(-10, 50), log_c=(-10, 10))
Q = dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5))
S = [(-10, 50), (-10, 10), (-10, 10)]
drw = lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B)
sho = lambda: terms.SHOTerm(np.log(50.0), np.log(3.0), np.log(0.9), bounds=S)
qpo = lambda: terms.ComplexTerm(log_a=np.log(100.0), log_c=-5.0, log_d=-0.46, bounds=Q)
CASES = {"null": (1000, 30, lambda: drw()), "alt": (1000, 30, lambda: qpo() + drw()), "c0": (1000, 32, lambda: drw()),
         "c1": (10000, 128, lambda: drw() + sho()), "c2": (10000, 256, lambda: drw() + sho() + qpo())}
name = sys.argv[1] if len(sys.argv) > 1 else "null"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
N, W, kernel = CASES[name]
t, y, dy = synth.make_lightcurves(N, 1, seed=3)
lc = GappyLightcurve(t, y[0], dy[0])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    np.random.seed(5)
    m = GPModelling(lc, kernel())
    m.derive_posteriors(fit=True, converge=False, max_steps=300, walkers=W, progress=False)
    np.random.seed(5)
    m = GPModelling(lc, kernel())
    t0 = time.perf_counter()
    m.derive_posteriors(fit=False, converge=False, max_steps=steps, walkers=W, progress=False)
    dt = time.perf_counter() - t0
print("%s: %.0f iterations/s (%.2f us per iteration), kernel %s" % (name, steps / dt, dt / steps * 1e6, m.gp._ensure_evaluator(m._y).engine.last_solver if hasattr(m.gp._ensure_evaluator(m._y), "engine") else "?"))
