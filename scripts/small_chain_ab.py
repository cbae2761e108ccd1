"""Iterations per second of small chains with a measurement knob of the library on (default) and off, in two processes, and
that the two chains are the same to the last bit.  The knobs exist in MTG_MEASURE builds only (scripts/build_variant.sh
measure): MTG_SAMPLER_LDS (the sampler's state in LDS), MTG_GLOBAL_TABLES (exp2 / cis tables copied from the context's
resident copy instead of computed per workgroup).
    scripts/build_variant.sh measure && python scripts/small_chain_ab.py [KNOB]      (default MTG_SAMPLER_LDS)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, warnings, hashlib; sys.path.insert(0, %r)
import numpy as np
from mind_the_gaps_amd import terms, synthetic as synth
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
B = dict(log_a=(-10, 50), log_c=(-10, 10))
cases = (("tutorial null N=1e3 W=30", 1000, 30, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B)),
         ("tutorial alt  N=1e3 W=30", 1000, 30, lambda: terms.ComplexTerm(log_a=np.log(100.0), log_c=-5.0, log_d=-0.46, bounds=dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5))) + terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B)),
         ("configs[0] DRW N=1e3 W=32", 1000, 32, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B)),
         ("configs[1] DRW+SHO N=1e4 W=128", 10000, 128, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B) + terms.SHOTerm(np.log(50.0), np.log(3.0), np.log(0.9), bounds=[(-10, 50), (-10, 10), (-10, 10)])),
         ("configs[2]-like N=1e4 W=256", 10000, 256, lambda: terms.RealTerm(np.log(100.0), np.log(0.3), bounds=B) + terms.SHOTerm(np.log(50.0), np.log(3.0), np.log(0.9), bounds=[(-10, 50), (-10, 10), (-10, 10)]) + terms.ComplexTerm(log_a=np.log(100.0), log_c=-5.0, log_d=-0.46, bounds=dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5)))))
for name, N, W, kernel in cases:
    t, y, dy = synth.make_lightcurves(N, 1, seed=3)
    lc = GappyLightcurve(t, y[0], dy[0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(5)
        m = GPModelling(lc, kernel())
        m.derive_posteriors(fit=True, converge=False, max_steps=300, walkers=W, progress=False)      # warm-up
        np.random.seed(5)
        m = GPModelling(lc, kernel())
        t0 = time.perf_counter()
        m.derive_posteriors(fit=False, converge=False, max_steps=4000, walkers=W, progress=False)
        dt = time.perf_counter() - t0
    digest = hashlib.sha256(np.ascontiguousarray(m.sampler.get_chain()).tobytes()).hexdigest()[:16]
    print("%%-32s %%8.0f iterations/s  chain %%s" %% (name, 4000 / dt, digest), flush=True)
''' % ROOT
KNOB = sys.argv[1] if len(sys.argv) > 1 else "MTG_SAMPLER_LDS"
VARIANT = os.path.join(ROOT, "mind_the_gaps_amd", "libmtg_var_measure.so")
if not os.path.exists(VARIANT):
    raise SystemExit("build the measurement variant first: scripts/build_variant.sh measure")
out = {}
for mode in ("1", "0"):
    env = dict(os.environ, MTG_HIP_LIB=VARIANT)
    env[KNOB] = mode
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    out[mode] = [l for l in r.stdout.splitlines() if "iterations/s" in l]
    if r.returncode:
        print(r.stderr[-2000:])
print("%-32s %14s %14s   same chain" % ("", KNOB + "=1", KNOB + "=0"))
for a, b in zip(out["1"], out["0"]):
    name = a[:32]
    ra, rb = float(a[32:].split()[0]), float(b[32:].split()[0])
    print("%-32s %14.0f %14.0f   %s (+%.1f %%)" % (name, ra, rb, a.split()[-1] == b.split()[-1], 100 * (ra / rb - 1)))
