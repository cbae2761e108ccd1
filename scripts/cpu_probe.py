import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp", oc.max_threads())
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
kinds = synth.ALT_MODEL
N = 10000
t, y, dy = synth.make_lightcurves(N, 4, seed=1)
for th in (1, 8, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    B = th * 16
    theta = synth.draw_thetas(kinds, B, seed=2)
    full = np.hstack([theta, np.full((B, 1), 100.0)])
    t0 = time.perf_counter(); oc.logprob_batch(t, y, dy, kinds, full, nthreads=th); el = time.perf_counter() - t0
    print("threads %d: %.0f evals/s (%.0f per thread)" % (th, B / el, B / el / th), flush=True)
