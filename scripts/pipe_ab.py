"""One line per model: serial sweep / pipeline device time at 250 light curves x 128 rows, N = 1e4 (min of 7, interleaved).
MTG_HIP_LIB=... python scripts/pipe_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
N, L, W = 10000, 250, 128
eng = Engine(0)
rng = np.random.default_rng(5)
line = [os.path.basename(os.environ.get("MTG_HIP_LIB", "libmtg_hip.so"))]
for name, kinds in (("J=3", synth.NULL_MODEL), ("J=5", synth.ALT_MODEL)):
    th = synth.truth(kinds)
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    t, y, dy = synth.make_lightcurves(N, L, seed=1)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((L * W, len(th)))
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    ms, outs = {}, {}
    eng.set_time_parallel(0)
    for rep in range(7):
        for pipe in (0, 1):
            eng.set_pipeline(pipe)
            outs[pipe], _ = eng.loglike(theta, lc)
            ms[pipe] = min(ms.get(pipe, 1e9), eng.last_kernel_ms)
    line.append("%s serial %.3f pipe %.3f ms same=%s" % (name, ms[0], ms[1], bool(np.array_equal(outs[0], outs[1]))))
print("  ".join(line), flush=True)
