"""The two-wave pipeline of the serial sweep (csrc/mtg_kernels_pipe.hip) against the one-lane-per-evaluation sweep and the
time-parallel kernels: device time of one batch of L light curves x W rows at N samples, interleaved, min of 5.
python scripts/pipe_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
eng = Engine(0)
rng = np.random.default_rng(5)
W = 128
for name, kinds in (("J=3 null", synth.NULL_MODEL), ("J=5 alt", synth.ALT_MODEL)):
    th = synth.truth(kinds)
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    for L in (64, 96, 125, 250, 256, 300, 500):
        t, y, dy = synth.make_lightcurves(N, L, seed=1)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
        B = L * W
        theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
        lc = np.repeat(np.arange(L, dtype=np.int32), W)
        ms, outs, names = {}, {}, {}
        modes = (("serial", 0, 0), ("pipe", 0, 1)) + ((("tp", 1, 0),) if B <= 16384 else ())
        for rep in range(5):
            for key, tp, pipe in modes:
                eng.set_time_parallel(tp); eng.set_pipeline(pipe)
                out, st = eng.loglike(theta, lc)
                ms[key] = min(ms.get(key, 1e9), eng.last_kernel_ms)
                outs[key] = out; names[key] = eng.last_solver
        same = bool(np.array_equal(outs["serial"], outs["pipe"]))
        print("%s N=%d rows=%6d  " % (name, N, B) + "  ".join("%s %.3f ms" % (k, ms[k]) for k, _, _ in modes)
              + "  pipe==serial bitwise: %s  [%s | %s]" % (same, names["serial"], names["pipe"]), flush=True)
eng.set_time_parallel(2); eng.set_pipeline(2)
