"""Two contexts on the two halves of the GPU's compute units (mtg_create_on_slice), each sweeping 250 x 128 rows from
its own host thread: do the launches overlap?  python scripts/slice_probe.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
N, L, W, reps = 10000, 250, 128, 100
t, y, dy = synth.make_lightcurves(N, L, seed=1)
rng = np.random.default_rng(5)


def setup(kinds, cu_slice):
    eng = Engine(0, cu_slice=cu_slice)
    th = synth.truth(kinds)
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, np.arange(len(th), dtype=np.int32), bounds)
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((L * W, len(th)))
    p0 = theta.reshape(L, W, -1)[:, :W]
    return eng, p0


def run(eng, p0, out, key):
    eng.ensemble_init(np.repeat(p0, 2, axis=1), seed=3)        # 256 walkers per light curve: half-steps of 250 x 128 rows
    t0 = time.perf_counter()
    eng.ensemble_run(reps)
    out[key] = time.perf_counter() - t0


for label, slices in (("whole GPU each, one after the other", (None, None)), ("halves, side by side", ((0, 2), (1, 2))),
                      ("whole GPU each, side by side", (None, None))):
    a, pa = setup(synth.NULL_MODEL, slices[0])
    b, pb = setup(synth.ALT_MODEL, slices[1])
    out = {}
    for eng, p0, k in ((a, pa, "w0"), (b, pb, "w1")):
        run(eng, p0, out, k)                                   # warm-up
    t0 = time.perf_counter()
    if "one after" in label:
        run(a, pa, out, "null"); run(b, pb, out, "alt")
    else:
        ths = [threading.Thread(target=run, args=(a, pa, out, "null")), threading.Thread(target=run, args=(b, pb, out, "alt"))]
        [x.start() for x in ths]; [x.join() for x in ths]
    wall = time.perf_counter() - t0
    print("%-40s null %.3f s  alt %.3f s  wall %.3f s  per iteration %.2f ms  [%s | %s]"
          % (label, out["null"], out["alt"], wall, wall / reps * 1e3, a.last_solver, b.last_solver), flush=True)
    a.close(); b.close()
