"""A resident set beyond 4 GiB on one MI355X (the sweep's buffer descriptors reach 4 GiB each; every
wave re-bases its own): L x N light curves tiled from a small set, so every result can be checked
bit for bit against the same evaluation in a small context.

    python scripts/bigmem_probe.py [L] [N] [tile]     default 30000 x 10000 (4.8 GB resident), tile 3000
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

L, N, tile = (int(a) for a in (sys.argv[1:4] + [30000, 10000, 3000][len(sys.argv) - 1:]))
kinds = synth.ALT_MODEL
t, y0, dy0 = synth.make_lightcurves(N, tile, seed=1)
reps = (L + tile - 1) // tile
y, dy = np.tile(y0, (reps, 1))[:L], np.tile(dy0, (reps, 1))[:L]
full, free, bounds = synth.model_spec(kinds, y0, per_lc_mean=True)
rng = np.random.default_rng(2)

small = Engine(0)
small.set_lightcurves(t, y0, dy0 + 1e-12, y_offset=y0.mean(axis=1))
small.set_model(kinds, full, free, bounds)
big = Engine(0)
t0 = time.perf_counter()
big.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
upload_s = time.perf_counter() - t0
big.set_model(kinds, full, free, bounds)
out = {"L": L, "N": N, "resident_bytes": int(L) * N * 16 + N * 16, "upload_s": upload_s}
for name, lc in (("grouped", np.repeat(np.arange(L, dtype=np.int32), 8)),
                 ("random order", rng.integers(0, L, 40000).astype(np.int32))):
    theta = synth.draw_thetas(kinds, len(lc), seed=3, percent=0.6)     # SHO on both sides of Q = 1/2: two structures
    for eng in (small, big):
        eng.set_time_parallel(0)
    got, st = big.loglike(theta, lc, add_prior=True)
    ms = big.last_kernel_ms
    want, wst = small.loglike(theta, (lc % tile).astype(np.int32), add_prior=True)
    same = bool(np.array_equal(st, wst) and np.array_equal(got[st == 0], want[wst == 0]))
    out[name] = {"evaluations": len(lc), "kernel_ms": ms, "evals_per_s": len(lc) / ms * 1e3,
                 "bit_identical_to_small_context": same, "ok": int((st == 0).sum())}
    if not same:
        bad = np.flatnonzero((st != wst) | (got != want))
        out[name]["first_mismatch"] = [int(bad[0]), int(lc[bad[0]]), float(got[bad[0]]), float(want[bad[0]]), int(st[bad[0]])]
print(json.dumps(out), flush=True)
