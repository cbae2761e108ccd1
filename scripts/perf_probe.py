"""Quick kernel-throughput probe (development aid; bench.py is the contract)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    eng = Engine(0)
    t, y, dy = synth.make_lightcurves(N, L, seed=20250708)
    eng.set_lightcurves(t, y, dy + 1e-12)
    for name, kinds in (("alt J=6", synth.ALT_MODEL), ("null J=3", synth.NULL_MODEL), ("drw J=1", [synth.K_DRW])):
        full, free, bounds = synth.model_spec(kinds, y)
        eng.set_model(kinds, full, free, bounds)
        B = L * W
        theta = synth.draw_thetas(kinds, B, seed=11)
        lc = np.repeat(np.arange(L), W).astype(np.int32)
        for rep in range(3):
            t0 = time.time()
            out, st = eng.loglike(theta, lc, add_prior=True)
            wall = time.time() - t0
            ms = eng.last_kernel_ms
            print("%-9s N=%d B=%d kernel %.2f ms -> %.3e evals/s (wall %.1f ms) ok=%d" % (
                name, N, B, ms, B / ms * 1e3, wall * 1e3, int((st == 0).sum())), flush=True)

main()
