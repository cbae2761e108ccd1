"""What would TIME sharding buy configs[4] (N = 2e5, five SHO terms, 256 rows per half-step) at 8 GPUs?  One GPU's part of
a time-sharded half-step is the composition + up-sweep of ALL 256 rows over N/8 samples -- which the shipped kernels run as
is on a light curve of N/8 samples -- followed by an all-gather of one filtering element per row and rank (235 doubles at
rank 10: 8 x 256 x 235 x 8 B = 3.85 MB per half-step) and three levels of combinations.  Measured here: that one-GPU part,
beside the walker-sharded share (32 rows x N samples) and the whole half-step (256 rows x N).
    python scripts/c5_time_shard_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

eng = Engine(0)
kinds = [synth.K_SHO] * 5
th = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))] for i in range(5)])
full = np.concatenate([th, [0.0]])
bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
rng = np.random.default_rng(5)


def half_step(N, B, reps=8):
    t, y, dy = synth.make_lightcurves(N, 1, seed=20250709)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, np.arange(15, dtype=np.int32), bounds)
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    ms = []
    for _ in range(reps):
        out, st = eng.loglike(theta)
        ms.append(eng.last_kernel_ms)
    return min(ms), eng.last_solver, int((st == 0).sum())


whole, k0, ok0 = half_step(200000, 256)
print("whole half-step            N = 200000, 256 rows: %.3f ms  %s (ok %d)" % (whole, k0, ok0))
for world in (2, 4, 8):
    walker, k1, _ = half_step(200000, 256 // world)
    shard, k2, _ = half_step(200000 // world, 256)
    elem_bytes = world * 256 * 235 * 8
    print("%d GPUs: walker shard (%3d rows x N) %.3f ms = %.2fx | time shard (256 rows x N/%d) %.3f ms = %.2fx BEFORE its exchange of %.2f MB "
          "(walker sharding exchanges %d B)  [%s | %s]" % (world, 256 // world, walker, whole / walker, world, shard, whole / shard,
                                                          elem_bytes / 1e6, 256 * 12, k1.split("C = ")[-1], k2.split("C = ")[-1]))
eng.close()
