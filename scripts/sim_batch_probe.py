"""Simulation of S series of BASELINE configs[3]'s grid (1 087 853 points) with the plan's batch forced through MTG_SIM_BATCH:
plan creation time and time per simulate call.   MTG_SIM_BATCH=16 python scripts/sim_batch_probe.py [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("config3_probe", os.path.join(ROOT, "scripts", "config3_probe.py"))
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t0 = time.perf_counter()
lc, sim, _ = probe._observed(10000, 0)
print("batch %s: first single series (plan + kernels cold) %.3f s" % (os.environ.get("MTG_SIM_BATCH", "default"), time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); sim.generate_lightcurve(); print("   second single series %.4f s" % (time.perf_counter() - t0), flush=True)
th = np.tile(sim._engine()[1].full[sim._engine()[1].free_index][None, :], (S, 1))
for rep in range(3):
    t0 = time.perf_counter(); out = sim.simulate(th, seed=5); print("   %d series: %.3f s" % (S, time.perf_counter() - t0), flush=True)
