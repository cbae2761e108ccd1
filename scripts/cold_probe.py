import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("config3_probe", os.path.join(ROOT, "scripts", "config3_probe.py"))
probe = importlib.util.module_from_spec(spec); spec.loader.exec_module(probe)
from mind_the_gaps_amd.gpmodelling import GPModelling
which = sys.argv[1]
t0 = time.perf_counter()
lc, _, _ = probe._observed(10000, 0)
print("observed light curve %.3f s" % (time.perf_counter() - t0), flush=True)
make = dict(zip(("null", "alt"), probe._kernels()))[which]
import mind_the_gaps_amd.device_sampler as ds
orig = ds._autocorr_time_where_it_is_cheapest
def traced(*a, **k):
    t1 = time.perf_counter(); r = orig(*a, **k); print("   autocorr check %.4f s" % (time.perf_counter() - t1), flush=True); return r
ds._autocorr_time_where_it_is_cheapest = traced
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        g = GPModelling(lc, make(), device=0, random_state=np.random.RandomState(11 + rep))
        t0 = time.perf_counter()
        g.fit()
        t1 = time.perf_counter()
        g.derive_posteriors(fit=False, max_steps=1000, walkers=256, progress=False, device_sampler=True)
        print("%s rep %d: fit %.3f s, chain %.3f s" % (which, rep, t1 - t0, time.perf_counter() - t1), flush=True)
