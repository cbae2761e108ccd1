"""mtg_chain_autocorr against the host's FFTs: first call for a shape (plans made) and a repeated one.
python scripts/acf_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd.engine import Engine
from mind_the_gaps_amd.sampler import _mean_autocorr_function
eng = Engine(0)
t0 = time.perf_counter(); eng.start_fft_warmup(); eng._join_fft_warmup()
print("hipFFT start-up: %.2f s" % (time.perf_counter() - t0))
rng = np.random.default_rng(1)
for n_t, W, P in ((200, 128, 5), (1000, 128, 5), (1000, 256, 8), (4000, 256, 8), (5000, 256, 8), (50000, 12, 5), (40, 512, 15)):
    x = rng.standard_normal((n_t, W, P)).cumsum(axis=0)
    t0 = time.perf_counter(); eng.chain_autocorr(x); first = time.perf_counter() - t0
    t0 = time.perf_counter(); a = eng.chain_autocorr(x); dev = time.perf_counter() - t0
    t0 = time.perf_counter(); b = _mean_autocorr_function(x); host = time.perf_counter() - t0
    print("%6d x %3d x %2d: device %.1f ms (first call for the shape %.1f ms), host %.1f ms, max diff %.1e"
          % (n_t, W, P, dev * 1e3, first * 1e3, host * 1e3, np.nanmax(np.abs(a - b))), flush=True)
