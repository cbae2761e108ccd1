"""Seed for rocFFT's run-time-compilation cache with the transforms mtg_chain_autocorr makes (D2Z / Z2D of every
power-of-two length 2^3 .. 2^21): on this ROCm build rocFFT compiles the kernels of a new length the first time a
plan asks for it, ~1.2 s each, and keeps them in the file ROCFFT_RTC_CACHE_PATH names (default: under ~/.cache).
Run on an MI355X:
    ROCFFT_RTC_CACHE_PATH=$PWD/gpurun_out/rocfft_cache_gfx950.db python scripts/make_rocfft_cache.py
and copy the file to mind_the_gaps_amd/rocfft_cache_gfx950.db; the script writes the name of the librocfft build next to
it (rocfft_cache_gfx950.db.version).  engine.load_library gives every process a private copy of the seed in the
temporary directory and points rocFFT at that copy -- never at the tracked file -- when the variable is not set and the
stamp matches the librocfft it finds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd.engine import Engine
assert os.environ.get("ROCFFT_RTC_CACHE_PATH"), "set ROCFFT_RTC_CACHE_PATH to the file to fill"
eng = Engine(0)
rng = np.random.default_rng(0)
for bits in range(3, 22):
    n2 = 1 << bits
    n_t = n2 // 2
    t0 = time.perf_counter()
    eng.chain_autocorr(rng.standard_normal((n_t, 2, 1)))
    print("length 2^%d: %.2f s" % (bits, time.perf_counter() - t0), flush=True)
print(os.path.getsize(os.environ["ROCFFT_RTC_CACHE_PATH"]), "bytes")
from mind_the_gaps_amd.engine import rocfft_library_version
open(os.environ["ROCFFT_RTC_CACHE_PATH"] + ".version", "w").write(rocfft_library_version() + "\n")
