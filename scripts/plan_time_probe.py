"""How long hipfftPlan1d takes for the Bluestein length of BASELINE configs[3] (1 087 853 points) by batch size, in a process
whose rocFFT kernels for that length are already compiled, and for the neighbouring power of two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mind_the_gaps_amd.engine import Engine
eng = Engine(0)
lib, ctx = eng._lib, eng._ctx
for nfft in (1087853, 1 << 20, 1 << 21):
    for batch in (1, 1, 2, 4, 16, 1):
        os.environ["MTG_SIM_BATCH"] = str(batch)
        t0 = time.perf_counter()
        rc = lib.mtg_simulate_plan(ctx, nfft)
        print("nfft %8d batch %2d: plan %.3f s (rc %d)" % (nfft, batch, time.perf_counter() - t0, rc), flush=True)
