"""Throughput kernel against the ORDER of the batch: L = 2000 resident light curves,
N = 1e4, alt model (J = 6); walkers per light curve in {1, 9, 32, 256}, rows grouped by light curve or shuffled, with the
sweep's own sort by (structure, light curve) off (round 2 behaviour) and on (default).  Device-resident inputs
(mtg_loglike_batch_device), HIP-event kernel time of the solve (prepare + sort in their own column).

    python scripts/order_sweep.py [L] [N]      ->  table on stdout (profiles/r03_order_sweep.txt)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
kinds = synth.ALT_MODEL
t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
eng = Engine(0)
eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
eng.set_model(kinds, full, free, bounds)
eng.set_time_parallel(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
rng = np.random.default_rng(3)
print("# L = %d light curves, N = %d, model DRW+SHO+Lorentzian; evals/s from the HIP-event time of the solve kernel(s);" % (L, N))
print("# prep_ms = theta -> coefficients (+ the sort when on)")
print("%-8s %-8s %-5s %10s %10s %10s %12s  %s" % ("wpl", "order", "sort", "B", "solve_ms", "prep_ms", "evals/s", "identical to grouped"))
for wpl in (1, 9, 32, 256):
    B = L * wpl
    theta = synth.draw_thetas(kinds, B, seed=20250704 + 40)
    lc = np.repeat(np.arange(L, dtype=np.int32), wpl)
    ref = None
    for order in ("grouped", "random"):
        perm = np.arange(B) if order == "grouped" else rng.permutation(B)
        d_theta = torch.from_numpy(theta[perm]).to(dev)
        d_lc = torch.from_numpy(lc[perm]).to(dev)
        d_out = torch.empty(B, dtype=torch.float64, device=dev)
        d_st = torch.empty(B, dtype=torch.int32, device=dev)
        for sort in (0, 1):
            eng.set_sort(sort)
            reps = 3
            for _ in range(2):
                eng.loglike_device(B, d_theta.data_ptr(), d_lc.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            eng.profile_begin(reps)
            for _ in range(reps):
                eng.loglike_device(B, d_theta.data_ptr(), d_lc.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            prep, solve = eng.profile_read()
            out = np.empty(B); out[perm] = d_out.cpu().numpy()
            if ref is None:
                ref = out
            same = bool(np.array_equal(out, ref))
            print("%-8d %-8s %-5s %10d %10.3f %10.3f %12.4e  %s" % (wpl, order, "on" if sort else "off", B, np.min(solve), np.min(prep),
                                                                  B / (np.min(solve) * 1e-3), same), flush=True)
eng.close()
