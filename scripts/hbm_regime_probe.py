"""The regime where the nominal HBM roofline of SURVEY 8(d) is a physical one: few walkers per light curve over so many
light curves that nothing is shared through L2 / MALL -- every row streams its own light curve's (y, sigma^2) from HBM
(16 N bytes; the 8 N bytes of t are the shared sampling's and stay in cache).  Light curves are born on the device
(mtg_set_lightcurves_device).  Prints evaluations/s, the bytes that MUST come from HBM per launch (rows' own light curves,
each counted once per launch when W walkers of a light curve sit in one wave) and their rate against 8 TB/s and against
the device-to-device copy bandwidth measured in the same process.

    python scripts/hbm_regime_probe.py [L ...]         default 131072 262144 524288; N = 10 000
Under rocprofv3 --pmc FETCH_SIZE the same launches give the measured traffic (scripts/hbm_regime.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

N = 10000
Ls = [int(a) for a in sys.argv[1:]] or [131072, 262144, 524288]
dev = torch.device("cuda", 0)
kinds = synth.ALT_MODEL
th = synth.truth(kinds)
P = len(th)
eng = Engine(0)
eng.set_time_parallel(0)
rng = np.random.default_rng(7)
t = torch.from_numpy(synth.make_times(N, rng)).to(dev)

# the copy bandwidth of this box (read + write bytes)
a = torch.empty(1 << 31, dtype=torch.uint8, device=dev).fill_(1)
b = torch.empty_like(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(6):
    e0.record(); b.copy_(a); e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1))
copy_gbs = 2 * a.numel() / best / 1e6
del a, b
print("# device-to-device copy of 2 GiB: %.0f GB/s (read + write)" % copy_gbs, flush=True)
print("# N = %d, model DRW+SHO+Lorentzian, one-lane sweep; own bytes = 16 N per light curve a launch touches" % N)
print("%8s %4s %9s %9s %11s %9s %9s %9s  %s" % ("L", "wpl", "rows", "solve_ms", "evals/s", "own GB", "GB/s", "of copy", "kernel"), flush=True)
for L in Ls:
    gen = torch.Generator(device=dev); gen.manual_seed(L)
    y = torch.empty((L, N), dtype=torch.float64, device=dev)
    dy = torch.empty((L, N), dtype=torch.float64, device=dev)
    for i in range(0, L, 8192):                              # in pieces: the generators' temporaries stay small
        j = min(L, i + 8192)
        y[i:j] = 100.0 + 10.0 * torch.randn((j - i, N), dtype=torch.float64, device=dev, generator=gen)
        dy[i:j] = 0.5 + 1.5 * torch.rand((j - i, N), dtype=torch.float64, device=dev, generator=gen) + 1e-12
    off = y.mean(dim=1).contiguous()
    torch.cuda.synchronize()
    eng.set_lightcurves_device(N, L, t.data_ptr(), y.data_ptr(), dy.data_ptr(), False, off.data_ptr())
    eng.synchronize()
    del y, dy
    torch.cuda.empty_cache()
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    eng.set_model(kinds, full, np.arange(P, dtype=np.int32), bounds)
    for wpl in (1, 2, 4, 8):
        B = L * wpl
        if B > 2 ** 22:
            continue
        theta = torch.from_numpy(th + 0.05 * np.abs(th) * rng.standard_normal((B, P))).to(dev)
        lc = torch.arange(L, dtype=torch.int32, device=dev).repeat_interleave(wpl).contiguous()
        out = torch.empty(B, dtype=torch.float64, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        ms = 1e9
        for rep in range(3):
            eng.profile_begin(1)
            eng.loglike_device(B, theta.data_ptr(), lc.data_ptr(), out.data_ptr(), st.data_ptr(), add_prior=True, stream=0)
            torch.cuda.synchronize()
            _, solve = eng.profile_read()
            ms = min(ms, float(solve[0]))
        ok = int((st == 0).sum().item())
        own = L * N * 16                                         # every light curve once per launch
        print("%8d %4d %9d %9.3f %11.4e %9.2f %9.0f %9.3f  %s" % (L, wpl, B, ms, ok / ms * 1e3, own / 1e9, own / ms / 1e6,
                                                                  own / ms / 1e6 / copy_gbs, eng.last_solver), flush=True)
        del theta, lc, out, st
eng.close()
