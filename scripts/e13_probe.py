"""The E13 flux-PDF adjustment on the device at the size of BASELINE configs[3] (N = 1e4 epochs, ~5e5-point segments): seconds
per simulated light curve and iterations to convergence, lognormal PDF; and the host loop on ONE of them for scale.
    python scripts/e13_probe.py [nsims]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.models import DampedRandomWalk
from mind_the_gaps_amd.simulator import Simulator

nsims = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t, y, dy = synth.make_lightcurves(10000, 1, seed=20250704 + 4)
kernel = DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 20.0), bounds=[(-10, 50), (-10, 10)])
expo = 0.5 * np.diff(t).min()
for where in ("device", "host"):
    sim = Simulator(kernel, t, expo, 100.0, "Lognormal", sigma_noise=1.0, extension_factor=2, random_state=1, adjust_on=where)
    S = nsims if where == "device" else 1
    model = sim._engine()[1]
    thetas = np.tile(model.full[model.free_index][None, :], (S, 1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if where == "device":
            sim.simulate(thetas[:2], seed=5)          # plans, rocFFT kernels
        t0 = time.perf_counter()
        out = sim.simulate(thetas, seed=6)
        dt = time.perf_counter() - t0
        if where == "device":       # once more with the plans of this batch made (rocFFT builds a Bluestein plan per length and batch)
            t0 = time.perf_counter()
            out = sim.simulate(thetas, seed=7)
            print("device, plans made: %.3f s = %.4f s per light curve (first call of this shape: %.3f s)" % (time.perf_counter() - t0, (time.perf_counter() - t0) / S, dt), flush=True)
    rep = getattr(sim, "last_adjustment", None)
    print("%-6s: %d light curves of %d epochs, segments of %d fine samples (grid %d): %.3f s = %.4f s per light curve%s"
          % (where, S, len(t), sim.seg_len, sim.fftndatapoints, dt, dt / S,
             "" if rep is None else ", %d iterations at most, %d not converged" % (rep["iterations"], rep["not_converged"])), flush=True)
