"""Soak of Simulator(stream="numpy") against the restated reference pipeline (tests/golden/make_notebook_data.py) over random
grids: regular and gappy sampling, scalar and per-epoch exposures, aliasing factors 1-3, extension factors 1-6 (1 = the
degenerate case of the celerite_variance notebook, where the series is shorter than the cut), odd and even grid lengths,
closed-form spectra, celerite kernels and plain callables.   python scripts/numpy_stream_soak.py [cases] [seed]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mind_the_gaps_amd import terms
from mind_the_gaps_amd.models.psd_models import BendingPowerlaw, Lorentzian as LorentzianPSD
from mind_the_gaps_amd.models.celerite_models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.simulator import Simulator

spec = importlib.util.spec_from_file_location("make_notebook_data", os.path.join(ROOT, "tests", "golden", "make_notebook_data.py"))
gold = importlib.util.module_from_spec(spec); spec.loader.exec_module(gold)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, worst, odd, even = 0, 0.0, 0, 0
for case in range(cases):
    n = int(rng.integers(20, 400))
    if rng.random() < 0.4:
        times = np.arange(n) * float(rng.uniform(0.5, 5.0))
    else:
        times = np.cumsum(rng.uniform(1.0, 4.0, n))
        if rng.random() < 0.5:
            times[n // 2:] += rng.uniform(10, 300)
    min_gap = np.min(np.diff(times))
    aliasing = int(rng.integers(1, 4))
    exposures = np.full(n, rng.uniform(0.2, 0.9) * min_gap) if rng.random() < 0.5 else rng.uniform(0.3, 0.9, n) * min_gap
    extension = float(rng.choice([1.0, 1.5, 2, 3, 6]))
    mean = float(rng.uniform(0, 100))
    w = 2 * np.pi / rng.uniform(5, 50)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        psd, name = BendingPowerlaw(S0=float(rng.uniform(0.5, 20)), omega0=w), "BPL"
    elif kind == 1:
        psd, name = LorentzianPSD(S0=float(rng.uniform(0.5, 20)), omega0=w, Q=float(rng.uniform(2, 50))), "Lorentzian"
    elif kind == 2:
        k = Lorentzian(np.log(rng.uniform(1, 20)), np.log(rng.uniform(2, 50)), np.log(w)) + DampedRandomWalk(np.log(rng.uniform(1, 20)), np.log(w / 3))
        psd, name = k.get_psd, "kernel.get_psd"
    else:
        beta = float(rng.uniform(0.5, 2.5))
        psd, name = (lambda om, beta=beta: (np.abs(om) + 1e-3) ** -beta), "power law callable"
    seed = int(rng.integers(1 << 30))
    try:
        np.random.seed(seed)
        want = gold.reference_lightcurve(psd, times, exposures, mean, extension, aliasing_factor=aliasing)
        np.random.seed(seed)
        sim = Simulator(psd, times, exposures, mean, pdf="Gaussian", sigma_noise=1.0, extension_factor=extension, aliasing_factor=aliasing, stream="numpy")
        got = sim.generate_lightcurve()
    except Exception as e:
        bad += 1
        print("CASE %d raised %r (n %d aliasing %d extension %g %s)" % (case, e, n, aliasing, extension, name), flush=True)
        continue
    odd += sim.fftndatapoints & 1; even += 1 - (sim.fftndatapoints & 1)
    scale = max(1.0, float(np.nanmax(np.abs(want - mean))))
    same_nan = np.array_equal(np.isnan(want), np.isnan(got))
    err = float(np.nanmax(np.abs(got - want))) / scale if same_nan and not np.all(np.isnan(want)) else (0.0 if same_nan else np.inf)
    worst = max(worst, err)
    if not err < 1e-8:
        bad += 1
        print("CASE %d MISMATCH %.3e (n %d nfft %d aliasing %d extension %g %s)" % (case, err, n, sim.fftndatapoints, aliasing, extension, name), flush=True)
print("numpy stream: %d cases (%d odd, %d even grid lengths), worst relative difference %.2e, %d bad" % (cases, odd, even, worst, bad), flush=True)
raise SystemExit(1 if bad else 0)
