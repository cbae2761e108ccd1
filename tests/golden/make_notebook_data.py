#!/usr/bin/env python3
"""Rebuilds the two light curves of the reference's docs/notebooks/celerite_variance.ipynb that ARE reproducible: cells 6
and 14 call np.random.seed(45) / np.random.seed(4) in the very cell that simulates, and the reference's simulator draws
everything from numpy's global legacy generator (simulator.py:468-501 get_fft: np.random.normal(size=(2, N // 2 + 1));
simulator.py:536-539 cut_random_segment: np.random.uniform), so the series depends on numpy alone (pyfftw's inverse
transform = numpy's to rounding).  The pipeline is restated below from the reference's text (simulator.py:196-258 grid
and windows, :369-394 TK95 series, :397-420 cut / shift / down-sample, :340-367 window rule); nothing of the reference is
imported.  That the restatement is the pipeline the notebook ran is CHECKED against numbers the notebook printed:

  cell 6  "Sample Variance: 0.97372";  cell 12 prints max_parameters[0] = -0.02605619 and exp(max_parameters[0]) / var =
          1.000578571036844  =>  var = 0.973716978816 +- 5e-9        (this script: difference -3.9e-9)
  cell 14 "Sample Variance: 0.96105";  cell 20: -0.08662917 and 0.9541861557182113  =>  var = 0.961046316320 +- 5e-9
          (this script: -2.8e-9)

(the cut keeps the sample at its stop time -- the one convention the printed digits decide: without it the variances are
2.4e-5 and 2.3e-6 off).  Output: tests/golden/notebook_variance_data.npz (the two series, 5000 float64 each).

    python tests/golden/make_notebook_data.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mind_the_gaps_amd.models.psd_models import BendingPowerlaw, Lorentzian   # pinned by the reference's own outputs (psd_golden.npz)


def reference_lightcurve(psd_model, times, exposures, mean, extension_factor, aliasing_factor=2, epsilon=1.001):
    """Simulator(psd_model, times, exposures, mean, pdf="Gaussian", extension_factor=...).generate_lightcurve() with numpy's
    global generator in the state the caller left it"""
    sim_dt = np.min(exposures) / aliasing_factor
    dt = np.diff(times)
    start_time, end_time = times[0] - dt[0] / 1.99, times[-1] + dt[-1]
    sim_duration = end_time - start_time
    grid = np.arange(start_time - sim_dt, start_time + (times[-1] - times[0]) * extension_factor + sim_dt, sim_dt)
    n = len(grid)
    windows = [(t - h, t + h) for t, h in zip(times, exposures / 2 * epsilon)]
    # one TK95 realisation on the fine grid
    omega = np.fft.rfftfreq(n, sim_dt) * 2 * np.pi
    re, im = np.random.normal(0, size=(2, n // 2 + 1))
    spectrum = np.empty(len(omega), dtype=complex)
    spectrum[1:] = (re + 1j * im)[1:] * np.sqrt(0.5 * psd_model(omega[1:]))
    spectrum[0] = 1e6
    if n % 2 == 0:
        spectrum[-1] = spectrum[-1].real
    rate = np.fft.irfft(spectrum, n=n) * np.sqrt(n * sim_dt * np.sqrt(2 * np.pi)) / sim_dt
    rate = rate - np.mean(rate) + mean
    # a random cut of the light curve's length (both end samples kept), moved onto the observing windows
    shift = np.random.uniform(grid[0], grid[-1] - sim_duration)
    first = np.flatnonzero(grid >= shift)[0]
    last = np.flatnonzero(grid <= shift + sim_duration)[-1]
    cut_t, cut_r = grid[first:last + 1], rate[first:last + 1]
    cut_t = cut_t + (windows[0][0] - (cut_t[0] - 0.5 * sim_dt))
    return np.array([np.mean(cut_r[(cut_t >= a) & (cut_t < b)]) for a, b in windows])


def reference_regular_series(psd_model, times, exposures, mean, extension_factor, aliasing_factor=2):
    """Simulator(...).simulate_regularly_sampled() (simulator.py:369-394): the whole fine-grid series -> (grid, rates)"""
    sim_dt = np.min(exposures) / aliasing_factor
    dt = np.diff(times)
    start_time = times[0] - dt[0] / 1.99
    grid = np.arange(start_time - sim_dt, start_time + (times[-1] - times[0]) * extension_factor + sim_dt, sim_dt)
    n = len(grid)
    omega = np.fft.rfftfreq(n, sim_dt) * 2 * np.pi
    re, im = np.random.normal(0, size=(2, n // 2 + 1))
    spectrum = np.empty(len(omega), dtype=complex)
    spectrum[1:] = (re + 1j * im)[1:] * np.sqrt(0.5 * psd_model(omega[1:]))
    spectrum[0] = 1e6
    if n % 2 == 0:
        spectrum[-1] = spectrum[-1].real
    rate = np.fft.irfft(spectrum, n=n) * np.sqrt(n * sim_dt * np.sqrt(2 * np.pi)) / sim_dt
    return grid, rate - np.mean(rate) + mean


def poisson_level_series():
    """docs/notebooks/poisson_level.ipynb cells 2 and 4 (executed one after the other, np.random.seed(42) in the first):
    1 728 002 points.  Cell 4 prints the kernel it builds from the series' variance, `DampedRandomWalk(0.017072777961537826,
    ...)`: log(np.var(lc.countrate)) with all its digits -- this restatement gives exactly that number.
    -> (time, noiseless rates, rates + the cell's Gaussian noise)"""
    np.random.seed(42)
    times = np.linspace(0, 1000, 1000) * 3600 * 24
    grid, rate = reference_regular_series(BendingPowerlaw(S0=1.0, omega0=np.exp(-13)), times, 1000 * np.ones(1000), 0, 10)
    return grid, rate, rate + np.random.normal(0, 0.5, size=len(rate))


def main():
    times = np.linspace(0, 5000, 5000)
    exposures = 0.5 * np.ones(5000)
    w0 = 2 * np.pi / 100
    np.random.seed(45)                                            # cell 6
    drw = reference_lightcurve(BendingPowerlaw(S0=1.0, omega0=w0), times, exposures, 0, 1.0)
    np.random.seed(4)                                             # cell 14
    lor = reference_lightcurve(Lorentzian(S0=1.0, omega0=w0, Q=5), times, exposures, 0, 1.0)
    for name, series, printed5, log_s, ratio in (("cell 6", drw, 0.97372, -0.02605619, 1.000578571036844),
                                                 ("cell 14", lor, 0.96105, -0.08662917, 0.9541861557182113)):
        var, target = np.var(series), np.exp(log_s) / ratio
        print("%s: variance %.12f, the notebook's %.5f and %.12f (difference %.1e)" % (name, var, printed5, target, var - target))
        assert "%.5f" % var == "%.5f" % printed5 and abs(var - target) < 6e-9
    np.savez(os.path.join(HERE, "notebook_variance_data.npz"), times=times, cell6_rates=drw, cell14_rates=lor)
    grid, rate, _ = poisson_level_series()
    print("poisson_level: %d points, log variance %r, the notebook's 0.017072777961537826" % (len(grid), float(np.log(np.var(rate)))))
    assert abs(np.log(np.var(rate)) - 0.017072777961537826) < 1e-14


if __name__ == "__main__":
    main()
