"""The reference's own known answers for the simulator, restated:
tests/simulator_test.py:192-253 (three down-sampling cases with explicit index lists), :255-304
(durations and sampling of the cut segment), :44-90 (slope recovery with a power-law PSD -- any
callable is a PSD model, simulator.py:272-280), plus the host-side parts this repository adds for
API completeness (Emmanoulopoulos et al. 2013 flux PDFs, Kraft noise).  The CPU tests exercise the
window rule and the host code; the GPU tests push known series and spectra through the device path."""
import warnings

import numpy as np
import pytest

from mind_the_gaps_amd.simulator import Simulator, kraft_interval, kraft_median


def power_law(amplitude, alpha):
    """astropy's PowerLaw1D(amplitude, alpha) as the reference's tests use it: amplitude * omega^-alpha."""
    return lambda omega: amplitude * np.asarray(omega, dtype=np.float64) ** -alpha


class FineLightcurve:
    """What the reference's tests build with stingray: times and count rates on a fine regular grid."""
    def __init__(self, time, countrate):
        self.time, self.countrate = time, countrate


DOWNSAMPLING_CASES = [   # exposure, expected indices of the fine grid per epoch (simulator_test.py:192-253)
    (0.5, [[3, 4, 5, 6, 7], [23, 24, 25, 26, 27], [43, 44, 45, 46, 47], [63, 64, 65, 66, 67]]),
    (0.6, [[2, 3, 4, 5, 6, 7, 8], [22, 23, 24, 25, 26, 27, 28], [42, 43, 44, 45, 46, 47, 48], [62, 63, 64, 65, 66, 67, 68]]),
    (0.1, [[5], [25], [45], [65]]),
]


@pytest.mark.parametrize("exposure,indices", DOWNSAMPLING_CASES)
def test_downsampling_known_answers(exposure, indices):
    timestamps = np.append(np.arange(1, 3.1, 2), np.arange(5, 7.1, 2))
    times = np.arange(0.5, 10.1, 0.1)
    countrates = np.linspace(5, 20, len(times)) / exposure
    simu = Simulator(power_law(10, 2), timestamps, exposure, 0, extension_factor=1.0, aliasing_factor=1)
    truerates = [np.mean(countrates[idx[0]:idx[-1] + 1]) for idx in indices]
    assert simu.downsample(FineLightcurve(times, countrates)) == truerates
    lo, hi = simu._windows(times)                                   # the index sets themselves
    assert [list(range(a, b)) for a, b in zip(lo, hi)] == indices


def test_constructor_grid_and_checks():
    """Grid arithmetic of simulator.py:213-238 and the argument checks of :197-226."""
    timestamps = np.arange(0.0, 10.0, 0.1)
    simu = Simulator(power_law(1, 1), timestamps, 0.1, 0.5, extension_factor=50, aliasing_factor=1)
    assert simu.sim_dt == 0.1
    assert np.allclose(np.diff(simu.sim_timestamps), 0.1)
    assert simu.sim_timestamps[-1] - simu.sim_timestamps[0] >= 50 * (timestamps[-1] - timestamps[0])
    # the cut segment covers the observed duration plus the margins of the first and last windows
    assert abs(simu.seg_len * simu.sim_dt - simu.sim_duration) <= simu.sim_dt
    assert simu.sim_duration > timestamps[-1] - timestamps[0]
    with pytest.raises(ValueError):
        Simulator(power_law(1, 1), timestamps, 0.1, 0.5, extension_factor=0.5)
    with pytest.raises(ValueError):
        Simulator(power_law(1, 1), timestamps, 0.1, 0.5, epsilon=0.9)
    with pytest.raises(ValueError):
        Simulator(power_law(1, 1), timestamps, 0.0, 0.5)
    with pytest.raises(ValueError):
        Simulator(power_law(1, 1), timestamps, 0.5, 0.5)          # exposures longer than the spacing
    with pytest.raises(ValueError):
        Simulator(power_law(1, 1), timestamps, 0.1, 0.5, pdf="cauchy")
    with pytest.raises(ValueError):
        Simulator("not callable", timestamps, 0.1, 0.5)


def test_kraft_posterior_helpers():
    from scipy import integrate
    for N, B in [(0, 0.5), (3, 1.2), (5, 0.0), (10, 4.0), (2, 6.0), (14, 2.5)]:
        f = lambda s: np.exp(-(s + B)) * (s + B) ** N
        Z = integrate.quad(f, 0, np.inf)[0]
        cdf = lambda s: integrate.quad(f, 0, s)[0] / Z
        lo, hi = kraft_interval(N, B, 0.68)
        assert abs(cdf(kraft_median(N, B)) - 0.5) < 1e-7
        assert abs(cdf(hi) - cdf(lo) - 0.68) < 1e-7
        assert lo == 0.0 and f(0.0) >= f(hi) or abs(f(lo) / f(hi) - 1.0) < 1e-6     # shortest interval
    # no background: Kraft's table 1, N = 5 -> [3.07, 7.61] at 68 %? (the shortest interval of a Gamma(6))
    lo, hi = kraft_interval(5, 0.0, 0.68)
    assert 3.0 < lo < 3.1 and 7.5 < hi < 7.7


def test_kraft_noise_on_the_host():
    """noise_models.py:81-150: Poisson counts with background; epochs below 15 counts get the posterior
    median and half the 68 % interval."""
    n = 400
    times = np.arange(n) * 1.0
    simu = Simulator(power_law(1, 1), times, 0.5, 30.0, bkg_rate=np.full(n, 4.0), bkg_rate_err=np.full(n, 0.3), random_state=2)
    assert simu.noise_name == "Kraft"
    bright = np.full(n, 200.0)                       # ~100 counts: plain Poisson branch
    r, e = simu.add_noise(bright)
    assert abs(r.mean() - 200.0) < 3.0 and np.allclose(e, np.sqrt((np.sqrt((r * 0.5 + 2.0)) / 0.5) ** 2 + 0.09))
    faint = np.full(n, 6.0)                          # ~3 + 2 counts: Bayesian branch
    state = simu.random_state.get_state()
    r, e = simu.add_noise(faint)
    simu.random_state.set_state(state)               # the same Poisson draw again, by hand
    total = simu.random_state.poisson(faint * 0.5 + 2.0)
    assert np.all(total < 15)
    for i in (0, 7, 123, n - 1):
        lo, hi = kraft_interval(int(total[i]), 2.0, 0.68)
        assert r[i] == kraft_median(int(total[i]), 2.0) / 0.5 and e[i] == (hi - lo) / 2 / 0.5
    assert np.all(r >= 0.0) and np.all(e > 0.0)


def test_flux_pdf_adjustment_on_the_host():
    """simulator.py:65-140: the adjusted series has exactly the values of a sample of the wanted PDF
    (mean = the simulator's, std = the segment's) and keeps the segment's Fourier amplitudes closely."""
    rng = np.random.default_rng(3)
    n = 4096
    red = np.cumsum(rng.standard_normal(n))
    red = 50.0 + 5.0 * (red - red.mean()) / red.std()
    times = np.arange(64) * 4.0
    for pdf in ("Lognormal", "Uniform"):
        simu = Simulator(power_law(1, 2), times, 1.0, 50.0, pdf, random_state=4, max_iter=300)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = simu._adjust_pdf(red)
        assert out.shape == red.shape and abs(out.mean() - 50.0) < 0.5 and abs(out.std() / 5.0 - 1.0) < 0.1
        amp_in, amp_out = np.abs(np.fft.rfft(red))[1:200], np.abs(np.fft.rfft(out))[1:200]
        assert np.corrcoef(np.log(amp_in), np.log(amp_out))[0, 1] > 0.9
        if pdf == "Uniform":
            assert out.min() >= 50.0 - np.sqrt(3) * 5.0 * 1.05 and out.max() <= 50.0 + np.sqrt(3) * 5.0 * 1.05
        else:
            assert np.all(out > 0) and np.mean((out - out.mean()) ** 3) > 0            # skewed to the right


# ---------------------------------------------------------------------------------------------------------
# device path
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("exposure,gap", [(0.5, 2.0), (0.6, 2.0), (0.1, 2.0), (0.3, 1.0)])
def test_device_windows_against_the_rule(engine, exposure, gap):
    """The observe kernel averages exactly the fine samples the reference's rule selects: series x_j = j
    and x_j = j^2 on the simulator's own segment grid determine both ends of every window."""
    timestamps = np.append(np.arange(1, 3.1, gap), np.arange(5 + gap, 9.1, gap))
    simu = Simulator(power_law(10, 2), timestamps, exposure, 0, extension_factor=2.0, aliasing_factor=3)
    seg_time = simu.segment_times
    want = [np.argwhere((seg_time >= start) & (seg_time < end)).ravel() for start, end in simu.strategy]
    assert all(len(w) > 0 for w in want)
    nfft, start = simu.fftndatapoints, 7
    j = np.arange(nfft, dtype=np.float64) - start                    # segment index of every fine sample
    n = len(timestamps)
    engine.set_lightcurves(timestamps, np.zeros((1, n)), np.ones((1, n)))
    got = engine.tk95_observe_series(np.vstack([j, j * j]), simu.seg_len, start, simu.win_lo, simu.win_hi)
    for e, idx in enumerate(want):
        assert got[0, e] == np.mean(idx.astype(float)) and got[1, e] == np.mean(idx.astype(float) ** 2)
    assert [list(range(a, b)) for a, b in zip(simu.win_lo, simu.win_hi)] == [list(w) for w in want]


@pytest.mark.gpu
def test_slope_recovery_with_a_power_law(engine):
    """simulator_test.py:44-61: the mean periodogram slope of an ensemble simulated from PowerLaw1D(alpha = 1)
    is -1 (a callable PSD: tabulated on the host, mtg_simulate_tk95's psd_table)."""
    dt, points, beta = 0.5, 500, 1.0
    timestamps = np.arange(0, points, dt) + dt / 2
    simu = Simulator(power_law(1, beta), timestamps, dt, 0, aliasing_factor=1, extension_factor=1.05, random_state=11)
    rates = simu.simulate(noise=False, nsims=250)["rates"]
    assert rates.shape == (250, len(timestamps)) and np.all(np.isfinite(rates))
    freqs = np.fft.rfftfreq(len(timestamps), dt)[1:-1]
    power = np.abs(np.fft.rfft(rates, axis=1)[:, 1:-1]) ** 2
    # log-periodogram regression (the bias of E[log chi^2_2] moves the intercept only)
    slopes = np.array([np.polyfit(np.log(freqs), np.log(p), 1)[0] for p in power])
    assert abs(np.mean(slopes) + beta) < np.std(slopes)              # the reference's criterion
    assert abs(np.mean(slopes) + beta) < 4 * np.std(slopes) / np.sqrt(len(slopes)) + 0.03


@pytest.mark.gpu
def test_lognormal_flux_pdf_end_to_end(engine):
    """simulator_test.py:64-90: lognormal flux PDF -- slope and mean are kept, the rates are positive."""
    dt, points, beta, mean = 0.5, 300, 1.0, 100.0
    timestamps = np.arange(0, points, dt) + dt / 2
    simu = Simulator(power_law(1, beta), timestamps, dt, mean, "Lognormal", aliasing_factor=1, extension_factor=1.05,
                     random_state=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rates = simu.simulate(noise=False, nsims=40)["rates"]
    assert rates.shape == (40, len(timestamps)) and np.all(rates > 0)
    freqs = np.fft.rfftfreq(len(timestamps), dt)[1:-1]
    power = np.abs(np.fft.rfft(rates, axis=1)[:, 1:-1]) ** 2
    slopes = np.array([np.polyfit(np.log(freqs), np.log(p), 1)[0] for p in power])
    assert abs(np.mean(slopes) + beta) < 3 * np.std(slopes)
    assert abs(rates.mean(axis=1).mean() - mean) < 3 * rates.mean(axis=1).std()
    skew = np.mean(((rates - rates.mean(axis=1, keepdims=True)) / rates.std(axis=1, keepdims=True)) ** 3)
    assert skew > 0.0


@pytest.mark.gpu
def test_kraft_noise_through_generate_from_posteriors(engine):
    """A light curve with background rates simulates (Kraft noise on the host), and the simulated set can
    still be made resident for the refits."""
    from mind_the_gaps_amd.models import DampedRandomWalk
    times = np.arange(0.5, 200.0, 1.0)
    kernel = DampedRandomWalk(np.log(4.0), np.log(0.3))
    simu = Simulator(kernel, times, 0.5, 20.0, "Gaussian", bkg_rate=np.full(len(times), 2.0),
                     bkg_rate_err=np.full(len(times), 0.2), extension_factor=2, random_state=6)
    out = simu.simulate(np.tile(kernel.get_parameter_vector(), (6, 1)), make_resident=True)
    assert out["rates"].shape == (6, len(times)) and np.all(np.isfinite(out["rates"])) and np.all(out["dy"] > 0)
    assert abs(out["rates"].mean() - 20.0) < 2.0


def test_noise_model_classes_are_the_simulators_noise():
    """mind_the_gaps/noise_models.py:14-184 by name: same numbers as Simulator(stream="numpy").add_noise for the same state of
    numpy's global generator (both draw from it, as the reference does), names and error shapes as there"""
    from mind_the_gaps_amd import noise_models as nm
    rng = np.random.default_rng(2)
    n = 40
    times = np.arange(n) * 10.0
    exposures = rng.uniform(1.0, 3.0, n)
    rates = rng.uniform(0.5, 30.0, n)
    rates[:8] = rng.uniform(0.0, 2.0, 8)                      # faint epochs: the Kraft branch
    bkg_rate, bkg_err = rng.uniform(0.1, 0.5, n), rng.uniform(0.01, 0.05, n)
    flat = lambda w: np.ones_like(w)
    cases = [(nm.GaussianNoise(exposures, 1.5), dict(sigma_noise=1.5), "Gaussian"),
             (nm.PoissonNoise(exposures), dict(), "Poisson"),
             (nm.KraftNoise(exposures, bkg_rate * exposures, bkg_err), dict(bkg_rate=bkg_rate, bkg_rate_err=bkg_err), "Kraft")]
    for model, kwargs, name in cases:
        sim = Simulator(flat, times, exposures, 10.0, stream="numpy", **kwargs)
        assert model.name == name == sim.noise_name
        np.random.seed(3)
        got, got_dy = model.add_noise(rates.copy())
        np.random.seed(3)
        want, want_dy = sim.add_noise(rates.copy())
        assert np.array_equal(got, want) and np.array_equal(got_dy, want_dy) and got.shape == got_dy.shape == (n,)
    with pytest.raises(NotImplementedError):
        nm.BaseNoise("x").add_noise(rates)


@pytest.mark.gpu
@pytest.mark.parametrize("pdf", ["Lognormal", "Uniform"])
def test_flux_pdf_of_a_million_point_light_curve(engine, pdf):
    """simulator_test.py:376-432 (test_pdf_lognormal / test_pdf_uniform) at the reference's own size -- 10^6 regular epochs,
    one fine sample each, a bending power law, max_iter 1000 -- through the DEVICE adjustment: the light curve has the wanted
    PDF with the simulator's mean and the TK95 segment's variance, to the reference's tolerances (mean 0.1 / 0.01, variance
    0.2 / 0.1), and still the segment's power spectrum."""
    from scipy import stats
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw
    dt, n, mean = 1.0, 1000000, 10.0
    timestamps = np.arange(0, n, dt)
    psd = BendingPowerlaw(S0=10, omega0=2 * np.pi / 1000)
    make = lambda kind: Simulator(psd, timestamps, dt, mean, kind, extension_factor=1.05, aliasing_factor=1, max_iter=1000, random_state=15)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = make("Gaussian").simulate(noise=False, nsims=1, seed=151)["rates"][0]      # the TK95 segment itself (same seed, same cut)
        simu = make(pdf)
        out = simu.simulate(noise=False, nsims=1, seed=151)["rates"][0]
    assert simu.last_adjustment["not_converged"] == 0 and simu.last_adjustment["iterations"] > 1
    input_var = np.var(plain)
    assert out.shape == (n,) and abs(out.mean() - mean) < (0.1 if pdf == "Lognormal" else 0.01)
    assert abs(np.var(out) - input_var) < (0.2 if pdf == "Lognormal" else 0.1)
    sub = out[::97]
    if pdf == "Lognormal":
        s = np.sqrt(np.log(input_var / mean ** 2 + 1.0))
        assert np.all(out > 0) and stats.kstest(sub, stats.lognorm(s, scale=mean ** 2 / np.sqrt(input_var + mean ** 2)).cdf).pvalue > 1e-3
    else:
        half = np.sqrt(3.0 * input_var)
        assert out.min() >= mean - half and out.max() <= mean + half
        assert stats.kstest(sub, stats.uniform(mean - half, 2 * half).cdf).pvalue > 1e-3
    # the adjustment keeps the amplitudes it was given: binned periodograms of the two series agree
    k = np.arange(1, 20001)
    p_in, p_out = (np.abs(np.fft.rfft(x - x.mean())[k]) ** 2 for x in (plain, out))
    bins = np.array_split(np.arange(len(k)), 40)
    ratio = np.array([p_out[b].mean() / p_in[b].mean() for b in bins])
    assert np.all(np.abs(np.log(ratio)) < 0.25), ratio
