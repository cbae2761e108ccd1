"""GPU tests of the device TK95 simulator: exact host replay of the
Philox-driven pipeline with numpy's FFT, the statistical properties the reference's own
tests check (tests/simulator_test.py: mean, variance, PSD shape), noise models, residency."""
import warnings

import numpy as np
import pytest

import philox_replay
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.simulator import Simulator

pytestmark = pytest.mark.gpu
SPECTRUM, SHIFT, NOISE = 8, 9, 10


def regular_pattern(n=400, dt=1.0, exposure=0.2):
    return np.arange(0.5, dt * n, dt), exposure


def host_replay(sim, kernel, thetas, seed):
    """The device pipeline restated with numpy (irfft) and the same Philox draws."""
    nfft, dt = sim.fftndatapoints, sim.sim_dt
    k = np.arange(1, nfft // 2 + 1)
    omega = 2 * np.pi * k / (nfft * dt)
    out = []
    for s, th in enumerate(thetas):
        kernel.set_parameter_vector(th)
        r = philox_replay.philox(k, SPECTRUM, s, k >> 32, seed)
        u1, u2 = 1.0 - philox_replay.u01(r[0], r[1]), philox_replay.u01(r[2], r[3])
        rad = np.sqrt(-2.0 * np.log(u1))
        X = np.zeros(nfft // 2 + 1, dtype=complex)
        X[1:] = rad * (np.cos(2 * np.pi * u2) + 1j * np.sin(2 * np.pi * u2)) * np.sqrt(0.5 * kernel.get_psd(omega))
        if nfft % 2 == 0:
            X[-1] = X[-1].real
        counts = np.fft.irfft(X, n=nfft) * np.sqrt(nfft * dt * np.sqrt(2 * np.pi))
        rate = counts / dt + sim.mean
        rs = philox_replay.philox(0, SHIFT, s, 0, seed)
        span = (nfft - 1) - sim.seg_len
        j0 = int(np.ceil(float(philox_replay.u01(rs[0], rs[1])) * span)) if span > 0 else 0
        j0 = max(0, min(j0, nfft - sim.seg_len))
        seg = rate[j0:j0 + sim.seg_len]
        # the windows by the reference's own rule (simulator.py:358-362), independently of sim.win_lo / win_hi:
        # the cut segment is shifted so that its first sample sits half a step after the first window opens
        seg_time = sim.strategy[0][0] + dt / 2 + dt * np.arange(sim.seg_len)
        out.append([np.mean(seg[np.argwhere((seg_time >= start) & (seg_time < end))]) for start, end in sim.strategy])
    return np.array(out)


def test_device_pipeline_replays_on_the_host():
    times, exposure = regular_pattern(300)
    times = np.delete(times, np.arange(100, 140))                     # a gap in the observing pattern
    kernel = DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 20)) + Lorentzian(np.log(30.0), np.log(20.0), np.log(0.7))
    sim = Simulator(kernel, times, exposure, 10.0, "Gaussian", sigma_noise=1.0, extension_factor=3, random_state=1)
    thetas = kernel.get_parameter_vector() + 0.05 * np.random.default_rng(0).standard_normal((4, 5))
    out = sim.simulate(thetas, noise=False, seed=0xABCDEF12345)
    want = host_replay(sim, kernel, thetas, 0xABCDEF12345)
    assert out["rates"].shape == (4, len(times)) and np.all(out["dy"] == 0)
    assert np.max(np.abs(out["rates"] - want)) < 1e-9 * np.std(want)
    assert np.allclose(out["means"], out["rates"].mean(axis=1), rtol=1e-12)
    # a different seed gives a different realisation; the same seed the same one
    again = sim.simulate(thetas, noise=False, seed=0xABCDEF12345)["rates"]
    other = sim.simulate(thetas, noise=False, seed=7)["rates"]
    assert np.array_equal(again, out["rates"]) and not np.allclose(other, out["rates"])


def test_mean_variance_and_psd_shape():
    """As tests/simulator_test.py: the input mean is recovered, the variance is the integral of
    the PSD (a e^{-c tau} has variance a), and the ensemble periodogram follows the model."""
    times, exposure = regular_pattern(1000)
    variance, bend = 100.0, 20.0
    kernel = DampedRandomWalk(np.log(variance), np.log(2 * np.pi / bend))
    sim = Simulator(kernel, times, exposure, 10.0, "Gaussian", sigma_noise=1.0, extension_factor=5,
                    aliasing_factor=2, random_state=3)
    S = 400
    rates = sim.simulate(np.tile(kernel.get_parameter_vector(), (S, 1)), noise=False)["rates"]
    assert rates.shape == (S, 1000) and np.all(np.isfinite(rates))
    assert abs(rates.mean() - 10.0) < 4 * np.sqrt(variance / S)                  # ensemble mean
    var = rates.var(axis=1)
    assert abs(var.mean() / variance - 1.0) < 0.12                                # long series: close to a
    # periodogram averaged over the ensemble vs the DRW PSD shape (bend recovered within 25 %)
    freqs = np.fft.rfftfreq(1000, 1.0)[1:-1]
    power = np.mean(np.abs(np.fft.rfft(rates - rates.mean(axis=1, keepdims=True), axis=1)[:, 1:-1]) ** 2, axis=0)
    w = 2 * np.pi * freqs
    c = 2 * np.pi / bend
    model = 1.0 / (1.0 + (w / c) ** 2)
    lo = w < c / 2
    hi = (w > 3 * c) & (freqs < 0.4)
    ratio = (power[lo].mean() / model[lo].mean()) / (power[hi].mean() / model[hi].mean())
    assert 0.75 < ratio < 1.33


def test_noise_models():
    times, exposure = regular_pattern(500, exposure=0.5)
    kernel = DampedRandomWalk(np.log(4.0), np.log(0.3))
    theta = np.tile(kernel.get_parameter_vector(), (200, 1))
    g = Simulator(kernel, times, exposure, 50.0, "Gaussian", sigma_noise=2.5, extension_factor=2, random_state=5)
    out = g.simulate(theta, want_clean=True)
    resid = out["rates"] - out["clean"]
    assert np.all(out["dy"] == 2.5) and abs(resid.std() - 2.5) < 0.03 and abs(resid.mean()) < 0.03
    assert abs(np.mean(resid ** 3)) < 0.6                                         # symmetric
    p = Simulator(kernel, times, exposure, 50.0, "Gaussian", extension_factor=2, random_state=6)   # Poisson
    assert p.noise_name == "Poisson"
    out = p.simulate(theta, want_clean=True)
    counts = out["rates"] * exposure
    assert np.allclose(counts, np.round(counts)) and np.all(counts >= 0)
    lam = out["clean"] * exposure                                                 # ~25 counts per bin (PTRS branch)
    assert abs(np.mean(counts - lam)) < 0.05 and abs(np.var(counts - lam) / lam.mean() - 1.0) < 0.03
    assert np.allclose(out["dy"], np.sqrt(counts) / exposure)
    low = Simulator(kernel, times, 0.04, 50.0, "Gaussian", extension_factor=2, random_state=7)     # ~2 counts (Knuth branch)
    out = low.simulate(theta[:50], want_clean=True)
    counts, lam = out["rates"] * 0.04, out["clean"] * 0.04
    assert abs(np.mean(counts - lam)) < 0.02 and abs(np.var(counts - lam) / lam.mean() - 1.0) < 0.05
    # host-side add_noise keeps the reference's single-light-curve API
    noisy, dy = g.add_noise(np.full(len(times), 50.0))
    assert noisy.shape == dy.shape == times.shape and np.all(dy == 2.5)


def test_constructor_checks_and_single_realisation():
    times, exposure = regular_pattern(100)
    kernel = DampedRandomWalk(0.0, 0.0)
    with pytest.raises(ValueError):
        Simulator(kernel, times, exposure, 1.0, extension_factor=0.5)
    with pytest.raises(ValueError):
        Simulator(kernel, times, 0.0, 1.0)
    with pytest.raises(ValueError):
        Simulator(kernel, times, 5.0, 1.0, aliasing_factor=1)        # exposure longer than the spacing
    assert Simulator(kernel, times, exposure, 1.0, pdf="Lognormal").pdf == "Lognormal"   # E13 on the host
    assert callable(Simulator(lambda w: 1.0 / (1.0 + w * w), times, exposure, 1.0).psd_model)  # any callable is a PSD
    with pytest.raises(ValueError):
        Simulator(3.0, times, exposure, 1.0)
    sim = Simulator(kernel.get_psd, times, exposure, 1.0, sigma_noise=0.1, random_state=2)
    rate = sim.generate_lightcurve()
    assert rate.shape == (100,) and np.all(np.isfinite(rate))
    assert sim.fftndatapoints == len(sim.sim_timestamps) and sim.sim_dt == 0.1


def test_simulated_set_stays_resident_and_generate_from_posteriors(engine):
    """make_resident: the simulated curves are fitted without leaving the GPU, and give the same
    likelihoods as uploading them; GPModelling.generate_from_posteriors returns light curves."""
    times, exposure = regular_pattern(300)
    th = synth.truth([synth.K_DRW])
    kernel = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)])
    sim = Simulator(kernel, times, exposure, 100.0, sigma_noise=1.5, extension_factor=2, random_state=4)
    S = 6
    thetas = np.tile(th, (S, 1))
    out = sim.simulate(thetas, make_resident=True, seed=11)
    from mind_the_gaps_amd.gp import get_engine
    eng = get_engine(0)
    lc = np.arange(S, dtype=np.int32)
    resident, st = eng.loglike(thetas, lc, add_prior=False)
    engine.set_lightcurves(times, out["rates"], out["dy"] + 1e-12, y_offset=out["rates"].mean(axis=1))
    full, free, bounds = synth.model_spec([synth.K_DRW], out["rates"], per_lc_mean=True)
    engine.set_model([synth.K_DRW], full, free, bounds)
    uploaded, st2 = engine.loglike(thetas, lc, add_prior=False)
    assert np.all(st == 0) and np.all(st2 == 0)
    assert np.max(np.abs(resident - uploaded) / np.abs(uploaded)) < 1e-12
    # the facade
    y = out["rates"][0]
    g = GPModelling(GappyLightcurve(times, y, out["dy"][0], exposures=exposure), kernel)
    np.random.seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g.derive_posteriors(fit=False, max_steps=60, convergence_steps=30, walkers=8, progress=False)
        lcs = g.generate_from_posteriors(nsims=5, sigma_noise=1.5)
    assert len(lcs) == 5 and all(isinstance(l, GappyLightcurve) for l in lcs)
    assert all(l.n == 300 and np.all(l.dy == 1.5) and np.all(np.isfinite(l.y)) for l in lcs)
    assert abs(np.mean([l.mean for l in lcs]) - np.mean(y)) < 15.0
    g0 = GPModelling(GappyLightcurve(times, y, out["dy"][0]), kernel)      # no exposures given
    g0._mcmc_samples = g._mcmc_samples
    with pytest.raises(ValueError):                       # "Some exposure times are 0!" (simulator.py:201-202)
        g0.generate_from_posteriors(2)


def test_closed_form_spectra_drive_the_simulator_like_their_terms():
    """The tutorials hand Simulator the closed-form spectra of models/psd_models.py
    (tutorial_ppp.ipynb: Lorentzian + BendingPowerlaw); they simulate through the celerite
    term they are the spectrum of: same seed, same light curves."""
    from mind_the_gaps_amd.models import psd_models as psd
    times, exposure = regular_pattern(200)
    spectrum = psd.Lorentzian(30.0, 20.0, 0.7) + psd.BendingPowerlaw(100.0, 2 * np.pi / 20)
    kernel = Lorentzian(np.log(30.0), np.log(20.0), np.log(0.7)) + DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 20))
    a = Simulator(spectrum, times, exposure, 10.0, "Gaussian", sigma_noise=1.0, extension_factor=3, random_state=1)
    b = Simulator(kernel, times, exposure, 10.0, "Gaussian", sigma_noise=1.0, extension_factor=3, random_state=1)
    ra, rb = a.simulate(noise=False, seed=77)["rates"], b.simulate(noise=False, seed=77)["rates"]
    assert np.array_equal(ra, rb)
    w = np.linspace(0.01, 5, 50)
    assert np.allclose(a.psd_model(w), spectrum(w), rtol=1e-10)
    # a spectrum without a celerite twin is simulated from its table (any callable is a PSD model)
    m52 = Simulator(psd.Matern52(), times, exposure, 10.0, "Gaussian", sigma_noise=1.0, extension_factor=3, random_state=1)
    r52 = m52.simulate(noise=False, nsims=3)["rates"]
    assert r52.shape == (3, len(times)) and np.all(np.isfinite(r52)) and r52.std() > 0


@pytest.mark.parametrize("nsims", [1, 2, 5])
def test_chirp_z_transform_is_the_library_transform(nsims):
    """A grid length with large prime factors is transformed by hand on power-of-two transforms (csrc/mtg_simulate.hip,
    chirp-z: hipFFT's own plan for such a length takes 0.9 s to BUILD); forced on and off (Simulator(transform=...), mtg_set_simulate_transform) for the same seed the
    two paths give the same series -- odd and even numbers of series (two share a complex transform), one alone."""
    rng = np.random.default_rng(11)
    times = synth.make_times(157, rng)                    # (the grid length is whatever the reference's arithmetic makes it)
    kernel = DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 10), bounds=[(-10, 50), (-10, 10)]) + \
        Lorentzian(np.log(50.0), np.log(20.0), np.log(2 * np.pi / 3.0), bounds=[(-10, 50), (-10, 10), (-10, 10)])
    got = {}
    for mode, transform in (("1", "chirp-z"), ("0", "library")):
        sim = Simulator(kernel, times, 0.04, 25.0, "Gaussian", sigma_noise=0.5, extension_factor=3, random_state=4, transform=transform)
        thetas = np.tile(sim._engine()[1].full[sim._engine()[1].free_index][None, :], (nsims, 1))
        got[mode] = (sim.fftndatapoints, sim.simulate(thetas, seed=2468, want_clean=True))
    n = got["1"][0]
    assert n == got["0"][0]
    a, b = got["1"][1], got["0"][1]
    scale = np.std(b["clean"])
    assert scale > 0
    assert np.max(np.abs(a["clean"] - b["clean"])) <= 1e-11 * scale
    assert np.max(np.abs(a["rates"] - b["rates"])) <= 1e-11 * scale


def test_chirp_z_transform_even_and_odd_lengths(monkeypatch):
    """The hand-made transform on grids of both parities (an even length has a real Nyquist entry, taken as real like the
    k = 0 entry) against the library's, three series each (a full pair and a lone series)."""
    kernel = DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 10), bounds=[(-10, 50), (-10, 10)])
    seen = set()
    for seed in (100, 101, 104, 105):         # sampling patterns whose grids come out at 56 989, 57 051, 56 414, 55 246 points
        times = synth.make_times(150, np.random.default_rng(seed))
        got = {}
        for mode, transform in (("1", "chirp-z"), ("0", "library")):
            sim = Simulator(kernel, times, 0.04, 25.0, "Gaussian", sigma_noise=0.5, extension_factor=2, random_state=4, transform=transform)
            thetas = np.tile(sim._engine()[1].full[sim._engine()[1].free_index][None, :], (3, 1))
            got[mode] = sim.simulate(thetas, seed=97531, want_clean=True)["clean"]
        seen.add(sim.fftndatapoints % 2)
        assert np.max(np.abs(got["1"] - got["0"])) <= 1e-11 * np.std(got["0"]), (seed, sim.fftndatapoints)
    assert seen == {0, 1}


def test_regularly_sampled_series_str_and_set_psd_params():
    """simulator.py:260-298, 369-394: the whole fine-grid realisation (mean = the simulator's, variance = the PSD's integral),
    down-sampled by the reference's rule it gives a light curve on the observing pattern; __str__; set_psd_params."""
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw
    times = np.arange(0.0, 600.0, 1.0)
    psd = BendingPowerlaw(S0=4.0, omega0=2 * np.pi / 30)
    sim = Simulator(psd, times, 1.0, 50.0, pdf="Gaussian", sigma_noise=1.0, extension_factor=20, random_state=3)
    lc = sim.simulate_regularly_sampled()
    assert lc.n == sim.fftndatapoints == len(lc.countrate) and lc.dt == sim.sim_dt and np.array_equal(lc.time, sim.sim_timestamps)
    assert abs(lc.meanrate - 50.0) < 1e-9 and abs(lc.tseg - lc.n * lc.dt) < 1e-9
    assert 0.5 * 4.0 < np.var(lc.countrate) < 1.6 * 4.0             # S0 is the variance of this model (one realisation, 20 x 600 / 30 bends)
    down = sim.downsample(lc)
    assert len(down) == len(times) and np.all(np.isfinite(down)) and abs(np.mean(down) - 50.0) < 2.0
    other = sim.simulate_regularly_sampled()
    assert not np.array_equal(other.countrate, lc.countrate)        # the simulator's generator moved on
    text = str(sim)
    assert text.startswith("Simulator(") and "PDF: Gaussian" in text and "Noise: Gaussian" in text
    sim.set_psd_params({"S0": 400.0})
    assert psd.S0 == 400.0 and np.var(sim.simulate_regularly_sampled().countrate) > 20 * 4.0
    # a celerite kernel as the PSD: the same entry
    from mind_the_gaps_amd.models import DampedRandomWalk
    sim2 = Simulator(DampedRandomWalk(np.log(4.0), np.log(2 * np.pi / 30)), times, 1.0, 50.0, sigma_noise=1.0, extension_factor=20, random_state=3)
    lc2 = sim2.simulate_regularly_sampled()
    assert lc2.n == sim2.fftndatapoints and abs(lc2.meanrate - 50.0) < 1e-9 and 0.5 * 4.0 < np.var(lc2.countrate) < 1.6 * 4.0


def _golden_module():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_notebook_data.py")
    spec = importlib.util.spec_from_file_location("make_notebook_data", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_numpy_stream_gives_the_reference_notebooks_light_curves():
    """stream="numpy": np.random.seed(45) / (4) and the calls of docs/notebooks/celerite_variance.ipynb cells 6 / 14 return
    the light curves the NOTEBOOK had -- tests/golden/notebook_variance_data.npz, whose variances match the digits the
    notebook printed (tests/test_notebook_known_answer.py) -- to the rounding of the transform (the reference carries a
    10^6 zero-frequency term through its own: ~1e-11 of noise)."""
    import os
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw as BPL, Lorentzian as Lor
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_variance_data.npz"))
    times = np.linspace(0, 5000, 5000)
    exposures = 0.5 * np.ones(5000)
    w0 = 2 * np.pi / 100
    for seed, psd_model, key, printed in ((45, BPL(S0=1.0, omega0=w0), "cell6_rates", "0.97372"), (4, Lor(S0=1.0, omega0=w0, Q=5), "cell14_rates", "0.96105")):
        np.random.seed(seed)
        simulator = Simulator(psd_model, times, exposures, mean=0, pdf="Gaussian", extension_factor=1.0, stream="numpy")
        rates = simulator.generate_lightcurve()
        assert np.max(np.abs(rates - data[key])) < 1e-9 and "%.5f" % np.var(rates) == printed      # "Sample Variance: ..."
        # the generator was left where the reference leaves it: the next draw is the same
        state_next = np.random.uniform()
        np.random.seed(seed)
        np.random.normal(0, size=(2, simulator.fftndatapoints // 2 + 1)); np.random.uniform()
        assert np.random.uniform() == state_next


def test_numpy_stream_against_the_restated_reference_pipeline():
    """an ordinary case (gaps, unequal exposures, the series 3x longer than the light curve, a celerite kernel's get_psd as
    the spectrum, a mean), against tests/golden/make_notebook_data.reference_lightcurve; then noise from the global
    generator as noise_models.py:71,182 draws it"""
    from mind_the_gaps_amd import terms
    from mind_the_gaps_amd.models.celerite_models import Lorentzian
    ref = _golden_module().reference_lightcurve
    rng = np.random.default_rng(8)
    times = np.cumsum(rng.uniform(2.0, 5.0, 300))
    times[150:] += 400.0
    exposures = rng.uniform(0.8, 1.6, 300)
    kernel = Lorentzian(np.log(40.0), np.log(30.0), np.log(2 * np.pi / 25)) + terms.RealTerm(np.log(60.0), np.log(2 * np.pi / 90))
    for trial in range(3):
        np.random.seed(100 + trial)
        want = ref(kernel.get_psd, times, exposures, 50.0, 3)
        want_noisy = want + np.random.normal(scale=2.0, size=len(want))
        np.random.seed(100 + trial)
        simulator = Simulator(kernel.get_psd, times, exposures, 50.0, pdf="Gaussian", sigma_noise=2.0, extension_factor=3, stream="numpy")
        rates = simulator.generate_lightcurve()
        noisy, dy = simulator.add_noise(rates)
        assert np.max(np.abs(rates - want)) < 1e-8 and np.max(np.abs(noisy - want_noisy)) < 1e-8 and np.all(dy == 2.0)
    # Poisson noise: np.random.poisson of the same expected counts
    np.random.seed(7)
    simulator = Simulator(kernel.get_psd, times, exposures, 50.0, extension_factor=3, stream="numpy")
    rates = simulator.generate_lightcurve()
    state = np.random.get_state()
    noisy, dy = simulator.add_noise(rates)
    np.random.set_state(state)
    counts = np.random.poisson(rates * exposures)
    assert np.array_equal(noisy, counts / exposures) and np.allclose(dy, np.sqrt(counts) / exposures)
    # the whole fine-grid series of simulate_regularly_sampled: same normals, no cut
    np.random.seed(11)
    lc = simulator.simulate_regularly_sampled()
    assert lc.n == simulator.fftndatapoints and abs(lc.meanrate - 50.0) < 1e-9
    with pytest.raises(NotImplementedError):
        simulator.simulate(nsims=2)
    with pytest.raises(NotImplementedError):
        Simulator(kernel.get_psd, times, exposures, 50.0, pdf="Lognormal", stream="numpy")


# ---- a non-Gaussian flux PDF: the E13 adjustment on the device (csrc/mtg_e13.hip) against the host loop -----------------
def _shaped_simulators(pdf, n_epochs=90, seed=5, **kw):
    rng = np.random.default_rng(seed)
    times = synth.make_times(n_epochs, rng)
    kernel = DampedRandomWalk(np.log(16.0), np.log(2 * np.pi / 12), bounds=[(-10, 50), (-10, 10)])
    make = lambda where: Simulator(kernel, times, 0.04, 25.0, pdf, sigma_noise=0.5, extension_factor=2, random_state=4,
                                   adjust_on=where, **kw)
    return make("device"), make("host")


@pytest.mark.parametrize("pdf", ["Lognormal", "Uniform"])
def test_flux_pdf_adjustment_on_the_device_is_the_host_loop(pdf):
    """simulator.py:65-140 on the device: started from the SAME white series (pdf_draws), the batched device loop -- hipFFT
    transforms, a segmented rank sort per iteration, np.allclose's test per segment -- ends on the light curves the numpy
    loop ends on, series by series (each stops at its own iteration: a converged segment is frozen)."""
    dev, host = _shaped_simulators(pdf)
    S = 5
    model = dev._engine()[1]
    thetas = np.tile(model.full[model.free_index][None, :], (S, 1)) + 0.05 * np.arange(S)[:, None]
    rng = np.random.default_rng(8)
    draws = 25.0 * np.exp(0.15 * rng.standard_normal((S, dev.seg_len))) if pdf == "Lognormal" \
        else rng.uniform(18.0, 32.0, size=(S, dev.seg_len))
    a = dev.simulate(thetas, seed=4242, noise=False, pdf_draws=draws)
    b = host.simulate(thetas, seed=4242, noise=False, pdf_draws=draws)
    assert dev.last_adjustment["not_converged"] == 0 and 2 <= dev.last_adjustment["iterations"] <= dev.max_iter
    assert a["rates"].shape == b["rates"].shape == (S, len(dev._times))
    assert np.max(np.abs(a["rates"] - b["rates"])) <= 1e-12 * np.max(np.abs(b["rates"]))
    # ... and it did something: the Gaussian (TK95) light curves of the same seed are others
    plain = Simulator(dev._kernel, dev._times, 0.04, 25.0, "Gaussian", sigma_noise=0.5, extension_factor=2, random_state=4)
    assert np.max(np.abs(plain.simulate(thetas, seed=4242, noise=False)["rates"] - a["rates"])) > 0.1


def test_flux_pdf_on_the_device_draws_blocks_and_noise():
    """The device's own draws: lognormal light curves are positive with the simulator's mean; a set simulated in blocks
    (index_base) is the set of one call; noise and make_resident follow the adjusted series without a host round trip; a
    run that is not given enough iterations says so as the reference does."""
    dev, _ = _shaped_simulators("Lognormal", n_epochs=70)
    S = 6
    model = dev._engine()[1]
    thetas = np.tile(model.full[model.free_index][None, :], (S, 1))
    whole = dev.simulate(thetas, seed=77, noise=False, index_base=0)
    assert np.all(whole["rates"] > 0) and abs(whole["rates"].mean() / 25.0 - 1.0) < 0.1
    assert dev.last_adjustment["not_converged"] == 0
    parts = [dev.simulate(thetas[lo:hi], seed=77, noise=False, index_base=lo)["rates"] for lo, hi in ((0, 2), (2, 3), (3, 6))]
    assert np.array_equal(np.vstack(parts), whole["rates"])
    eng = dev._engine()[0]
    noisy = dev.simulate(thetas, seed=77, noise=True, want_clean=True, make_resident=True, index_base=0)
    assert eng.L == S                    # the adjusted, noisy set is the resident set (no host round trip)
    assert np.array_equal(noisy["clean"], whole["rates"]) and np.all(noisy["dy"] == 0.5)
    assert 0.3 < np.std(noisy["rates"] - noisy["clean"]) < 0.7
    short, _ = _shaped_simulators("Lognormal", n_epochs=70, max_iter=1)
    with pytest.warns(UserWarning, match="did not converge after 1 iterations"):
        short.simulate(thetas[:2], seed=77, noise=False)
    assert short.last_adjustment == {"not_converged": 2, "iterations": 2}


def test_kraft_noise_on_the_device_is_add_noise_epoch_by_epoch():
    """noise_models.py:81-150 on the device (mtg_set_simulate_kraft): every (rate, dy) pair the device returns is what the
    host's add_noise formulas give for an integer number of total counts -- the Poisson branch for 15 counts or more, the
    tabulated Kraft-Burrows-Nousek median and half-interval below --, the counts scatter around rate x exposure + background
    as Poisson counts do, and a set simulated in blocks is the set of one call."""
    rng = np.random.default_rng(4)
    times = synth.make_times(150, rng)
    kernel = DampedRandomWalk(np.log(900.0), np.log(2 * np.pi / 12), bounds=[(-10, 50), (-10, 10)])
    sim = Simulator(kernel, times, 0.04, 400.0, "Gaussian", bkg_rate=30.0, bkg_rate_err=2.0, extension_factor=2, random_state=1)
    assert sim.noise_name == "Kraft" and sim.adjust_on == "device"
    S = 40
    model = sim._engine()[1]
    thetas = np.tile(model.full[model.free_index][None, :], (S, 1))
    out = sim.simulate(thetas, seed=31, want_clean=True, index_base=0)
    expo, bkg, err = sim._exposures, sim._bkg_counts, sim._bkg_rate_err
    med, half = sim._kraft_tables()
    K = med.shape[1]
    assert K == 15 and np.all(np.diff(med, axis=1) > 0) and np.all(half > 0)
    rates, dy, clean = out["rates"], out["dy"], out["clean"]
    assert np.all(np.isfinite(rates)) and np.all(dy > 0)
    totals = np.full(rates.shape, -1.0)
    # faint epochs: (rate, dy) = (median, half)[epoch][c] / exposure for exactly one c < 15
    for c in range(K):
        hit = (rates == med[:, c][None, :] / expo[None, :]) & (dy == half[:, c][None, :] / expo[None, :])
        assert not np.any(hit & (totals >= 0))
        totals[hit] = c
    faint = totals >= 0
    # the others: an integer total >= 15 through the Poisson formulas
    t_guess = np.round(rates * expo + bkg)
    poisson = (~faint) & (t_guess >= 15) & (np.abs(rates - (t_guess - bkg) / expo) <= 1e-9 * np.abs(rates) + 1e-9) \
        & (np.abs(dy - np.sqrt((np.sqrt(t_guess) / expo) ** 2 + err ** 2)) <= 1e-9 * dy)
    totals[poisson] = t_guess[poisson]
    assert np.all(totals >= 0), "an epoch whose (rate, dy) is neither branch of add_noise"
    assert faint.mean() > 0.05 and poisson.mean() > 0.05          # both branches exercised
    lam = clean * expo + bkg
    z = (totals - lam) / np.sqrt(lam)
    assert abs(z.mean()) < 0.1 and 0.9 < z.std() < 1.1              # Poisson counts around the clean rates
    parts = [sim.simulate(thetas[lo:hi], seed=31, index_base=lo) for lo, hi in ((0, 6), (6, 7), (7, 40))]
    assert np.array_equal(np.vstack([p["rates"] for p in parts]), rates) and np.array_equal(np.vstack([p["dy"] for p in parts]), dy)
    # the host's own add_noise on the same tables agrees with them (the table IS add_noise's per-epoch computation)
    host = Simulator(kernel, times, 0.04, 400.0, "Gaussian", bkg_rate=30.0, bkg_rate_err=2.0, extension_factor=2, random_state=1,
                     adjust_on="host")
    r_h, dy_h = host.add_noise(np.full(len(times), 250.0))           # 10 + 1.2 counts expected: mostly faint
    for n in np.flatnonzero(r_h * expo + bkg < 14.5)[:20]:
        c = int(np.argmin(np.abs(med[n] / expo[n] - r_h[n])))
        assert r_h[n] == med[n, c] / expo[n] and dy_h[n] == half[n, c] / expo[n]
