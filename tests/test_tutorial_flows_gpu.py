"""The reference's two tutorials driven through this package with the calls, keywords and attribute names the notebooks
use (docs/notebooks/tutorial_model_selection.ipynb cells 0-10, tutorial_ppp.ipynb cells 0-15), shortened chains: a user
who pastes a tutorial and swaps the imports must get through it.  The numbers are checked where the tutorial's own
logic fixes them (the true model wins the AICc comparison by a wide margin; residuals of the right model look normal;
the likelihood-ratio statistic is non-negative up to the optimiser's slack)."""
import warnings

import numpy as np
import pytest
from scipy.stats import ks_1samp, norm, percentileofscore

from mind_the_gaps_amd import terms
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
# (the notebook imports its PSD models from `mind_the_gaps.models`, as an older release exported them; today's
# models/__init__.py:1-2 exports the celerite terms under those names and the PSD models live in models/psd_models.py)
from mind_the_gaps_amd.models.psd_models import BendingPowerlaw, Lorentzian as LorentzianPSD
from mind_the_gaps_amd.models.celerite_models import Lorentzian
from mind_the_gaps_amd.simulator import Simulator
from mind_the_gaps_amd.stats import aicc

pytestmark = pytest.mark.gpu


def test_model_selection_tutorial():
    np.random.seed(10)
    times = np.arange(0, 1000)
    exposure = np.diff(times)[0]
    mean, rms = 100, 0.1
    variance_drw = (mean * rms) ** 2
    w, w_bend = 2 * np.pi / 25, 2 * np.pi / 40
    kernel = Lorentzian(log_S0=np.log(variance_drw), log_Q=np.log(80), log_omega0=np.log(w)) \
        + terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend)) \
        + terms.Matern32Term(np.log(np.sqrt(variance_drw)), np.log(10 / 2 / np.pi), eps=1e-8)
    truth = kernel.get_parameter_vector()
    assert len(truth) == 7 and len(kernel.terms) == 3
    psd_model = kernel.get_psd                                     # a bound method as the simulator's PSD
    simulator = Simulator(psd_model, times, np.ones(len(times)) * exposure, mean, pdf="Gaussian", sigma_noise=10, extension_factor=2,
                          random_state=10)      # (the notebook leaves the simulator unseeded: np.random.seed does not reach its RandomState)
    countrates = simulator.generate_lightcurve()
    noisy_countrates, dy = simulator.add_noise(countrates)
    input_lc = GappyLightcurve(times, noisy_countrates, dy, exposures=exposure)
    freqs = np.arange(1 / input_lc.duration, 1 / (2 * exposure), 1 / input_lc.duration)
    total = sum(term.get_psd(2 * np.pi * freqs) for term in kernel.terms)
    np.testing.assert_allclose(total, psd_model(2 * np.pi * freqs), rtol=1e-12)
    lc_variance = np.var(input_lc.y)

    def bounds_variance(variance, margin=15):
        return np.log(variance / margin), np.log(variance * margin)
    variance_bounds = bounds_variance(lc_variance)
    bend_bounds = (np.log(2 * np.pi / input_lc.duration), np.log(1 / (2 * exposure) * 2 * np.pi))
    sigma_bounds = bounds_variance(np.sqrt(lc_variance))
    timescale_bounds = (np.log(exposure), np.log(input_lc.duration))
    Q_bounds = (np.log(1.5), np.log(1000))
    log_var = np.log(lc_variance)
    realterm = terms.RealTerm(log_var, np.log(2 * np.pi / 50), bounds=[variance_bounds, bend_bounds])
    lorentzian = Lorentzian(log_var, np.log(100), np.log(2 * np.pi / 10), bounds=[variance_bounds, Q_bounds, bend_bounds])
    matern = terms.Matern32Term(np.log(np.sqrt(lc_variance)), np.log(10), bounds=[sigma_bounds, timescale_bounds], eps=1e-8)
    models = [realterm, matern, lorentzian + realterm, lorentzian + realterm + matern]
    cpus, aiccs, pvalues, gps = 12, [], [], []
    for k in models:
        assert str(k)                                              # the tutorial prints every kernel
        gp = GPModelling(input_lc, k)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                        # "The chains did not converge ..."
            gp.derive_posteriors(fit=True, max_steps=600, walkers=2 * cpus, cores=cpus, progress=False)
        gp.gp.set_parameter_vector(gp.max_parameters)
        std_res = gp.standarized_residuals()
        pvalues.append(ks_1samp(std_res, norm.cdf).pvalue)
        aiccs.append(aicc(gp.max_loglikelihood, gp.k, input_lc.n))
        gps.append(gp)
    assert np.all(np.isfinite(aiccs)) and [g.k for g in gps] == [2, 2, 5, 7]
    # the QPO is in the data: both models holding a Lorentzian beat both that do not, by far (the notebook: ~115)
    assert max(aiccs[2:]) < min(aiccs[:2]) - 10
    best_gp = gps[int(np.argmin(aiccs))]
    assert best_gp.mcmc_samples.shape[1] == len(best_gp.gp.get_parameter_names())
    best_gp.gp.set_parameter_vector(best_gp.max_parameters)
    assert np.array_equal(best_gp.gp.get_parameter_vector(), best_gp.max_parameters)
    pred_mean, pred_var = best_gp.gp.predict(input_lc.y, return_var=True)
    assert pred_mean.shape == pred_var.shape == times.shape and np.all(pred_var > 0)
    assert pvalues[int(np.argmin(aiccs))] > 1e-3                   # residuals of the preferred model are compatible with N(0, 1)


def test_ppp_tutorial():
    np.random.seed(10)
    cpus = 15
    times = np.arange(0, 400)
    dt = np.diff(times)[0]
    mean = 100
    variance_drw = (mean * 0.1) ** 2
    w_bend = 2 * np.pi / 20
    psd_model = BendingPowerlaw(variance_drw, w_bend)
    simulator = Simulator(psd_model, times, np.ones(len(times)) * dt, mean, pdf="Gaussian", extension_factor=2, random_state=10)
    countrates = simulator.generate_lightcurve()
    noisy_countrates, dy = simulator.add_noise(countrates)
    input_lc = GappyLightcurve(times, noisy_countrates, dy, exposures=dt)

    bounds_drw = dict(log_a=(-10, 50), log_c=(-10, 10))
    null_kernel = terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend), bounds=bounds_drw)
    null_model = GPModelling(input_lc, null_kernel)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        null_model.derive_posteriors(max_steps=1500, fit=True, cores=cpus)
    assert null_model.mcmc_samples.shape[1] == 2 and list(null_model.gp.get_parameter_names()) == ["kernel:log_a", "kernel:log_c"]
    autocorr = null_model.autocorr
    assert len(autocorr) >= 1 and np.all(np.isfinite(autocorr))

    w = 2 * np.pi / 10
    bounds_qpo = dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5))
    alternative_kernel = terms.ComplexTerm(log_a=np.log(variance_drw), log_c=np.log(0.5 * w / 80), log_d=np.log(w), bounds=bounds_qpo) \
        + terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend), bounds=bounds_drw)
    alternative_model = GPModelling(input_lc, alternative_kernel)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        alternative_model.derive_posteriors(max_steps=1500, fit=True, cores=cpus)
    assert alternative_model.mcmc_samples.shape[1] == 5

    Nsims = 6
    lcs = null_model.generate_from_posteriors(Nsims, cpus=cpus)
    assert len(lcs) == Nsims
    likelihoods_null, likelihoods_alt = [], []
    for lc in lcs:
        assert np.array_equal(lc.times, times) and lc.y.shape == lc.dy.shape == times.shape
        null_modelling = GPModelling(lc, null_kernel)
        null_modelling.derive_posteriors(fit=True, cores=cpus, walkers=2 * cpus, max_steps=200, progress=False)
        likelihoods_null.append(null_modelling.max_loglikelihood)
        alternative_modelling = GPModelling(lc, alternative_kernel)
        alternative_modelling.derive_posteriors(fit=True, cores=cpus, walkers=2 * cpus, max_steps=200, progress=False)
        likelihoods_alt.append(alternative_modelling.max_loglikelihood)
    T_dist = -2 * (np.array(likelihoods_null) - np.array(likelihoods_alt))
    T_obs = -2 * (null_model.max_loglikelihood - alternative_model.max_loglikelihood)
    perc = percentileofscore(T_dist, T_obs)
    assert np.all(np.isfinite(T_dist)) and np.isfinite(T_obs) and 0.0 <= 1 - perc / 100 <= 1.0
    assert np.all(T_dist > -1.0) and T_obs > -1.0                   # nested models: the alternative cannot fit much worse

    # second part of the tutorial (cells 19-20): a sum of PSD models, max_iter keyword
    psd_sum = LorentzianPSD(variance_drw, 80, w) + BendingPowerlaw(variance_drw, w_bend)
    sim2 = Simulator(psd_sum, times, np.ones(len(times)) * dt, mean, pdf="Gaussian", max_iter=500, random_state=11)
    rates = sim2.generate_lightcurve()
    noisy_rates, dy2 = sim2.add_noise(rates)
    assert rates.shape == noisy_rates.shape == dy2.shape == times.shape


def test_celerite_variance_notebook_flows():
    """docs/notebooks/celerite_variance.ipynb: error bars of 1e-12 on a regular grid (cells 8, 16), bounds given as a dict,
    a frozen kernel parameter and celerite.GP + scipy's minimize driven by hand (cells 35-38)."""
    from scipy.optimize import minimize
    from mind_the_gaps_amd.gp import GP
    from mind_the_gaps_amd.models import BendingPowerlaw as BPL_celerite, DampedRandomWalk as DRW
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw as BPL
    np.random.seed(45)
    Npoints = 1500
    times = np.linspace(0, 1500, Npoints)
    exposures = 0.5 * np.ones(Npoints)
    w0 = 2 * np.pi / 100
    simulator = Simulator(BPL(S0=1.0, omega0=w0), times, exposures, mean=0, pdf="Gaussian", extension_factor=1.0, random_state=45)
    rates = simulator.generate_lightcurve()
    S0 = np.var(rates)
    bounds = dict(log_S0=(-10, 10), log_omega0=(-10, 10))
    kernel = DRW(log_S0=np.log(S0), log_omega0=np.log(w0), bounds=bounds)
    gpmodel = GPModelling(GappyLightcurve(times, rates, dy=np.ones(len(rates)) * 1e-12), kernel)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gpmodel.derive_posteriors(max_steps=1000, fit=True, cores=12)
    assert gpmodel.gp.parameter_names == ("kernel:log_S0", "kernel:log_omega0", "mean:value")     # cell 12's print
    ratio_amplitude, ratio_break = np.exp(gpmodel.max_parameters[0]) / S0, np.exp(gpmodel.max_parameters[1]) / w0
    assert 0.5 < ratio_amplitude < 2.0 and 0.4 < ratio_break < 2.5                                # the notebook finds 1.0006 / 0.868
    kernel.set_parameter_vector(gpmodel.max_parameters)
    assert np.all(kernel.get_psd(2 * np.pi * np.arange(1e-3, 1.0, 1e-3)) > 0)

    # cells 35-38: a BendingPowerlaw term with log_Q frozen, the GP and the optimiser driven by hand
    y, time = rates, times
    Q = 1 / 2
    kernel = BPL_celerite(log_S0=np.log(np.var(y) / (w0 * Q)) + 3.0, log_Q=np.log(Q), log_omega0=np.log(w0),
                          bounds=dict(log_S0=(0, 15), log_omega0=(-20, 5)))   # a > b inside the box: K stays positive definite
    kernel.freeze_parameter("log_Q")
    gp = GP(kernel, mean=np.mean(y))
    gp.compute(time)
    initial_params = gp.get_parameter_vector()
    assert list(gp.get_parameter_names()) == ["kernel:log_S0", "kernel:log_omega0"] and len(initial_params) == 2
    assert gp.get_parameter_bounds() == [(0, 15), (-20, 5)]

    def neg_log_like(params, y, gp):
        gp.set_parameter_vector(params)
        return -gp.log_likelihood(y)
    solution = minimize(neg_log_like, initial_params, method="L-BFGS-B", bounds=gp.get_parameter_bounds(), args=(y, gp))
    assert np.isfinite(solution.fun) and solution.fun <= neg_log_like(initial_params, y, gp)
    gp.set_parameter_vector(solution.x)
    assert np.all(gp.kernel.get_psd(2 * np.pi * np.arange(1e-3, 1.0, 1e-3)) > 0)
    assert kernel.get_parameter_vector(include_frozen=True)[1] == np.log(Q)                       # the frozen one stayed
