"""The one log-likelihood NUMBER the reference's own material holds (SURVEY §8c said there was none: it is in a notebook
output, not in a test): docs/notebooks/celerite_variance.ipynb, cells 24-27.  A white-noise light curve of 1000 points is
given to `celerite.GP(JitterTerm(log_sigma=log 3), mean=np.mean(y))`, `gp.compute(time)` (celerite's default yerr,
1.123e-12), and `scipy.optimize.minimize(neg_log_like, method="L-BFGS-B")` prints

      fun: 3433.545361920608        x: [2.01460783]          (cell 26 / cell 27 output)

next to "Sample Variance: 56.21669" of the same series (cell 24 output).  The series itself is not reproducible here
(unseeded pyfftw simulation), but with a white kernel and the mean fixed at the sample mean the likelihood depends on
the data only through N and the sample variance, both printed: -lnL(x) = N/2 [ var / s2 + ln s2 + ln 2 pi ],
s2 = exp(2 x) + yerr^2.  The printed variance has 5 decimals (+-5e-6), which bounds the printed minimum to
[3433.545332, 3433.545422]: the reference's celerite value pins the normalisation of this build's lnL (the 1/2, the
N ln 2 pi, jitter = exp(2 log_sigma), the default yerr) to 1.3e-8 relative -- celerite run by the reference's author,
not by this build.  Any series with that N, mean and variance must give that number through the oracle (CPU) and
through the HIP path (GPU)."""
import numpy as np
import pytest

N, SAMPLE_VARIANCE, PRINTED_FUN, PRINTED_X = 1000, 56.21669, 3433.545361920608, 2.01460783
ROUNDING = 0.5 * N * 5e-6 / SAMPLE_VARIANCE * 1.02          # of `fun`, from the 5 printed decimals of the variance


def series(seed=0):
    """a series of the notebook's length, sampling and sample variance (np.var, ddof 0), any mean"""
    z = np.random.default_rng(seed).standard_normal(N)
    z = (z - z.mean()) * np.sqrt(SAMPLE_VARIANCE / np.var(z))
    times = np.linspace(0, 1000, N) * 3600 * 24                # cell 24: seconds
    return times, z + 0.3


def test_the_printed_value_is_inside_the_interval_its_printed_variance_allows():
    s2 = np.exp(2 * PRINTED_X) + 1.123e-12 ** 2
    lo, hi = (0.5 * N * (v / s2 + np.log(s2) + np.log(2 * np.pi)) for v in (SAMPLE_VARIANCE - 5e-6, SAMPLE_VARIANCE + 5e-6))
    assert lo < PRINTED_FUN < hi and hi - lo < 1e-4
    # the printed optimum against the maximum-likelihood sigma: L-BFGS-B stopped with jac = 2.6e-3 (cell 26), curvature 2 N
    assert abs(0.5 * np.log(SAMPLE_VARIANCE) - PRINTED_X) < 2.0 * 2.592e-3 / (2 * N)


def test_oracle_gives_the_notebooks_number():
    from oracle import dense
    from oracle import celerite as oc
    t, y = series()
    dy = np.full(N, 1.123e-12 - 1e-12)        # the oracle adds the facade's 1e-12 (gpmodelling.py:54); celerite's default is 1.123e-12
    co = dense.build_coeffs([dense.K_JITTER], [PRINTED_X])
    want = -dense.dense_loglike(t, y, dy, co, 0, [np.mean(y)])
    assert abs(want - PRINTED_FUN) < ROUNDING
    got, st = oc.logprob_batch(t, y, dy, [dense.K_JITTER], np.array([PRINTED_X, np.mean(y)]))
    assert st[0] == 0 and abs(-got[0] - PRINTED_FUN) < ROUNDING


@pytest.mark.gpu
def test_hip_path_gives_the_notebooks_number_as_the_notebook_drives_it():
    """cells 26-27 with this build's GP / JitterTerm in place of celerite's"""
    from scipy.optimize import minimize
    from mind_the_gaps_amd import terms
    from mind_the_gaps_amd.gp import GP
    for seed in (0, 1):
        time, y = series(seed)
        kernel = terms.JitterTerm(log_sigma=np.log(3))
        gp = GP(kernel, mean=np.mean(y))
        gp.compute(time)
        assert gp.get_parameter_names() == ("kernel:log_sigma",) and gp.parameter_names == ("kernel:log_sigma", "mean:value")

        def neg_log_like(params, y, gp):
            gp.set_parameter_vector(params)
            return -gp.log_likelihood(y)
        assert abs(neg_log_like(np.array([PRINTED_X]), y, gp) - PRINTED_FUN) < ROUNDING
        solution = minimize(neg_log_like, gp.get_parameter_vector(), method="L-BFGS-B", bounds=gp.get_parameter_bounds(), args=(y, gp))
        assert solution.success and abs(solution.fun - PRINTED_FUN) < ROUNDING
        assert abs(solution.x[0] - PRINTED_X) < 2e-4          # the notebook's own jac is 2.6e-3 at its stopping point


# ---- the white kernel beyond the notebook's case: celerite takes a JitterTerm alone, so must the HIP path ------------------
@pytest.mark.gpu
def test_white_kernel_against_the_dense_algebra(engine):
    """unequal errors, several light curves, a batch larger than a workgroup, the prior's verdicts"""
    from oracle import dense
    rng = np.random.default_rng(4)
    Nn, L, B = 333, 3, 700
    t = np.cumsum(0.05 + rng.exponential(1.0, Nn))
    y = 5.0 + rng.standard_normal((L, Nn)) * 2.0
    dy = rng.uniform(0.3, 1.5, (L, Nn))
    thetas = rng.uniform(-1.0, 1.5, (B, 1))
    lc = rng.integers(0, L, B).astype(np.int32)
    mean = 4.9
    engine.set_lightcurves(t, y, dy + 1e-12)
    bounds = np.array([[-0.8, 1.4], [-np.inf, np.inf]])
    engine.set_model([dense.K_JITTER], np.array([0.0, mean]), np.array([0], np.int32), bounds)
    out, st = engine.loglike(thetas, lc_index=lc, add_prior=True)
    inside = (thetas[:, 0] >= -0.8) & (thetas[:, 0] <= 1.4)
    assert np.array_equal(st == 0, inside) and np.array_equal(st == 1, ~inside) and np.all(np.isneginf(out[~inside]))
    assert "white" in engine.last_solver
    for b in np.flatnonzero(inside)[:60]:
        want = dense.dense_loglike(t, y[lc[b]], dy[lc[b]], dense.build_coeffs([dense.K_JITTER], thetas[b]), 0, [mean])
        assert abs(out[b] - want) <= 1e-12 * abs(want)
    closed = [-0.5 * np.sum((y[l] - mean) ** 2 / ((dy[l] + 1e-12) ** 2 + np.exp(2 * th)) + np.log((dy[l] + 1e-12) ** 2 + np.exp(2 * th))
                            + np.log(2 * np.pi)) for th, l in zip(thetas[inside, 0], lc[inside])]
    np.testing.assert_allclose(out[inside], closed, rtol=1e-12)


@pytest.mark.gpu
def test_white_kernel_through_the_facade():
    """GPModelling on a JitterTerm alone: fit, chains, prediction (mean = the constant, variance = the jitter)"""
    import warnings
    from mind_the_gaps_amd import terms
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    time, y = series(2)
    lc = GappyLightcurve(time, y, np.full(N, 1e-3))
    gpm = GPModelling(lc, terms.JitterTerm(log_sigma=np.log(3.0), bounds=[(-5.0, 8.0)]))
    assert abs(gpm._neg_log_like(np.array([PRINTED_X])) - PRINTED_FUN) < 1e-3       # dy = 1e-3 instead of 1.123e-12
    sol = gpm.fit()
    assert sol.success and abs(sol.x[0] - 0.5 * np.log(SAMPLE_VARIANCE)) < 1e-4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gpm.derive_posteriors(fit=True, converge=False, max_steps=300, walkers=12, progress=False)
    assert abs(gpm.max_parameters[0] - 0.5 * np.log(SAMPLE_VARIANCE)) < 0.01
    assert abs(-gpm.max_loglikelihood - PRINTED_FUN) < 0.01
    assert abs(np.std(gpm.mcmc_samples[:, 0]) - 1.0 / np.sqrt(2 * N)) < 0.01       # the posterior width of a variance: 1 / sqrt(2 N)


# ---- celerite_variance.ipynb cells 6-12 and 14-20: the reference's celerite posterior maxima on two reproducible data sets ----
# tests/golden/notebook_variance_data.npz holds the two simulated light curves of those cells, rebuilt from numpy's seeded
# global generator by tests/golden/make_notebook_data.py and verified against the notebook's own printed numbers (their
# variances to 8 digits, rechecked below).  The notebook then runs the reference's GPModelling.derive_posteriors (celerite +
# emcee) on them and prints `max_parameters`, the chain sample of largest posterior.  No likelihood VALUE is printed, so
# this pins the likelihood SURFACE: the sample celerite scored highest among N samples of a P-dimensional Gaussian-like
# posterior lies within ~(1/N)^(2/P) of the top in lnL; a likelihood that differed from celerite's in location or shape
# would put it visibly lower.  Measured with the oracle: 1.1e-4 below this build's maximum for the DRW (cell 12: converged
# after 4500 steps of 12 walkers), 2.0e-2 for the Lorentzian (cell 20: 50 000 steps without convergence, Q barely constrained).
W0 = 2 * np.pi / 100
NOTEBOOK_CASES = {
    # name: (rates key, sample variance the notebook printed, log S0 it printed, printed exp(log S0) / variance, printed max_parameters,
    #        allowed drop of lnL below the maximum, allowed distance of the maximum from the printed sample)
    "drw": ("cell6_rates", 0.97372, -0.02605619, 1.000578571036844, [-0.02605619, -2.90922302], 1e-3, 5e-3),
    "lorentzian": ("cell14_rates", 0.96105, -0.08662917, 0.9541861557182113, [-0.08662917, 1.68480537, -2.77819922], 0.1, 0.1),
}


def notebook_case(name):
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebook_variance_data.npz"))
    key, printed5, log_s, ratio, theta, drop, dist = NOTEBOOK_CASES[name]
    return data["times"], data[key], printed5, np.exp(log_s) / ratio, np.array(theta), drop, dist


@pytest.mark.parametrize("name", list(NOTEBOOK_CASES))
def test_fixture_is_the_notebooks_light_curve(name):
    times, rates, printed5, variance, theta, _, _ = notebook_case(name)
    assert len(times) == len(rates) == 5000 and "%.5f" % np.var(rates) == "%.5f" % printed5      # "Sample Variance: ..."
    assert abs(np.var(rates) - variance) < 6e-9                                                  # from "Ratio ampltiudes" and max_parameters[0]
    if name == "drw":                                                                            # "Ratio breaks" (cell 12)
        assert abs(np.exp(theta[1]) / W0 - 0.867682082322184) < 1e-8


@pytest.mark.parametrize("name", list(NOTEBOOK_CASES))
def test_oracle_puts_celerites_best_sample_at_the_top(name):
    from scipy.optimize import minimize
    from oracle import celerite as oc
    from oracle import dense
    times, rates, _, _, theta, drop, dist = notebook_case(name)
    kinds = [dense.K_DRW] if name == "drw" else [dense.K_LORENTZIAN]
    dy, mean = np.full(len(times), 1e-12), float(np.mean(rates))         # cell 8: dy = 1e-12 (+ the facade's 1e-12); mean frozen at the sample mean
    nll = lambda th: -oc.logprob_batch(times, rates, dy, kinds, np.append(th, mean))[0][0]
    top = minimize(nll, theta, method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-9, maxiter=3000))
    assert top.success and 0.0 <= nll(theta) - top.fun < drop
    assert np.max(np.abs(top.x - theta)) < dist


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(NOTEBOOK_CASES))
def test_hip_path_puts_celerites_best_sample_at_the_top(name):
    """cells 8 / 16 with this package's classes; the optimiser is the facade's own fit()"""
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk as DRW, Lorentzian
    times, rates, _, variance, theta, drop, dist = notebook_case(name)
    if name == "drw":
        kernel = DRW(log_S0=np.log(variance), log_omega0=np.log(W0), bounds=dict(log_S0=(-10, 10), log_omega0=(-10, 10)))
    else:
        kernel = Lorentzian(log_S0=np.log(variance), log_omega0=np.log(W0), log_Q=np.log(200),
                            bounds=dict(log_S0=(-10, 10), log_omega0=(-10, 1), log_Q=(np.log(1.5), np.log(5000))))
    gpmodel = GPModelling(GappyLightcurve(times, rates, dy=np.ones(len(rates)) * 1e-12), kernel)
    assert gpmodel.gp.parameter_names[-1] == "mean:value" and len(gpmodel.gp.get_parameter_vector()) == len(theta)
    at_printed = gpmodel._neg_log_like(theta)
    best = min((gpmodel.fit(start) for start in (theta, gpmodel.initial_params)), key=lambda s: s.fun)
    assert 0.0 <= at_printed - best.fun + 1e-6 and at_printed - best.fun < drop
    assert np.max(np.abs(best.x - theta)) < dist
    # and the value itself against the oracle at the printed sample
    from oracle import celerite as oc
    from oracle import dense
    kinds = [dense.K_DRW] if name == "drw" else [dense.K_LORENTZIAN]
    want = -oc.logprob_batch(times, rates, np.full(len(times), 1e-12), kinds, np.append(theta, np.mean(rates)))[0][0]
    assert abs(at_printed - want) <= 1e-9 * abs(want)


# ---- poisson_level.ipynb cells 2-6: a 1 728 002-point series, rebuilt to the last printed digit -------------------------------
POISSON_LOG_VARIANCE = 0.017072777961537826      # cell 4 prints the kernel built from log(np.var(lc.countrate))
POISSON_SOLUTION = [np.log(0.9657523627905847), np.log(1.0372544336869347) - 13.0, -0.69427256]   # cell 6: "Ratio ..." lines, solution.x


def _poisson_level_series():
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_notebook_data.py")
    spec = importlib.util.spec_from_file_location("make_notebook_data", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.poisson_level_series()


def test_poisson_level_series_is_rebuilt_to_the_last_printed_digit():
    """numpy alone (tests/golden/make_notebook_data.py): the restated pipeline IS the one the notebook ran"""
    grid, rate, noisy = _poisson_level_series()
    assert len(grid) == 1728002 and "%.5f" % np.var(rate) == "1.01722"           # "LC variance: 1.01722"
    assert abs(np.log(np.var(rate)) - POISSON_LOG_VARIANCE) < 1e-14


@pytest.mark.gpu
def test_poisson_level_notebook_on_the_device():
    """cells 2-6 with this package: (a) the device simulator, fed numpy's draws (stream="numpy"), returns the notebook's series
    -- its log variance is the 16 digits the notebook printed; (b) celerite.GP + minimize as the notebook drives them: the
    L-BFGS-B solution celerite reached (printed) scores 89.83 better than the starting point the notebook printed and lies
    within 3 of this build's maximum -- 2e-6 of lnL = -1.33e6 (finite-difference gradients are noise at this size: neither
    optimiser run, the notebook's or this one, ends at the top; the oracle says the same: 89.829 and 2.54)."""
    from scipy.optimize import minimize
    from mind_the_gaps_amd import terms
    from mind_the_gaps_amd.gp import GP
    from mind_the_gaps_amd.models import DampedRandomWalk as DRW
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw as BPL
    from mind_the_gaps_amd.simulator import Simulator
    np.random.seed(42)
    times = np.linspace(0, 1000, 1000) * 3600 * 24
    simulator = Simulator(BPL(S0=1.0, omega0=np.exp(-13)), times, 1000 * np.ones(1000), mean=0, pdf="Gaussian", extension_factor=10,
                          aliasing_factor=2, stream="numpy")
    lc = simulator.simulate_regularly_sampled()
    assert lc.n == 1728002 and abs(np.log(np.var(lc.countrate)) - POISSON_LOG_VARIANCE) < 1e-11
    signoise = 0.5
    y = lc.countrate + np.random.normal(0, signoise, size=lc.n)                    # cell 4
    grid, rate, noisy = _poisson_level_series()
    assert np.max(np.abs(lc.countrate - rate)) < 1e-8 and np.max(np.abs(y - noisy)) < 1e-8
    time = lc.time
    w0 = 2 * np.pi / (30 * 86400.0)
    kernel = DRW(log_S0=np.log(np.var(lc.countrate)), log_omega0=np.log(w0), bounds=dict(log_S0=(-30, 15), log_omega0=(-25, -1))) \
        + terms.JitterTerm(log_sigma=np.log(signoise), bounds=dict(log_sigma=(-10, 20)))
    assert abs(kernel.get_parameter_vector()[1] - (-12.930063270044956)) < 1e-12   # the kernel cell 4 prints
    gp = GP(kernel, mean=np.mean(y), fit_mean=False, fit_white_noise=False)
    gp.compute(time, yerr=1e-12)
    assert gp.parameter_names == ("kernel:terms[0]:log_S0", "kernel:terms[0]:log_omega0", "kernel:terms[1]:log_sigma", "mean:value")

    def neg_log_like(params, y, gp):
        gp.set_parameter_vector(params)
        return -gp.log_likelihood(y)
    initial_params = gp.get_parameter_vector()
    at_start, at_printed = neg_log_like(initial_params, y, gp), neg_log_like(np.array(POISSON_SOLUTION), y, gp)
    assert abs(at_start - 1334134.906415) < 1e-3 and abs(at_printed - 1334045.077138) < 1e-3      # the oracle's values
    top = minimize(neg_log_like, np.array(POISSON_SOLUTION), method="Nelder-Mead", args=(y, gp), options=dict(xatol=1e-6, fatol=1e-4, maxiter=400))
    assert 0.0 < at_printed - top.fun < 3.0 and np.max(np.abs(top.x - POISSON_SOLUTION)) < 0.1     # (a shallow valley along log S0 - log omega0)
    solution = minimize(neg_log_like, initial_params, method="L-BFGS-B", bounds=gp.get_parameter_bounds(), args=(y, gp))
    assert solution.fun < at_start                                                                 # the notebook's call runs


# ---- docs/notebooks/tutorial_ppp.ipynb, cells 0-15: the one J > 0 likelihood RATIO the reference's material prints -------------
# Executions 14 -> 21 of that notebook are sequential after `np.random.seed(10)` (cell 0) and print "Observed LRT_stat: 5.257"
# (cell 15): T_obs = -2 (max lnL_null - max lnL_alt) of a RealTerm (J = 1) against ComplexTerm + RealTerm (J = 3) on a
# 1000-point light curve, each maximum taken over an emcee chain of 12 walkers that celerite evaluated.  The light curve
# is reproducible from numpy alone (the reference's simulator and its Poisson noise draw from numpy's global generator);
# the chains are not a replay (the author's scipy, Pool and package version differ: convergence after 4500 / 29 000
# iterations there, 9000 / 31 500 here), so this is a CONSISTENCY check, not a pin: a 0.3 spread in T is 1e-4 of lnL.
# What it does establish, on the CPU with the oracle: (i) the same mode, the same size of T; (ii) celerite's chain maximum
# cannot beat the likelihood's maximum, so this build's TRUE maxima must give T >= 5.257 (up to the print's rounding).
# `tutorial_model_selection.ipynb` (np.random.seed at execution 1, the simulating cell at execution 14) is NOT
# reproducible; no other celerite J > 0 number exists in /root/reference: the parity cap on the recurrence is final.
PPP_PRINTED_T_OBS = 5.257


class _OracleGP:
    """gp.GP's batch entry point answered by oracle/celerite_ref.c (tests only: the product never does this)."""

    def __init__(self, real, t, dy):
        self._real, self._t, self._dy = real, np.asarray(t, dtype=np.float64), np.asarray(dy, dtype=np.float64)

    def __getattr__(self, name):
        return getattr(self._real, name)

    def log_probability_batch(self, theta, y, add_prior=True):
        from oracle import celerite as oc
        m = self._real._device_model()
        theta = np.atleast_2d(theta)
        full = np.tile(m.full, (len(theta), 1))
        full[:, m.free_index] = theta
        y = np.asarray(y, dtype=np.float64) - (m.y_offset or 0.0)
        return oc.logprob_batch(self._t, y, self._dy, m.kinds, full, bounds=m.bounds, extra=m.extra, add_prior=add_prior,
                                nthreads=4, fused=True)


def _tutorial_ppp_lightcurve():
    """cells 0-5: np.random.seed(10); Simulator(BendingPowerlaw(100, 2 pi / 20), arange(1000), exposures 1, mean 100,
    extension_factor=2).generate_lightcurve(); add_noise (PoissonNoise: noise_models.py:49-78)"""
    import importlib.util
    import os
    from mind_the_gaps_amd.models.psd_models import BendingPowerlaw
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_notebook_data", os.path.join(here, "golden", "make_notebook_data.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    times = np.arange(0, 1000).astype(np.float64)
    exposures = np.ones(len(times))
    np.random.seed(10)
    rates = mk.reference_lightcurve(BendingPowerlaw(100.0, 2 * np.pi / 20), times, exposures, 100, 2)
    counts = np.random.poisson(rates * exposures)
    return times, counts / exposures, np.sqrt(counts) / exposures


@pytest.mark.timeout(900)
def test_tutorial_ppp_observed_lrt_is_consistent_with_the_notebooks_print():
    import warnings
    from scipy.optimize import minimize
    from mind_the_gaps_amd import terms
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    times, y, dy = _tutorial_ppp_lightcurve()
    assert 95 < y.mean() < 105 and 8 < y.std() < 20          # mean 100, DRW variance 100 + Poisson variance 100
    lc = GappyLightcurve(times, y, dy, exposures=1.0)
    variance_drw, w_bend, w = 100.0, 2 * np.pi / 20, 2 * np.pi / 10
    bounds_drw = dict(log_a=(-10, 50), log_c=(-10, 10))
    bounds_qpo = dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5))

    def null_kernel():
        return terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend), bounds=bounds_drw)

    def alt_kernel():
        return terms.ComplexTerm(log_a=np.log(variance_drw), log_c=np.log(0.5 * w / 80), log_d=np.log(w), bounds=bounds_qpo) + null_kernel()

    best, chain_max = {}, {}
    for name, make in (("null", null_kernel), ("alt", alt_kernel)):     # cells 7 and 9, in the notebook's order
        g = GPModelling(lc, make())
        g.gp = _OracleGP(g.gp, times, dy)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g.derive_posteriors(max_steps=50000, fit=True, progress=False, device_sampler=False)
        chain_max[name] = g.max_loglikelihood
        # this build's TRUE maximum: polish the chain's best sample (and the fit's) with L-BFGS-B on the oracle
        cands = [g.max_parameters, g.fit(g.initial_params).x]
        tops = []
        for x0 in cands:
            sol = minimize(lambda x: g._neg_log_like_and_grad(x, *(np.array(b, dtype=float) for b in zip(*g.gp.get_parameter_bounds())))[0],
                           x0, method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-9, maxiter=4000))
            tops.append(-sol.fun)
        best[name] = max(tops + [chain_max[name]])
    T_chain = -2 * (chain_max["null"] - chain_max["alt"])
    T_true = -2 * (best["null"] - best["alt"])
    # (i) same mode, same size: this build's replay of the notebook's recipe lands around the printed value
    print("tutorial_ppp replay: T over the chains %.3f, T at the polished maxima %.3f (notebook: 5.257)" % (T_chain, T_true))
    assert 4.8 < T_chain < 6.3, (T_chain, T_true)
    # (ii) a chain's maximum is below the likelihood's: max lnL_alt gains more from polishing than max lnL_null (5 against 2
    # parameters), and celerite's printed T cannot exceed the T of the true maxima by more than the null chain's own shortfall
    assert best["null"] >= chain_max["null"] and best["alt"] >= chain_max["alt"]
    assert T_true >= PPP_PRINTED_T_OBS - 1e-2, (T_true, T_chain)
    assert T_true < 7.5
