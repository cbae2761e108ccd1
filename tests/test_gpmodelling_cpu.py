"""CPU tests of GPModelling's host logic.  The four spread_walkers tests restate
/root/reference/tests/gpmodelling_test.py:8-114 against this package."""
import unittest
import warnings

import numpy as np
import pytest

from mind_the_gaps_amd import engine
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import ExposureTimeError, GappyLightcurve
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian


def make(bounds_drw, bounds_lor):
    drw_params, lor_params = [5.0, 10.0], [10, 5, -5]
    kernel = DampedRandomWalk(drw_params[0], drw_params[1], bounds=bounds_drw) + \
        Lorentzian(lor_params[0], lor_params[1], lor_params[2], bounds=bounds_lor)
    lc = GappyLightcurve(np.arange(100), np.arange(100), np.arange(100))
    return GPModelling(lc, kernel), drw_params + lor_params, bounds_drw + bounds_lor


class TestGPModelling(unittest.TestCase):
    def test_parameters_within_bounds(self):
        gpmodel, parameters, bounds = make([(4.0, 6.0), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for percent, attempts in ((0.1, 100), (0.9, 2)):
                samples = gpmodel.spread_walkers(100, parameters, bounds, percent=percent, max_attempts=attempts)
                for i, sample in enumerate(samples.T):
                    self.assertTrue(np.all(np.logical_and(bounds[i][0] <= sample, sample <= bounds[i][1])))

    def test_infinite_bounds(self):
        gpmodel, parameters, bounds = make([(None, None), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
        samples = gpmodel.spread_walkers(100, parameters, bounds, percent=0.1, max_attempts=50)
        self.assertTrue(np.all(np.isfinite(samples[:, 0])))
        for bounds_i, sample in zip(bounds[1:], samples.T[1:]):
            self.assertTrue(np.all(np.logical_and(bounds_i[0] <= sample, sample <= bounds_i[1])))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            samples = gpmodel.spread_walkers(100, parameters, bounds, percent=0.99, max_attempts=5)
        self.assertTrue(np.all(np.isfinite(samples[:, 0])))

    def test_zero_percent(self):
        gpmodel, parameters, bounds = make([(None, None), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
        samples = gpmodel.spread_walkers(100, parameters, bounds, percent=0, max_attempts=50)
        np.testing.assert_array_equal(samples, np.array([parameters] * 100))

    def test_max_attempts(self):
        p = [5.0, 10.0, 10, 5, -5]
        gpmodel, parameters, bounds = make([(p[0] - 0.01, p[0] + 0.01), (p[1] - 0.01, p[1] + 0.01)],
                                           [(v - 0.01, v + 0.01) for v in p[2:]])
        samples = gpmodel.spread_walkers(100, parameters, bounds, percent=0, max_attempts=50)
        for i, sample in enumerate(samples.T):
            self.assertTrue(np.all(sample == parameters[i]))


def test_percent_out_of_range_raises():
    gpmodel, parameters, bounds = make([(4.0, 6.0), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
    with pytest.raises(ValueError):
        gpmodel.spread_walkers(10, parameters, bounds, percent=1.5)


def test_clamping_rule():
    """Walkers still outside after max_attempts go to bound*1.05 / bound*0.95 by the bound's sign
    (gpmodelling.py:327-328,346-349)."""
    gpmodel, _, _ = make([(4.0, 6.0), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
    np.random.seed(3)
    bounds = [(100.0, 200.0), (-50.0, -40.0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = gpmodel.spread_walkers(4, np.array([1.0, 1.0]), bounds, percent=0.1, max_attempts=3)
    assert np.allclose(s[:, 0], 105.0) and np.allclose(s[:, 1], -40.0 * 1.05)


def test_constructor_state_and_mean_models():
    lc = GappyLightcurve(np.arange(50.0), 10 + np.sin(np.arange(50.0)), np.full(50, 0.1))
    k = DampedRandomWalk(1.0, -1.0, bounds=[(-10, 50), (-10, 10)])
    g = GPModelling(lc, k)
    assert g.k == 2 and g.parameter_names == ("kernel:log_S0", "kernel:log_omega0")
    assert np.array_equal(g.initial_params, [1.0, -1.0]) and g.autocorr == []
    assert g.gp.mean.value == pytest.approx(np.mean(lc.y))
    assert g.gp.mean.get_parameter_bounds(include_frozen=True) == [(np.min(lc.y), np.max(lc.y))]
    gc = GPModelling(lc, DampedRandomWalk(1.0, -1.0), mean_model="constant")
    assert gc.k == 3 and gc.parameter_names[-1] == "mean:value"
    gl = GPModelling(lc, DampedRandomWalk(1.0, -1.0), mean_model="Linear")
    assert gl.k == 4 and gl.parameter_names[-2:] == ("mean:slope", "mean:intercept")
    assert list(gl.initial_params[-2:]) == [0.0, 1.5]
    with pytest.raises(ValueError):
        GPModelling(lc, DampedRandomWalk(1.0, -1.0), mean_model="quadratic")
    with pytest.raises(ValueError):                      # reference quirk, SURVEY Appendix C.2
        GPModelling(lc, DampedRandomWalk(1.0, -1.0), mean_model="gaussian")
    for prop in ("loglikelihoods", "mcmc_samples", "max_loglikelihood", "max_parameters",
                 "median_parameters", "tau", "sampler"):
        with pytest.raises(AttributeError):
            getattr(g, prop)
    with pytest.raises(ValueError):
        g.get_rstat()
    with pytest.raises(RuntimeError):
        g.generate_from_posteriors(3)


def test_unsorted_times_rejected():
    with pytest.raises(ValueError):
        GPModelling(GappyLightcurve(np.array([0.0, 2.0, 1.0]), np.zeros(3), np.ones(3)), DampedRandomWalk(1.0, -1.0))


def test_lightcurve_container():
    lc = GappyLightcurve(np.array([0.0, 1.0, 3.0]), np.array([1.0, 2.0, 3.0]), np.ones(3), exposures=0.5)
    assert lc.n == 3 and lc.duration == 3.0 and lc.mean == 2.0 and np.all(lc.exposures == 0.5)
    with pytest.raises(ExposureTimeError):
        GappyLightcurve(np.array([0.0, 0.1]), np.zeros(2), np.ones(2), exposures=1.0)


@pytest.mark.skipif(engine.device_count() > 0, reason="a GPU is present")
def test_likelihood_fails_loudly_without_gpu():
    """No CPU fallback: evaluating on a box without an MI355X raises."""
    lc = GappyLightcurve(np.arange(20.0), np.zeros(20), np.ones(20))
    g = GPModelling(lc, DampedRandomWalk(1.0, -1.0))
    with pytest.raises(engine.EngineUnavailable):
        g._log_probability(g.initial_params)


class _StubGP:
    """Stands in for gp.GP inside GPModelling: log_probability_batch from a table of statuses."""
    def __init__(self, real, statuses):
        self._real, self._statuses, self.calls = real, list(statuses), 0

    def __getattr__(self, name):
        return getattr(self._real, name)

    def log_probability_batch(self, theta, y, add_prior=True):
        theta = np.atleast_2d(theta)
        st = self._statuses[min(self.calls, len(self._statuses) - 1)]
        self.calls += 1
        out = -np.sum((theta - 1.0) ** 2, axis=1) - 10.0
        status = np.full(len(theta), st, dtype=np.int32)
        return np.where(status == 0, out, -np.inf), status


def test_fit_raises_when_the_starting_point_cannot_be_factorised():
    """gpmodelling.py:192: celerite's LinAlgError aborts the reference's fit; a flat plateau would make
    L-BFGS-B report success at the starting point instead."""
    gpmodel, parameters, bounds = make([(4.0, 6.0), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
    gpmodel.gp = _StubGP(gpmodel.gp, [engine.ST_NOTPD])
    from mind_the_gaps_amd.gp import LinAlgError
    with pytest.raises(LinAlgError):
        gpmodel.fit()
    # quiet=True: no exception, but the result says it did not succeed
    gpmodel._quiet = True
    gpmodel.gp = _StubGP(gpmodel.gp._real, [engine.ST_NOTPD])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        sol = gpmodel.fit()
    assert not sol.success and any("factorised" in str(w.message) for w in caught)
    # a bad point LATER in the line search is just a bad point
    gpmodel._quiet = False
    gpmodel.gp = _StubGP(gpmodel.gp._real, [engine.ST_OK, engine.ST_NOTPD, engine.ST_OK])
    sol = gpmodel.fit()
    assert np.isfinite(sol.fun)


def test_best_loglikelihood_is_the_maximum_over_the_whole_chain():
    gpmodel, _, _ = make([(4.0, 6.0), (8.0, 12.0)], [(5, 15), (1, 6), (-7, -1)])
    with pytest.raises(AttributeError):
        gpmodel.best_loglikelihood

    class _Sampler:
        state = {"best_log_prob": np.array([-1.5])}

        def get_log_prob(self, flat=False, discard=0, thin=1):
            return np.array([-9.0, -2.0, -5.0, -7.0])[discard::thin]

    gpmodel._sampler = _Sampler()
    gpmodel._loglikelihoods = np.array([-5.0, -7.0])      # what survived burn-in and thinning
    gpmodel._mcmc_samples = np.zeros((2, 5))
    assert gpmodel.max_loglikelihood == -5.0 and gpmodel.best_loglikelihood == -1.5
    del _Sampler.state
    assert gpmodel.best_loglikelihood == -2.0
