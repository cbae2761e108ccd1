"""GPU tests of the drop-in facade: GPModelling on the HIP engine against the oracle."""
import warnings

import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd import terms
from mind_the_gaps_amd.gp import GP, LinAlgError
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models import BendingPowerlaw, DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.sampler import EnsembleSampler
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

AMP, OTHER = (-10, 50), (-10, 10)


def alt_kernel():
    th = synth.truth(synth.ALT_MODEL)
    return (DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER])
            + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])
            + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER]))


def oracle_lnp(t, y, dy, kinds, theta, mean, bounds, add_prior=True):
    theta = np.atleast_2d(theta)
    full = np.hstack([theta, np.full((len(theta), 1), mean)])
    b = np.vstack([bounds, [[np.min(y), np.max(y)]]])
    return oracle_c.logprob_batch(t, y, dy, kinds, full, bounds=b, add_prior=add_prior, nthreads=4)[0]


def test_log_probability_single_and_batch():
    N = 800
    t, y, dy = synth.make_lightcurves(N, 1, seed=5)
    y, dy = y[0], dy[0]
    g = GPModelling(GappyLightcurve(t, y, dy), alt_kernel())
    theta = synth.draw_thetas(synth.ALT_MODEL, 40, seed=6)
    theta[3, 1] = 10.5                                     # outside the box -> -inf (gpmodelling.py:150-151)
    ref = oracle_lnp(t, y, dy, synth.ALT_MODEL, theta, np.mean(y), synth.bounds_for(synth.ALT_MODEL))
    batch = g._log_probability(theta)
    assert batch.shape == (40,) and np.isneginf(batch[3]) and np.isneginf(ref[3])
    ok = np.isfinite(ref)
    assert np.max(np.abs(batch[ok] - ref[ok]) / np.abs(ref[ok])) < 1e-8
    one = g._log_probability(theta[0])
    assert isinstance(one, float) and one == batch[0]
    assert g._log_probability(theta[3]) == -np.inf
    nll = g._neg_log_like(theta[3])                        # no prior on this path (gpmodelling.py:168-169)
    ref_nll = -oracle_lnp(t, y, dy, synth.ALT_MODEL, theta[3], np.mean(y),
                          synth.bounds_for(synth.ALT_MODEL), add_prior=False)[0]
    assert abs(nll - ref_nll) / abs(ref_nll) < 1e-8
    # the facade is pure in theta: the GP's own parameter vector is untouched
    assert np.array_equal(g.gp.get_parameter_vector(), g.initial_params)


def test_gp_log_likelihood_celerite_style():
    t, y, dy = synth.make_lightcurves(300, 1, seed=8)
    y, dy = y[0], dy[0]
    k = DampedRandomWalk(np.log(100.0), np.log(0.3)) + BendingPowerlaw(4.0, 1.0, -1.0)
    gp = GP(k, mean=float(np.mean(y)))
    gp.compute(t, dy + 1e-12)
    v = gp.get_parameter_vector()
    want = oracle_c.logprob_batch(t, y, dy, [synth.K_DRW, synth.K_BPL], np.append(v, np.mean(y)))[0][0]
    assert abs(gp.log_likelihood(y) - want) / abs(want) < 1e-8
    gp.set_parameter_vector(v + 0.1)
    want2 = oracle_c.logprob_batch(t, y, dy, [synth.K_DRW, synth.K_BPL], np.append(v + 0.1, np.mean(y)))[0][0]
    assert abs(gp.log_likelihood(y) - want2) / abs(want2) < 1e-8


def test_user_defined_term_goes_through_raw_coefficients():
    """A Python Term without a device tag (celerite_models.py-style override) is
    expanded on the host and evaluated by mtg_loglike_coeffs."""
    class MyDRW(terms.Term):
        parameter_names = ("log_S0", "log_omega0")

        def get_real_coefficients(self, params):
            return np.exp(params[0]), np.exp(params[1])

    t, y, dy = synth.make_lightcurves(400, 1, seed=9)
    y, dy = y[0], dy[0]
    th = synth.truth([synth.K_DRW])
    g_user = GPModelling(GappyLightcurve(t, y, dy), MyDRW(th[0], th[1], bounds=[AMP, OTHER]))
    g_dev = GPModelling(GappyLightcurve(t, y, dy), DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]))
    theta = synth.draw_thetas([synth.K_DRW], 10, seed=1)
    theta[2, 1] = 11.0
    a, b = g_user._log_probability(theta), g_dev._log_probability(theta)
    assert np.isneginf(a[2]) and np.isneginf(b[2])
    ok = np.isfinite(b)
    assert np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok])) < 1e-12
    assert abs(g_user.gp.log_likelihood(y) - g_dev.gp.log_likelihood(y)) < 1e-9 * abs(g_dev.gp.log_likelihood(y))


def test_not_positive_definite_raises_like_celerite():
    t = np.arange(20.0)
    lc = GappyLightcurve(t, np.zeros(20), np.full(20, 1e-3))
    with pytest.raises(ValueError):                                # celerite: "non-finite log prior value"
        terms.ComplexTerm(np.log(1.0), np.log(50.0), np.log(0.01), np.log(1.0))
    g = GPModelling(lc, terms.ComplexTerm(0.0, -5.0, 0.0, 0.0))
    bad = np.log([1.0, 50.0, 0.01, 1.0])                          # a c < b d: not a valid kernel
    assert g._log_probability(bad) == -np.inf                     # ComplexTerm's prior vetoes it
    with pytest.raises(LinAlgError):
        g._neg_log_like(bad)                                      # no prior: the solver fails loudly
    assert GPModelling(lc, terms.ComplexTerm(0.0, -5.0, 0.0, 0.0), quiet=True)._neg_log_like(bad) == np.inf


def test_fit_improves_and_matches_oracle_at_optimum():
    N = 600
    t, y, dy = synth.make_lightcurves(N, 1, seed=12)
    y, dy = y[0], dy[0]
    th = synth.truth(synth.NULL_MODEL)
    k = DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4],
                                                                            bounds=[AMP, OTHER, OTHER])
    g = GPModelling(GappyLightcurve(t, y, dy), k)
    f0 = g._neg_log_like(g.initial_params)
    sol = g.fit()
    assert sol.fun <= f0 and np.all(np.isfinite(sol.x))
    lo, hi = np.array(g.gp.get_parameter_bounds()).T
    assert np.all(sol.x >= lo) and np.all(sol.x <= hi)
    ref = -oracle_lnp(t, y, dy, synth.NULL_MODEL, sol.x, np.mean(y), synth.bounds_for(synth.NULL_MODEL),
                      add_prior=False)[0]
    assert abs(sol.fun - ref) / abs(ref) < 1e-8
    # gradient from the batched launch == the same forward differences on the oracle
    f, grad = g._neg_log_like_and_grad(sol.x, lo.astype(float), hi.astype(float), step=1e-6)
    pts = np.vstack([sol.x] + [sol.x + 1e-6 * np.eye(5)[i] for i in range(5)])
    rv = -oracle_lnp(t, y, dy, synth.NULL_MODEL, pts, np.mean(y), synth.bounds_for(synth.NULL_MODEL), False)
    assert np.allclose(grad, (rv[1:] - rv[0]) / 1e-6, atol=2e-3)


def test_derive_posteriors_matches_oracle_driven_chain():
    """Same sampler, same seed: the chain driven by the HIP likelihood equals the chain
    driven by the oracle likelihood (accept/reject decisions agree)."""
    N = 300
    t, y, dy = synth.make_lightcurves(N, 1, seed=21)
    y, dy = y[0], dy[0]
    th = synth.truth([synth.K_DRW])
    g = GPModelling(GappyLightcurve(t, y, dy), DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]))
    np.random.seed(123)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g.derive_posteriors(fit=False, max_steps=120, convergence_steps=60, walkers=12, cores=1, progress=False,
                            device_sampler=False)
    assert g.sampler.iteration == 120 and len(g.autocorr) == 2 and not g.converged
    assert g.mcmc_samples.shape[1] == 2 and len(g.loglikelihoods) == len(g.mcmc_samples)
    assert g.max_loglikelihood == np.max(g.loglikelihoods)
    assert np.array_equal(g.max_parameters, g.mcmc_samples[np.argmax(g.loglikelihoods)])
    assert g.median_parameters.shape == (2,) and g.tau.shape == (2,) and g.get_rstat().shape == (12, 2)

    np.random.seed(123)
    p0 = g.spread_walkers(12, g.initial_params, np.array(g.gp.get_parameter_bounds()))
    ref = EnsembleSampler(12, 2, lambda p: oracle_lnp(t, y, dy, [synth.K_DRW], p, np.mean(y),
                                                      synth.bounds_for([synth.K_DRW])))
    ref.run_mcmc(p0, 120)
    assert np.allclose(g.sampler.get_chain(), ref.get_chain(), rtol=0, atol=1e-12)
    assert np.allclose(g.sampler.get_log_prob(), ref.get_log_prob(), rtol=1e-10)


def test_config1_drw_n1000_32_walkers():
    """BASELINE configs[0]: single DRW, N = 1000, 32 walkers, fit() + a short chain."""
    t, y, dy = synth.make_lightcurves(1000, 1, seed=20250704)
    y, dy = y[0], dy[0]
    th = synth.truth([synth.K_DRW])
    g = GPModelling(GappyLightcurve(t, y, dy), DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]))
    np.random.seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g.derive_posteriors(fit=True, max_steps=200, convergence_steps=100, walkers=32, progress=False)
    best = g.max_parameters
    ref = oracle_lnp(t, y, dy, [synth.K_DRW], best, np.mean(y), synth.bounds_for([synth.K_DRW]))[0]
    assert abs(g.max_loglikelihood - ref) / abs(ref) < 1e-8
    assert g.max_loglikelihood >= -g.fit().fun - 0.5      # the chain stays at the mode found by the fit


def test_derive_posteriors_with_device_sampler():
    """Same workflow with walkers, random numbers and accept/reject resident on the GPU."""
    N = 300
    t, y, dy = synth.make_lightcurves(N, 1, seed=21)
    y, dy = y[0], dy[0]
    th = synth.truth([synth.K_DRW])
    g = GPModelling(GappyLightcurve(t, y, dy), DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]))
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g.derive_posteriors(fit=True, max_steps=250, convergence_steps=100, walkers=12, progress=False,
                            device_sampler=True)
    assert g.sampler.iteration == 250 and len(g.autocorr) == 2
    chain, lnp = g.sampler.get_chain(), g.sampler.get_log_prob()
    assert chain.shape == (250, 12, 2) and lnp.shape == (250, 12)
    # every stored log-probability is the oracle's value of the stored sample
    ref = oracle_lnp(t, y, dy, [synth.K_DRW], chain[-1], np.mean(y), synth.bounds_for([synth.K_DRW]))
    assert np.max(np.abs(lnp[-1] - ref) / np.abs(ref)) < 1e-9
    assert g.max_loglikelihood == np.max(g.loglikelihoods) and g.get_rstat().shape == (12, 2)
    assert 0.2 < g.sampler.acceptance_fraction.mean() < 0.95
    # reproducible from numpy's global seed, like the host sampler
    g2 = GPModelling(GappyLightcurve(t, y, dy), DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]))
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g2.derive_posteriors(fit=True, max_steps=250, convergence_steps=100, walkers=12, progress=False,
                             device_sampler=True)
    assert np.array_equal(g2.sampler.get_chain(), chain)
    # host and device samplers agree on the posterior (not on the trajectory)
    np.random.seed(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g2.derive_posteriors(fit=True, max_steps=250, convergence_steps=100, walkers=12, progress=False,
                             device_sampler=False)
    a, b = chain[100:].reshape(-1, 2), g2.sampler.get_chain()[100:].reshape(-1, 2)
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) < 3 * (a.std(axis=0) + b.std(axis=0)) / np.sqrt(40))


def test_predict_and_standarized_residuals():
    """GP.predict(return_var=True) / standarized_residuals (gpmodelling.py:353-370) against
    dense algebra: conditional mean and variance at the training times."""
    from oracle import dense
    N = 500
    t, y, dy = synth.make_lightcurves(N, 1, seed=17)
    y, dy = y[0], dy[0]
    kernel = alt_kernel() + terms.JitterTerm(np.log(0.8))
    g = GPModelling(GappyLightcurve(t, y, dy), kernel)
    theta = g.gp.get_parameter_vector()
    kinds = synth.ALT_MODEL + [synth.K_JITTER]
    co = dense.build_coeffs(kinds, theta)
    mu_ref, var_ref = dense.dense_predict(t, y, dy, co, 0, [np.mean(y)])
    mu, var = g.gp.predict(y, return_var=True, return_cov=False)
    assert np.max(np.abs(mu - mu_ref)) < 1e-8 * np.max(np.abs(mu_ref))
    assert np.max(np.abs(var - var_ref) / var_ref) < 1e-7
    res = g.standarized_residuals(include_noise=True)
    assert np.allclose(res, (y - mu_ref) / np.sqrt(var_ref + co[6]), rtol=1e-6, atol=1e-9)
    assert np.array_equal(g.gp.predict(y, return_cov=False, return_var=False), mu)
    # fitted linear mean and an over-damped SHO (two real terms)
    k2 = DampedRandomWalk(np.log(50.0), np.log(0.2)) + terms.SHOTerm(np.log(20.0), np.log(0.2), np.log(0.5))
    g2 = GPModelling(GappyLightcurve(t, y + 0.01 * t, dy), k2, mean_model="linear")
    v = g2.gp.get_parameter_vector()
    v[-2:] = [0.01, float(np.mean(y))]
    g2.gp.set_parameter_vector(v)
    mu2, var2 = g2.gp.predict(y + 0.01 * t, return_var=True, return_cov=False)
    co2 = dense.build_coeffs([synth.K_DRW, synth.K_SHO], v[:5])
    mu2_ref, var2_ref = dense.dense_predict(t, y + 0.01 * t, dy, co2, 1, v[-2:])
    assert np.max(np.abs(mu2 - mu2_ref)) < 1e-8 * np.max(np.abs(mu2_ref))
    assert np.max(np.abs(var2 - var2_ref) / var2_ref) < 1e-7
    # other times: see test_apply_inverse_and_predict_at_new_times
    mu_new, var_new = g.gp.predict(y, t=t[:2] + 0.25, return_var=True, return_cov=False)
    assert mu_new.shape == (2,) and np.all(var_new > 0)


def test_product_kernel_against_the_dense_definition():
    """celerite's ``k1 * k2`` (terms.TermProduct) has no device tag: host-side coefficients through
    mtg_loglike_coeffs; lnL against the dense covariance of k1(tau) k2(tau)."""
    from oracle import dense
    t, y, dy = synth.make_lightcurves(300, 1, seed=13)
    y, dy = y[0], dy[0]
    th = synth.truth(synth.NULL_MODEL)
    k1 = DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER])
    k2 = terms.SHOTerm(np.log(1.0), th[3], th[4], bounds=[AMP, OTHER, OTHER])
    kernel = k1 * k2 + DampedRandomWalk(np.log(20.0), np.log(0.05), bounds=[AMP, OTHER])
    g = GPModelling(GappyLightcurve(t, y, dy), kernel)
    theta = g.initial_params + 0.05 * np.random.default_rng(0).standard_normal((6, len(g.initial_params)))
    got = g._log_probability(theta)
    for row, val in zip(theta, got):
        kernel.set_parameter_vector(row)
        coeffs = tuple(kernel.coefficients) + (0.0,)
        want = dense.dense_loglike(t, y, dy, coeffs, 0, (float(np.mean(y)),))
        assert abs(val - want) <= 1e-8 * abs(want)


def test_apply_inverse_and_predict_at_new_times():
    """celerite.GP.apply_inverse and GP.predict(y, t, ...) at times other than the training ones
    (mean, variance, full covariance) against dense linear algebra on the same covariance."""
    N = 300
    t, y, dy = synth.make_lightcurves(N, 1, seed=17)
    y, dy = y[0], dy[0]
    th = synth.truth(synth.ALT_MODEL)
    kernel = (DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])
              + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER]) + terms.JitterTerm(np.log(0.7)))
    for mean_model in (None, "linear"):
        g = GPModelling(GappyLightcurve(t, y, dy), kernel, mean_model=mean_model)
        gp = g.gp
        K = kernel.get_value(t[:, None] - t[None, :]) + np.diag((dy + 1e-12) ** 2 + kernel.jitter)
        rng = np.random.default_rng(1)
        b1, b3 = rng.standard_normal(N), rng.standard_normal((N, 3))
        assert np.allclose(gp.apply_inverse(b1), np.linalg.solve(K, b1), rtol=1e-8, atol=1e-12)
        assert np.allclose(gp.apply_inverse(b3), np.linalg.solve(K, b3), rtol=1e-8, atol=1e-12)
        ts = np.sort(np.concatenate([rng.uniform(t[0] - 5, t[-1] + 5, 40), t[[3, 77]]]))   # between, beyond and ON samples
        resid = y - gp.mean.get_value(t)
        ks = kernel.get_value(ts[:, None] - t[None, :])
        want_mu = gp.mean.get_value(ts) + ks @ np.linalg.solve(K, resid)
        want_cov = kernel.get_value(ts[:, None] - ts[None, :]) - ks @ np.linalg.solve(K, ks.T)
        mu = gp.predict(y, ts, return_cov=False)
        mu_v, var = gp.predict(y, ts, return_var=True)
        mu_c, cov = gp.predict(y, ts)
        scale = np.max(np.abs(want_mu))
        for m in (mu, mu_v, mu_c):
            assert np.max(np.abs(m - want_mu)) < 1e-8 * scale
        assert np.max(np.abs(cov - want_cov)) < 1e-7 * np.max(np.abs(want_cov))
        assert np.max(np.abs(var - np.diag(want_cov))) < 1e-7 * np.max(np.abs(want_cov))
        # training times through the same route (dense N x N, celerite's default return_cov=True)
        mu_t, cov_t = gp.predict(y)
        mu_dev, var_dev = gp.predict(y, return_var=True)
        assert np.max(np.abs(mu_t - mu_dev)) < 1e-8 * scale
        assert np.max(np.abs(np.diag(cov_t) - var_dev)) < 1e-6 * np.max(np.abs(var_dev))
    with pytest.raises(ValueError):
        gp.apply_inverse(np.zeros(N + 1))
