"""CPU tests of the lock-step (many light curves) driver with analytic targets."""
import numpy as np
import pytest

from mind_the_gaps_amd.ppp import EnsembleBatchSampler, batched_minimize


def test_lockstep_sampler_recovers_per_lightcurve_gaussians():
    L, W, P = 6, 16, 2
    mus = np.arange(L)[:, None] * np.array([1.0, -2.0])          # a different mean per light curve
    sig = 0.5 + 0.25 * np.arange(L)
    calls = []

    def logp(x, lc):
        calls.append(len(x))
        return -0.5 * np.sum(((x - mus[lc]) / sig[lc, None]) ** 2, axis=1)

    s = EnsembleBatchSampler(L, W, P, logp, seed=3)
    rng = np.random.default_rng(0)
    s.run(mus[:, None, :] + 0.1 * rng.standard_normal((L, W, P)), 1500)
    assert calls[0] == L * W and set(calls[1:]) == {L * W // 2} and len(calls) == 1 + 2 * 1500
    chain = s.get_chain()
    assert chain.shape == (1500, L, W, P) and s.get_log_prob(2).shape == (1500, W)
    for l in range(L):
        flat = chain[300:, l].reshape(-1, P)
        assert np.allclose(flat.mean(axis=0), mus[l], atol=0.15 * sig[l] + 0.05)
        assert np.allclose(flat.std(axis=0), sig[l], rtol=0.15)
    assert np.all(s.best_lnp <= 0) and np.all(s.best_lnp > -0.1)
    assert np.allclose(s.best_coords, mus, atol=0.5)
    tau = s.get_autocorr_time()
    assert tau.shape == (L, P) and np.all(tau > 1)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    # light curves never mix: the stored log-probs are those of their own target
    k = 4
    assert np.allclose(s.get_log_prob(k)[-1], logp(chain[-1, k], np.full(W, k)))


def test_lockstep_sampler_is_reproducible_and_checks_input():
    def logp(x, lc):
        return -0.5 * np.sum(x ** 2, axis=1)
    a = EnsembleBatchSampler(3, 8, 2, logp, seed=5, store_chain=False)
    b = EnsembleBatchSampler(3, 8, 2, logp, seed=5, store_chain=False)
    p0 = np.random.default_rng(1).standard_normal((3, 8, 2))
    ca, _ = a.run(p0, 50)
    cb, _ = b.run(p0, 50)
    assert np.array_equal(ca, cb) and np.array_equal(a.best_lnp, b.best_lnp)
    with pytest.raises(RuntimeError):
        EnsembleBatchSampler(3, 2, 2, logp)
    with pytest.raises(ValueError):
        EnsembleBatchSampler(3, 7, 2, logp)
    with pytest.raises(ValueError):
        a.run(np.zeros((2, 8, 2)), 1)
    bad = p0.copy(); bad[0, 0, 0] = np.nan
    with pytest.raises(ValueError):
        EnsembleBatchSampler(3, 8, 2, logp).run(bad, 1)


def test_batched_minimize_box_constrained_quadratics():
    L, P = 7, 3
    rng = np.random.default_rng(2)
    centers = rng.uniform(-2, 2, (L, P))
    scales = rng.uniform(0.5, 3.0, (L, P))
    lower, upper = np.array([-1.0, -np.inf, -1.5]), np.array([1.0, np.inf, 0.5])

    def fun(x, lc):
        return np.sum(scales[lc] * (x - centers[lc]) ** 2, axis=1) + 3.0

    x, f, it = batched_minimize(fun, np.zeros((L, P)), lower, upper)
    want = np.clip(centers, lower, upper)                      # separable: the box solution is the clip
    assert np.allclose(x, want, atol=2e-4)
    assert np.allclose(f, fun(want, np.arange(L)), atol=1e-6) and it < 60


def test_batched_minimize_keeps_the_last_good_point_when_the_gradient_batch_rejects_a_step():
    """The line search (L rows) and the gradient batch (L (P + 1) rows) may run on different solver families on the
    device; a point accepted by the first and found unfactorisable by the second must not be adopted."""
    L, P = 5, 2
    centers = np.arange(L * P, dtype=float).reshape(L, P) / 5.0 + 0.3

    def fun(x, lc):
        f = np.sum((x - centers[lc]) ** 2, axis=1) + 1.0
        if len(x) == L * (P + 1):                                # the gradient batch: problem 3 fails away from 0
            f = np.where((lc == 3) & np.any(np.abs(x) > 1e-3, axis=1), np.inf, f)
        return f

    x, f, _ = batched_minimize(fun, np.zeros((L, P)), np.full(P, -5.0), np.full(P, 5.0))
    assert np.all(np.isfinite(f)) and np.all(np.isfinite(x))
    assert np.array_equal(x[3], np.zeros(P)) and np.isclose(f[3], 1.0 + np.sum(centers[3] ** 2))
    keep = np.arange(L) != 3
    assert np.allclose(x[keep], centers[keep], atol=2e-4)
