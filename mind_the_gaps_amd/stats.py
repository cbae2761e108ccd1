"""Model-selection statistics fed by the hot path's log-likelihoods.

Mirrors /root/reference/mind_the_gaps/stats.py:155-195 (``bic``, ``aic``, ``aicc``: same
names, argument order and formulas) and the likelihood-ratio post-processing of the
Protassov test, docs/notebooks/tutorial_ppp.ipynb:406-411:
``T = -2 (lnL_null - lnL_alt)`` for the observed data against its distribution over the
simulated light curves.  Pure host arithmetic on a few numbers per light curve.
"""
import numpy as np

__all__ = ["bic", "aic", "aicc", "lrt_statistic", "lrt_pvalue"]


def bic(loglikehood, n, k):
    """Bayesian Information Criterion for ``n`` data points and ``k`` parameters."""
    return -2.0 * loglikehood + k * np.log(n)


def aic(loglikehood, k):
    """Akaike Information Criterion."""
    return 2 * k - 2 * loglikehood


def aicc(loglikehood, n, k):
    """AIC corrected for finite sample size."""
    return aic(loglikehood, k) + 2 * k * (k + 1) / (n - k - 1)


def lrt_statistic(loglike_null, loglike_alt):
    """T_LRT = -2 (ln L_null - ln L_alt); arrays broadcast (one value per light curve)."""
    return -2.0 * (np.asarray(loglike_null, dtype=np.float64) - np.asarray(loglike_alt, dtype=np.float64))


def lrt_pvalue(t_observed, t_simulated):
    """Posterior-predictive p-value: fraction of simulated statistics >= the observed one
    (Protassov et al. 2002), with the +1 correction so that it is never exactly 0."""
    t_sim = np.asarray(t_simulated, dtype=np.float64).ravel()
    return (1.0 + np.count_nonzero(t_sim >= t_observed)) / (1.0 + t_sim.size)
