"""Model-selection statistics fed by the hot path's log-likelihoods.

Mirrors /root/reference/mind_the_gaps/stats.py:155-195 (``bic``, ``aic``, ``aicc``: same
names, argument order and formulas) and the likelihood-ratio post-processing of the
Protassov test, docs/notebooks/tutorial_ppp.ipynb:406-411:
``T = -2 (lnL_null - lnL_alt)`` for the observed data against its distribution over the
simulated light curves.  Pure host arithmetic on a few numbers per light curve.
"""
import numpy as np
from scipy import special, stats as _scipy_stats

__all__ = ["bic", "aic", "aicc", "lrt_statistic", "lrt_pvalue", "lrt_pvalue_percentile", "kraft_pdf", "lognormal",
           "create_log_normal", "create_uniform_distribution", "neg_log_like"]


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


def lrt_pvalue_percentile(t_observed, t_simulated):
    """The tutorial's own expression (docs/notebooks/tutorial_ppp.ipynb cell 15):
    ``1 - scipy.stats.percentileofscore(T_dist, T_obs) / 100``."""
    return 1.0 - _scipy_stats.percentileofscore(np.asarray(t_simulated, dtype=np.float64).ravel(), t_observed) / 100.0


# ---- the distributions the simulator's flux PDFs and the Kraft noise model are built from (reference stats.py:10-29,
# 116-146; same names, same parametrisation) ------------------------------------------------------------------------------
class kraft_pdf(_scipy_stats.rv_continuous):
    """Posterior of the source counts s >= 0 given N observed and B expected background counts (Kraft et al. 1991):
    f(s | N, B) = C e^-(s+B) (s+B)^N / N!,  1/C = sum_{n<=N} e^-B B^n / n!."""

    def _argcheck(self, N, B):
        return (N >= 0) and (B >= 0)

    def _pdf(self, x, N, B):
        n = np.arange(N + 1)
        norm = 1.0 / np.sum(np.exp(-B) * B ** n / special.factorial(n))
        return norm * np.exp(-x - B) * (x + B) ** N / special.factorial(N)


class lognormal(_scipy_stats.rv_continuous):
    """Log-normal density with ``center`` and ``sigma`` those of ln x."""

    def _argcheck(self, center, sigma):
        return sigma >= 0

    def _pdf(self, x, center, sigma):
        return np.exp(-(np.log(x) - center) ** 2 / (2.0 * sigma ** 2)) / (sigma * x * np.sqrt(2.0 * np.pi))


def create_log_normal(mean, std):
    """A frozen scipy log-normal with the given mean and standard deviation (of x, not of ln x)."""
    var = std ** 2
    mu = np.log(mean ** 2 / np.sqrt(var + mean ** 2))
    sigma = np.sqrt(np.log(var / mean ** 2 + 1.0))
    return _scipy_stats.lognorm(sigma, scale=np.exp(mu))


def create_uniform_distribution(mean, std):
    """A frozen scipy uniform distribution with the given mean and standard deviation."""
    upper = np.sqrt(3.0 * std ** 2) + mean
    lower = 2.0 * mean - upper
    return _scipy_stats.uniform(loc=lower, scale=upper - lower)


def neg_log_like(params, y, gp):
    """-ln L of ``gp`` at ``params`` (reference stats.py:149-152; what the notebooks hand to scipy's minimize)."""
    gp.set_parameter_vector(params)
    return -gp.log_likelihood(y)
