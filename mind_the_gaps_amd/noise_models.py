"""The reference's noise models by name (mind_the_gaps/noise_models.py:14-184): ``PoissonNoise``, ``KraftNoise``,
``GaussianNoise`` with the reference's constructors, ``name`` and ``add_noise(rates) -> (noisy rates, uncertainties)``, for one
light curve on the host and -- as in the reference -- from numpy's GLOBAL generator (noise_models.py:71,182), so
``np.random.seed`` fixes them.  The batched Protassov path adds Gaussian and Poisson noise on the device instead
(``mtg_simulate_tk95``, counter-based streams); ``Simulator.add_noise`` is the same arithmetic.
"""
import numpy as np

__all__ = ["BaseNoise", "PoissonNoise", "KraftNoise", "GaussianNoise"]


class BaseNoise:
    def __init__(self, name):
        self.name = name

    def add_noise(self, rates):
        raise NotImplementedError("This method should be implemented by subclasses")


class PoissonNoise(BaseNoise):
    """Counting noise: total counts ~ Poisson(rate x exposure + background), background subtracted again
    (noise_models.py:29-78)."""

    def __init__(self, exposures, background_counts=None, bkg_rate_err=None):
        super().__init__(name="Poisson")
        self.exposures = exposures
        n = len(exposures)
        self.background_counts = np.zeros(n, dtype=int) if background_counts is None else background_counts
        self.bkg_rate_err = np.zeros(n) if bkg_rate_err is None else bkg_rate_err

    def add_noise(self, rates):
        observed = np.random.poisson(rates * self.exposures + self.background_counts)
        dy = np.sqrt((np.sqrt(observed) / self.exposures) ** 2 + self.bkg_rate_err ** 2)
        return (observed - self.background_counts) / self.exposures, dy


class KraftNoise(PoissonNoise):
    """Poisson noise with the Bayesian treatment of Kraft, Burrows & Nousek (1991) for epochs with fewer than
    ``kraft_counts`` total counts: net counts = the posterior's median, uncertainty = half its 68 % interval
    (noise_models.py:81-150; the reference takes both from scipy / astropy numerically, here they are solved directly,
    simulator.kraft_median / kraft_interval)."""

    def __init__(self, exposures, background_counts=None, bkg_rate_err=None, kraft_counts=15):
        super().__init__(exposures, background_counts, bkg_rate_err)
        self.name = "Kraft"
        self.kraft_counts = kraft_counts

    def add_noise(self, rates):
        from .simulator import kraft_interval, kraft_median
        net_rates, dy = super().add_noise(rates)
        exposures = np.broadcast_to(self.exposures, net_rates.shape)
        background = np.broadcast_to(self.background_counts, net_rates.shape)
        total = net_rates * exposures + background
        for i in np.flatnonzero(total < self.kraft_counts):
            counts = int(round(total[i]))
            net_rates[i] = kraft_median(counts, background[i]) / exposures[i]
            lower, upper = kraft_interval(int(total[i]), background[i], 0.68)
            dy[i] = (upper - lower) / 2.0 / exposures[i]
        return net_rates, dy


class GaussianNoise(BaseNoise):
    """White Gaussian noise of standard deviation ``sigma_noise`` (noise_models.py:152-184)."""

    def __init__(self, exposures, sigma_noise):
        super().__init__(name="Gaussian")
        self.sigma_noise = sigma_noise

    def add_noise(self, rates):
        return rates + np.random.normal(scale=self.sigma_noise, size=len(rates)), self.sigma_noise * np.ones(len(rates))
