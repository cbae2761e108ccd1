"""Closed-form power spectra of the stochastic processes, as light-weight callables.

API of /root/reference/mind_the_gaps/models/psd_models.py:7-85 (``SHO``, ``Lorentzian``,
``BendingPowerlaw``, ``Matern32``, ``Matern52``, ``Jitter``): ``model(omega)`` evaluates the
one-sided PSD at angular frequencies ``omega`` with celerite's normalisation, ``a + b`` adds two
spectra, parameters are attributes that can be reassigned.  The reference builds them with
astropy's ``custom_model``; astropy is not a dependency here, so these are plain classes with the
same constructor signatures (positional or keyword, same defaults).

What they are for in this package: they are what the reference's users hand to ``Simulator``
(``docs/notebooks/tutorial_ppp.ipynb``: ``Lorentzian(S0, Q, w0) + BendingPowerlaw(S0, w0)``).
The device simulator draws its spectrum from celerite coefficient columns, so every model that IS
the spectrum of a celerite term knows that term (``to_term``) and ``Simulator`` converts it;
``Matern52`` and ``Jitter`` have no such term and only evaluate.  The parity tests check the
values against the outputs of the reference's own functions (``tests/golden/psd_golden.npz``).
"""
from math import pi, sqrt

import numpy as np

__all__ = ["PSDModel", "SHO", "Lorentzian", "BendingPowerlaw", "Matern", "Matern32", "Matern52", "Jitter"]


class PSDModel:
    """One closed-form spectrum: ``param_names`` with defaults, ``evaluate(x, *params)``."""

    param_names = ()
    defaults = ()

    def __init__(self, *args, **kwargs):
        if len(args) > len(self.param_names):
            raise TypeError("%s takes at most %d parameters" % (type(self).__name__, len(self.param_names)))
        values = dict(zip(self.param_names, self.defaults))
        values.update(zip(self.param_names, args))
        for key, value in kwargs.items():
            if key not in values:
                raise TypeError("%s has no parameter %r" % (type(self).__name__, key))
            values[key] = value
        for key, value in values.items():
            setattr(self, key, float(value))

    @property
    def parameters(self):
        return np.array([getattr(self, n) for n in self.param_names], dtype=np.float64)

    def __call__(self, x):
        return self.evaluate(np.asarray(x, dtype=np.float64), *self.parameters)

    def __add__(self, other):
        if not isinstance(other, PSDModel):
            return NotImplemented
        return CompoundPSD(self._leaves() + other._leaves())

    def _leaves(self):
        return [self]

    def to_term(self, bounds=None):
        """The celerite term whose ``get_psd`` is this spectrum (log parameters), or ValueError."""
        raise ValueError("%s is not the spectrum of a celerite term" % type(self).__name__)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % (n, getattr(self, n)) for n in self.param_names))


class CompoundPSD(PSDModel):
    """Sum of spectra (what ``a + b`` of two astropy models is in the reference)."""

    def __init__(self, leaves):
        self.leaves = list(leaves)

    @property
    def parameters(self):
        return np.concatenate([leaf.parameters for leaf in self.leaves])

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        return sum(leaf(x) for leaf in self.leaves)

    def _leaves(self):
        return list(self.leaves)

    def to_term(self, bounds=None):
        term = self.leaves[0].to_term()
        for leaf in self.leaves[1:]:
            term = term + leaf.to_term()
        return term

    def __repr__(self):
        return " + ".join(repr(leaf) for leaf in self.leaves)


class SHO(PSDModel):
    """Stochastically driven damped harmonic oscillator, Foreman-Mackey+2017 eq. 20
    (psd_models.py:7-11); the spectrum of celerite's ``SHOTerm``."""

    param_names = ("S0", "Q", "omega0")
    defaults = (1, 10, 1)

    @staticmethod
    def evaluate(x, S0, Q, omega0):
        return sqrt(2 / pi) * S0 * omega0 ** 4 / ((x ** 2 - omega0 ** 2) ** 2 + (x ** 2) * (omega0 ** 2) / Q ** 2)

    def to_term(self, bounds=None):
        from ..terms import SHOTerm
        return SHOTerm(np.log(self.S0), np.log(self.Q), np.log(self.omega0))


class Lorentzian(PSDModel):
    """Lorentzian of variance S0, quality factor Q, centroid omega0, Foreman-Mackey+2017 eq. 11
    (psd_models.py:14-32); the spectrum of the ``Lorentzian`` celerite term."""

    param_names = ("S0", "Q", "omega0")
    defaults = (1, 10, 1)

    @staticmethod
    def evaluate(x, S0, Q, omega0):
        a = S0
        c = omega0 / 2 / Q
        return sqrt(1 / 2 / pi) * a / c * (1 / (1 + ((x - omega0) / c) ** 2) + 1 / (1 + ((x + omega0) / c) ** 2))

    def to_term(self, bounds=None):
        from .celerite_models import Lorentzian as LorentzianTerm
        return LorentzianTerm(np.log(self.S0), np.log(self.Q), np.log(self.omega0))


class BendingPowerlaw(PSDModel):
    """Spectrum of the damped random walk (psd_models.py:35-46): flat below the bend
    ``c = omega0 / 2Q``, slope -2 above."""

    param_names = ("S0", "omega0", "Q")
    defaults = (1, 1, 1 / 2)

    @staticmethod
    def evaluate(x, S0, omega0, Q):
        a = S0
        c = 0.5 * omega0 / Q
        return sqrt(2 / pi) * a / c * (1 / (1 + (x / c) ** 2))

    def to_term(self, bounds=None):
        from ..terms import RealTerm
        from .celerite_models import DampedRandomWalk
        if self.Q == 0.5:
            return DampedRandomWalk(np.log(self.S0), np.log(self.omega0))
        return RealTerm(np.log(self.S0), np.log(0.5 * self.omega0 / self.Q))


def Matern(x, sigma: float = 1, rho: float = 1, n: int = 1, nu=3 / 2):
    """General Matern spectrum (psd_models.py:48-60; Matern-3/2 by default)."""
    from scipy.special import gamma
    x = np.asarray(x, dtype=np.float64)
    return 1 / sqrt(2 * pi) * sigma ** 2 * 2 ** n * pi ** (n / 2) * gamma(nu + n / 2) * (2 * nu) ** nu / (
        gamma(nu) * rho ** (2 * nu)) * (2 * nu / rho ** 2 + x ** 2) ** -(nu + n / 2)


class Matern32(PSDModel):
    """Matern-3/2 (psd_models.py:63-67); celerite's ``Matern32Term`` approaches it as eps -> 0."""

    param_names = ("sigma", "rho", "n")
    defaults = (1, 1, 1)

    @staticmethod
    def evaluate(x, sigma, rho, n=1):
        return 1 / sqrt(2 * pi) * sigma ** 2 * 4 / sqrt(3) * rho * (1 / (1 + (x * rho / sqrt(3)) ** 2)) ** 2

    def to_term(self, bounds=None, eps=0.01):
        from ..terms import Matern32Term
        return Matern32Term(np.log(self.sigma), np.log(self.rho), eps=eps)


class Matern52(PSDModel):
    """Matern-5/2 (psd_models.py:69-73)."""

    param_names = ("sigma", "rho")
    defaults = (1, 1)

    @staticmethod
    def evaluate(x, sigma, rho):
        return 1 / sqrt(2 * pi) * sigma ** 2 * 16 / 3 / sqrt(5) * rho * (1 / (1 + (x * rho / sqrt(5)) ** 2)) ** 3


class Jitter(PSDModel):
    """White noise of variance sigma^2 spread over the N frequencies given (psd_models.py:75-85);
    the frequencies must be evenly spaced."""

    param_names = ("sigma",)
    defaults = (1,)

    @staticmethod
    def evaluate(x, sigma):
        N = len(x)
        df = np.diff(x)[0]
        normalization_factor = 2 / sqrt(2 * pi)
        return np.ones(N) * sigma ** 2 / normalization_factor / df / N
