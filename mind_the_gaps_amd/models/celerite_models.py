"""The reference's own celerite terms, with device kind tags.

Mirror of /root/reference/mind_the_gaps/models/celerite_models.py:7-90: same
class names, parameter names, coefficient formulas and ``log_prior`` rules.  On
the hot path the formulas below are NOT what runs: ``mtg_kind`` tells the
prepare kernel (csrc/mtg_kernels.hip) to expand theta on the device; the Python
builders serve ``coefficients`` / ``get_psd`` and the parity tests.
"""
import numpy as np

from .. import engine as _engine
from ..terms import Term


class Lorentzian(Term):
    """celerite_models.py:7-34: a (0, 0) real term plus the complex term
    (S0, 0, w0 / 2Q, w0)."""

    parameter_names = ("log_S0", "log_Q", "log_omega0")
    mtg_kind = _engine.TERM_LORENTZIAN

    def get_real_coefficients(self, params):
        return 0, 0

    def get_complex_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        w0 = np.exp(log_omega0)
        return np.exp(log_S0), 0, 0.5 * w0 / np.exp(log_Q), w0

    def __repr__(self):
        return "Lorentzian({0.log_S0}, {0.log_Q}, {0.log_omega0})".format(self)


class Cosinus(Term):
    """celerite_models.py:36-52: undamped cosine, complex term (S0, 0, 0, w0)."""

    parameter_names = ("log_S0", "log_omega0")
    mtg_kind = _engine.TERM_COSINUS

    def get_complex_coefficients(self, params):
        log_S0, log_omega0 = params
        return np.exp(log_S0), 0, 0, np.exp(log_omega0)


class DampedRandomWalk(Term):
    """celerite_models.py:55-68 (Foreman-Mackey+2017 eq. 13): real term
    (S0, 0.5 w0 / Q) with Q = 1/2."""

    parameter_names = ("log_S0", "log_omega0")
    mtg_kind = _engine.TERM_DRW

    def get_real_coefficients(self, params):
        log_S0, log_omega0 = params
        Q = 1 / 2
        return np.exp(log_S0), 0.5 * np.exp(log_omega0) / Q

    def __repr__(self):
        return "DampedRandomWalk({0.log_S0}, {0.log_omega0})".format(self)


class BendingPowerlaw(Term):
    """celerite_models.py:71-90: complex term (e^log_S0, e^log_Q, w0, w0) with the
    extra prior log_S0 >= log_Q."""

    parameter_names = ("log_S0", "log_Q", "log_omega0")
    mtg_kind = _engine.TERM_BPL

    def get_complex_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        w0 = np.exp(log_omega0)
        return np.exp(log_S0), np.exp(log_Q), w0, w0

    def log_prior(self):
        if self.log_S0 < self.log_Q:
            return -np.inf
        return super().log_prior()
