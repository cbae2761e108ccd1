"""The reference's own celerite terms as device-expandable terms.

API of /root/reference/mind_the_gaps/models/celerite_models.py:7-90 (class names,
parameter names, coefficients, ``BendingPowerlaw``'s extra prior), declared here as a
table: every term is (device kind tag, log-parameter names, coefficient rule in linear
parameters).  On the hot path none of these Python rules run -- ``mtg_kind`` tells
``mtg_prepare_kernel`` (csrc/mtg_kernels.hip) to expand theta on the GPU; the rules serve
``Term.coefficients`` / ``get_psd`` and the parity tests of the two expansions.
"""
import numpy as np

from .. import engine as _engine
from ..terms import Term

__all__ = ["Lorentzian", "Cosinus", "DampedRandomWalk", "BendingPowerlaw"]


def _device_term(name, kind, log_names, real=None, complex_=None, ordered=None, doc=""):
    """Build a Term subclass whose coefficient rules take the exponentiated parameters.

    real(*linear) -> (a, c); complex_(*linear) -> (a, b, c, d); ``ordered = (hi, lo)``
    adds the prior ``hi >= lo`` on two of the log parameters."""
    def linear(params):
        return [float(v) for v in np.exp(np.asarray(params, dtype=np.float64))]

    body = {"parameter_names": tuple(log_names), "mtg_kind": kind, "__doc__": doc, "__module__": __name__}
    if real is not None:
        body["get_real_coefficients"] = lambda self, params: real(*linear(params))
    if complex_ is not None:
        body["get_complex_coefficients"] = lambda self, params: complex_(*linear(params))
    if ordered is not None:
        hi, lo = ordered

        def log_prior(self):
            if getattr(self, hi) < getattr(self, lo):
                return -np.inf
            return Term.log_prior(self)
        body["log_prior"] = log_prior
    body["__repr__"] = lambda self: "%s(%s)" % (name, ", ".join(repr(float(getattr(self, n))) for n in log_names))
    return type(name, (Term,), body)


# celerite_models.py:7-34 -- the reference also returns a real term (a = 0, c = 0); it never
# enters the likelihood (U = 0) and the device expansion drops it.
Lorentzian = _device_term(
    "Lorentzian", _engine.TERM_LORENTZIAN, ("log_S0", "log_Q", "log_omega0"),
    real=lambda S0, Q, w0: (0, 0),
    complex_=lambda S0, Q, w0: (S0, 0, w0 / (2.0 * Q), w0),
    doc="Lorentzian of centroid w0 and quality factor Q: complex term (S0, 0, w0 / 2Q, w0).")

# celerite_models.py:36-52
Cosinus = _device_term(
    "Cosinus", _engine.TERM_COSINUS, ("log_S0", "log_omega0"),
    complex_=lambda S0, w0: (S0, 0, 0, w0),
    doc="Undamped cosine: complex term (S0, 0, 0, w0).")

# celerite_models.py:55-68 (Foreman-Mackey+2017 eq. 13 with Q = 1/2: c = 0.5 w0 / Q = w0)
DampedRandomWalk = _device_term(
    "DampedRandomWalk", _engine.TERM_DRW, ("log_S0", "log_omega0"),
    real=lambda S0, w0: (S0, w0),
    doc="Damped random walk (Ornstein-Uhlenbeck): real term (S0, w0).")

# celerite_models.py:71-90
BendingPowerlaw = _device_term(
    "BendingPowerlaw", _engine.TERM_BPL, ("log_S0", "log_Q", "log_omega0"),
    complex_=lambda S0, Q, w0: (S0, Q, w0, w0),
    ordered=("log_S0", "log_Q"),
    doc="Bending power law: complex term (S0, Q, w0, w0); prior log_S0 >= log_Q.")
