"""celerite kernel terms as the hot path sees them.

Host-side mirror of ``celerite.terms`` (third-party dependency of the
reference; semantics restated in SURVEY.md Appendix A.2) -- the objects users
hand to ``GPModelling(lightcurve, kernel)``
(/root/reference/mind_the_gaps/gpmodelling.py:29-34,51;
tests/models_test.py:9,37; docs/notebooks/tutorial_ppp.ipynb:167,253).

Each built-in term carries ``mtg_kind``, the integer tag of include/mtg.h, so
that theta -> (a, b, c, d) runs ON THE DEVICE inside the prepare kernel.  The
Python ``get_*_coefficients`` below exist for the rest of the celerite API
(``coefficients``, ``get_psd``, ``get_value``) and for user-defined terms, which
override them exactly as with celerite (celerite_models.py:9,17); a kernel
containing such a term is evaluated through the raw-coefficient entry point
``mtg_loglike_coeffs``.
"""
import numpy as np

from . import engine as _engine
from .modeling import Model, ModelSet

__all__ = ["Term", "TermSum", "TermProduct", "RealTerm", "ComplexTerm", "SHOTerm", "Matern32Term", "JitterTerm"]


class Term(Model):
    """Base class: a sum of real (a e^{-c tau}) and complex
    (e^{-c tau}[a cos d tau + b sin d tau]) celerite terms."""

    mtg_kind = None  # MTG_TERM_* tag when the device can expand this term itself

    @property
    def terms(self):
        return [self]

    # -- overridable coefficient builders ------------------------------------
    def get_real_coefficients(self, params):
        return np.empty(0), np.empty(0)

    def get_complex_coefficients(self, params):
        return np.empty(0), np.empty(0), np.empty(0), np.empty(0)

    def get_jitter(self, params):
        return 0.0

    @property
    def jitter(self):
        return self.get_jitter(self.get_parameter_vector(include_frozen=True))

    def get_all_coefficients(self, params=None):
        if params is None:
            params = self.get_parameter_vector(include_frozen=True)
        r = self.get_real_coefficients(params)
        c = self.get_complex_coefficients(params)
        if len(c) == 3:  # (a, c, d) means b = 0
            a, cc, d = c
            c = (a, np.zeros_like(np.atleast_1d(np.asarray(a, dtype=np.float64))), cc, d)
        return tuple(np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in tuple(r) + tuple(c))

    @property
    def coefficients(self):
        """(a_real, c_real, a_comp, b_comp, c_comp, d_comp), each 1-d."""
        return self.get_all_coefficients()

    def mtg_extra(self):
        return 0.0

    # -- derived quantities ----------------------------------------------------
    def get_value(self, tau):
        ar, cr, ac, bc, cc, dc = self.coefficients
        tau = np.abs(np.asarray(tau, dtype=np.float64))
        k = np.zeros_like(tau)
        for a, c in zip(ar, cr):
            k = k + a * np.exp(-c * tau)
        for a, b, c, d in zip(ac, bc, cc, dc):
            k = k + np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))
        return k

    def get_psd(self, omega):
        """Power spectral density, celerite convention (sqrt(2/pi) prefactor)."""
        ar, cr, ac, bc, cc, dc = self.coefficients
        w2 = np.asarray(omega, dtype=np.float64) ** 2
        p = np.zeros_like(w2)
        for a, c in zip(ar, cr):
            p = p + a * c / (c * c + w2)
        for a, b, c, d in zip(ac, bc, cc, dc):
            w02 = c * c + d * d
            p = p + ((a * c + b * d) * w02 + (a * c - b * d) * w2) / (
                w2 * w2 + 2.0 * (c * c - d * d) * w2 + w02 * w02)
        return np.sqrt(2.0 / np.pi) * p

    def __add__(self, other):
        if not isinstance(other, Term):
            return NotImplemented
        return TermSum(*(self.terms + other.terms))

    def __radd__(self, other):
        if other == 0:
            return self
        return NotImplemented

    def __mul__(self, other):
        if not isinstance(other, Term):
            return NotImplemented
        return TermProduct(self, other)


class TermSum(ModelSet, Term):
    """``k1 + k2 + ...``: parameters named ``terms[i]:name`` in `+` order."""

    def __init__(self, *terms):
        ModelSet.__init__(self, [("terms[{0}]".format(i), t) for i, t in enumerate(terms)])

    @property
    def terms(self):
        return list(self.models.values())

    def get_all_coefficients(self, params=None):
        if params is not None:
            raise ValueError("TermSum coefficients are taken from its terms")
        parts = [t.get_all_coefficients() for t in self.terms]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(6))

    @property
    def jitter(self):
        return float(sum(t.jitter for t in self.terms))

    def log_prior(self):
        return ModelSet.log_prior(self)

    def __repr__(self):
        # celerite's form, as the notebooks print it: "(Lorentzian(...) + RealTerm(...) + Matern32Term(..., eps=1e-08))"
        return "(" + " + ".join(repr(t) for t in self.terms) + ")"


class TermProduct(ModelSet, Term):
    """``k1 * k2`` (celerite's TermProduct): the product of two sums of exponentials is again
    one -- real x real gives a real term (a1 a2, c1 + c2), real x complex a complex term
    (a1 a2, a1 b2, c1 + c2, d2), complex x complex two complex terms at the difference and the sum
    of the frequencies.  Parameters are named ``k1:...`` and ``k2:...``.  No device tag: the
    coefficients are expanded on the host and evaluated through ``mtg_loglike_coeffs``."""

    def __init__(self, k1, k2):
        if k1.jitter != 0.0 or k2.jitter != 0.0:
            raise ValueError("jitter terms cannot be multiplied")
        ModelSet.__init__(self, [("k1", k1), ("k2", k2)])

    def get_all_coefficients(self, params=None):
        if params is not None:
            raise ValueError("TermProduct coefficients are taken from its factors")
        r1, q1, a1, b1, c1, d1 = self.models["k1"].get_all_coefficients()
        r2, q2, a2, b2, c2, d2 = self.models["k2"].get_all_coefficients()
        ar = [x * y for x in r1 for y in r2]
        cr = [x + y for x in q1 for y in q2]
        ac, bc, cc, dc = [], [], [], []
        for rr, qq, aa, bb, c_, dd in ((r1, q1, a2, b2, c2, d2), (r2, q2, a1, b1, c1, d1)):   # real x complex
            for x, cx in zip(rr, qq):
                for a, b, c, d in zip(aa, bb, c_, dd):
                    ac.append(x * a); bc.append(x * b); cc.append(cx + c); dc.append(d)
        for aj, bj, cj, dj in zip(a1, b1, c1, d1):                                             # complex x complex
            for ak, bk, ck, dk in zip(a2, b2, c2, d2):
                ac.append(0.5 * (aj * ak + bj * bk)); bc.append(0.5 * (bj * ak - aj * bk))
                cc.append(cj + ck); dc.append(dj - dk)
                ac.append(0.5 * (aj * ak - bj * bk)); bc.append(0.5 * (bj * ak + aj * bk))
                cc.append(cj + ck); dc.append(dj + dk)
        return tuple(np.asarray(v, dtype=np.float64) for v in (ar, cr, ac, bc, cc, dc))

    @property
    def jitter(self):
        return 0.0

    def log_prior(self):
        return ModelSet.log_prior(self)

    def __repr__(self):
        return "({0!r}) * ({1!r})".format(self.models["k1"], self.models["k2"])


class RealTerm(Term):
    r"""k(tau) = a e^{-c tau}, parameters ``log_a``, ``log_c``."""

    parameter_names = ("log_a", "log_c")
    mtg_kind = _engine.TERM_REAL

    def get_real_coefficients(self, params):
        log_a, log_c = params
        return np.exp(log_a), np.exp(log_c)

    def __repr__(self):
        return "RealTerm({0.log_a}, {0.log_c})".format(self)


class ComplexTerm(Term):
    r"""k(tau) = e^{-c tau}[a cos(d tau) + b sin(d tau)]; ``log_b`` optional (b = 0)."""

    def __init__(self, *args, **kwargs):
        if len(args) == 4 or "log_b" in kwargs:
            self.fit_b = True
            self.parameter_names = ("log_a", "log_b", "log_c", "log_d")
        else:
            self.fit_b = False
            self.parameter_names = ("log_a", "log_c", "log_d")
        super().__init__(*args, **kwargs)

    # parameter_names is per instance here: route attribute access accordingly
    def __getattr__(self, name):
        d = self.__dict__
        names = d.get("parameter_names", ())
        if name in names and "parameter_vector" in d:
            return d["parameter_vector"][names.index(name)]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        d = self.__dict__
        if name in d.get("parameter_names", ()) and "parameter_vector" in d:
            d["parameter_vector"][d["parameter_names"].index(name)] = value
            d["dirty"] = True
        else:
            object.__setattr__(self, name, value)

    @property
    def mtg_kind(self):
        return _engine.TERM_COMPLEX4 if self.fit_b else _engine.TERM_COMPLEX3

    def get_complex_coefficients(self, params):
        if self.fit_b:
            log_a, log_b, log_c, log_d = params
            return np.exp(log_a), np.exp(log_b), np.exp(log_c), np.exp(log_d)
        log_a, log_c, log_d = params
        return np.exp(log_a), 0.0, np.exp(log_c), np.exp(log_d)

    def log_prior(self):
        # celerite rejects a complex term that is not positive definite on its own
        lp = super().log_prior()
        if not np.isfinite(lp):
            return -np.inf
        a, b, c, d = self.get_complex_coefficients(self.get_parameter_vector(include_frozen=True))
        if a * c < b * d:
            return -np.inf
        return lp

    def __repr__(self):
        return "ComplexTerm({0})".format(", ".join(repr(float(v)) for v in self.parameter_vector))


class SHOTerm(Term):
    r"""Stochastically driven damped harmonic oscillator,
    S(w) = sqrt(2/pi) S0 w0^4 / ((w^2 - w0^2)^2 + w0^2 w^2 / Q^2)."""

    parameter_names = ("log_S0", "log_Q", "log_omega0")
    mtg_kind = _engine.TERM_SHO

    def get_real_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        Q = np.exp(log_Q)
        if Q >= 0.5:
            return np.empty(0), np.empty(0)
        S0, w0 = np.exp(log_S0), np.exp(log_omega0)
        f = np.sqrt(1.0 - 4.0 * Q * Q)
        return (0.5 * S0 * w0 * Q * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f]),
                0.5 * w0 / Q * np.array([1.0 - f, 1.0 + f]))

    def get_complex_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        Q = np.exp(log_Q)
        if Q < 0.5:
            return np.empty(0), np.empty(0), np.empty(0), np.empty(0)
        S0, w0 = np.exp(log_S0), np.exp(log_omega0)
        f = np.sqrt(4.0 * Q * Q - 1.0)
        return S0 * w0 * Q, S0 * w0 * Q / f, 0.5 * w0 / Q, 0.5 * w0 / Q * f

    def __repr__(self):
        return "SHOTerm({0.log_S0}, {0.log_Q}, {0.log_omega0})".format(self)


class Matern32Term(Term):
    r"""Matern-3/2 approximated by a complex term with small ``eps``."""

    parameter_names = ("log_sigma", "log_rho")
    mtg_kind = _engine.TERM_MATERN32

    def __init__(self, *args, **kwargs):
        eps = kwargs.pop("eps", 0.01)
        object.__setattr__(self, "eps", float(eps))
        super().__init__(*args, **kwargs)

    def mtg_extra(self):
        return self.eps

    def get_complex_coefficients(self, params):
        log_sigma, log_rho = params
        w0 = np.sqrt(3.0) * np.exp(-log_rho)
        S0 = np.exp(2.0 * log_sigma) / w0
        return w0 * S0, w0 * w0 * S0 / self.eps, w0, self.eps

    def __repr__(self):
        return "Matern32Term({0.log_sigma}, {0.log_rho}, eps={0.eps})".format(self)


class JitterTerm(Term):
    r"""White noise added to the diagonal: jitter = exp(2 log_sigma)."""

    parameter_names = ("log_sigma",)
    mtg_kind = _engine.TERM_JITTER

    def get_jitter(self, params):
        return float(np.exp(2.0 * params[0]))

    def __repr__(self):
        return "JitterTerm({0.log_sigma})".format(self)
