"""CPU tests of the host-side model protocol (celerite.modeling / celerite.terms
semantics restated in SURVEY.md Appendix A.2) and of the reference's own terms."""
import numpy as np
import pytest

from mind_the_gaps_amd import terms
from mind_the_gaps_amd.gp import GP, DeviceModel
from mind_the_gaps_amd.modeling import ConstantModel
from mind_the_gaps_amd.models import (BendingPowerlaw, Cosinus, DampedRandomWalk, LinearModel, Lorentzian)
from mind_the_gaps_amd.models import celerite_models
from oracle import dense

from test_oracle import reference_psd  # outputs of the reference's own psd_models.py (golden fixture)


# ---- the reference's tests/models_test.py, against THIS package's classes ------
def test_DRW():
    cel = celerite_models.DampedRandomWalk(log_S0=np.log(10), log_omega0=np.log(5))
    w, ref = reference_psd("drw", 10.0, 5.0)
    np.testing.assert_array_almost_equal(ref, cel.get_psd(w))


@pytest.mark.parametrize("Q", [10, 1, 1 / np.sqrt(2), 0.1])
def test_SHO(Q):
    cel = terms.SHOTerm(log_S0=np.log(10), log_Q=np.log(Q), log_omega0=np.log(5))
    w, ref = reference_psd("sho", 10.0, Q, 5.0)
    np.testing.assert_array_almost_equal(ref, cel.get_psd(w))


@pytest.mark.parametrize("rho", [1, 10, 20])
def test_materns(rho):
    cel = terms.Matern32Term(log_sigma=np.log(10), log_rho=np.log(rho), eps=1e-15)
    w, ref = reference_psd("matern32", 10.0, rho)
    np.testing.assert_array_almost_equal(ref, cel.get_psd(w))


@pytest.mark.parametrize("Q", [10, 1, 1 / np.sqrt(2), 0.1])
@pytest.mark.parametrize("S", [10, 5, 1])
def test_Lorentzian(Q, S):
    cel = celerite_models.Lorentzian(log_S0=np.log(S), log_Q=np.log(Q), log_omega0=np.log(5))
    w, ref = reference_psd("lorentzian", S, Q, 5.0)
    np.testing.assert_array_almost_equal(ref, cel.get_psd(w))


# ---- coefficients: product classes == oracle restatement -----------------------------
def test_coefficients_match_oracle():
    cases = [
        (DampedRandomWalk(1.0, -2.0), [dense.K_DRW]),
        (Lorentzian(2.0, 3.0, -1.0), [dense.K_LORENTZIAN]),
        (Cosinus(0.5, -0.3), [dense.K_COSINUS]),
        (BendingPowerlaw(2.0, 1.0, -1.5), [dense.K_BPL]),
        (terms.RealTerm(0.3, -1.0), [dense.K_REAL]),
        (terms.ComplexTerm(0.3, -1.0, 0.2), [dense.K_COMPLEX3]),
        (terms.ComplexTerm(1.3, 0.1, -1.0, 0.2), [dense.K_COMPLEX4]),
        (terms.SHOTerm(1.0, np.log(3.0), 0.1), [dense.K_SHO]),
        (terms.SHOTerm(1.0, np.log(0.2), 0.1), [dense.K_SHO]),
        (terms.Matern32Term(0.5, 1.5), [dense.K_MATERN32]),
        (DampedRandomWalk(1.0, -2.0) + terms.SHOTerm(1.0, 1.0, 0.1) + Lorentzian(2.0, 3.0, -1.0)
         + terms.JitterTerm(-0.4), [dense.K_DRW, dense.K_SHO, dense.K_LORENTZIAN, dense.K_JITTER]),
    ]
    for term, kinds in cases:
        want = dense.build_coeffs(kinds, term.get_parameter_vector(include_frozen=True))
        got = term.coefficients
        for g, w in zip(got, want[:6]):
            np.testing.assert_allclose(g, w, rtol=1e-15, atol=0)
        assert term.jitter == pytest.approx(want[6], rel=1e-15)
        assert [t.mtg_kind for t in term.terms] == kinds


def test_lorentzian_keeps_null_real_term():
    ar, cr, ac, bc, cc, dc = Lorentzian(1.0, 2.0, 0.5).coefficients   # celerite_models.py:12-15
    assert list(ar) == [0.0] and list(cr) == [0.0] and len(ac) == 1 and bc[0] == 0.0


# ---- parameter protocol --------------------------------------------------------------
def test_model_vector_bounds_freeze():
    t = DampedRandomWalk(log_S0=1.0, log_omega0=-2.0, bounds=[(-10, 50), (None, 10)])
    assert t.get_parameter_names() == ("log_S0", "log_omega0")
    assert t.get_parameter_bounds() == [(-10, 50), (None, 10)]
    assert t.log_S0 == 1.0 and t.log_omega0 == -2.0
    t.set_parameter_vector([2.0, 3.0])
    assert list(t.get_parameter_vector()) == [2.0, 3.0] and t.log_prior() == 0.0
    t.set_parameter_vector([2.0, 11.0])
    assert t.log_prior() == -np.inf
    t.set_parameter_vector([-1e9 if False else 2.0, 10.0])   # bounds are inclusive
    assert t.log_prior() == 0.0
    t.freeze_parameter("log_S0")
    assert t.get_parameter_names() == ("log_omega0",) and len(t) == 1 and t.full_size == 2
    t.set_parameter_vector([4.0])
    assert list(t.get_parameter_vector(include_frozen=True)) == [2.0, 4.0]
    t.thaw_all_parameters()
    assert t.vector_size == 2
    with pytest.raises(ValueError):
        DampedRandomWalk(1.0, 20.0, bounds=[(-10, 50), (-10, 10)])   # "non-finite log prior value"
    with pytest.raises(ValueError):
        DampedRandomWalk(1.0)
    d = DampedRandomWalk(1.0, 2.0, bounds={"log_omega0": (0, 5)})
    assert d.get_parameter_bounds() == [(None, None), (0, 5)]


def test_bending_powerlaw_prior():
    b = BendingPowerlaw(2.0, 1.0, 0.0, bounds=[(-10, 50), (-10, 10), (-10, 10)])
    assert b.log_prior() == 0.0
    b.set_parameter_vector([0.5, 1.0, 0.0])       # log_S0 < log_Q  (celerite_models.py:85-90)
    assert b.log_prior() == -np.inf


def test_termsum_and_gp_naming():
    k = DampedRandomWalk(1.0, -2.0, bounds=[(-10, 50), (-10, 10)]) + Lorentzian(2.0, 3.0, -1.0)
    assert isinstance(k, terms.TermSum) and len(k.terms) == 2
    assert k.get_parameter_names() == ("terms[0]:log_S0", "terms[0]:log_omega0", "terms[1]:log_S0",
                                       "terms[1]:log_Q", "terms[1]:log_omega0")
    k3 = k + terms.JitterTerm(0.0)
    assert len(k3.terms) == 3 and k3.jitter == pytest.approx(1.0)
    gp = GP(k, mean=ConstantModel(3.0, bounds=[(0, 10)]), fit_mean=False)
    assert gp.get_parameter_names() == tuple("kernel:" + n for n in k.get_parameter_names())
    assert gp.parameter_names[-1] == "mean:value" and len(gp.parameter_names) == 6
    assert len(gp.get_parameter_vector()) == 5
    assert gp.get_parameter_bounds()[:2] == [(-10, 50), (-10, 10)]
    gp.set_parameter_vector([1.5, -2.5, 2.0, 3.0, -1.0])
    assert k.terms[0].log_S0 == 1.5 and gp.log_prior() == 0.0
    gp.set_parameter_vector([51.0, -2.5, 2.0, 3.0, -1.0])
    assert gp.log_prior() == -np.inf
    single = GP(DampedRandomWalk(1.0, -2.0), mean=2.0, fit_mean=True)
    assert single.get_parameter_names() == ("kernel:log_S0", "kernel:log_omega0", "mean:value")
    # frozen mean outside its bounds still vetoes the prior (celerite counts frozen parameters)
    veto = GP(DampedRandomWalk(1.0, -2.0), mean=ConstantModel(3.0, bounds=[(0, 10)]))
    veto.mean.set_parameter_vector([11.0], include_frozen=True)
    assert veto.log_prior() == -np.inf


def test_device_model_flattening():
    k = DampedRandomWalk(1.0, -2.0, bounds=[(-10, 50), (-10, 10)]) + terms.SHOTerm(0.5, 1.0, 0.2) \
        + terms.Matern32Term(0.1, 0.2, eps=0.05)
    k.freeze_parameter("terms[1]:log_Q")
    gp = GP(k, mean=LinearModel(0.1, 2.0), fit_mean=True)
    m = DeviceModel(gp.kernel, gp.mean, gp.mean.unfrozen_mask)
    assert m.device_terms and m.kinds == [6, 3, 4] and m.extra[2] == 0.05 and m.mean_kind == 1
    assert list(m.free_index) == [0, 1, 2, 4, 5, 6, 7, 8]
    assert m.full[3] == 1.0 and np.isinf(m.bounds[2]).all() and list(m.bounds[0]) == [-10.0, 50.0]
    assert np.array_equal(m.full[m.free_index], gp.get_parameter_vector())

    class UserTerm(terms.Term):                      # no mtg_kind: host-evaluated coefficients
        parameter_names = ("log_a",)

        def get_real_coefficients(self, params):
            return np.exp(params[0]), 0.3

    m2 = DeviceModel(UserTerm(0.0) + DampedRandomWalk(1.0, -2.0), gp.mean, gp.mean.unfrozen_mask)
    assert not m2.device_terms


def test_gp_compute_validates():
    gp = GP(DampedRandomWalk(1.0, -2.0))
    with pytest.raises(ValueError):
        gp.compute(np.array([0.0, 2.0, 1.0]), 0.1)             # unsorted
    with pytest.raises(RuntimeError):
        gp.log_likelihood(np.zeros(3))                          # compute first
    gp.compute(np.array([0.0, 1.0, 2.0]), 0.1)
    with pytest.raises(ValueError):
        gp.log_likelihood(np.zeros(4))                          # dimension mismatch


def test_term_product_is_the_product_of_the_kernels():
    """celerite's ``k1 * k2``: the coefficient algebra must reproduce k1(tau) k2(tau) for every
    pairing (real x real, real x complex, complex x complex, sums)."""
    from mind_the_gaps_amd import terms
    tau = np.linspace(0.0, 40.0, 400)
    real = terms.RealTerm(np.log(3.0), np.log(0.2))
    comp = terms.ComplexTerm(np.log(2.0), np.log(0.1), np.log(0.5), np.log(1.3))
    sho = terms.SHOTerm(np.log(1.5), np.log(4.0), np.log(0.9))
    for k1, k2 in ((real, real), (real, comp), (comp, real), (comp, sho), (real + comp, sho + real)):
        prod = k1 * k2
        np.testing.assert_allclose(prod.get_value(tau), k1.get_value(tau) * k2.get_value(tau), rtol=1e-12, atol=1e-14)
        assert prod.get_parameter_names()[0].startswith("k1:") and prod.get_parameter_names()[-1].startswith("k2:")
        assert len(prod.get_parameter_vector()) == len(k1.get_parameter_vector()) + len(k2.get_parameter_vector())
    prod = real * comp
    prod.set_parameter_vector(prod.get_parameter_vector() + 0.1)          # parameters reach the factors
    np.testing.assert_allclose(prod.get_value(tau), prod.models["k1"].get_value(tau) * prod.models["k2"].get_value(tau), rtol=1e-12)
    total = real + comp * sho                                              # a product inside a sum
    np.testing.assert_allclose(total.get_value(tau), real.get_value(tau) + comp.get_value(tau) * sho.get_value(tau), rtol=1e-12)
    with pytest.raises(ValueError):
        real * terms.JitterTerm(0.0)


def test_kernels_print_as_the_notebooks_print_them():
    """celerite's repr of a sum of terms, copied from the stored output of docs/notebooks/tutorial_model_selection.ipynb cell 8"""
    from mind_the_gaps_amd.models.celerite_models import Lorentzian
    k = Lorentzian(4.792317694191546, 4.88245622156523, -1.381704165724039) + terms.RealTerm(5.3258366603280365, -0.9582493463338226) \
        + terms.Matern32Term(2.9179435023277605, 0.9755365010911353, eps=1e-08)
    assert repr(k) == ("(Lorentzian(4.792317694191546, 4.88245622156523, -1.381704165724039) + RealTerm(5.3258366603280365, "
                       "-0.9582493463338226) + Matern32Term(2.9179435023277605, 0.9755365010911353, eps=1e-08))")
    assert repr(terms.RealTerm(6.125107037258277, -2.0741459390188006)) == "RealTerm(6.125107037258277, -2.0741459390188006)"
    assert repr(terms.Matern32Term(3.0625535186291386, 2.3025850929940455, eps=1e-08)) == "Matern32Term(3.0625535186291386, 2.3025850929940455, eps=1e-08)"
