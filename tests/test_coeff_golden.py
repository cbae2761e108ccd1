"""The coefficient builders against the REFERENCE's own outputs.

tests/golden/coeff_golden.npz holds what the ``get_real_coefficients`` / ``get_complex_coefficients`` methods of
/root/reference/mind_the_gaps/models/celerite_models.py returned for 40 parameter vectors per class over the tutorials'
prior box (generator: tests/golden/make_coeff_golden.py, which compiles those methods from the reference file).  Three
things must reproduce them: the oracle's restatement (oracle/dense.py: what the C oracle and every parity test build
on), the product's host classes (mind_the_gaps_amd/models), and -- on the GPU -- the device's expansion of theta
(csrc/mtg_prepare.h), checked through the likelihood: the model entry point against the raw-coefficient entry point fed
with the reference's coefficients.
"""
import os

import numpy as np
import pytest

from mind_the_gaps_amd import models, synthetic as synth
from oracle import dense

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "coeff_golden.npz"))
KEYS = ("a_real", "c_real", "a_comp", "b_comp", "c_comp", "d_comp")
KIND = {"Lorentzian": dense.K_LORENTZIAN, "Cosinus": dense.K_COSINUS, "DampedRandomWalk": dense.K_DRW,
        "BendingPowerlaw": dense.K_BPL}
CLASSES = [str(c) for c in GOLD["classes"]]


@pytest.mark.parametrize("name", CLASSES)
def test_oracle_restatement_reproduces_the_reference_builders(name):
    for i, p in enumerate(GOLD[name + "/params"]):
        got = dense.build_coeffs([KIND[name]], p)
        for j, key in enumerate(KEYS):
            assert np.array_equal(np.atleast_1d(got[j]), GOLD["%s/%s" % (name, key)][i]), (name, i, key)


@pytest.mark.parametrize("name", CLASSES)
def test_product_classes_reproduce_the_reference_builders(name):
    cls = getattr(models, name)
    checked = 0
    for i, p in enumerate(GOLD[name + "/params"]):
        if name == "BendingPowerlaw" and p[0] < p[1]:
            with pytest.raises(ValueError):      # its prior rule (celerite_models.py:85-90): celerite refuses such a start
                cls(*p)
            continue
        checked += 1
        got = cls(*p).coefficients
        for j, key in enumerate(KEYS):
            assert np.array_equal(np.atleast_1d(got[j]), GOLD["%s/%s" % (name, key)][i]), (name, i, key)
    assert checked >= 15


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Lorentzian", "DampedRandomWalk", "BendingPowerlaw"])
def test_device_expansion_agrees_with_the_reference_builders(engine, name):
    """theta -> coefficients on the device (mtg_prepare_kernel), seen through lnL: the same rows through the model entry
    point and through mtg_loglike_coeffs with the reference's own coefficients.  (Cosinus alone has c = 0 -- a kernel that
    does not decay --, not a covariance one fits on its own; the device's exp is OCML's, numpy's is libm's: an ulp apart.)"""
    kinds = [{"Lorentzian": synth.K_LORENTZIAN, "DampedRandomWalk": synth.K_DRW, "BendingPowerlaw": synth.K_BPL}[name]]
    params = GOLD[name + "/params"]
    # rows whose amplitudes and frequencies leave a well-conditioned covariance on this sampling
    keep = params[:, 0] < 14        # (e^14 against unit noise: beyond that both entry points drown in the same conditioning)
    if name == "BendingPowerlaw":
        keep &= params[:, 0] >= params[:, 1]                    # its own prior rule (celerite_models.py:85-90)
    assert keep.sum() >= 6
    t, y, dy = synth.make_lightcurves(500, 1, seed=9)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    theta = params[keep]
    got, gst = engine.loglike(theta, add_prior=False)
    cols = [GOLD["%s/%s" % (name, key)][keep] for key in KEYS]
    if name == "Lorentzian":      # its null real term (a = 0, c = 0) carries nothing: handed over or not, the same lnL
        cols[0], cols[1] = np.empty((keep.sum(), 0)), np.empty((keep.sum(), 0))
    want, wst = engine.loglike_coeffs(*cols)
    ok = (gst == 0) & (wst == 0)
    assert np.array_equal(gst == 0, wst == 0) and ok.sum() >= 5
    assert np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok])) <= 1e-9
