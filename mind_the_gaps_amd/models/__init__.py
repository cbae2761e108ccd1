"""Kernel and mean models of mind_the_gaps (mirror of
/root/reference/mind_the_gaps/models/__init__.py:1-2)."""
from .mean_models import LinearModel, GaussianModel, SineModel
from .celerite_models import Lorentzian, Cosinus, DampedRandomWalk, BendingPowerlaw
from . import psd_models  # closed-form spectra (reference models/psd_models.py); not re-exported by name,
#                           like the reference: Lorentzian / BendingPowerlaw above are the celerite terms

__all__ = ["LinearModel", "GaussianModel", "SineModel", "Lorentzian", "Cosinus",
           "DampedRandomWalk", "BendingPowerlaw"]
