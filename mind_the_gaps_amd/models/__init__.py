"""Kernel and mean models of mind_the_gaps (mirror of
/root/reference/mind_the_gaps/models/__init__.py:1-2)."""
from .mean_models import LinearModel, GaussianModel, SineModel
from .celerite_models import Lorentzian, Cosinus, DampedRandomWalk, BendingPowerlaw

__all__ = ["LinearModel", "GaussianModel", "SineModel", "Lorentzian", "Cosinus",
           "DampedRandomWalk", "BendingPowerlaw"]
