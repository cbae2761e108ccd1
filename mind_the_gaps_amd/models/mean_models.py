"""Mean functions (API of /root/reference/mind_the_gaps/models/mean_models.py:6-31).

``GPModelling`` reaches the constant and the linear mean only (gpmodelling.py:27,83-111);
both are evaluated inside the kernels (MTG_MEAN_CONSTANT / MTG_MEAN_LINEAR, ``mtg_mean_kind``).
The Gaussian and sine profiles exist for API parity and evaluate on the host.
"""
import numpy as np

from .. import engine as _engine
from ..modeling import Model

__all__ = ["LinearModel", "GaussianModel", "SineModel"]


class LinearModel(Model):
    """slope * t + intercept, fitted on the device (mean_models.py:24-31)."""

    parameter_names = ("slope", "intercept")
    mtg_mean_kind = _engine.MEAN_LINEAR

    def get_value(self, x):
        return np.add(np.multiply(self.slope, x), self.intercept)

    def compute_gradient(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.vstack([x, np.ones_like(x)])          # d/d slope, d/d intercept


class GaussianModel(Model):
    """Gaussian bump on a constant level; note the reference's normalisation
    amplitude / (2 pi sigma) (mean_models.py:9-10), kept as is."""

    parameter_names = ("mean", "sigma", "amplitude", "constant")

    def get_value(self, x):
        u = (np.asarray(x, dtype=np.float64) - self.mean) / self.sigma
        peak = self.amplitude / (2.0 * np.pi * self.sigma)
        return self.constant + peak * np.exp(-0.5 * u * u)


class SineModel(Model):
    """constant + amplitude sin(frequency t + phase) (mean_models.py:12-16)."""

    parameter_names = ("constant", "amplitude", "frequency", "phase")

    def get_value(self, x):
        return self.constant + self.amplitude * np.sin(np.multiply(self.frequency, x) + self.phase)
