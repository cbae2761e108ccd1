"""Mean models (mirror of /root/reference/mind_the_gaps/models/mean_models.py:6-31).

Only the constant and linear means are reachable from ``GPModelling``
(gpmodelling.py:27,83-111) and evaluated on the device (MTG_MEAN_CONSTANT /
MTG_MEAN_LINEAR); the others are kept for API parity and evaluate on the host.
"""
import numpy as np

from .. import engine as _engine
from ..modeling import Model


class GaussianModel(Model):
    parameter_names = ("mean", "sigma", "amplitude", "constant")

    def get_value(self, x):
        return self.amplitude / (2 * np.pi * self.sigma) * np.exp(
            -(x - self.mean) ** 2 / (2 * self.sigma ** 2)) + self.constant


class SineModel(Model):
    parameter_names = ("constant", "amplitude", "frequency", "phase")

    def get_value(self, x):
        return self.constant + self.amplitude * np.sin(self.frequency * x + self.phase)


class LinearModel(Model):
    """mean_models.py:24-31."""

    parameter_names = ("slope", "intercept")
    mtg_mean_kind = _engine.MEAN_LINEAR

    def get_value(self, x):
        return self.slope * x + self.intercept

    def compute_gradient(self, x):
        return np.array([np.ones_like(x) * x, np.ones_like(x)])
