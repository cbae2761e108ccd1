"""Parameter protocol of the models the hot path consumes.

Host-side mirror of the ``celerite.modeling`` interface that
/root/reference/mind_the_gaps/gpmodelling.py relies on (celerite is a
third-party dependency of the reference, absent from this image; semantics
restated in SURVEY.md Appendix A.2): named parameters, (lo, hi) bounds with
``None`` for an open side, freeze/thaw, a box ``log_prior`` and composite
``ModelSet`` vectors with ``prefix:name`` parameter names.

Used by the reference at gpmodelling.py:51-55 (construction,
``get_parameter_vector``), :147-151 (``set_parameter_vector``/``log_prior``),
:193,239 (``get_parameter_bounds``), :454 (``get_parameter_names``).
"""
from collections import OrderedDict

import numpy as np

__all__ = ["Model", "ModelSet", "ConstantModel"]


class Model:
    """A set of named scalar parameters with bounds and a frozen/thawed mask."""

    parameter_names = tuple()

    def __init__(self, *args, **kwargs):
        names = self.parameter_names
        bounds = kwargs.pop("bounds", None)
        if len(args):
            if len(args) != len(names):
                raise ValueError("expected {0} arguments but got {1}".format(len(names), len(args)))
            if len(kwargs):
                raise ValueError("parameters must be fully specified by arguments or keyword "
                                 "arguments, not both")
            values = list(args)
        else:
            values = []
            for name in names:
                if name not in kwargs:
                    raise ValueError("missing parameter '{0}'".format(name))
                values.append(kwargs.pop(name))
            if len(kwargs):
                raise ValueError("unrecognized parameter(s) {0}".format(sorted(kwargs)))
        self.parameter_vector = np.array(values, dtype=np.float64)
        self.unfrozen_mask = np.ones(len(names), dtype=bool)
        if bounds is None:
            self.parameter_bounds = [(None, None) for _ in names]
        elif isinstance(bounds, dict):
            self.parameter_bounds = [tuple(bounds.get(n, (None, None))) for n in names]
        else:
            bounds = list(bounds)
            if len(bounds) != len(names):
                raise ValueError("the number of bounds must equal the number of parameters")
            self.parameter_bounds = [(None, None) if b is None else tuple(b) for b in bounds]
        for b in self.parameter_bounds:
            if len(b) != 2:
                raise ValueError("invalid bounds: each entry must be a (lower, upper) pair")
        self.dirty = True
        if not np.isfinite(self.log_prior()):
            raise ValueError("non-finite log prior value")

    # -- attribute access by parameter name (celerite_models.py:34,68,88) ----
    def __getattr__(self, name):
        if name.startswith("__") or name in ("parameter_names", "parameter_vector"):
            raise AttributeError(name)
        names = type(self).parameter_names
        if name in names and "parameter_vector" in self.__dict__:
            return self.__dict__["parameter_vector"][names.index(name)]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        names = type(self).parameter_names
        if name in names and "parameter_vector" in self.__dict__:
            self.parameter_vector[names.index(name)] = value
            self.dirty = True
        else:
            object.__setattr__(self, name, value)

    # -- sizes -------------------------------------------------------------
    def __len__(self):
        return self.vector_size

    @property
    def full_size(self):
        return len(self.parameter_names)

    @property
    def vector_size(self):
        return int(np.sum(self.unfrozen_mask))

    # -- vectors -------------------------------------------------------------
    def get_parameter_names(self, include_frozen=False):
        if include_frozen:
            return tuple(self.parameter_names)
        return tuple(n for n, m in zip(self.parameter_names, self.unfrozen_mask) if m)

    def get_parameter_bounds(self, include_frozen=False):
        if include_frozen:
            return list(self.parameter_bounds)
        return [b for b, m in zip(self.parameter_bounds, self.unfrozen_mask) if m]

    def get_parameter_vector(self, include_frozen=False):
        if include_frozen:
            return self.parameter_vector.copy()
        return self.parameter_vector[self.unfrozen_mask]

    def set_parameter_vector(self, vector, include_frozen=False):
        v = np.asarray(vector, dtype=np.float64)
        if include_frozen:
            if v.shape != self.parameter_vector.shape:
                raise ValueError("dimension mismatch")
            self.parameter_vector[:] = v
        else:
            if v.shape != (self.vector_size,):
                raise ValueError("dimension mismatch")
            self.parameter_vector[self.unfrozen_mask] = v
        self.dirty = True

    def get_parameter_dict(self, include_frozen=False):
        return OrderedDict(zip(self.get_parameter_names(include_frozen),
                               self.get_parameter_vector(include_frozen)))

    def _index(self, name):
        try:
            return list(self.parameter_names).index(name)
        except ValueError:
            raise ValueError("unknown parameter '{0}'".format(name))

    def get_parameter(self, name):
        return self.parameter_vector[self._index(name)]

    def set_parameter(self, name, value):
        self.parameter_vector[self._index(name)] = value
        self.dirty = True

    def freeze_parameter(self, name):
        self.unfrozen_mask[self._index(name)] = False

    def thaw_parameter(self, name):
        self.unfrozen_mask[self._index(name)] = True

    def freeze_all_parameters(self):
        self.unfrozen_mask[:] = False

    def thaw_all_parameters(self):
        self.unfrozen_mask[:] = True

    # -- prior ---------------------------------------------------------------
    def log_prior(self):
        """0.0 when every parameter (frozen ones included) is inside its bounds, else -inf."""
        for (lo, hi), v in zip(self.parameter_bounds, self.parameter_vector):
            if lo is not None and v < lo:
                return -np.inf
            if hi is not None and v > hi:
                return -np.inf
        return 0.0

    def get_value(self, *args, **kwargs):
        raise NotImplementedError("overloaded by subclasses")


class ModelSet(Model):
    """Ordered named collection of models exposed as one parameter vector."""

    def __init__(self, models):
        self.models = OrderedDict(models)
        self.dirty = True

    # composite views (no parameter_vector of its own)
    @property
    def parameter_names(self):
        return tuple("{0}:{1}".format(k, n) for k, m in self.models.items()
                     for n in m.get_parameter_names(include_frozen=True))

    @property
    def parameter_vector(self):
        parts = [m.get_parameter_vector(include_frozen=True) for m in self.models.values()]
        return np.concatenate(parts) if parts else np.empty(0)

    @parameter_vector.setter
    def parameter_vector(self, v):
        self.set_parameter_vector(v, include_frozen=True)

    @property
    def unfrozen_mask(self):
        parts = [np.atleast_1d(m.unfrozen_mask) for m in self.models.values()]
        return np.concatenate(parts) if parts else np.empty(0, dtype=bool)

    @property
    def parameter_bounds(self):
        return [b for m in self.models.values() for b in m.get_parameter_bounds(include_frozen=True)]

    def __getattr__(self, name):
        models = self.__dict__.get("models")
        if models is not None and name in models:
            return models[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)

    @property
    def full_size(self):
        return sum(m.full_size for m in self.models.values())

    @property
    def vector_size(self):
        return sum(m.vector_size for m in self.models.values())

    def get_parameter_names(self, include_frozen=False):
        return tuple("{0}:{1}".format(k, n) for k, m in self.models.items()
                     for n in m.get_parameter_names(include_frozen=include_frozen))

    def get_parameter_bounds(self, include_frozen=False):
        return [b for m in self.models.values()
                for b in m.get_parameter_bounds(include_frozen=include_frozen)]

    def get_parameter_vector(self, include_frozen=False):
        parts = [m.get_parameter_vector(include_frozen=include_frozen) for m in self.models.values()]
        return np.concatenate(parts) if parts else np.empty(0)

    def set_parameter_vector(self, vector, include_frozen=False):
        v = np.asarray(vector, dtype=np.float64)
        size = self.full_size if include_frozen else self.vector_size
        if v.shape != (size,):
            raise ValueError("dimension mismatch")
        i = 0
        for m in self.models.values():
            n = m.full_size if include_frozen else m.vector_size
            m.set_parameter_vector(v[i:i + n], include_frozen=include_frozen)
            i += n
        self.dirty = True

    def _split(self, name):
        head, _, tail = name.partition(":")
        if head not in self.models or not tail:
            raise ValueError("unknown parameter '{0}'".format(name))
        return self.models[head], tail

    def get_parameter(self, name):
        m, tail = self._split(name)
        return m.get_parameter(tail)

    def set_parameter(self, name, value):
        m, tail = self._split(name)
        m.set_parameter(tail, value)
        self.dirty = True

    def freeze_parameter(self, name):
        m, tail = self._split(name)
        m.freeze_parameter(tail)

    def thaw_parameter(self, name):
        m, tail = self._split(name)
        m.thaw_parameter(tail)

    def freeze_all_parameters(self):
        for m in self.models.values():
            m.freeze_all_parameters()

    def thaw_all_parameters(self):
        for m in self.models.values():
            m.thaw_all_parameters()

    def log_prior(self):
        lp = 0.0
        for m in self.models.values():
            lp += m.log_prior()
            if not np.isfinite(lp):
                return -np.inf
        return lp


class ConstantModel(Model):
    """``celerite.modeling.ConstantModel`` (gpmodelling.py:83-97)."""

    parameter_names = ("value",)

    def get_value(self, x):
        return self.value + np.zeros_like(np.asarray(x, dtype=np.float64))

    def compute_gradient(self, x):
        return np.ones((1, len(np.atleast_1d(x))))
