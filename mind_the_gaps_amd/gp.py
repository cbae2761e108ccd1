"""``GP``: the celerite.GP interface of the hot path, evaluated on MI355X.

Host-side mirror of what /root/reference/mind_the_gaps/gpmodelling.py does with
``celerite.GP`` (third-party; semantics restated in SURVEY.md Appendix A.1/A.2):

    gp = GP(kernel, mean=meanmodel, fit_mean=fit_mean)        gpmodelling.py:51
    gp.compute(times, dy + 1e-12)                             gpmodelling.py:54
    gp.get_parameter_vector() / set_parameter_vector(theta)   gpmodelling.py:55,147
    gp.log_prior(); gp.log_likelihood(y)                      gpmodelling.py:149-152
    gp.get_parameter_bounds() / get_parameter_names()         gpmodelling.py:193,454

Every likelihood value comes from the HIP kernels through the C-ABI
(engine.Engine).  There is no host fallback: without libmtg_hip.so or a GPU the
first evaluation raises ``EngineUnavailable``.

Beyond celerite's one-theta-at-a-time calls, ``LogProbEvaluator`` exposes the
batched form the ensemble sampler and the finite-difference gradient use:
``evaluate(theta[B, P], lc_index[B]) -> (lnP[B], status[B])`` over L resident
light curves.
"""
import numpy as np

from . import engine as _engine
from .modeling import ConstantModel, Model, ModelSet
from .terms import Term

__all__ = ["GP", "LinAlgError", "LogProbEvaluator", "DeviceModel", "get_engine"]


class LinAlgError(Exception):
    """The covariance matrix is not positive definite (celerite.solver.LinAlgError)."""


_engines = {}


def get_engine(device=0):
    """Process-wide engine (one ``mtg_ctx``) per GPU."""
    eng = _engines.get(device)
    if eng is None:
        eng = _engine.Engine(device)
        eng.bound_to = None
        _engines[device] = eng
    return eng


_side_engines = {}


def get_side_engine(device=0, k=0):
    """The k-th extra context of a GPU, kept for the life of the process: work that runs BESIDE the process-wide engine's
    (or beside another side engine's) from another host thread -- the two models of the Protassov test.  Kept and reused
    rather than made per call: HIP deals its streams round the hardware queues in the order they are created (four queues
    by default), and two contexts whose streams land on the same queue run their kernels strictly one after the other
    (measured: the refits of a 250-light-curve block side by side 3.56 s, on one queue 4.09 s = one after the other)."""
    key = (device, int(k))
    eng = _side_engines.get(key)
    if eng is None:
        eng = _engine.Engine(device)
        eng.bound_to = None
        _side_engines[key] = eng
    return eng


def _inf_bounds(bounds):
    out = np.empty((len(bounds), 2), dtype=np.float64)
    for i, (lo, hi) in enumerate(bounds):
        out[i, 0] = -np.inf if lo is None else lo
        out[i, 1] = np.inf if hi is None else hi
    return out


class DeviceModel:
    """Flattened (kernel, mean) description handed to ``mtg_set_model``.

    ``device_terms`` is False when some term has no ``mtg_kind`` (a user-defined
    Python term): coefficients are then evaluated on the host per theta and sent
    through ``mtg_loglike_coeffs``.
    """

    def __init__(self, kernel, mean, mean_unfrozen):
        self.terms = list(kernel.terms)
        self.kinds = [t.mtg_kind for t in self.terms]
        self.device_terms = all(k is not None for k in self.kinds)
        self.extra = [float(t.mtg_extra()) for t in self.terms]
        if isinstance(mean, ConstantModel):
            self.mean_kind = _engine.MEAN_CONSTANT
        elif getattr(mean, "mtg_mean_kind", None) is not None:
            self.mean_kind = mean.mtg_mean_kind
        else:
            self.mean_kind = None  # host-evaluated mean: only a frozen one is supported
        kfull = kernel.get_parameter_vector(include_frozen=True)
        kmask = np.atleast_1d(kernel.unfrozen_mask).astype(bool)
        kbounds = kernel.get_parameter_bounds(include_frozen=True)
        if self.mean_kind is None:
            mfull, mmask, mbounds = np.zeros(1), np.zeros(1, dtype=bool), [(None, None)]
        else:
            mfull = mean.get_parameter_vector(include_frozen=True)
            mmask = np.atleast_1d(mean_unfrozen).astype(bool)
            mbounds = mean.get_parameter_bounds(include_frozen=True)
        self.nk = len(kfull)
        self.full = np.concatenate([kfull, mfull]).astype(np.float64)
        self.mask = np.concatenate([kmask, mmask])
        self.free_index = np.flatnonzero(self.mask).astype(np.int32)
        self.bounds = _inf_bounds(list(kbounds) + list(mbounds))
        # A frozen constant mean -- the reference default, ConstantModel(lightcurve.mean)
        # with fit_mean=False (gpmodelling.py:83-87) -- is subtracted from y once at upload
        # (mtg_set_lightcurves y_offset); the device model then carries a mean of exactly 0
        # and its bounds move with it, so the box prior still vetoes an out-of-range value.
        self.y_offset = None
        if self.mean_kind == _engine.MEAN_CONSTANT and not self.mask[self.nk]:
            self.y_offset = float(self.full[self.nk])
            self.bounds[self.nk] -= self.y_offset
            self.full[self.nk] = 0.0

    def signature(self):
        frozen = self.full[~self.mask]
        return (tuple(-1 if k is None else k for k in self.kinds), tuple(self.extra), self.mean_kind,
                self.mask.tobytes(), frozen.tobytes(), self.bounds.tobytes(), self.y_offset)


class LogProbEvaluator:
    """L light curves resident on one GPU + one model; batched lnP / lnL.

    ``t``: [N] shared sampling (gpmodelling.py:538 gives every simulated light
    curve the observed ``times``) or [L][N]; ``y``, ``yerr``: [N] or [L][N];
    ``yerr`` is what the reference passes to ``gp.compute``, i.e. ``dy + 1e-12``
    (gpmodelling.py:54); the device squares it.
    """

    def __init__(self, t, y, yerr, device=0, y_offset=None, own_engine=False):
        """``y_offset``: [L] frozen per-light-curve means; when None a model with a frozen
        constant mean supplies its value for every light curve.  ``own_engine``: a context of its own instead of
        the process-wide one of ``get_engine`` (for work that runs beside other work on the same GPU, from another
        host thread; ``close()`` frees it)."""
        self.t = np.ascontiguousarray(t, dtype=np.float64)
        self.y = np.atleast_2d(np.ascontiguousarray(y, dtype=np.float64))
        self.yerr = np.atleast_2d(np.ascontiguousarray(yerr, dtype=np.float64))
        if self.yerr.shape != self.y.shape:
            raise ValueError("dimension mismatch")
        self.device = device
        self.y_offset = None if y_offset is None else np.ascontiguousarray(y_offset, dtype=np.float64)
        self._model_sig = None
        self._bound_offset = None
        self._token = object()
        self._own = None
        self._own_is_mine = False
        if isinstance(own_engine, tuple) and own_engine and own_engine[0] == "side":
            self._own = get_side_engine(device, own_engine[1])      # ("side", k): the process's k-th extra context, reused
        elif own_engine:     # True, or (part, parts): a context of its own on that slice of the compute units
            self._own = _engine.Engine(device, cu_slice=None if own_engine is True else tuple(own_engine))
            self._own.bound_to = None
            self._own_is_mine = True

    def close(self):
        if self._own is not None:
            if self._own_is_mine:
                self._own.close()
            elif self._own.bound_to is self._token:
                self._own.bound_to = None       # a side engine goes back to the pool; its resident set is nobody's now
            self._own = None

    @property
    def n_lightcurves(self):
        return self.y.shape[0]

    def _bind_lightcurves(self):
        """The engine with these light curves resident and no particular model (tabulated spectra)."""
        class _NoModel:
            device_terms = ()
        return self._bind(_NoModel())

    def _bind(self, model):
        eng = self._own if self._own is not None else get_engine(self.device)
        offset = self.y_offset if self.y_offset is not None else getattr(model, "y_offset", None)
        key = None if offset is None else np.asarray(offset, dtype=np.float64).tobytes()
        if eng.bound_to is not self._token or key != self._bound_offset:
            eng.set_lightcurves(self.t, self.y, self.yerr, y_offset=offset)
            eng.bound_to = self._token
            self._bound_offset = key
            self._model_sig = None
        if model.device_terms:
            sig = model.signature()
            if sig != self._model_sig:
                eng.set_model(model.kinds, model.full, model.free_index, model.bounds,
                              mean_kind=model.mean_kind, extra=model.extra)
                self._model_sig = sig
        return eng

    def evaluate(self, model, theta, lc_index=None, add_prior=True):
        """theta [B][P] -> (lnP [B], status [B]); ``model`` is a DeviceModel."""
        if not model.device_terms:
            raise ValueError("batched evaluation needs device-expandable terms")
        if model.mean_kind is None:
            raise NotImplementedError("only constant and linear mean models run on the device")
        eng = self._bind(model)
        return eng.loglike(theta, lc_index, add_prior=add_prior)

    def evaluate_coefficients(self, model, coeffs, jitter, mean_params, lc_index=None):
        """Host-evaluated celerite coefficients [B][j]; mean_params [B][1 or 2] are the
        device-side mean parameters (0 for a frozen constant mean, see DeviceModel)."""
        eng = self._bind(model)
        ar, cr, ac, bc, cc, dc = coeffs
        return eng.loglike_coeffs(ar, cr, ac, bc, cc, dc, jitter=jitter, mean_kind=model.mean_kind,
                                  mean_params=mean_params, lc_index=lc_index)


class GP(ModelSet):
    """celerite.GP look-alike; parameters are ``kernel:*`` then ``mean:*``."""

    def __init__(self, kernel, mean=0.0, fit_mean=False, log_white_noise=None, fit_white_noise=False, device=0, own_engine=False):
        if not isinstance(kernel, Term):
            raise TypeError("kernel must be a mind_the_gaps_amd Term")
        if log_white_noise is not None:
            # celerite's extra diagonal model; the reference never passes one (its notebooks give fit_white_noise=False
            # with the default None, docs/notebooks/poisson_level.ipynb cell 5): white noise goes in as a JitterTerm
            raise NotImplementedError("log_white_noise: add a JitterTerm to the kernel instead")
        try:
            mean = ConstantModel(float(mean))
        except TypeError:
            if not isinstance(mean, Model):
                raise
        if not fit_mean:
            mean.freeze_all_parameters()
        super().__init__([("kernel", kernel), ("mean", mean)])
        self.device = device
        self._own_engine = own_engine    # a device context of this GP's own (work that runs beside other work, from another thread)
        self._t = None
        self._yerr = None
        self._evaluator = None
        self._y_bound = None

    # -- celerite.GP API ---------------------------------------------------------
    @property
    def mean(self):
        return self.models["mean"]

    @property
    def kernel(self):
        return self.models["kernel"]

    @property
    def computed(self):
        return self._t is not None

    def compute(self, t, yerr=1.123e-12, check_sorted=True):
        """Bind the sampling ``t`` and the per-point standard deviations ``yerr``."""
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        if t.ndim != 1:
            raise ValueError("dimension mismatch")
        if check_sorted and np.any(np.diff(t) < 0.0):
            raise ValueError("the input coordinates must be sorted")
        self._t = t
        self._yerr = np.empty_like(t)
        self._yerr[:] = yerr
        self._evaluator = None

    def _device_model(self):
        return DeviceModel(self.kernel, self.mean, self.mean.unfrozen_mask)

    def release_engine(self):
        """Give a context of this GP's own back (``own_engine=True``); later calls use the process-wide engine."""
        self._own_engine = False
        if self._evaluator is not None:
            self._evaluator.close()

    def _ensure_evaluator(self, y):
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")
        y = np.asarray(y, dtype=np.float64)
        if y.shape != self._t.shape:
            raise ValueError("dimension mismatch")
        if self._evaluator is None or self._y_bound is None or not np.array_equal(self._y_bound, y):
            if self._evaluator is not None:
                self._evaluator.close()
            self._evaluator = LogProbEvaluator(self._t, y, self._yerr, device=self.device, own_engine=self._own_engine)
            self._y_bound = y.copy()
        return self._evaluator

    def log_likelihood(self, y, quiet=False):
        """ln L of the current parameter vector (gpmodelling.py:152,169)."""
        ev = self._ensure_evaluator(y)
        model = self._device_model()
        if model.mean_kind is None:
            raise NotImplementedError("only constant and linear mean models run on the device")
        if model.device_terms:
            out, status = ev.evaluate(model, model.full[model.free_index][None, :], add_prior=False)
        else:
            coeffs = tuple(c[None, :] for c in self.kernel.coefficients)
            out, status = ev.evaluate_coefficients(
                model, coeffs, np.array([self.kernel.jitter]), model.full[model.nk:][None, :])
        if status[0] == _engine.ST_NOTPD:
            if quiet:
                return -np.inf
            raise LinAlgError("failed to factorize or solve matrix")
        return float(out[0])

    def log_probability_batch(self, theta, y, add_prior=True):
        """lnP of B free-parameter vectors at once -> (lnP[B], status[B]); pure in theta."""
        ev = self._ensure_evaluator(y)
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        model = self._device_model()
        if model.device_terms:
            return ev.evaluate(model, theta, add_prior=add_prior)
        return self._host_coefficient_batch(ev, model, theta, add_prior)

    def _host_coefficient_batch(self, ev, model, theta, add_prior):
        """User-defined Python terms: coefficients on the host, recurrence on the device."""
        saved = self.get_parameter_vector()
        B = theta.shape[0]
        rows, jit, means, keep = [], [], [], np.ones(B, dtype=bool)
        try:
            for b in range(B):
                self.set_parameter_vector(theta[b])
                if add_prior and not np.isfinite(self.log_prior()):
                    keep[b] = False
                    continue
                rows.append(self.kernel.coefficients)
                jit.append(self.kernel.jitter)
                mvec = self.mean.get_parameter_vector(include_frozen=True).copy()
                if model.y_offset is not None:
                    mvec[0] = 0.0      # folded into y at upload
                means.append(mvec)
        finally:
            self.set_parameter_vector(saved)
        out = np.full(B, -np.inf)
        status = np.full(B, _engine.ST_PRIOR, dtype=np.int32)
        if rows:
            shapes = {tuple(len(c) for c in r) for r in rows}
            if len(shapes) != 1:
                raise ValueError("host-evaluated terms must keep one structure across the batch")
            coeffs = tuple(np.array([r[i] for r in rows]).reshape(len(rows), -1) for i in range(6))
            o, s = ev.evaluate_coefficients(model, coeffs, np.array(jit), np.array(means))
            out[keep], status[keep] = o, s
        return out, status

    def _bound_engine(self, y):
        ev = self._ensure_evaluator(y)
        model = self._device_model()
        if not model.device_terms or model.mean_kind is None:
            raise NotImplementedError("needs device-expandable terms and a constant or linear mean")
        return ev._bind(model), model

    @staticmethod
    def _raise_for(status):
        if status == _engine.ST_NOTPD:
            raise LinAlgError("failed to factorize or solve matrix")
        if status == _engine.ST_PRIOR:
            raise ValueError("the current parameter vector has zero prior probability")

    def apply_inverse(self, y):
        """``K^-1 y`` for ``y`` of shape (N,) or (N, nrhs) (celerite.GP.apply_inverse), K the
        covariance at the current parameters: O(N J^2 + N J nrhs) on the device."""
        y = np.asarray(y, dtype=np.float64)
        if y.shape[0] != len(self._t):
            raise ValueError("dimension mismatch")
        eng, model = self._bound_engine(self._y_bound if self._y_bound is not None else np.zeros(len(self._t)))
        x, status = eng.apply_inverse(model.full[model.free_index], y)
        self._raise_for(status)
        return x

    def predict(self, y, t=None, return_cov=True, return_var=False):
        """Conditional mean and (co)variance, celerite.GP.predict.

        At the training times (``t=None``) with ``return_var=True`` -- the reference's call,
        gpmodelling.py:366 -- mean and variance both come from the O(N J^2) factorisation on the
        device (celerite forms the dense N x N cross-covariance for the variance).  At other times,
        or for the full covariance, celerite's own expressions ``mu = mean(t) + K_* K^-1 r``,
        ``cov = K_** - K_* K^-1 K_*^T`` are assembled on the host from ``apply_inverse`` (one
        device call with 1 + N_* right-hand sides).  Like celerite's, the variances are those of
        the noise-free process (no jitter, no measurement errors)."""
        if t is None and (return_var or not return_cov):
            eng, model = self._bound_engine(y)
            mu, var, status = eng.predict(model.full[model.free_index][None, :])
            self._raise_for(status[0])
            mu = mu[0] + (model.y_offset or 0.0)
            return (mu, var[0]) if return_var else mu
        eng, model = self._bound_engine(y)
        y = np.asarray(y, dtype=np.float64)
        xs = self._t if t is None else np.atleast_1d(np.asarray(t, dtype=np.float64))
        resid = y - self.mean.get_value(self._t)
        if not (return_cov or return_var):
            alpha, status = eng.apply_inverse(model.full[model.free_index], resid)
            self._raise_for(status)
            return self.mean.get_value(xs) + self.kernel.get_value(xs[:, None] - self._t[None, :]) @ alpha
        kxs = self.kernel.get_value(xs[:, None] - self._t[None, :])                # [N_*][N]
        sol, status = eng.apply_inverse(model.full[model.free_index], np.column_stack([resid, kxs.T]))
        self._raise_for(status)
        mu = self.mean.get_value(xs) + kxs @ sol[:, 0]
        if return_var:
            return mu, self.kernel.get_value(0.0) - np.sum(kxs.T * sol[:, 1:], axis=0)
        return mu, self.kernel.get_value(xs[:, None] - xs[None, :]) - kxs @ sol[:, 1:]
