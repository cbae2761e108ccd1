"""``GPModelling``: the reference's workflow facade on the MI355X engine.

Drop-in mirror of /root/reference/mind_the_gaps/gpmodelling.py:23-539 for the
log-likelihood hot path: same constructor, method names, defaults, attributes
and error behaviour; every likelihood comes from the HIP kernels (gp.GP ->
engine.Engine -> libmtg_hip.so).  What changes underneath:

* ``_log_probability`` / ``_neg_log_like`` accept one theta (P,) -> float exactly
  like the reference, or a batch (B, P) -> array, which is one GPU launch;
* ``fit`` gives L-BFGS-B the value and the forward-difference gradient from ONE
  batched launch of P + 1 evaluations (scipy's own serial finite differences,
  gpmodelling.py:192, cost P + 1 sequential likelihood calls);
* ``derive_posteriors`` runs the stretch move with the whole half-ensemble per
  launch; ``cores`` (a multiprocessing.Pool size in the reference,
  gpmodelling.py:245) is accepted and ignored.
"""
import time
import warnings
from typing import List, Tuple

import numpy as np
from scipy.optimize import minimize

from . import engine as _engine
from . import walkers as _walkers
from .gp import GP, LinAlgError
from .lightcurves import GappyLightcurve
from .modeling import ConstantModel
from .models import GaussianModel, LinearModel
from .device_sampler import DeviceEnsembleSampler
from .sampler import EnsembleSampler

__all__ = ["GPModelling"]


class GPModelling:
    """The interface for Gaussian Process (GP) modelling on MI355X."""

    meanmodels = ["linear", "constant", "gaussian"]

    def __init__(self, lightcurve: GappyLightcurve, kernel, mean_model: str = None, device: int = 0,
                 quiet: bool = False, random_state=None, own_engine: bool = False):
        """GP of ``kernel`` (a ``mind_the_gaps_amd.terms.Term``) on ``lightcurve``, factorised once with
        ``yerr = dy + 1e-12`` as the reference does (gpmodelling.py:54).

        ``mean_model``: None keeps the mean frozen at the light curve's average; "constant", "linear" or
        "gaussian" name a mean that is fitted along with the kernel (the table in ``_build_mean_model``).
        New and optional: ``device`` = GPU ordinal; ``quiet`` = a covariance that is not positive definite gives
        -inf instead of ``LinAlgError`` (the reference raises, gpmodelling.py:152); ``random_state`` = a
        ``numpy.random.RandomState`` for the walkers' starting points and the device sampler's key instead of numpy's
        global generator (the reference's and emcee's source; a private one lets two models be sampled from two threads
        with the numbers ``np.random.seed`` would have given each); ``own_engine`` = a device context of this object's
        own (``gp.release_engine()`` gives it back)."""
        self._lightcurve = lightcurve
        meanmodel, fit_mean = self._build_mean_model(mean_model)
        self.gp = GP(kernel, mean=meanmodel, fit_mean=fit_mean, device=device, own_engine=own_engine)
        self._random = random_state
        self.gp.compute(self._lightcurve.times, np.asarray(self._lightcurve.dy, dtype=np.float64) + 1e-12)
        self.initial_params = self.gp.get_parameter_vector()
        self._ndim = len(self.initial_params)
        self._autocorr = []
        self._loglikelihoods = None
        self._mcmc_samples = None
        self._quiet = quiet
        self._fit_evaluations = 1   # fit() resets it: only the first evaluation of a fit may raise LinAlgError
        self._y = np.asarray(self._lightcurve.y, dtype=np.float64)

    # mean kind -> (model factory taking the light curve, fitted?).  The contract of gpmodelling.py:62-124: without a
    # name the mean is a ConstantModel frozen at the light curve's average, boxed by the data's range; a name selects a
    # fitted mean -- "constant" the same model, "linear" LinearModel(slope 0, intercept 1.5) without bounds (the
    # reference works slope estimates out and then does not use them), "gaussian" a bump centred on the middle epoch,
    # half the duration wide.  The reference hands GaussianModel three values for its four parameters, so that branch
    # raises ValueError there and here -- kept, not fixed.
    @staticmethod
    def _constant_mean(lc):
        return ConstantModel(lc.mean, bounds=[(np.min(lc.y), np.max(lc.y))])

    @staticmethod
    def _linear_mean(lc):
        return LinearModel(0, 1.5, bounds=[(-np.inf, np.inf)] * 2)

    @staticmethod
    def _gaussian_mean(lc):
        span, top, root2pi = lc.duration, np.max(lc.y), np.sqrt(2 * np.pi)
        width = span / 2
        return GaussianModel(lc.times[len(lc.times) // 2], width, (top - np.min(lc.y)) * root2pi * width,
                             bounds=[(lc.times[0], lc.times[-1]), (0, span), (top * root2pi * span, 50 * top * root2pi * span)])

    _MEAN_KINDS = {None: ("_constant_mean", False), "constant": ("_constant_mean", True),
                   "linear": ("_linear_mean", True), "gaussian": ("_gaussian_mean", True)}

    def _build_mean_model(self, meanmodel: str = None):
        """(mean model, fit_mean) for a mean kind (see the table above)."""
        kind = meanmodel if meanmodel is None else meanmodel.lower()
        if kind not in self._MEAN_KINDS:
            raise ValueError("Input mean model %s not implemented! Only \n %s \n are available"
                             % (meanmodel, "\t".join(GPModelling.meanmodels)))
        factory, fitted = self._MEAN_KINDS[kind]
        return getattr(self, factory)(self._lightcurve), fitted

    # -- the hot path --------------------------------------------------------------
    def _batch(self, params, add_prior):
        theta = np.asarray(params, dtype=np.float64)
        single = theta.ndim == 1
        out, status = self.gp.log_probability_batch(np.atleast_2d(theta), self._y, add_prior=add_prior)
        if not self._quiet and np.any(status == _engine.ST_NOTPD):
            raise LinAlgError("failed to factorize or solve matrix")
        return (float(out[0]) if single else out)

    def _log_probability(self, params):
        """Logarithm of the posterior: box prior + GP log-likelihood
        (gpmodelling.py:127-152).  ``params``: (P,) -> float or (B, P) -> array."""
        return self._batch(params, add_prior=True)

    def _neg_log_like(self, params):
        """Negative log-likelihood, no prior (gpmodelling.py:155-169)."""
        r = self._batch(params, add_prior=False)
        return -r

    def _neg_log_like_and_grad(self, x, lower, upper, step=1e-8):
        """-lnL and its forward-difference gradient from one batched launch.

        Same scheme scipy applies for L-BFGS-B without a jacobian (absolute step
        ``eps`` = 1e-8, flipped or shrunk where x + h would leave the bounds)."""
        x = np.asarray(x, dtype=np.float64)
        h = np.full_like(x, step)
        room_up, room_dn = upper - x, x - lower
        out_of_box = x + h > upper
        flip = out_of_box & (np.abs(h) <= np.maximum(room_dn, room_up))
        h[flip] *= -1.0
        shrink = out_of_box & ~flip
        h[shrink & (room_up >= room_dn)] = room_up[shrink & (room_up >= room_dn)]
        h[shrink & (room_up < room_dn)] = -room_dn[shrink & (room_up < room_dn)]
        pts = np.vstack([x[None, :], x[None, :] + np.diag(h)])
        # Inside the optimiser a numerically singular corner of the box (e.g. log_S0 at its
        # upper bound) is a very bad point, not a fatal one: celerite's LinAlgError would
        # abort the reference's fit there; here the line search simply backs off.
        out, status = self.gp.log_probability_batch(pts, self._y, add_prior=False)
        vals = np.where(status == _engine.ST_OK, -out, np.inf)
        f0 = float(vals[0])
        if not np.isfinite(f0):
            if self._fit_evaluations == 0 and not self._quiet:
                # the STARTING point itself cannot be factorised: returning a plateau would make
                # L-BFGS-B "converge" there at once; the reference raises here (gpmodelling.py:192)
                raise LinAlgError("failed to factorize or solve matrix")
            self._fit_evaluations += 1
            return 1e300, np.zeros_like(x)
        self._fit_evaluations += 1
        dx = pts[1:].diagonal() - x
        with np.errstate(divide="ignore", invalid="ignore"):
            grad = np.where((dx != 0.0) & np.isfinite(vals[1:]), (vals[1:] - f0) / dx, 0.0)
        return f0, grad

    def fit(self, initial_params=None):
        """L-BFGS-B minimisation of the negative log-likelihood within the
        parameter bounds (gpmodelling.py:172-194).  Returns scipy's OptimizeResult."""
        if initial_params is None:
            initial_params = self.initial_params
        bounds = self.gp.get_parameter_bounds()
        lower = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
        upper = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
        self._fit_evaluations = 0
        solution = minimize(self._neg_log_like_and_grad, initial_params, args=(lower, upper), jac=True,
                            method="L-BFGS-B", bounds=bounds)
        if not np.isfinite(solution.fun) or solution.fun >= 1e300:   # quiet mode: say so instead of "success"
            solution.success = False
            solution.message = "the covariance could not be factorised at the starting point"
            warnings.warn("fit(): " + solution.message)
        return solution

    def derive_posteriors(self, initial_chain_params=None, fit: bool = True, converge: bool = True,
                          max_steps: int = 10000, convergence_steps: int = 500, walkers: int = 12,
                          cores: int = 6, progress: bool = True, device_sampler: bool = None,
                          shard_walkers: bool = False, group=None):
        """Derive GP posteriors (gpmodelling.py:197-286): optional fit, walker
        spreading, stretch-move MCMC with an autocorrelation-time convergence check
        every ``convergence_steps`` iterations, then burn-in and thinning.

        ``device_sampler`` (new, optional): keep walkers, random numbers and the
        accept/reject step on the GPU (``mtg_ensemble_*``); the host only sees the chain
        every ``convergence_steps`` iterations.  Needs an even number of walkers and terms the
        device can expand.  Default (None): on whenever that holds -- 2-3x the iteration rate of
        the host-side sampler; like emcee's, the run is reproducible from ``np.random.seed``
        (the Philox key is drawn from numpy's global generator).  False: the host-side sampler
        with emcee's own use of numpy's global generator.

        ``shard_walkers`` (new, optional): inside a ``torch.distributed`` job (one process per
        GPU, every rank holding this light curve and calling this method with the same
        arguments) split every half-ensemble across the ranks: each GPU evaluates its rows,
        one all-gather of the log-probabilities per half-step (``group``: the process group,
        default the world), identical accept/reject everywhere -- every rank ends with the
        same chain.  With the device sampler the exchange is an ``ncclAllGather`` on the engine's
        stream (``mtg_ensemble_shard_rccl``; a host-staged exchange over ``group`` when its backend
        is not nccl); with the host-side sampler (``device_sampler=False``) it is
        ``distributed.WalkerShardedLogProb``.  Rank 0's starting ensemble and random state / seed are
        broadcast first."""
        if device_sampler is None:
            model = self.gp._device_model()
            device_sampler = walkers % 2 == 0 and bool(model.device_terms) and model.mean_kind is not None
        if initial_chain_params is None:
            if not fit:
                initial_params = self.initial_params
            else:
                solution = self.fit(self.initial_params)
                initial_params = solution.x
            initial_chain_params = self.spread_walkers(walkers, initial_params,
                                                       np.array(self.gp.get_parameter_bounds()))
        every_samples = convergence_steps
        old_tau = np.inf
        self.converged = False
        tau = None
        started = time.perf_counter()
        if device_sampler:
            sampler = self._device_sampler(walkers, shard_group=group if shard_walkers else False)
            first, done = initial_chain_params, 0

            def iterations():
                nonlocal first, done
                while done < max_steps:
                    chunk = min(every_samples - done % every_samples, max_steps - done)
                    sampler.run_mcmc(first, chunk)
                    if not self._quiet and sampler.state["n_not_pd"]:
                        raise LinAlgError("failed to factorize or solve matrix")
                    first, done = None, done + chunk
                    yield done
            steps_iter = iterations()
        else:
            log_prob_fn = self._log_probability
            if shard_walkers:
                from .distributed import WalkerShardedLogProb, lockstep
                log_prob_fn = WalkerShardedLogProb(self._log_probability, group=group)
            sampler = EnsembleSampler(walkers, self._ndim, log_prob_fn, vectorize=True)
            if shard_walkers:
                initial_chain_params = lockstep(sampler, initial_chain_params, group=group)
            steps_iter = sampler.sample(initial_chain_params, iterations=max_steps, progress=progress)
        for sample in steps_iter:
            if sampler.iteration % every_samples:
                continue
            # tol=0: always get an estimate, even an untrustworthy one
            tau = sampler.get_autocorr_time(tol=0)
            if shard_walkers and not device_sampler:   # (the device sampler broadcasts rank 0's value itself)
                from .distributed import broadcast_array
                tau = broadcast_array(tau, group)
            self.autocorr.append(np.mean(tau))
            if np.all(tau * 100 < sampler.iteration) and np.all(np.abs(old_tau - tau) / tau < 0.01) and converge:
                print("Convergence reached after %d samples!" % sampler.iteration)
                self.converged = True
                break
            old_tau = tau
        if tau is None:
            # max_steps < convergence_steps: the reference hits a NameError here
            # (SURVEY.md Appendix C.1); estimate once from the chain we have
            tau = sampler.get_autocorr_time(tol=0)
        self._tau = tau
        mean_tau = np.mean(tau)
        if not np.isfinite(mean_tau):
            # a parameter that never moved (very short chains) has an undefined autocorrelation
            # time; the reference would crash on int(nan) below
            warnings.warn("The autocorrelation time could not be estimated; keeping every sample")
            mean_tau = 0.0
        if not self.converged:
            warnings.warn(f"The chains did not converge after {sampler.iteration} iterations!")
            thin = int(mean_tau / 4)
            discard = int(mean_tau) * 5
        else:
            discard = int(mean_tau * 40)
            if discard > max_steps:
                discard = int(mean_tau * 10)
            thin = int(mean_tau / 2)
        thin = max(thin, 1)  # int(mean_tau / k) can be 0 for short chains (Appendix C.1)
        discard = min(max(discard, 0), max(sampler.iteration - thin, 0))
        self._loglikelihoods = sampler.get_log_prob(discard=discard, thin=thin, flat=True)
        self._mcmc_samples = sampler.get_chain(discard=discard, thin=thin, flat=True)
        self._sampler = sampler
        # metrics: log-probability evaluations per second of this run, convergence checks included
        self.evals_per_second = sampler.iteration * walkers / max(time.perf_counter() - started, 1e-12)

    def _device_sampler(self, walkers, shard_group=False):
        """One device-resident ensemble on this light curve (see device_sampler.py)."""
        ev = self.gp._ensure_evaluator(self._y)
        model = self.gp._device_model()
        if not model.device_terms or model.mean_kind is None:
            raise ValueError("the device sampler needs device-expandable terms and a constant or linear mean")
        seed = None if self._random is None else int(self._random.randint(0, 2 ** 62))   # (None: numpy's global generator)
        return DeviceEnsembleSampler(lambda: ev._bind(model), walkers, self._ndim, n_ensembles=1,
                                     shard_group=shard_group, seed=seed)

    def spread_walkers(self, walkers: int, parameters, bounds: List[Tuple[float, float]],
                       percent: float = 0.1, max_attempts: int = 20):
        """Spread the walkers with a Gaussian around ``parameters`` (gpmodelling.py:289-350).
        Draws from numpy's global generator in the reference's order (walker by walker,
        ``walkers.spread_reference_order``): after ``np.random.seed(k)`` the array returned is
        the reference's, value for value (tests/golden/spread_golden.npz, made by the
        reference's function; tests/test_spread_golden.py)."""
        bounds = np.array([(-np.inf if lower is None else lower, np.inf if upper is None else upper)
                           for lower, upper in bounds], dtype=np.float64)
        normal = np.random.normal if getattr(self, "_random", None) is None else self._random.normal
        return _walkers.spread_reference_order(normal, np.asarray(parameters, dtype=np.float64), bounds[:, 0],
                                               bounds[:, 1], walkers, percent=percent, max_attempts=max_attempts)

    def standarized_residuals(self, include_noise: bool = True):
        """Standardised residuals (gpmodelling.py:353-370) need ``GP.predict`` -- outside
        the log-likelihood hot path."""
        pred_mean, pred_var = self.gp.predict(self._lightcurve.y, return_var=True, return_cov=False)
        if include_noise:
            pred_var += self.gp.kernel.jitter
        return (self._lightcurve.y - pred_mean) / np.sqrt(pred_var)

    def get_rstat(self, burnin: int = None):
        """Gelman-Rubin-like statistic as the reference computes it
        (gpmodelling.py:373-403): within-chain over total variance."""
        if getattr(self, "_sampler", None) is None:
            raise ValueError("Posteriors have not been derived. Please run derive_posteriors prior to "
                             "populate the attributes.")
        if burnin is None:
            burnin = int(np.mean(self.tau)) * 10
        samples = self._sampler.get_chain(discard=burnin)
        whithin_chain_variances = np.var(samples, axis=0)
        samples = self._sampler.get_chain(flat=True, discard=burnin)
        between_chain_variances = np.var(samples, axis=0)
        return whithin_chain_variances / between_chain_variances[np.newaxis, :]

    # -- result accessors (gpmodelling.py:405-475) -----------------------------------
    _NOT_DERIVED = ("Posteriors have not been derived. Please run derive_posteriors prior to "
                    "populate the attributes.")

    @property
    def loglikelihoods(self):
        if self._loglikelihoods is None:
            raise AttributeError(self._NOT_DERIVED)
        return self._loglikelihoods

    @property
    def autocorr(self):
        return self._autocorr

    @property
    def sampler(self):
        if self._loglikelihoods is None:
            raise AttributeError(self._NOT_DERIVED)
        return self._sampler

    @property
    def mcmc_samples(self):
        if self._mcmc_samples is None:
            raise AttributeError(self._NOT_DERIVED)
        return self._mcmc_samples

    @property
    def max_loglikelihood(self):
        if self._loglikelihoods is None:
            raise AttributeError(self._NOT_DERIVED)
        return np.max(self._loglikelihoods)

    @property
    def best_loglikelihood(self):
        """Largest log-posterior over EVERYTHING the chains stored, burn-in included (new; the
        reference has only ``max_loglikelihood``, the maximum over the burned-in, thinned chain).  It
        is the estimator the lock-step refits of ``ppp.derive_posteriors_batch`` keep when they do not
        store chains, so a likelihood-ratio test compares like with like (``ppp.protassov_test``)."""
        if self._loglikelihoods is None:
            raise AttributeError(self._NOT_DERIVED)
        best = float(np.max(self._sampler.get_log_prob(flat=True)))
        state = getattr(self._sampler, "state", None)
        if isinstance(state, dict) and "best_log_prob" in state:   # the device sampler also saw the starting ensemble
            best = max(best, float(np.max(state["best_log_prob"])))
        return best

    @property
    def max_parameters(self):
        if self._mcmc_samples is None:
            raise AttributeError(self._NOT_DERIVED)
        return self._mcmc_samples[np.argmax(self._loglikelihoods)]

    @property
    def median_parameters(self):
        if self._mcmc_samples is None:
            raise AttributeError(self._NOT_DERIVED)
        return np.median(self._mcmc_samples, axis=0)

    @property
    def parameter_names(self):
        return self.gp.get_parameter_names()

    @property
    def k(self) -> int:
        return self._ndim

    @property
    def tau(self):
        if self._mcmc_samples is None:
            raise AttributeError(self._NOT_DERIVED)
        return self._tau

    def generate_from_posteriors(self, nsims: int = 10, cpus: int = 8, pdf: str = "Gaussian",
                                 extension_factor: int = 2, sigma_noise=None):
        """Light curves drawn from the MCMC posteriors (gpmodelling.py:478-539): ``nsims``
        random posterior samples, one Timmer & Koenig realisation each on this light
        curve's sampling, noise and error bars from its exposures (or ``sigma_noise``).
        All simulations run in one device call (``cpus`` is accepted and ignored)."""
        if self._mcmc_samples is None:
            raise RuntimeError("Posteriors have not been derived. Please run derive_posteriors prior to "
                               "calling this method.")
        if nsims >= len(self._mcmc_samples):
            warnings.warn("The number of simulation requested (%d) is higher than the number of posterior "
                          "samples (%d), so many samples will be drawn more than once"
                          % (nsims, len(self._mcmc_samples)))
        from .simulator import Simulator
        param_samples = self._mcmc_samples[np.random.randint(len(self._mcmc_samples), size=nsims)]
        lc = self._lightcurve
        simulator = Simulator(self.gp.kernel, lc.times, lc.exposures, lc.mean, pdf, lc.bkg_rate, lc.bkg_rate_err,
                              sigma_noise=sigma_noise, extension_factor=extension_factor,
                              random_state=np.random.randint(0, 2 ** 31 - 1), device=self.gp.device)
        nk = self.gp.kernel.vector_size               # the PSD depends on the kernel parameters only
        out = simulator.simulate(param_samples[:, :nk])
        return [GappyLightcurve(lc.times, out["rates"][i], out["dy"][i]) for i in range(nsims)]
