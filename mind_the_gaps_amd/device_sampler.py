"""Device-resident ensemble sampler with emcee's read surface.

Same stretch move as ``sampler.EnsembleSampler`` (the emcee 3.1.4 semantics the
reference relies on, /root/reference/mind_the_gaps/gpmodelling.py:247-285), but the
walkers, the random numbers (Philox4x32-10) and the accept/reject step live on the
GPU (``mtg_ensemble_*`` in include/mtg.h): ``run_mcmc`` enqueues all its iterations
without a host round trip, which is what keeps a many-light-curve sweep at kernel speed.
The chain comes back in emcee's layout, so ``get_autocorr_time``, ``get_chain`` and
``get_log_prob`` behave as the reference expects.
"""
import numpy as np

from .sampler import integrated_time

__all__ = ["DeviceEnsembleSampler"]

# The convergence check runs on the whole chain every time it is made.  On the device (Engine.chain_autocorr) it
# costs ~1 ms where the host's FFTs take 10-200 ms (1000 x 256 x 8: 0.8 against 35 ms; 50 000 x 12 x 5: 1.1 against
# 53 ms) -- once hipFFT is up (~0.8 s the first time in a process; the sampler starts that on a helper thread when it
# starts sampling) and has a plan for the padded length: ~15 ms with the kernels in rocFFT's cache, ~1.2 s when
# rocFFT has to compile them first (it does that at run time for every new length on this ROCm build; the package
# ships a seed of the cache with the power-of-two lengths, engine._seed_rocfft_cache).  So every padded length is
# rented before it is bought: the host does the checks of a length until their ESTIMATED cost (a fixed rate per
# chain point -- not a measured time) has reached what the plan would cost, or a single check would -- at most
# twice the cost of always choosing right.  The choice is a pure function of the sequence of chain shapes checked
# so far: the same run takes the same path (and gets the same tau, to the last bit) every time, on every rank.
HOST_SECONDS_PER_POINT = 1.1e-8      # measured: 7 ms for 6.4e5 points, 30 ms for 2e6, 220 ms for 2e7


def _plan_seconds():
    from . import engine as _engine
    return 0.015 if _engine.rocfft_cache_seeded else 1.2


def _autocorr_time_where_it_is_cheapest(engine, chain, kwargs):
    """chain[n_t, W, P] -> tau[P]; or chain[n_t, E, W, P] (independent ensembles) -> tau[E, P]."""
    n_t = chain.shape[0]
    many = chain.ndim == 4

    def taus(rho):
        if not many:
            return integrated_time(chain, acf=None if rho is None else (lambda _x: rho), **kwargs)
        return np.array([integrated_time(chain[:, e], acf=None if rho is None else (lambda _x, e=e: rho[:, e]), **kwargs)
                         for e in range(chain.shape[1])])

    key = (1 << max(n_t - 1, 1).bit_length(),) + chain.shape[1:]
    # "planned": the shapes the library holds plans for, least recently used first (it keeps four: the tutorial's loop
    # checks the null and the alternative model's chains in turn)
    state = engine.__dict__.setdefault("_acf_state", {"planned": [], "rented": {}})
    estimate = HOST_SECONDS_PER_POINT * chain.size
    rented = state["rented"].get(key, 0.0)
    plan = _plan_seconds()
    on_device = n_t >= 2 and (key in state["planned"] or estimate > plan or rented > plan)
    if on_device:
        try:
            rho = engine.chain_autocorr(chain)   # (waits for the hipFFT warm-up thread if it is still at it)
        except Exception:        # (too large for the device's workspace, hipFFT refusing a plan ...): the host can always
            rho = None
        else:                    # (engine.chain_autocorr keeps state["planned"] itself)
            return taus(rho)
    state["rented"][key] = rented + estimate
    return taus(None)


class DeviceEnsembleSampler:
    """E lock-step ensembles of ``nwalkers`` walkers on one engine (E = 1 mirrors emcee).

    ``bind`` is a zero-argument callable returning the engine with the right light curves
    and model resident (``LogProbEvaluator._bind``); ``lc_of_ensemble`` maps ensembles to
    light curves (default: ensemble e -> light curve e, or all -> 0 for one light curve).
    """

    def __init__(self, bind, nwalkers, ndim, n_ensembles=1, lc_of_ensemble=None, seed=None,
                 store_chain=True, shard_group=False, shard_transport=None, index_base=0):
        if nwalkers < 2 * ndim:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than "
                               "twice the number of dimensions.")
        if nwalkers % 2:
            raise ValueError("the device sampler needs an even number of walkers")
        self._bind = bind
        self.nwalkers, self.ndim, self.E = int(nwalkers), int(ndim), int(n_ensembles)
        self.lc_of_ensemble = lc_of_ensemble
        # like emcee, reproducible from numpy's global state when no seed is given
        self.seed = int(np.random.randint(0, 2 ** 62)) if seed is None else int(seed)
        self.store_chain = store_chain
        # walker sharding: False = none; None = the default process group; or a torch.distributed group.  Every
        # rank then holds the same chain (distributed.shard_device_ensemble).
        self.shard_group, self.shard_transport = shard_group, shard_transport
        # global index of ensemble 0 (mtg_set_stream_base): with the same seed, ensembles [index_base, index_base + E)
        # of a job split over several GPUs get the random numbers they would get in one sampler holding them all
        self.index_base = int(index_base)
        self.iteration = 0
        self._chain = np.empty((0, self.E, self.nwalkers, self.ndim))
        self._log_prob = np.empty((0, self.E, self.nwalkers))
        self._started = False

    def run_mcmc(self, initial_state, nsteps):
        eng = self._bind()
        if self.store_chain:         # the convergence checks may want the device (see _autocorr_time_where_it_is_cheapest)
            eng.start_fft_warmup()
        if initial_state is not None:
            p0 = np.asarray(initial_state, dtype=np.float64)
            if p0.ndim == 2:
                p0 = p0[None]
            if p0.shape != (self.E, self.nwalkers, self.ndim):
                raise ValueError("incompatible input dimensions {0}".format(p0.shape))
            if not np.all(np.isfinite(p0)):
                raise ValueError("At least one parameter value was infinite or NaN")
            if self.shard_group is not False:
                from .distributed import broadcast_start, shard_device_ensemble
                p0, self.seed = broadcast_start(p0, self.seed, self.shard_group)
            self._init(eng, p0)
            if self.shard_group is not False:
                self.transport = shard_device_ensemble(eng, self.shard_group, self.shard_transport)
            self._started = True
            self.iteration = 0
            self._chain = np.empty((0, self.E, self.nwalkers, self.ndim))
            self._log_prob = np.empty((0, self.E, self.nwalkers))
        elif not self._started:
            raise ValueError("cannot continue a run that was never started")
        chain, lnp = eng.ensemble_run(int(nsteps), store_chain=self.store_chain)
        if self.store_chain:
            self._chain = np.concatenate([self._chain, chain], axis=0)
            self._log_prob = np.concatenate([self._log_prob, lnp], axis=0)
        self.iteration += int(nsteps)
        self._state = eng.ensemble_state()
        return self._state

    def _init(self, eng, coords):
        eng.set_stream_base(self.index_base)      # read by mtg_ensemble_init, kept with the resident ensembles
        try:
            eng.ensemble_init(coords, seed=self.seed, lc_of_ensemble=self.lc_of_ensemble)
        finally:
            eng.set_stream_base(0)

    # -- checkpoint / resume (the reference keeps its chains in memory only) -----------------
    def save(self, path):
        """Everything needed to continue this run in another process: state, Philox seed, iteration, chain."""
        np.savez(path, seed=np.uint64(self.seed), iteration=self.iteration, chain=self._chain, log_prob=self._log_prob,
                 index_base=self.index_base, shape=np.array([self.E, self.nwalkers, self.ndim]), **{"state_" + k: v for k, v in self._state.items()})

    def load(self, path):
        """Continue the run ``save`` wrote: the next ``run_mcmc(None, n)`` produces the iterations the original run
        would have produced (random numbers are functions of (seed, iteration, walker); the saved log-probabilities
        go back as they are -- evaluating the saved coordinates again, in one batch instead of half-steps, may take
        another kernel and differ in the last bits, enough to flip an accept decision)."""
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        if tuple(z["shape"]) != (self.E, self.nwalkers, self.ndim):
            raise ValueError("the checkpoint holds %s ensembles x walkers x parameters, this sampler %s"
                             % (tuple(z["shape"]), (self.E, self.nwalkers, self.ndim)))
        eng = self._bind()
        self.seed = int(z["seed"])
        self.index_base = int(z["index_base"]) if "index_base" in z.files else 0
        self._init(eng, z["state_coords"])
        eng.ensemble_restore(int(z["iteration"]), z["state_log_prob"], z["state_naccept"], z["state_best_log_prob"],
                             z["state_best_coords"])
        if self.shard_group is not False:
            from .distributed import shard_device_ensemble
            self.transport = shard_device_ensemble(eng, self.shard_group, self.shard_transport)
        self.iteration = int(z["iteration"])
        self._chain, self._log_prob = z["chain"], z["log_prob"]
        self._state = eng.ensemble_state()
        self._started = True
        return self._state

    # -- emcee-style views (ensemble 0 unless ``ensemble`` is given) --------------------
    def _get(self, arr, flat, thin, discard, ensemble):
        v = arr[discard + thin - 1::thin, ensemble]
        return v.reshape((-1,) + v.shape[2:]) if flat else v

    def get_chain(self, flat=False, thin=1, discard=0, ensemble=0):
        return self._get(self._chain, flat, thin, discard, ensemble)

    def get_log_prob(self, flat=False, thin=1, discard=0, ensemble=0):
        return self._get(self._log_prob, flat, thin, discard, ensemble)

    def get_autocorr_time(self, discard=0, thin=1, ensemble=0, broadcast=True, **kwargs):
        """emcee's ``get_autocorr_time``.  In a walker-sharded run (``shard_group`` given) this is a COLLECTIVE: every
        rank must call it, at the same point, or the callers wait for ever in the broadcast of rank 0's value (which
        is what keeps the ranks' convergence decisions identical).  ``broadcast=False``: this rank's own value, no
        communication -- for diagnostics from one rank only, never for a decision that changes what the ranks do
        next."""
        chain = self.get_chain(discard=discard, thin=thin, ensemble=ensemble)
        if "acf" not in kwargs:
            tau = thin * _autocorr_time_where_it_is_cheapest(self._bind(), chain, kwargs)
        else:
            tau = thin * integrated_time(chain, **kwargs)
        if self.shard_group is not False and broadcast:
            # every rank holds the same chain, but host and device FFTs agree to ~1e-11 only and a device failure
            # falls back to the host on that rank alone: the ranks must take the SAME convergence decision (one that
            # stops sampling while the others enter the next all-gather hangs the job), so rank 0's value is
            # everybody's
            from .distributed import broadcast_array
            tau = broadcast_array(tau, self.shard_group)
        return tau

    @property
    def acceptance_fraction(self):
        return self._state["naccept"] / float(max(self.iteration, 1))

    @property
    def state(self):
        """dict(coords, log_prob, best_log_prob, best_coords, naccept, iteration, n_not_pd)."""
        return self._state
