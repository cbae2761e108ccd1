"""Lock-step posteriors for MANY light curves: the Protassov loop as a batch axis.

The reference refits every simulated light curve one after the other
(/root/reference/docs/notebooks/tutorial_ppp.ipynb:326-343):

    for lc in lcs:
        gpm = GPModelling(lc, kernel)
        gpm.derive_posteriors(fit=True, max_steps=500, walkers=2 * cpus, cores=cpus)
        likelihoods.append(gpm.max_loglikelihood)

Every light curve is an independent problem of identical shape, so here all L
ensembles advance together: one stretch-move half-step of every ensemble is ONE
launch of L * W/2 evaluations (`LogProbEvaluator.evaluate` over the L resident
light curves).  The move, the acceptance rule, the autocorrelation estimate and
the burn-in / thinning arithmetic are those of `sampler.EnsembleSampler` and
`GPModelling.derive_posteriors` (gpmodelling.py:197-286), applied per light curve;
random numbers come from one vectorised generator instead of L private streams.
"""
import warnings

import numpy as np

from . import engine as _engine
from . import walkers as _walkers
from .gp import DeviceModel, LinAlgError, LogProbEvaluator
from .modeling import ConstantModel
from .sampler import integrated_time

__all__ = ["EnsembleBatchSampler", "BatchPosteriors", "derive_posteriors_batch", "batched_minimize",
           "protassov_test", "derive_posteriors_sharded"]


class EnsembleBatchSampler:
    """L independent affine-invariant ensembles (stretch move, a = 2) in lock-step.

    ``log_prob_fn(coords[B, ndim], lc_index[B]) -> lnP[B]``.  ``store_chain=False`` keeps
    only the running best sample of every light curve (what the LRT needs) instead of
    the [steps, L, W, ndim] chain.
    """

    def __init__(self, nlc, nwalkers, ndim, log_prob_fn, seed=None, a=2.0, store_chain=True):
        if nwalkers < 2 * ndim:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than "
                               "twice the number of dimensions.")
        if nwalkers % 2:
            raise ValueError("the lock-step sampler needs an even number of walkers")
        self.L, self.W, self.ndim = int(nlc), int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.a = float(a)
        self.rng = np.random.default_rng(seed)
        self.store_chain = store_chain
        self.iteration = 0
        self._chain = []
        self._lnp = []
        self.coords = None
        self.lnp = None
        self.best_lnp = np.full(self.L, -np.inf)
        self.best_coords = np.full((self.L, self.ndim), np.nan)
        self.n_accepted = np.zeros((self.L, self.W))

    def _evaluate(self, coords):
        """coords [L, K, ndim] -> lnP [L, K] in one call."""
        L, K, _ = coords.shape
        flat = coords.reshape(L * K, self.ndim)
        if not np.all(np.isfinite(flat)):
            raise ValueError("At least one parameter value was infinite or NaN")
        lc = np.repeat(np.arange(L, dtype=np.int32), K)
        lnp = np.asarray(self.log_prob_fn(flat, lc), dtype=np.float64)
        if np.any(np.isnan(lnp)):
            raise ValueError("Probability function returned NaN")
        return lnp.reshape(L, K)

    def _track_best(self):
        j = np.argmax(self.lnp, axis=1)
        cand = self.lnp[np.arange(self.L), j]
        better = cand > self.best_lnp
        self.best_lnp[better] = cand[better]
        self.best_coords[better] = self.coords[np.arange(self.L), j][better]

    def run(self, initial_state, steps, progress=False):
        """Advance every ensemble ``steps`` iterations from ``initial_state`` [L, W, ndim]
        (or continue when it is None)."""
        L, W, ndim, half = self.L, self.W, self.ndim, self.W // 2
        if initial_state is not None:
            p0 = np.array(initial_state, dtype=np.float64)
            if p0.shape != (L, W, ndim):
                raise ValueError("incompatible input dimensions {0}".format(p0.shape))
            self.coords = p0
            self.lnp = self._evaluate(p0)
            self._track_best()
        rows = np.arange(L)[:, None]
        for _ in range(int(steps)):
            # red/blue split, independently shuffled for every light curve
            order = np.argsort(self.rng.random((L, W)), axis=1)
            red, blue = order[:, :half], order[:, half:]
            for first, other in ((red, blue), (blue, red)):
                s = self.coords[rows, first]                                   # [L, half, ndim]
                zz = ((self.a - 1.0) * self.rng.random((L, half)) + 1.0) ** 2 / self.a
                partner = other[rows, self.rng.integers(half, size=(L, half))]
                c = self.coords[rows, partner]
                q = c - (c - s) * zz[:, :, None]
                new_lnp = self._evaluate(q)                                    # <- ONE launch
                lnpdiff = (ndim - 1.0) * np.log(zz) + new_lnp - self.lnp[rows, first]
                accept = lnpdiff > np.log(self.rng.random((L, half)))
                li, wi = np.nonzero(accept)
                tgt = first[li, wi]
                self.coords[li, tgt] = q[li, wi]
                self.lnp[li, tgt] = new_lnp[li, wi]
                self.n_accepted[li, tgt] += 1
            self.iteration += 1
            self._track_best()
            if self.store_chain:
                self._chain.append(self.coords.copy())
                self._lnp.append(self.lnp.copy())
        return self.coords, self.lnp

    # -- per-light-curve views -------------------------------------------------------
    def get_chain(self, lc=None):
        chain = np.asarray(self._chain)                     # [steps, L, W, ndim]
        return chain if lc is None else chain[:, lc]

    def get_log_prob(self, lc=None):
        lnp = np.asarray(self._lnp)
        return lnp if lc is None else lnp[:, lc]

    def get_autocorr_time(self, tol=0):
        """tau[L, ndim]: integrated autocorrelation time per light curve and parameter."""
        chain = self.get_chain()
        return np.array([integrated_time(chain[:, l], tol=tol, quiet=True) for l in range(self.L)])

    @property
    def acceptance_fraction(self):
        return self.n_accepted / float(max(self.iteration, 1))


def batched_minimize(fun, x0, lower, upper, max_iter=60, history=8, fd_step=1e-6, gtol=1e-5, ftol=1e-10):
    """Projected L-BFGS for L independent box-constrained problems in lock-step.

    ``fun(X[M, P], lc[M]) -> f[M]``; ``x0`` [L, P]; ``lower``/``upper`` [P].  Each
    iteration costs one launch for the forward-difference gradients (L * (P + 1)
    evaluations) plus one per backtracking round (L evaluations).  It plays the role of
    ``scipy.optimize.minimize(method="L-BFGS-B")`` in GPModelling.fit (gpmodelling.py:192)
    for the lock-step driver: a good starting point for the walkers, not a bit-identical
    optimiser path.  Returns (x[L, P], f[L], iterations).
    """
    x = np.clip(np.array(x0, dtype=np.float64), lower, upper)
    L, P = x.shape
    lcs = np.arange(L, dtype=np.int32)

    def value_and_grad(x):
        h = np.where(x + fd_step > upper, -fd_step, fd_step)            # step inward at the upper bound
        pts = np.repeat(x[:, None, :], P + 1, axis=1)                   # [L, P+1, P]
        pts[:, 1:, :] += np.eye(P)[None] * h[:, None, :]
        vals = fun(pts.reshape(-1, P), np.repeat(lcs, P + 1)).reshape(L, P + 1)
        return vals[:, 0], (vals[:, 1:] - vals[:, :1]) / h

    f, g = value_and_grad(x)
    S, Y = [], []
    active = np.ones(L, dtype=bool)
    it = 0
    for it in range(1, max_iter + 1):
        # projected gradient: components pushing out of the box are dropped
        blocked = ((x <= lower) & (g > 0)) | ((x >= upper) & (g < 0))
        pg = np.where(blocked, 0.0, g)
        active &= np.max(np.abs(pg), axis=1) > gtol
        if not active.any():
            break
        # two-loop recursion, vectorised over the L problems
        q = pg.copy()
        alphas = []
        for s, yv in zip(reversed(S), reversed(Y)):
            rho = 1.0 / np.maximum(np.einsum("lp,lp->l", yv, s), 1e-300)
            a = rho * np.einsum("lp,lp->l", s, q)
            q -= a[:, None] * yv
            alphas.append((a, rho))
        if S:
            sy = np.einsum("lp,lp->l", S[-1], Y[-1])
            yy = np.maximum(np.einsum("lp,lp->l", Y[-1], Y[-1]), 1e-300)
            q *= np.where(sy > 0, sy / yy, 1.0)[:, None]
        for (a, rho), s, yv in zip(reversed(alphas), S, Y):
            b = rho * np.einsum("lp,lp->l", yv, q)
            q += (a - b)[:, None] * s
        d = -np.where(blocked, 0.0, q)
        bad_dir = np.einsum("lp,lp->l", d, pg) >= 0                      # not a descent direction
        d[bad_dir] = -pg[bad_dir]
        # backtracking (Armijo) on the projected path: steps 1, 1/2, 1/4, ... (twenty of them), every problem takes
        # the first that passes.  The steps are tried in THREE launches, not twenty -- 1; then 1/2, 1/4, 1/8 for whoever
        # failed; then all the rest for the stubborn few -- because a round is a launch of a handful of rows (a
        # latency, ~0.2 ms) and the stragglers of 250 light curves kept 13 rounds per iteration going: 715 of the 775
        # launches of a fit.  Same trial points, same test, same choice as one step per round.
        x_new, f_new = x.copy(), f.copy()
        todo = active.copy()
        for ks in ((0,), (1, 2, 3), tuple(range(4, 20))):
            if not todo.any():
                break
            idx = np.flatnonzero(todo)
            steps = 0.5 ** np.asarray(ks, dtype=np.float64)
            trial = np.clip(x[idx, None, :] + steps[None, :, None] * d[idx, None, :], lower, upper)      # [rows, steps, P]
            ft = fun(trial.reshape(-1, P), np.repeat(lcs[idx], len(ks))).reshape(len(idx), len(ks))
            ok = np.isfinite(ft) & (ft <= f[idx, None] + 1e-4 * np.einsum("lp,lkp->lk", pg[idx], trial - x[idx, None, :]))
            passed = ok.any(axis=1)
            first = np.argmax(ok, axis=1)                                 # the largest step that passes
            rows = np.flatnonzero(passed)
            x_new[idx[rows]] = trial[rows, first[rows]]
            f_new[idx[rows]] = ft[rows, first[rows]]
            todo[idx[rows]] = False
        active &= ~todo                                                   # line search failed: stop there
        f_old = f
        g_old = g
        f_try, g_try = value_and_grad(x_new)
        # The line search and the gradient batch differ in size and may run on different solver families (time-parallel
        # scan / serial sweep), which can disagree on positive-definiteness at the very edge: a point the line search
        # accepted and the gradient batch rejects is not taken.
        lost = ~np.isfinite(f_try) | ~np.all(np.isfinite(g_try), axis=1)
        if lost.any():
            x_new[lost], f_try[lost], g_try[lost] = x[lost], f[lost], g[lost]
        moved = np.any(x_new != x, axis=1)
        s_vec, y_vec = x_new - x, g_try - g_old
        good = np.einsum("lp,lp->l", s_vec, y_vec) > 1e-12
        S.append(np.where(good[:, None], s_vec, 0.0))
        Y.append(np.where(good[:, None], y_vec, 0.0))
        if len(S) > history:
            S.pop(0), Y.pop(0)
        x, f, g = x_new, f_try, g_try
        active &= moved & (np.abs(f_old - f) > ftol * np.maximum(np.abs(f), 1.0))
    return x, f, it


class BatchPosteriors:
    """Per-light-curve results of :func:`derive_posteriors_batch` (the accessors of
    GPModelling, gpmodelling.py:405-475, with a leading light-curve axis)."""

    def __init__(self, sampler, tau, discard, thin, fit_params, fit_loglike, parameter_names):
        self.sampler = sampler
        self.tau = tau                              # [L, ndim]
        self.discard, self.thin = discard, thin     # [L]
        self.fit_parameters = fit_params            # [L, ndim] or None
        self.fit_loglikelihood = fit_loglike        # [L] or None
        self.parameter_names = parameter_names
        L = sampler.L
        if sampler.store_chain:
            lnp = sampler.get_log_prob()            # [steps, L, W]
            chain = sampler.get_chain()
            self.max_loglikelihood = np.empty(L)
            self.max_parameters = np.empty((L, sampler.ndim))
            self.median_parameters = np.empty((L, sampler.ndim))
            for l in range(L):
                sl = slice(discard[l] + thin[l] - 1, None, thin[l])
                lp = lnp[sl, l].reshape(-1)
                ch = chain[sl, l].reshape(-1, sampler.ndim)
                k = int(np.argmax(lp))
                self.max_loglikelihood[l] = lp[k]
                self.max_parameters[l] = ch[k]
                self.median_parameters[l] = np.median(ch, axis=0)
        else:
            self.max_loglikelihood = sampler.best_lnp.copy()
            self.max_parameters = sampler.best_coords.copy()
            self.median_parameters = None


# rows per half-step of a WHOLE set of light curves up to which derive_posteriors_batch(index_base=...) keeps the chains on
# the batch-independent time-parallel kernel (mtg_set_time_parallel 3); see its docstring
REPRODUCIBLE_TP_ROWS = 16384


def _spread(rng, centers, lower, upper, walkers, percent=0.1, max_attempts=20):
    """spread_walkers (gpmodelling.py:289-350) for every light curve at once."""
    return _walkers.spread(rng.normal, centers, lower, upper, walkers, percent=percent, max_attempts=max_attempts)


class _DeviceBatch:
    """Adapter giving DeviceEnsembleSampler the read surface BatchPosteriors uses."""

    def __init__(self, dev):
        self.dev = dev
        self.L, self.W, self.ndim = dev.E, dev.nwalkers, dev.ndim
        self.store_chain = dev.store_chain
        self.iteration = dev.iteration
        self.best_lnp = dev.state["best_log_prob"]
        self.best_coords = dev.state["best_coords"]
        self.n_accepted = dev.state["naccept"]

    def get_chain(self, lc=None):
        return self.dev._chain if lc is None else self.dev._chain[:, lc]

    def get_log_prob(self, lc=None):
        return self.dev._log_prob if lc is None else self.dev._log_prob[:, lc]

    def get_autocorr_time(self, tol=0):
        from .device_sampler import _autocorr_time_where_it_is_cheapest
        return _autocorr_time_where_it_is_cheapest(self.dev._bind(), self.dev._chain, dict(tol=tol, quiet=True))

    @property
    def acceptance_fraction(self):
        return self.n_accepted / float(max(self.iteration, 1))


def derive_posteriors_batch(times, Y, DY, kernel, walkers=12, max_steps=500, fit=True, seed=None,
                            device=0, store_chain=True, initial_params=None, quiet=False,
                            evaluate=None, device_sampler=True, own_engine=False, index_base=None, before_sampling=None,
                            total_lightcurves=None):
    """GPModelling(lc, kernel).derive_posteriors(...) for L light curves at once.

    times [N] (shared sampling, gpmodelling.py:538); Y, DY [L, N]; ``kernel`` a
    mind_the_gaps_amd Term (its current parameter vector is the common starting point,
    as in the tutorial loop).  The mean of every light curve is frozen at its own average
    (the reference default, gpmodelling.py:83-87).  ``evaluate`` overrides the engine call
    ``(theta[B, P], lc[B], add_prior) -> (lnP, status)`` (used by the multi-GPU driver);
    ``device_sampler`` keeps the L ensembles on the GPU between iterations
    (``mtg_ensemble_*``) instead of proposing and accepting on the host.  ``own_engine``: a device context for this
    call alone (closed before it returns), so that several calls can run side by side from different host threads.

    ``before_sampling``: called once, between the starting fit and the chains (two calls that run side by side meet there,
    so that their chains -- the long, regular part -- overlap from the first iteration to the last).

    ``index_base`` (an integer; needs ``seed`` and the device sampler): these are light curves [index_base,
    index_base + L) of a larger set that is being fitted in blocks, and every one of them must get the result it
    would get in ONE call for the whole set, bit for bit -- whatever the blocks.  Three things then stop depending on
    L: the walkers' starting points (one generator per light curve, keyed by seed and global index), the sampler's
    random numbers (Philox counters by global ensemble index, ``mtg_set_stream_base``) and the kernels (the number
    of rows in a batch otherwise picks between kernels whose sums differ in the last bits: the starting fit runs on
    the one-wave time-parallel kernel whatever the batch; the chains on the one-lane sweep and its pipelined form,
    which agree bit for bit -- or, when the WHOLE set is small, on that one-wave kernel as well;
    ``mtg_set_time_parallel`` 3 and 0).  ``total_lightcurves``: the size of the whole set, which is what that choice
    may depend on (not L, the block's): up to ``REPRODUCIBLE_TP_ROWS`` rows per half-step of the whole set every
    block, whatever the split, is inside the range where the time-parallel kernel is the fast one (0.35-3.6 ms against
    2.4-3.4 ms for the sweep at N = 1e4); unknown or larger, the sweep.  Models of rank above 6 keep the sweep in
    either mode (their time-parallel path sizes its chunks by the batch).
    """
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    DY = np.atleast_2d(np.asarray(DY, dtype=np.float64))
    L = Y.shape[0]
    model = DeviceModel(kernel, ConstantModel(0.0), np.zeros(1, dtype=bool))
    if not model.device_terms:
        raise ValueError("the lock-step driver needs device-expandable terms")
    model.y_offset = None                         # offsets are per light curve, owned by the evaluator
    import time
    clock = [("start", time.perf_counter())]
    ev = None
    if evaluate is None:
        ev = LogProbEvaluator(times, Y, DY + 1e-12, device=device, y_offset=Y.mean(axis=1), own_engine=own_engine)
        clock.append(("upload", time.perf_counter()))

        def evaluate(theta, lc, add_prior):
            return ev.evaluate(model, theta, lc, add_prior=add_prior)

    def checked(theta, lc, add_prior):
        out, status = evaluate(theta, lc, add_prior)
        if not quiet and np.any(status == _engine.ST_NOTPD):
            raise LinAlgError("failed to factorize or solve matrix")
        return out

    P = len(model.free_index)
    lower, upper = model.bounds[model.free_index, 0], model.bounds[model.free_index, 1]
    start = model.full[model.free_index] if initial_params is None else np.asarray(initial_params, float)
    centers = np.broadcast_to(start, (L, P)).copy()
    fit_x = fit_f = None
    block_free = index_base is not None
    if block_free and (seed is None or ev is None or not device_sampler):
        raise ValueError("index_base needs a seed and the device-resident sampler on this call's own evaluator")

    def kernels(mode):          # which kernel family the evaluator's engine may pick from (see the docstring)
        if block_free:
            ev._bind(model).set_time_parallel(mode)

    chain_mode = 3 if (total_lightcurves is not None
                       and int(total_lightcurves) * (walkers // 2) <= REPRODUCIBLE_TP_ROWS) else 0

    try:
        if fit:
            kernels(3)
            fit_x, fit_f, _ = batched_minimize(lambda x, lc: -checked(x, lc, False), centers, lower, upper)
            centers, fit_f = fit_x, -fit_f
            clock.append(("fit", time.perf_counter()))
        kernels(chain_mode)
        if block_free:
            p0 = np.concatenate([_spread(np.random.default_rng([int(seed), 1, int(index_base) + l]), centers[l:l + 1],
                                         lower, upper, walkers) for l in range(L)])
            sampler_seed = int(np.random.default_rng([int(seed), 2]).integers(0, 2 ** 62))
        else:
            rng = np.random.default_rng(seed)
            p0 = _spread(rng, centers, lower, upper, walkers)
        clock.append(("spread", time.perf_counter()))
        if before_sampling is not None:
            before_sampling()
            clock.append(("wait", time.perf_counter()))
        if device_sampler and ev is not None:
            from .device_sampler import DeviceEnsembleSampler
            dev = DeviceEnsembleSampler(lambda: ev._bind(model), walkers, P, n_ensembles=L,
                                        seed=sampler_seed if block_free else int(rng.integers(0, 2 ** 62)),
                                        store_chain=store_chain, index_base=index_base or 0)
            dev.run_mcmc(p0, max_steps)
            if not quiet and dev.state["n_not_pd"]:
                raise LinAlgError("failed to factorize or solve matrix")
            sampler = _DeviceBatch(dev)
        else:
            sampler = EnsembleBatchSampler(L, walkers, P, lambda x, lc: checked(x, lc, True), seed=rng,
                                           store_chain=store_chain)
            sampler.run(p0, max_steps)
    finally:
        kernels(2)
    clock.append(("sample", time.perf_counter()))
    if store_chain:
        tau = sampler.get_autocorr_time(tol=0)
        mean_tau = np.mean(tau, axis=1)
        # not-converged branch of gpmodelling.py:272-276 (a fixed-length PPP run never "converges")
        thin = np.maximum((mean_tau / 4).astype(int), 1)
        discard = np.minimum(mean_tau.astype(int) * 5, np.maximum(max_steps - thin, 0))
    else:
        tau = None
        thin = np.ones(L, dtype=int)
        discard = np.zeros(L, dtype=int)
    names = tuple("kernel:" + n for n in kernel.get_parameter_names())
    res = BatchPosteriors(sampler, tau, discard, thin, fit_x, fit_f, names)
    if own_engine and ev is not None:
        ev.close()      # (everything BatchPosteriors needs has been copied to the host by now)
    clock.append(("collect", time.perf_counter()))
    res.seconds = {b[0]: b[1] - a[1] for a, b in zip(clock[:-1], clock[1:])}   # wall time of each phase
    return res


def protassov_test(lightcurve, null_kernel, alt_kernel, nsims=100, walkers=12, max_steps=500, sim_walkers=None,
                   sim_steps=500, sigma_noise=None, extension_factor=2, seed=None, device=0, progress=False,
                   sharded=False, group=None, concurrent_refits="auto", split="auto", reproducible=None,
                   observed_side_by_side=True, observed_split=True, pdf="Gaussian", max_iter=400):
    """The whole posterior-predictive likelihood-ratio test of the reference's workflow
    (README.md:38-41, docs/notebooks/tutorial_ppp.ipynb) on the GPU:

    1. posteriors of the null and the alternative kernel on the observed light curve,
       ``T_obs = -2 (max lnL_null - max lnL_alt)``;
    2. ``nsims`` light curves simulated from the null posteriors (device TK95);
    3. both kernels refitted to every simulated light curve in lock-step;
    4. p-value of ``T_obs`` in the simulated distribution.

    Returns dict(T_obs, T_sim[nsims], p_value ((1 + #{T_sim >= T_obs}) / (1 + nsims)), p_value_percentile (the tutorial's own
    ``1 - percentileofscore(T_sim, T_obs) / 100``), null, alt, sim_null, sim_alt, lightcurves, seconds, split, reproducible) --
    ``seconds``:
    wall time of the observed chains, the simulation and the two refits on this process.

    ``concurrent_refits``: the null and the alternative refits of step 3 side by side on the device (two contexts, two
    host threads) instead of one after the other; the results are the same either way (``seconds`` then gives the two
    refits' common wall time under "refit_null" and 0 under "refit_alt").  "auto" (default): side by side when a
    half-step leaves the GPU room -- at most ~40 000 rows, e.g. one GPU's 250 light curves x 128 proposals at 8 GPUs,
    where the two models' launches interleave and their tails and sampler kernels overlap (7.1 ms per iteration of both
    against 8.1 ms one after the other); with the GPU full (2000 x 128 rows) there is nothing to gain (26.30 s against
    26.36 s) and the refits run one after the other.  For measurements: "unpaired" = side by side without
    ``mtg_pair_contexts`` (each context launches its own pipelined half-steps), "slices" = each context on its own half of
    the compute units (``mtg_create_on_slice``); same results all ways.

    ``observed_side_by_side`` (default on; even walker counts): the observed light curve's two chains of step 1 from two
    host threads, each model on a device context and a random generator of its own -- two single-light-curve chains
    leave the GPU nearly empty; the chains are the ones that running them one after the other gives.

    ``sharded`` (inside a ``torch.distributed`` job, one process per GPU, every rank calling with the same
    arguments and its own ``device``; BASELINE configs[3]): steps 2 and 3 -- the loop over simulated light curves
    of tutorial_ppp.ipynb:326-343 -- are cut into contiguous blocks, one per rank (``distributed.LightcurveShard``):
    rank r simulates and refits only its block, nothing is exchanged meanwhile, and ONE all-gather per model of the
    maxima of lnL (8 bytes per light curve) gives every rank the whole ``T_sim``.  Step 1 is split by model
    (``observed_split``, two ranks or more): rank 0 runs the null model's chain on the observed light curve, rank 1 the
    alternative's, nobody else any; rank 1's maximum and rank 0's maximum, posterior samples and seeds are broadcast,
    so that all ranks test the same thing -- the chains are the ones one process runs (each model has a generator of its
    own), hence the same ``T_obs`` to the last bit.  **The returned ``null`` is then rank 0's alone and ``alt`` rank 1's
    (``None`` on every other rank** -- a caller that reads ``res["null"].sampler`` on all ranks passes
    ``observed_split=False``); a chain that fails on rank 0 or 1 raises on every rank (a ``RuntimeError`` naming the rank
    on the others); ``observed_split=False``: every rank runs both and keeps both, rank 0's are everybody's test.
    ``sim_null``, ``sim_alt`` and ``lightcurves`` hold the rank's own block.  ``split``: "lightcurves" as just described,
    "models" -- the first half of the ranks refits the null model, the second half the alternative, each over all the
    light curves (a rank then holds ``sim_null`` or ``sim_alt``, not both) --, "auto" picks by the rows a half-step
    leaves each rank (``_split_by_model``).

    ``reproducible`` (default: on when ``sharded`` and it costs nothing, see below): ``T_sim`` and the p-value do not
    depend on the number of ranks or on the split -- a run on 8 GPUs can be CHECKED against a run on one, bit for bit.  Every simulated light curve
    is then a function of (seed, its global index) alone: the simulator's noise stream and each refit's Philox
    counters are keyed by global index (``mtg_set_stream_base``), the walkers start from a generator of the light
    curve's own, and the refits keep to kernels whose results do not depend on the batch a row travels in
    (``derive_posteriors_batch(index_base=..., total_lightcurves=nsims)``; what the host draws -- Kraft noise, a
    non-Gaussian flux PDF -- comes from a generator per light curve as well).  Off, every block draws from a stream of
    its own and every batch takes whichever kernel is fastest for its size: the same statistics, not the same numbers.
    The price of "on" is the kernel choice: a set beyond ``REPRODUCIBLE_TP_ROWS`` rows per half-step keeps its chains
    on the sweep, which a rank whose own share is small (under ~8000 rows: the time-parallel kernels' range) pays
    with 2.4-3.4 ms per half-step instead of 0.35-1.8 ms at N = 1e4.  The default is therefore "on" only where that
    does not happen -- the set is small enough for the batch-independent time-parallel kernel everywhere, or every
    rank's share is beyond the time-parallel range anyway (BASELINE configs[3]: 250 x 128 rows per rank) -- and the
    returned dict says which it was (``reproducible``).  (The observed light curve's chains are rank 0's either way.)
    """
    from .gpmodelling import GPModelling
    from .simulator import Simulator
    from .stats import lrt_pvalue, lrt_pvalue_percentile, lrt_statistic
    rng = np.random.default_rng(seed)

    seeds = [int(rng.integers(0, 2 ** 31 - 1)) for _ in range(2)]

    def observed(kernel, seed, own_engine=False):
        """The observed light curve's chain for one model, from a generator of its own (the stream np.random.seed(seed)
        would give: nothing here touches numpy's global generator unless the walkers are odd -- the host-side sampler
        copies the global state, as emcee does)."""
        g = GPModelling(lightcurve, kernel, device=device, own_engine=own_engine,
                        random_state=np.random.RandomState(seed) if walkers % 2 == 0 else None)
        state = np.random.get_state()
        if walkers % 2:
            np.random.seed(seed)
        try:
            g.derive_posteriors(fit=True, max_steps=max_steps, walkers=walkers, progress=progress,
                                device_sampler=walkers % 2 == 0)
        finally:
            if walkers % 2:
                np.random.set_state(state)
        return g

    import threading
    import time
    clock = [time.perf_counter()]
    # the simulator's transform plan is built beside the observed chains (its grid depends on the sampling alone)
    # (``pdf``: the flux PDF of the simulated light curves, simulator.py:149-150 -- "Lognormal" / "Uniform" go through the
    # E13 adjustment on the device, csrc/mtg_e13.hip, ``max_iter`` iterations at most: the loop still never leaves the GPU)
    sim = Simulator(null_kernel, lightcurve.times, lightcurve.exposures, lightcurve.mean, pdf,
                    lightcurve.bkg_rate, lightcurve.bkg_rate_err, sigma_noise=sigma_noise,
                    extension_factor=extension_factor, max_iter=max_iter, random_state=0, device=device)
    sim.warm_up()
    shard = None
    if sharded:
        from .distributed import LightcurveShard, all_gather_rows, block_bounds, broadcast_array
        shard = LightcurveShard(nsims, group=group)
    by_model = shard is not None and shard.world >= 2 and bool(observed_split)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if by_model:
            # one model's chain per rank (0: null, 1: alternative), on the process's own context
            null = alt = obs_failure = None
            try:
                if shard.rank == 0:
                    null = observed(null_kernel, seeds[0])
                elif shard.rank == 1:
                    alt = observed(alt_kernel, seeds[1])
            except Exception as exc:      # said to everybody below: nobody may wait in a broadcast for a chain that died
                obs_failure = exc
            obs_failed = all_gather_rows(np.array([0.0 if obs_failure is None else 1.0]), np.ones(shard.world, dtype=int), group)
            if obs_failure is not None:
                raise obs_failure
            if obs_failed.any():
                raise RuntimeError("protassov_test: the observed light curve's chain failed on rank(s) %s"
                                   % np.flatnonzero(obs_failed).tolist())
        elif walkers % 2 == 0 and observed_side_by_side:
            # two single-light-curve chains leave the GPU nearly empty: the two models side by side, each on a context
            # and a generator of its own -- the same chains as one after the other
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=2) as pool:
                null, alt = pool.map(lambda a: observed(a[0], a[1], own_engine=("side", a[2])),
                                     ((null_kernel, seeds[0], 0), (alt_kernel, seeds[1], 1)))
            null.gp.release_engine()
            alt.gp.release_engine()
        else:
            null, alt = observed(null_kernel, seeds[0]), observed(alt_kernel, seeds[1])
    clock.append(time.perf_counter())
    # one estimator on both sides of the test: the largest log-posterior over everything the chains
    # visited (the refits below store no chains and keep exactly that; the maximum over the burned-in,
    # thinned chain is systematically smaller, the more so the more parameters a model has)
    if by_model:
        alt_best = float(broadcast_array(np.array([alt.best_loglikelihood if alt is not None else 0.0]), group, src=1)[0])
        null_best = float(null.best_loglikelihood) if null is not None else 0.0     # (rank 0's goes out with the head below)
        ndim_null = null_kernel.vector_size if null is None else null.mcmc_samples.shape[1]
        samples = null.mcmc_samples[rng.integers(len(null.mcmc_samples), size=nsims)] if null is not None \
            else np.zeros((nsims, ndim_null))
    else:
        null_best, alt_best = float(null.best_loglikelihood), float(alt.best_loglikelihood)
        samples = null.mcmc_samples[rng.integers(len(null.mcmc_samples), size=nsims)]
    t_obs = float(lrt_statistic(null_best, alt_best))
    # (seeds below 2^52: they travel as float64 in the broadcast)
    sim_seed, fit_seeds = int(rng.integers(0, 2 ** 31 - 1)), [int(rng.integers(0, 2 ** 52)) for _ in range(2)]
    sw = sim_walkers or walkers
    lo, hi, models = 0, nsims, (0, 1)
    if sharded:
        if reproducible is None:
            reproducible = _reproducible_is_free(split, nsims, sw, shard.world)
    reproducible = bool(reproducible)
    if sharded:
        # rank 0's test is everybody's test: its T_obs, its posterior samples and its seeds (every rank drew its own
        # from its own chains' generator state; under the model split two ranks must simulate the SAME light curves)
        head = broadcast_array(np.concatenate([[null_best, float(sim_seed)], np.asarray(fit_seeds, dtype=np.float64),
                                               [t_obs], samples.ravel()]), group)
        sim_seed, fit_seeds = int(head[1]), [int(head[2]), int(head[3])]
        # split observed chains: rank 0's null maximum with rank 1's alternative maximum; otherwise rank 0's own T_obs
        t_obs = float(lrt_statistic(float(head[0]), alt_best)) if by_model else float(head[4])
        samples = head[5:].reshape(samples.shape)
        if _split_by_model(split, nsims, sw, shard.world):
            # half of the ranks refit the null model, the other half the alternative, each half over ALL the light
            # curves: twice the rows per rank and one model's half-steps instead of both one after the other
            half = shard.world // 2
            models = (0,) if shard.rank < half else (1,)
            block = shard.rank % half
            bounds = block_bounds(nsims, half)
            lo, hi = int(bounds[block]), int(bounds[block + 1])
            if shard.rank >= 2 * half:                           # an odd rank out takes no part in the refits
                models, lo, hi = (), 0, 0
        else:
            block, bounds = shard.rank, shard.bounds
            lo, hi = shard.lo, shard.hi
        if not reproducible:
            sim_seed = (sim_seed + 7919 * block) % (2 ** 31 - 1)   # independent noise on every block (the two ranks
            fit_seeds = [f + 7919 * block for f in fit_seeds]       # of a block under the model split draw the same)
    out, fits, best, failure, pair_stats = None, [None, None], [np.empty(0), np.empty(0)], None, None
    # each model on a context of its own ("slices": and on its own half of the compute units, mtg_create_on_slice --
    # measured no faster: 7.24 against 7.14 ms per iteration)
    if concurrent_refits not in (True, False, "auto", "unpaired", "slices"):
        raise ValueError("concurrent_refits must be True, False, 'auto', 'unpaired' or 'slices'")
    side_by_side = len(models) == 2 and (concurrent_refits in (True, "unpaired", "slices") or
                                         (concurrent_refits == "auto" and (hi - lo) * (sw // 2) <= 40000 and hi - lo > 1))
    if side_by_side and concurrent_refits == "slices":
        side_by_side = "slices"
    if hi > lo:
        try:
            sim.random_state = np.random.RandomState(sim_seed)
            if reproducible:
                # Every series with the partner it has in the whole set (the simulator's transform packs series 2p and
                # 2p + 1 together): a block that starts or ends inside a pair simulates the partner too -- its parameters
                # are at hand, every rank holds all the posterior samples -- and drops it.  At most two extra series.
                lo_e, hi_e = lo - (lo & 1), min(nsims, hi + (hi & 1))
                out = sim.simulate(samples[lo_e:hi_e, :null_kernel.vector_size], index_base=lo_e, pair_series=True)
                keep = slice(lo - lo_e, lo - lo_e + (hi - lo))
                out = {k: (v[keep] if v is not None else None) for k, v in out.items()}
            else:
                out = sim.simulate(samples[lo:hi, :null_kernel.vector_size])
            clock.append(time.perf_counter())
            meet = threading.Barrier(2) if side_by_side else None
            # side by side AND paired: from the chains on, the two contexts' pipelined half-steps go out in ONE launch
            # (mtg_pair_contexts: eight waves per compute unit on one table set, two per SIMD -- a pipelined sweep alone
            # takes the whole compute unit, so unpaired launches alternate rather than share SIMDs).  Paired between the
            # starting fits and the chains: the fits' batches come at each model's own pace and must not wait for each other.
            paired = side_by_side is True and concurrent_refits != "unpaired"

            def meet_then_pair():
                first = meet.wait() == 0
                if paired:
                    if first:
                        from .gp import get_side_engine
                        for k in (0, 1):
                            get_side_engine(device, k).unpair()      # (whatever an interrupted run may have left)
                        get_side_engine(device, 0).pair_with(get_side_engine(device, 1))
                    meet.wait()

            def refit(k):
                kernel = (null_kernel, alt_kernel)[k]
                try:
                    return derive_posteriors_batch(lightcurve.times, out["rates"], out["dy"], kernel, walkers=sw,
                                                   max_steps=sim_steps, fit=True, seed=fit_seeds[k], device=device,
                                                   store_chain=False, quiet=True,
                                                   # side by side: model k on the process's k-th extra context (gp.get_side_engine)
                                                   own_engine=((k, 2) if side_by_side == "slices" else ("side", k) if side_by_side
                                                               else False),
                                                   index_base=lo if reproducible else None,
                                                   total_lightcurves=nsims if reproducible else None,
                                                   # (no timeout: a partner that fails aborts the barrier, below)
                                                   before_sampling=meet_then_pair if meet is not None else None)
                except BaseException:
                    if meet is not None:
                        meet.abort()        # the partner thread must not wait at the barrier for a refit that has failed
                    raise

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if side_by_side:
                    # The two models' refits are independent: each on its own context and stream, driven by its own host
                    # thread (the library calls release the GIL).  Measured at configs[3]'s sizes: 26.30 s side by side
                    # against 8.40 + 17.96 s one after the other -- both sweeps are bound by FP64 issue and a half-step
                    # leaves no idle issue slots for the other model to fill; on a block that leaves the GPU room (a rank's
                    # 250 light curves at 8 GPUs) the two chains interleave and gain 16 %: "auto" (docstring).
                    from concurrent.futures import ThreadPoolExecutor
                    from .gp import get_side_engine
                    try:
                        with ThreadPoolExecutor(max_workers=2) as pool:
                            futures = [pool.submit(refit, k) for k in (0, 1)]
                    finally:
                        if paired:     # (both threads have returned: nobody is inside a paired call)
                            pair_stats = get_side_engine(device, 0).pair_stats()
                            get_side_engine(device, 0).unpair()
                    errors = [f.exception() for f in futures if f.exception() is not None]
                    if errors:   # the refit that failed, not the partner it left at the barrier
                        real = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
                        raise (real or errors)[0]
                    fits = [f.result() for f in futures]
                    clock += [time.perf_counter()] * 2
                else:
                    for k in (0, 1):
                        if k in models:
                            fits[k] = refit(k)
                        clock.append(time.perf_counter())
                best = [np.empty(0) if f is None else f.max_loglikelihood for f in fits]
        except BaseException as exc:     # (simulation or refits)
            failure = exc
            if not sharded:
                raise
    if sharded:
        # a rank whose refits failed says so BEFORE the gather: the others must not wait in it for maxima that will not come
        failed = all_gather_rows(np.array([0.0 if failure is None else 1.0]), np.ones(shard.world, dtype=int), group)
        if failure is not None:
            raise failure
        if failed.any():
            raise RuntimeError("protassov_test: the refits failed on rank(s) %s" % np.flatnonzero(failed).tolist())
        # the only exchange of the loop: the maxima of lnL, one all-gather per model
        if len(models) == 2:
            best = [shard.gather(b) for b in best]
        else:
            # the null rank and the alternative rank of a block pair lnL values of what must be the same light curves,
            # simulated on two GPUs: a checksum per rank says so, or the test stops here (a block without light curves
            # -- fewer of them than pairs of ranks -- has nothing to compare)
            mine = np.array([float(np.sum(out["rates"])) + float(np.sum(out["dy"])) if out is not None else 0.0])
            sums = all_gather_rows(mine, np.ones(shard.world, dtype=int), group)
            for b in range(shard.world // 2):
                if bounds[b + 1] > bounds[b] and sums[b] != sums[b + shard.world // 2]:
                    raise RuntimeError("ranks %d and %d simulated different light curves for block %d (checksums %r, %r)"
                                       % (b, b + shard.world // 2, b, sums[b], sums[b + shard.world // 2]))
            half, sizes = shard.world // 2, np.diff(bounds)
            counts = [np.concatenate([sizes, 0 * sizes]), np.concatenate([0 * sizes, sizes])]
            if shard.world % 2:                                  # an odd rank out takes no part in the refits
                counts = [np.append(c, 0) for c in counts]
            best = [all_gather_rows(best[k] if k in models else np.empty(0), counts[k], group) for k in (0, 1)]
    t_sim = lrt_statistic(best[0], best[1])
    clock.append(time.perf_counter())
    seconds = dict(zip(("observed_chains", "simulate", "refit_null", "refit_alt", "gather"), np.diff(clock))) \
        if len(clock) == 6 else {"observed_chains": clock[1] - clock[0]}
    return dict(T_obs=t_obs, T_sim=t_sim, p_value=lrt_pvalue(t_obs, t_sim), p_value_percentile=lrt_pvalue_percentile(t_obs, t_sim), null=null, alt=alt,
                sim_null=fits[0], sim_alt=fits[1], lightcurves=out, seconds=seconds, reproducible=reproducible,
                paired_launches=pair_stats,
                split=None if not sharded else ("models" if len(models) < 2 else "lightcurves"))


def _reproducible_is_free(split, nsims, walkers, world):
    """protassov_test(sharded=True, reproducible=None): world-size-independent results by default exactly where they do not
    cost a rank the time-parallel kernels (docstring there): the whole set is within ``REPRODUCIBLE_TP_ROWS`` rows per
    half-step -- every block then runs the batch-independent time-parallel kernel --, or every rank's own share is beyond
    the time-parallel range (8192 rows), where the sweep is what it would run anyway."""
    by_model = _split_by_model(split, nsims, walkers, world)
    rows_per_rank = -(-nsims // (world // 2 if by_model else world)) * (walkers // 2)
    return bool(nsims * (walkers // 2) <= REPRODUCIBLE_TP_ROWS or rows_per_rank > 8192)


def _split_by_model(split, nsims, walkers, world):
    """How protassov_test(sharded=True) divides the refits: by light curve (every rank refits both models on its block)
    or by model (half of the ranks each).  By light curve whenever a rank's half-step fits the pipelined sweep (at most
    32 768 rows: one workgroup of 128 rows per compute unit): the two models' chains then run side by side on the rank
    and its share of BASELINE configs[3] at 8 GPUs takes 3.8 s.  Beyond that a half-step of the
    one-lane sweep costs one wave's latency over the N samples until a rank has about one wave per SIMD (65 536 rows),
    so two half-steps of both models one after the other take twice as long as one half-step of one model on twice
    the rows: by model, when asked for, or -- "auto" -- when the rows of a half-step per rank stay under that mark
    either way (at 8 GPUs that share is ~4.2 s: the alternative's 64 000-row half-steps at 3.7 ms)."""
    if split == "models":
        if world < 2:
            raise ValueError("split='models' needs at least two ranks")
        return True
    if split == "lightcurves" or world < 2:
        return False
    if split != "auto":
        raise ValueError("split must be 'auto', 'lightcurves' or 'models'")
    rows_by_lightcurve = -(-nsims // world) * (walkers // 2)
    if rows_by_lightcurve <= 32768:
        return False
    rows_by_model = -(-nsims // (world // 2)) * (walkers // 2)
    return rows_by_model <= 70000


def derive_posteriors_sharded(times, Y, DY, kernel, group=None, device=None, **kwargs):
    """:func:`derive_posteriors_batch` with the light curves sharded over the ranks of a
    ``torch.distributed`` job (one process per GPU; BASELINE configs[3]: 2000 simulated light
    curves over the 8 GPUs of a node).

    Every rank calls this with the SAME ``Y, DY`` [L, N]; rank r fits only its contiguous block
    of light curves (``distributed.LightcurveShard``) on its own GPU -- the ensembles are
    independent, so nothing is exchanged while sampling -- and ONE all-gather of
    ``max_loglikelihood`` (8 bytes per light curve, RCCL over xGMI) gives every rank the full
    vector for the LRT.  Returns (max_loglikelihood[L], local BatchPosteriors, shard).
    ``kwargs`` go to derive_posteriors_batch; a ``seed`` is offset by the rank.
    """
    from .distributed import LightcurveShard
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    DY = np.atleast_2d(np.asarray(DY, dtype=np.float64))
    shard = LightcurveShard(Y.shape[0], group=group)
    if kwargs.get("seed") is not None:
        kwargs["seed"] = int(kwargs["seed"]) + 7919 * shard.rank
    local = None
    best = np.empty(0)
    if len(shard):
        local = derive_posteriors_batch(times, Y[shard.lo:shard.hi], DY[shard.lo:shard.hi], kernel, **kwargs)
        best = local.max_loglikelihood
    return shard.gather(best, device=device), local, shard
