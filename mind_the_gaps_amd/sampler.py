"""Affine-invariant ensemble sampler driving the batched log-probability.

The reference hands ``self._log_probability`` to ``emcee.EnsembleSampler``
(/root/reference/mind_the_gaps/gpmodelling.py:247-248) and reads
``sampler.iteration``, ``get_autocorr_time(tol=0)``, ``get_chain`` and
``get_log_prob`` (gpmodelling.py:250-285, 397-401).  emcee (pin 3.1.4,
pyproject.toml:13) is a third-party dependency that is not installed in this
image; its published algorithm (Goodman & Weare 2010 stretch move with a = 2 and
a red/blue split; Sokal's windowed integrated autocorrelation time) is restated
here with the same call surface, so that
``derive_posteriors`` reads like the reference.  The one structural difference:
``log_prob_fn`` is called ONCE per half-ensemble with a [W/2, ndim] array
(emcee's ``vectorize=True`` contract) -- that call is the GPU launch.
"""
import os

import numpy as np
from scipy import fft as _fft

__all__ = ["EnsembleSampler", "integrated_time", "AutocorrError"]


class AutocorrError(Exception):
    """The chain is shorter than ``tol`` autocorrelation times."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _next_pow_two(n):
    i = 1
    while i < n:
        i <<= 1
    return i


def _mean_autocorr_function(x):
    """Walker-averaged normalised autocorrelation function of x[n_t, n_walkers, n_dim] -> [n_t, n_dim] (emcee's
    ``function_1d`` of every walker, each normalised by its own lag-0 value, then averaged) by zero-padded FFT.
    The convergence check runs on the whole chain every ``convergence_steps`` iterations, so for long chains this
    is the host's hot spot.  The inverse transform is linear, so the power spectra are normalised (lag 0 of a
    series is the sum of its squares) and averaged over the walkers BEFORE it: one inverse transform per
    dimension instead of one per walker and dimension; real transforms over contiguous series."""
    n_t = x.shape[0]
    n = _next_pow_two(n_t)
    series = np.ascontiguousarray(np.moveaxis(x - np.mean(x, axis=0), 0, -1))     # [n_walkers, n_dim, n_t]
    # (threads only where they pay: below a few million points the pool's start-up and hand-over cost more than
    # the transforms -- 640 series of 1000 steps: 60 ms on 8 workers, 12 ms on one)
    workers = max(1, min(8, os.cpu_count() or 1)) if series.size * 2 * n // max(n_t, 1) > (1 << 23) else 1
    f = _fft.rfft(series, n=2 * n, axis=-1, workers=workers)
    power = f.real ** 2 + f.imag ** 2
    with np.errstate(invalid="ignore", divide="ignore"):   # a walker that never moved: 0 / 0 = nan, as emcee has it
        power /= np.sum(series * series, axis=-1)[..., None]
    acf = _fft.irfft(np.mean(power, axis=0), n=2 * n, axis=-1)[..., :n_t]          # [n_dim, n_t]
    return np.ascontiguousarray(acf.T)


def integrated_time(x, c=5, tol=50, quiet=False, acf=None):
    """Integrated autocorrelation time of x[n_t, n_walkers, n_dim] per dimension:
    walker-averaged autocorrelation function, tau(M) = 2 sum_{k<=M} rho_k - 1, window
    M = first index with M >= c tau(M) (Sokal).  ``acf``: a callable chain[n_t, W, P] -> rho[n_t, P] to use
    instead of the host FFTs (``Engine.chain_autocorr``: the same numbers from the device)."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, None, None]
    if x.ndim == 2:
        x = x[:, :, None]
    if x.ndim != 3:
        raise ValueError("invalid dimensions")
    n_t, n_w, n_d = x.shape
    rho = _mean_autocorr_function(x) if acf is None else acf(x)   # [n_t, n_d]
    taus = 2.0 * np.cumsum(rho, axis=0) - 1.0
    tau_est = np.empty(n_d)
    lags = np.arange(n_t)
    for d in range(n_d):
        m = lags < c * taus[:, d]
        window = np.argmin(m) if np.any(m) else n_t - 1
        tau_est[d] = taus[window, d]
    flag = tol * tau_est > n_t
    if np.any(flag):
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} "
               "parameter(s). Use this estimate with caution and run a longer chain!\n"
               "N/{0} = {2:.0f};\ntau: {3}").format(tol, np.sum(flag), n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
    return tau_est


class EnsembleSampler:
    """Stretch-move ensemble sampler with emcee's call surface.

    ``log_prob_fn(coords[B, ndim]) -> lnP[B]`` when ``vectorize`` (the default here);
    with ``vectorize=False`` it is called row by row (or through ``pool.map``).
    """

    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, vectorize=True, a=2.0):
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        self.log_prob_fn = log_prob_fn
        self.pool = pool
        self.vectorize = vectorize
        self.a = float(a)
        # like emcee: a private generator seeded with a COPY of numpy's global state,
        # so np.random.seed(k) before derive_posteriors makes chains reproducible
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(np.random.get_state())
        self.reset()

    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    def reset(self):
        self.iteration = 0
        self._chain = np.empty((0, self.nwalkers, self.ndim))
        self._log_prob = np.empty((0, self.nwalkers))
        self._accepted = np.zeros(self.nwalkers)
        self._coords = None
        self._lnp = None

    # -- log-probability -------------------------------------------------------
    def compute_log_prob(self, coords):
        p = np.asarray(coords, dtype=np.float64)
        if np.any(np.isinf(p)):
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(p)):
            raise ValueError("At least one parameter value was NaN")
        if self.vectorize:
            lp = np.asarray(self.log_prob_fn(p), dtype=np.float64)
        else:
            mapper = self.pool.map if self.pool is not None else map
            lp = np.array([float(v) for v in mapper(self.log_prob_fn, (row for row in p))])
        if lp.shape != (p.shape[0],):
            raise ValueError("incompatible input dimensions: log_prob_fn returned shape "
                             "{0}".format(lp.shape))
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return lp

    # -- sampling ----------------------------------------------------------------
    def _check_initial(self, p0, skip_initial_state_check):
        p0 = np.array(p0, dtype=np.float64)
        if p0.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions {0}".format(p0.shape))
        if np.any(np.isinf(p0)):
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(p0)):
            raise ValueError("At least one parameter value was NaN")
        if self.nwalkers < 2 * self.ndim:
            raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than "
                               "twice the number of dimensions.")
        if not skip_initial_state_check:
            c = p0 - np.mean(p0, axis=0)[None, :]
            cmax = np.max(np.abs(c), axis=0)
            ok = not np.any(cmax == 0)
            if ok:
                c = c / cmax
                cnorm = np.sqrt(np.sum(c ** 2, axis=0))
                ok = np.linalg.cond((c / cnorm).astype(float)) <= 1e8
            if not ok:
                raise ValueError("Initial state has a large condition number. Make sure that "
                                 "your walkers are linearly independent for the best performance")
        return p0

    def sample(self, initial_state, iterations=1, progress=False, skip_initial_state_check=False,
               log_prob0=None):
        """Generator advancing the ensemble; yields (coords, log_prob) after every iteration."""
        if self._coords is None or initial_state is not None:
            coords = self._check_initial(initial_state, skip_initial_state_check)
            lnp = self.compute_log_prob(coords) if log_prob0 is None else np.array(log_prob0, float)
        else:
            coords, lnp = self._coords, self._lnp
        W, ndim = self.nwalkers, self.ndim
        grow = int(iterations)
        self._chain = np.concatenate([self._chain, np.empty((grow, W, ndim))], axis=0)
        self._log_prob = np.concatenate([self._log_prob, np.empty((grow, W))], axis=0)
        bar = _progress_bar(grow) if progress else None
        rng = self._random
        for _ in range(grow):
            # emcee picks the move of every iteration with RandomState.choice(moves, p=weights),
            # which draws one uniform even when the stretch move is the only one: consume it, so that
            # the stream stays aligned with emcee's for the same seed
            rng.random_sample()
            # red/blue split: shuffled alternating labels, each half moved given the other
            inds = np.arange(W) % 2
            rng.shuffle(inds)
            for split in (0, 1):
                mine = inds == split
                s = coords[mine]
                c = coords[~mine]
                ns, nc = len(s), len(c)
                zz = ((self.a - 1.0) * rng.rand(ns) + 1.0) ** 2.0 / self.a
                factors = (ndim - 1.0) * np.log(zz)
                partner = c[rng.randint(nc, size=(ns,))]
                q = partner - (partner - s) * zz[:, None]
                new_lnp = self.compute_log_prob(q)          # <- one GPU launch, W/2 evaluations
                lnpdiff = factors + new_lnp - lnp[mine]
                accepted = lnpdiff > np.log(rng.rand(ns))
                idx = np.flatnonzero(mine)[accepted]
                coords[idx] = q[accepted]
                lnp[idx] = new_lnp[accepted]
                self._accepted[idx] += 1
            self._chain[self.iteration] = coords
            self._log_prob[self.iteration] = lnp
            self.iteration += 1
            self._coords, self._lnp = coords, lnp
            if bar is not None:
                bar.update(1)
            yield coords, lnp
        if bar is not None:
            bar.close()
        self._chain = self._chain[:self.iteration]
        self._log_prob = self._log_prob[:self.iteration]

    def run_mcmc(self, initial_state, nsteps, **kwargs):
        results = None
        for results in self.sample(initial_state, iterations=nsteps, **kwargs):
            pass
        return results

    # -- results -----------------------------------------------------------------
    def _get(self, arr, flat, thin, discard):
        v = arr[:self.iteration][discard + thin - 1::thin]
        if flat:
            return v.reshape((-1,) + v.shape[2:])
        return v

    def get_chain(self, flat=False, thin=1, discard=0):
        return self._get(self._chain, flat, thin, discard)

    def get_log_prob(self, flat=False, thin=1, discard=0):
        return self._get(self._log_prob, flat, thin, discard)

    @property
    def acceptance_fraction(self):
        return self._accepted / float(max(self.iteration, 1))

    def get_autocorr_time(self, discard=0, thin=1, **kwargs):
        return thin * integrated_time(self.get_chain(discard=discard, thin=thin), **kwargs)


class _NullBar:
    def update(self, n):
        pass

    def close(self):
        pass


def _progress_bar(total):
    try:
        from tqdm import tqdm
        return tqdm(total=total)
    except Exception:
        return _NullBar()
