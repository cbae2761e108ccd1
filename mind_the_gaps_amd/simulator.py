"""Light-curve simulation from a power spectrum, on the device.

Mirror of /root/reference/mind_the_gaps/simulator.py as the Protassov loop uses it:
``Simulator(psd_model, times, exposures, mean, pdf, bkg_rate, bkg_rate_err, sigma_noise,
aliasing_factor, extension_factor, epsilon, ...)`` (:143-275) with ``generate_lightcurve``
(:397-420: Timmer & Koenig 1995 on a fine, extended regular grid, random segment cut,
bin-averaging onto the observing pattern) and ``add_noise`` (:300-338).  The grid arithmetic
below is the reference's; the spectrum draw, the inverse FFT (hipFFT), the cut, the down-sampling
and Gaussian / Poisson noise run on the GPU for ALL requested simulations at once
(``mtg_simulate_tk95``), and the result can stay resident as the light-curve set of the next
fitting sweep.

What drives the spectrum:
* a celerite ``Term`` (or its bound ``get_psd``, what gpmodelling.py:509 passes), or one of the
  closed-form spectra of ``models.psd_models`` that has a celerite twin: the device evaluates
  the PSD of every posterior sample from its coefficients;
* ANY other callable ``psd(omega)`` (simulator.py:149,272-280; the reference's tests use astropy's
  ``PowerLaw1D``): evaluated once on the host at the grid's angular frequencies and uploaded as a
  table.

The Emmanoulopoulos et al. (2013) amplitude / rank adjustment for ``pdf="lognormal" | "uniform"``
(simulator.py:65-131) runs on the device too (round 6: ``mtg_set_simulate_pdf``, csrc/mtg_e13.hip -- batched
transforms and one segmented sort per iteration over all the segments of a chunk), between the cut and the
down-sampling, so a posterior-predictive run with a non-Gaussian flux PDF stays on the GPU; ``adjust_on="host"``
keeps the numpy implementation (``_adjust_pdf``: the reference's loop, at the reference's kind of speed), which the
tests hold the device against.  The Kraft et al. (1991) treatment of low-count epochs with background
(noise_models.py:81-150) as well: the posterior median and 68 % interval of the source counts depend on (total counts,
background) alone, so the simulator tabulates them once per epoch for totals below ``kraft_counts`` and the device looks
them up after its Poisson draw (``mtg_set_simulate_kraft``; ``adjust_on="host"``: numpy, per light curve).

Random numbers.  Default (``stream="philox"``): counter-based streams on the device, keyed by (seed, series index) -- what
the batched Protassov loop needs.  ``stream="numpy"``: the reference draws everything from numpy's GLOBAL generator
(get_fft, simulator.py:468-501; cut_random_segment, :536-539; the noise classes, noise_models.py:71,182), so
``np.random.seed(s)`` fixes its light curves; in this mode the host makes those draws in the reference's order and the
device does the arithmetic with them (``mtg_set_simulate_draws``): ``generate_lightcurve`` / ``add_noise`` then return the
REFERENCE'S light curve for a seed, to the rounding of the transform (tests/test_simulator_gpu.py: the two light curves of
docs/notebooks/celerite_variance.ipynb, whose variances the notebook prints).  Gaussian flux PDF only.
"""
import warnings

import numpy as np

from .gp import DeviceModel, LogProbEvaluator
from .modeling import ConstantModel
from .models.psd_models import PSDModel
from .terms import Term

__all__ = ["Simulator", "RegularLightcurve", "kraft_median", "kraft_interval"]


# -- Kraft, Burrows & Nousek (1991): posterior of the source counts s given N total and B background --
def kraft_median(N, B):
    """Median of f(s | N, B) = C e^-(s+B) (s+B)^N / N!, s >= 0 (reference stats.py:10-19 ``kraft_pdf``):
    its distribution function is [P(N+1, s+B) - P(N+1, B)] / Q(N+1, B) with the regularised incomplete
    gamma functions, so the median is one inverse incomplete gamma away."""
    from scipy.special import gammainc, gammaincinv
    p0 = gammainc(N + 1.0, B)                   # P(N + 1, B): the mass "below zero", cut away
    return max(float(gammaincinv(N + 1.0, p0 + 0.5 * (1.0 - p0))) - B, 0.0)


def kraft_interval(N, B, confidence_level=0.68):
    """Shortest interval [s_lo, s_hi] holding ``confidence_level`` of that posterior: equal density at
    both ends, or s_lo = 0 when the density there is already higher (what astropy's
    ``poisson_conf_interval(..., "kraft-burrows-nousek")`` computes, noise_models.py:139-141)."""
    from scipy.optimize import brentq
    from scipy.special import gammainc, gammaincinv
    N, B = float(N), float(B)
    p0 = gammainc(N + 1.0, B)
    norm = 1.0 - p0

    def cdf(s):
        return (gammainc(N + 1.0, s + B) - p0) / norm

    def logpdf(s):
        return -(s + B) + N * np.log(s + B) if s + B > 0 else (0.0 if N == 0 else -np.inf)

    def upper_for(lo):   # the upper end that makes the interval hold the requested mass
        target = cdf(lo) + confidence_level
        return np.inf if target >= 1.0 else float(gammaincinv(N + 1.0, p0 + target * norm)) - B

    mode = max(N - B, 0.0)
    hi0 = upper_for(0.0)
    if mode == 0.0 or logpdf(0.0) >= logpdf(hi0):
        return 0.0, hi0
    # lo in (0, mode): density at lo rises with lo, density at the matching upper end falls
    lo_max = mode
    while not np.isfinite(upper_for(lo_max)):
        lo_max *= 0.999
    lo = brentq(lambda s: logpdf(s) - logpdf(upper_for(s)), 0.0, lo_max, xtol=1e-10)
    return lo, upper_for(lo)


class RegularLightcurve:
    """What ``Simulator.simulate_regularly_sampled`` returns: the attributes of stingray's Lightcurve the reference's
    own code reads (``time``, ``countrate``, ``dt``, ``n``, ``tseg``, ``meanrate``; simulator.py:340-420, 503-539)."""

    def __init__(self, time, countrate, dt):
        self.time, self.countrate, self.dt = np.asarray(time), np.asarray(countrate), float(dt)

    n = property(lambda self: len(self.time))
    tseg = property(lambda self: self.n * self.dt)
    meanrate = property(lambda self: float(np.mean(self.countrate)))


class Simulator:
    """Simulate light curves with a given power spectrum and flux probability density."""

    def __init__(self, psd_model, times, exposures, mean, pdf="gaussian", bkg_rate=None, bkg_rate_err=None,
                 sigma_noise=None, aliasing_factor=2, extension_factor=10, epsilon=1.001, max_iter=400,
                 random_state=None, device=0, kraft_counts=15, stream="philox", transform="auto", adjust_on="device"):
        if stream not in ("philox", "numpy"):
            raise ValueError("stream must be 'philox' or 'numpy'")
        if transform not in ("auto", "library", "chirp-z"):
            raise ValueError("transform must be 'auto', 'library' or 'chirp-z'")
        if adjust_on not in ("device", "host"):
            raise ValueError("adjust_on must be 'device' or 'host'")
        self.adjust_on = adjust_on        # where a non-Gaussian flux PDF is imposed on the segments (module docstring)
        # which inverse transform the device takes (mtg_set_simulate_transform): "auto" picks by grid length; the other
        # two exist so that tests can hold one against the other
        self.transform = transform
        if stream == "numpy" and pdf.lower() != "gaussian":
            raise NotImplementedError("stream='numpy' reproduces the reference's Gaussian (TK95) light curves only")
        self.stream = stream
        if extension_factor < 1:
            raise ValueError("Extension factor must be greater than 1")
        if epsilon < 1:
            raise ValueError("Epsilon needs to be greater than 1!")
        if np.any(np.asarray(exposures) == 0):
            raise ValueError("Some exposure times are 0!")
        times = np.asarray(times, dtype=np.float64)
        n = len(times)
        self._exposures = np.full(n, exposures, dtype=np.float64) if np.isscalar(exposures) \
            else np.asarray(exposures, dtype=np.float64)
        if pdf.lower() not in ("gaussian", "lognormal", "uniform"):
            raise ValueError("%s not implemented! Currently implemented: Gaussian, Uniform or Lognormal" % pdf)
        self.pdf = pdf
        self.max_iter = int(max_iter)
        self.random_state = np.random.RandomState(random_state)
        self.sim_dt = float(np.min(self._exposures) / aliasing_factor)
        dt = np.diff(times)
        wrong = np.count_nonzero(dt < self.sim_dt * 0.99)
        if wrong > 0:
            raise ValueError("%d timestamps differences are below the exposure integration time! Either reduce "
                             "the exposure times, or space your observations" % wrong)
        start_time = times[0] - dt[0] / 1.99
        end_time = times[-1] + dt[-1]
        self.sim_duration = end_time - start_time
        duration = (times[-1] - times[0]) * extension_factor
        # fine regular grid, longer than the observed light curve (red-noise leakage)
        self.sim_timestamps = np.arange(start_time - self.sim_dt, start_time + duration + self.sim_dt, self.sim_dt)
        self.fftndatapoints = len(self.sim_timestamps)
        self.psd_model = psd_model
        self._times = times
        self.mean = float(mean)
        self.device = device
        # noise model (simulator.py:255-262): Gaussian if a sigma is given, else Poisson, with the
        # Kraft treatment of faint epochs when there is a background
        self._bkg_counts = np.zeros(n)
        self._bkg_rate_err = np.zeros(n)
        self.kraft_counts = kraft_counts
        if sigma_noise is not None:
            self.noise_name, self._noise_kind, self.sigma_noise = "Gaussian", 1, float(sigma_noise)
        elif bkg_rate is None or np.all(np.asarray(bkg_rate) == 0):
            self.noise_name, self._noise_kind, self.sigma_noise = "Poisson", 2, 0.0
        else:
            self.noise_name, self._noise_kind, self.sigma_noise = "Kraft", 3, 0.0
            self._bkg_counts = np.broadcast_to(np.asarray(bkg_rate, dtype=np.float64), (n,)) * self._exposures
            if bkg_rate_err is not None:
                self._bkg_rate_err = np.broadcast_to(np.asarray(bkg_rate_err, dtype=np.float64), (n,)).copy()
        # observing windows ("strategy", simulator.py:265-267) as index ranges of the cut segment, whose
        # first fine sample sits sim_dt / 2 after the start of the first window
        half_bins = self._exposures / 2 * epsilon
        self.strategy = [(t - h, t + h) for t, h in zip(times, half_bins)]
        self.seg_len = min(int(np.ceil(self.sim_duration / self.sim_dt)), self.fftndatapoints)
        self.segment_times = self.strategy[0][0] + self.sim_dt / 2 + np.arange(self.seg_len) * self.sim_dt
        self.win_lo, self.win_hi = self._windows(self.segment_times)
        self._evaluator = None

    # -- down-sampling rule ------------------------------------------------------------
    def _windows(self, grid_times):
        """[lo, hi) index ranges of ``grid_times`` (ascending) that fall in every epoch's window
        ``start <= time < end`` (simulator.py:358-362)."""
        starts = np.array([s for s, _ in self.strategy])
        ends = np.array([e for _, e in self.strategy])
        lo = np.searchsorted(grid_times, starts, side="left")     # first time >= start
        hi = np.searchsorted(grid_times, ends, side="left")       # first time >= end: excluded
        return lo.astype(np.int32), hi.astype(np.int32)

    def downsample(self, lc, countrate=None):
        """Average of a regularly sampled light curve over every epoch's window (simulator.py:340-367).
        ``lc``: an object with ``time`` and ``countrate`` (stingray's Lightcurve in the reference), or the
        times with the rates as second argument.  Returns the list of mean rates."""
        time = np.asarray(lc.time if countrate is None else lc, dtype=np.float64)
        rate = np.asarray(lc.countrate if countrate is None else countrate, dtype=np.float64)
        lo, hi = self._windows(time)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                        # an empty window is NaN, as numpy's mean of nothing
            return [float(np.mean(rate[a:b])) for a, b in zip(lo, hi)]

    # -- PSD model ---------------------------------------------------------------------
    @property
    def psd_model(self):
        """the object the caller gave (simulator.py:267-270); a celerite term, which is not callable: its ``get_psd``"""
        return self._psd_given if callable(self._psd_given) else self._psd_given.get_psd

    @psd_model.setter
    def psd_model(self, new_psd_model):
        if not callable(new_psd_model) and not isinstance(new_psd_model, Term):
            raise ValueError("PSD model must be callable (e.g., a function or Astropy model).")
        self._psd_given = new_psd_model
        kernel = getattr(new_psd_model, "__self__", new_psd_model)
        if isinstance(kernel, PSDModel):
            try:
                # the reference's closed-form spectra (models/psd_models.py): every one that is the spectrum
                # of a celerite term is simulated through that term's coefficients
                kernel = kernel.to_term()
            except (ValueError, NotImplementedError, AttributeError):
                kernel = None
        self._kernel = kernel if isinstance(kernel, Term) else None
        self._psd_callable = None if self._kernel is not None else new_psd_model

    def _engine(self):
        if self._evaluator is None:
            n = len(self._times)
            self._evaluator = LogProbEvaluator(self._times, np.zeros(n), np.ones(n), device=self.device)
        if self._kernel is None:
            eng = self._evaluator._bind_lightcurves()
            eng.set_simulate_transform(self.transform)
            return eng, None
        model = DeviceModel(self._kernel, ConstantModel(0.0), np.zeros(1, dtype=bool))
        if not model.device_terms:
            raise ValueError("the device simulator needs device-expandable terms")
        eng = self._evaluator._bind(model)
        eng.set_simulate_transform(self.transform)     # (engines are shared per device: every simulator says which)
        return eng, model

    def warm_up(self):
        """Start building the inverse-transform plan of this simulator's grid on a helper thread (the engine keeps one
        plan per length): call it before work that does not need the simulator yet, e.g. the chains of the observed
        light curve."""
        from .gp import get_engine
        get_engine(self.device).start_simulate_warmup(self.fftndatapoints)

    def _psd_table(self):
        """The callable PSD on the grid's angular frequencies; the k = 0 entry is not used (the mean of
        the series is set afterwards) and a power law would be infinite there."""
        omega = np.fft.rfftfreq(self.fftndatapoints, self.sim_dt) * 2 * np.pi
        table = np.zeros(len(omega))
        table[1:] = np.asarray(self._psd_callable(omega[1:]), dtype=np.float64)
        if not np.all(np.isfinite(table)) or np.any(table < 0):
            raise ValueError("the PSD model returned negative or non-finite power")
        return table[None, :]

    # -- simulation --------------------------------------------------------------------
    def simulate(self, thetas=None, noise=True, want_clean=False, make_resident=False, seed=None, nsims=None,
                 index_base=None, pair_series=None, pdf_draws=None):
        """Light curves for S kernel parameter vectors ``thetas`` [S][P] (default: the kernel's
        current one; with a callable PSD: ``nsims`` realisations of it) in one device call ->
        dict(rates[S][N], dy[S][N], means[S], clean[S][N] | None).  ``index_base``: global index of the first of
        these series (mtg_set_stream_base) -- with the same ``seed``, series [index_base, index_base + S) of a set
        simulated in blocks are the ones a single call for the whole set would make.  That holds for what is drawn on the
        host as well (Kraft noise, a non-Gaussian flux PDF): given an ``index_base``, every series draws from a generator
        of its own, keyed by (seed, global index), instead of this simulator's one ``random_state``.
        ``pair_series``: the device's hand-made transform (grid lengths with large prime factors) packs series 2p and
        2p + 1 of a call into one complex transform, which ties a series' last bits to its neighbour's.  Default: on
        without an ``index_base``, off with one (a block's series must not depend on where the block was cut).  A caller
        that cuts at EVEN global indices -- so that every series keeps the partner it has in the whole set -- may turn it
        on and keep both the speed and the invariance (``ppp.protassov_test`` does).
        ``pdf_draws`` [S][seg_len] (tests): the white series a non-Gaussian flux PDF's adjustment starts from, instead of
        the device's own draws (``adjust_on="device"``) or this simulator's generator (``"host"``)."""
        if self.stream == "numpy":
            raise NotImplementedError("stream='numpy' serves generate_lightcurve / simulate_regularly_sampled / add_noise (one light "
                                      "curve per call, as the reference); the batched simulate() draws on the device")
        eng, model = self._engine()
        if seed is None:
            seed = int(self.random_state.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(self.random_state.randint(0, 2 ** 31 - 1))
        shaped = self.pdf.lower() != "gaussian"
        on_device = shaped and self.adjust_on == "device"
        host_adjust = shaped and not on_device
        host_side = host_adjust or (noise and self._noise_kind == 3 and self.adjust_on == "host")
        if noise and self._noise_kind == 3 and not host_side:
            eng.set_simulate_kraft(self._bkg_counts, self._bkg_rate_err, *self._kraft_tables(), self.kraft_counts)
        kw = dict(noise_kind=0 if (host_side or not noise) else self._noise_kind, sigma_noise=self.sigma_noise,
                  exposures=self._exposures, want_clean=want_clean and not host_side,
                  make_resident=make_resident and not host_side, want_segments=host_adjust)
        eng.set_simulate_pdf(self.pdf if on_device else 0, self.max_iter)
        if pdf_draws is not None and on_device:
            eng.set_simulate_pdf_draws(pdf_draws)
        eng.set_stream_base(index_base or 0)
        if pair_series and index_base is not None and int(index_base) % 2:
            raise ValueError("pair_series with an index_base needs an even index_base (series 2p and 2p + 1 share a transform)")
        # blocks of a larger set: no series shares a transform with a neighbour, unless the caller cut at even indices
        eng.set_simulate_pairs(index_base is None if pair_series is None else bool(pair_series))
        try:
            if model is None:
                out = eng.simulate_tk95(int(nsims or 1), seed, self.fftndatapoints, self.sim_dt, self.mean, self.seg_len,
                                        self.win_lo, self.win_hi, psd_table=self._psd_table(), **kw)
            else:
                if thetas is None:
                    thetas = np.tile(model.full[model.free_index][None, :], (int(nsims or 1), 1))
                out = eng.simulate_tk95(thetas, seed, self.fftndatapoints, self.sim_dt, self.mean, self.seg_len,
                                        self.win_lo, self.win_hi, **kw)
        finally:
            eng.set_stream_base(0)
            eng.set_simulate_pairs(True)
            eng.set_simulate_pdf(0)
            eng.set_simulate_pdf_draws(None)
            if make_resident:
                eng.bound_to = None    # whatever happened, the engine no longer holds this evaluator's dummy data
        if on_device:
            self.last_adjustment = eng.simulate_pdf_report()
            if self.last_adjustment["not_converged"]:      # the reference's warning (simulator.py:125-126), once per call
                warnings.warn("Lightcurve did not converge after %d iterations, PDF might be inaccurate. Try increase the "
                              "maximum number of iterations (%d of %d)" % (self.max_iter, self.last_adjustment["not_converged"], len(out["rates"])))
        if host_side:
            out = self._finish_on_host(out, noise, want_clean, None if index_base is None else (int(seed), int(index_base)),
                                       adjust=host_adjust, draws=pdf_draws)
            if make_resident:          # the refits want the set resident: upload what the host produced
                eng.set_lightcurves(self._times, out["rates"], out["dy"] + 1e-12, y_offset=out["means"])
        out.pop("segments", None)
        return out

    def _finish_on_host(self, out, noise, want_clean, keyed=None, adjust=None, draws=None):
        """Flux-PDF adjustment of the fine-grid segments, down-sampling and noise for the cases the device
        kernels do not cover (module docstring).  ``keyed`` = (seed, index_base): series l draws from
        RandomState([seed, index_base + l]) -- its values then do not depend on which block it was simulated in."""
        S = len(out["rates"])
        shared = self.random_state
        if adjust is None:
            adjust = self.pdf.lower() != "gaussian"

        def own_stream(l, phase):
            if keyed is not None:
                seed, g = keyed[0], keyed[1] + l
                self.random_state = np.random.RandomState(np.array(
                    [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, g & 0xFFFFFFFF, g >> 32, phase], dtype=np.uint32))

        rates, dy = [out["rates"][l] for l in range(S)], []
        try:
            # (the adjustment of every series comes before the noise of any, as it always did on the shared generator)
            if adjust:
                for l in range(S):
                    own_stream(l, 0)
                    rates[l] = self.downsample(self.segment_times, self._adjust_pdf(out["segments"][l], None if draws is None else draws[l]))
            clean = np.array(rates) if want_clean else None
            if noise:
                for l in range(S):
                    own_stream(l, 1)
                    rates[l], e = self.add_noise(rates[l])
                    dy.append(e)
        finally:
            self.random_state = shared
        rates = np.array(rates)
        dy = np.array(dy) if noise else np.zeros_like(rates)
        return dict(rates=rates, dy=dy, means=rates.mean(axis=1), clean=clean)

    def _kraft_tables(self):
        """(median[N][K], half[N][K]): Kraft, Burrows & Nousek's posterior median of the source counts and half the width of
        its 68 % interval for total counts 0 .. K - 1 (K = the first integer >= kraft_counts) at every epoch's background --
        what ``add_noise`` computes per faint epoch, once per distinct background."""
        if getattr(self, "_kraft_cache", None) is None:
            K = max(int(np.ceil(self.kraft_counts)), 1)
            med, half = np.empty((len(self._bkg_counts), K)), np.empty((len(self._bkg_counts), K))
            for b in np.unique(self._bkg_counts):
                rows = self._bkg_counts == b
                m = [kraft_median(c, b) for c in range(K)]
                h = [(lambda lo_hi: (lo_hi[1] - lo_hi[0]) / 2.0)(kraft_interval(c, b, 0.68)) for c in range(K)]
                med[rows], half[rows] = m, h
            self._kraft_cache = (med, half)
        return self._kraft_cache

    def _adjust_pdf(self, segment, draws=None):
        """Emmanoulopoulos et al. (2013), as simulator.py:65-140 runs it: a white series drawn from the
        wanted flux PDF (mean = the simulator's, standard deviation = the segment's) repeatedly takes the
        Fourier amplitudes of the TK95 segment and gives its values back by rank, until it stops changing."""
        from scipy import stats
        mean, std = self.mean, float(np.std(segment))
        if self.pdf.lower() == "lognormal":     # stats.py:116-129
            var = std ** 2
            pdf = stats.lognorm(np.sqrt(np.log(var / mean ** 2 + 1.0)), scale=mean ** 2 / np.sqrt(var + mean ** 2))
        else:                                   # uniform with that mean and variance, stats.py:132-146
            half = np.sqrt(3.0) * std
            pdf = stats.uniform(loc=mean - half, scale=2.0 * half)
        n = len(segment)
        amplitudes = np.abs(np.fft.rfft(segment))
        if draws is None:
            values = np.sort(pdf.rvs(size=n, random_state=self.random_state))[::-1]     # the flux values, descending
            current = self.random_state.permutation(values)
        else:                                   # the caller's white series (tests: the same one on the device)
            current = np.asarray(draws, dtype=np.float64).copy()
            values = np.sort(current)[::-1]
        for iteration in range(self.max_iter + 1):
            spectrum = amplitudes * np.exp(1j * np.angle(np.fft.rfft(current)))
            adjusted = np.fft.irfft(spectrum, n=n)
            new = np.empty(n)
            new[np.argsort(-adjusted)] = values                                       # same ranks, the PDF's values
            if np.allclose(new, current, rtol=1e-4):
                return new
            current = new
        warnings.warn("Lightcurve did not converge after %d iterations, PDF might be inaccurate. Try increase the "
                      "maximum number of iterations" % self.max_iter)
        return current

    def __str__(self):
        return "Simulator(\n  PSD Model: %s\n  PDF: %s\n) Noise: %s" % (self.psd_model, self.pdf, self.noise_name)

    def set_psd_params(self, psd_params):
        """Set attributes of the PSD model by name before the next ``generate_lightcurve`` (simulator.py:282-298)."""
        for name, value in psd_params.items():
            setattr(self._psd_given, name, value)
        self.psd_model = self._psd_given     # a closed-form spectrum is simulated through its celerite term: derive it again

    def simulate_regularly_sampled(self):
        """One TK95 realisation on the whole fine grid -- ``sim_timestamps``, longer and finer than the observed light
        curve -- with the mean set to the simulator's (simulator.py:369-394).  The reference returns stingray's Lightcurve;
        here an object with its ``time``, ``countrate``, ``dt``, ``n``, ``tseg`` and ``meanrate``."""
        if self.stream == "numpy":
            return RegularLightcurve(self.sim_timestamps, self._generate_with_reference_draws(whole_grid=True)["segments"][0], self.sim_dt)
        eng, model = self._engine()
        seed = int(self.random_state.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(self.random_state.randint(0, 2 ** 31 - 1))
        lo, hi = self._windows(self.sim_timestamps)
        kw = dict(noise_kind=0, want_segments=True)
        if model is None:
            out = eng.simulate_tk95(1, seed, self.fftndatapoints, self.sim_dt, self.mean, self.fftndatapoints, lo, hi,
                                    psd_table=self._psd_table(), **kw)
        else:
            out = eng.simulate_tk95(model.full[model.free_index][None, :], seed, self.fftndatapoints, self.sim_dt, self.mean,
                                    self.fftndatapoints, lo, hi, **kw)
        return RegularLightcurve(self.sim_timestamps, out["segments"][0], self.sim_dt)

    def _reference_draws(self):
        """The reference's draws for one light curve, from numpy's global generator in its order: the spectrum's standard
        normals (simulator.py:486) and the cut (:538, then stingray's truncate: first sample at or after the drawn start,
        last sample at or before start + sim_duration -- both kept, which the variances printed in
        docs/notebooks/celerite_variance.ipynb decide).  -> normals [1][2][nk], first index, samples in the cut"""
        grid = self.sim_timestamps
        normals = np.random.normal(0, size=(2, self.fftndatapoints // 2 + 1))
        shift = np.random.uniform(grid[0], grid[-1] - self.sim_duration)
        first = int(np.flatnonzero(grid >= shift)[0])
        last = int(np.flatnonzero(grid <= shift + self.sim_duration)[-1])
        return normals[None, :, :], first, last - first + 1

    def _generate_with_reference_draws(self, whole_grid=False):
        eng, model = self._engine()
        normals, first, count = self._reference_draws() if not whole_grid else \
            (np.random.normal(0, size=(2, self.fftndatapoints // 2 + 1))[None, :, :], 0, self.fftndatapoints)
        cut_times = self.strategy[0][0] + self.sim_dt / 2 + np.arange(count) * self.sim_dt
        lo, hi = self._windows(self.sim_timestamps if whole_grid else cut_times)
        eng.set_simulate_draws(normals, [first])
        try:
            if model is None:
                return eng.simulate_tk95(1, 0, self.fftndatapoints, self.sim_dt, self.mean, count, lo, hi,
                                         psd_table=self._psd_table(), noise_kind=0, want_segments=whole_grid)
            return eng.simulate_tk95(model.full[model.free_index][None, :], 0, self.fftndatapoints, self.sim_dt, self.mean, count,
                                     lo, hi, noise_kind=0, want_segments=whole_grid)
        finally:
            eng.set_simulate_draws(None, None)

    def generate_lightcurve(self):
        """One noise-free realisation on the observing pattern (simulator.py:397-420)."""
        if self.stream == "numpy":
            return self._generate_with_reference_draws()["rates"][0]
        return self.simulate(noise=False)["rates"][0]

    def add_noise(self, rates):
        """Noisy rates and their uncertainties (simulator.py:300-338; noise_models.py) for ONE light curve,
        on the host with this simulator's RandomState; the batched path adds Gaussian and Poisson noise on
        the device."""
        rates = np.asarray(rates, dtype=np.float64)
        rng = np.random if self.stream == "numpy" else self.random_state      # the reference's noise classes: the global generator
        if self._noise_kind == 1:
            return rates + rng.normal(scale=self.sigma_noise, size=len(rates)), \
                self.sigma_noise * np.ones(len(rates))
        expo, bkg = self._exposures, self._bkg_counts
        total = rng.poisson(rates * expo + bkg).astype(np.float64)
        net_rates = (total - bkg) / expo
        dy = np.sqrt((np.sqrt(total) / expo) ** 2 + self._bkg_rate_err ** 2)
        if self._noise_kind == 3:
            faint = total < self.kraft_counts          # Bayesian treatment of the faint epochs
            for i in np.nonzero(faint)[0]:
                n_i = int(round(total[i]))
                net_rates[i] = kraft_median(n_i, bkg[i]) / expo[i]
                lo, hi = kraft_interval(n_i, bkg[i], 0.68)
                dy[i] = (hi - lo) / 2.0 / expo[i]
        return net_rates, dy
