"""Light-curve simulation from the GP's power spectrum, on the device.

Mirror of the part of /root/reference/mind_the_gaps/simulator.py that the Protassov
loop uses: ``Simulator(psd_model, times, exposures, mean, pdf, bkg_rate, bkg_rate_err,
sigma_noise, aliasing_factor, extension_factor, epsilon, ...)`` (:143-275) with
``generate_lightcurve`` (:397-420: Timmer & Koenig 1995 on a fine, extended regular grid,
random segment cut, bin-averaging onto the observing pattern) and ``add_noise`` (:300-338;
noise_models.py Gaussian :152-184 and Poisson :29-78).  The grid / window arithmetic
below is the reference's; the spectrum draw, the inverse FFT (hipFFT), the cut, the
downsampling and the noise run on the GPU for ALL requested simulations at once
(``mtg_simulate_tk95``), and the result can stay resident as the light-curve set of the
next fitting sweep.

Not provided: the Emmanoulopoulos et al. (2013) amplitude-adjustment loop for non-Gaussian
flux PDFs (``pdf="lognormal" | "uniform"``) and the Kraft low-count posterior noise.
"""
import numpy as np

from .gp import DeviceModel, LogProbEvaluator
from .modeling import ConstantModel
from .models.psd_models import PSDModel
from .terms import Term

__all__ = ["Simulator"]


class Simulator:
    """Simulate light curves with the PSD of a celerite kernel and Gaussian flux PDF."""

    def __init__(self, psd_model, times, exposures, mean, pdf="gaussian", bkg_rate=None, bkg_rate_err=None,
                 sigma_noise=None, aliasing_factor=2, extension_factor=10, epsilon=1.001, max_iter=400,
                 random_state=None, device=0):
        """``psd_model``: the kernel ``Term`` whose PSD drives the simulation, its bound
        ``get_psd`` (what gpmodelling.py:509 passes), or one of the closed-form spectra of
        ``models.psd_models`` (what the tutorials pass).  Other arguments as in the reference."""
        if extension_factor < 1:
            raise ValueError("Extension factor must be greater than 1")
        if epsilon < 1:
            raise ValueError("Epsilon needs to be greater than 1!")
        if np.any(np.asarray(exposures) == 0):
            raise ValueError("Some exposure times are 0!")
        times = np.asarray(times, dtype=np.float64)
        self._exposures = np.full(len(times), exposures, dtype=np.float64) if np.isscalar(exposures) \
            else np.asarray(exposures, dtype=np.float64)
        if pdf.lower() not in ["gaussian", "lognormal", "uniform"]:
            raise ValueError("%s not implemented! Currently implemented: Gaussian, Uniform or Lognormal" % pdf)
        if pdf.lower() != "gaussian":
            raise NotImplementedError("only the Gaussian flux PDF (Timmer & Koenig 1995) runs on the device; "
                                      "the E13 amplitude adjustment is not provided")
        self.pdf = pdf
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
        # noise model (simulator.py:255-262)
        if sigma_noise is None:
            if bkg_rate is None or np.all(np.asarray(bkg_rate) == 0):
                self.noise_name, self._noise_kind, self.sigma_noise = "Poisson", 2, 0.0
            else:
                raise NotImplementedError("Kraft noise (background counts) is not provided on the device")
        else:
            self.noise_name, self._noise_kind, self.sigma_noise = "Gaussian", 1, float(sigma_noise)
        # observing windows ("strategy", simulator.py:265-267) as index ranges of the cut segment,
        # whose first fine sample sits sim_dt / 2 after the start of the first window
        half_bins = self._exposures / 2 * epsilon
        self.strategy = [(t - h, t + h) for t, h in zip(times, half_bins)]
        self.seg_len = min(int(np.ceil(self.sim_duration / self.sim_dt)), self.fftndatapoints)
        seg_times = self.strategy[0][0] + self.sim_dt / 2 + np.arange(self.seg_len) * self.sim_dt
        self.win_lo = np.searchsorted(seg_times, times - half_bins, side="left").astype(np.int32)
        self.win_hi = np.searchsorted(seg_times, times + half_bins, side="left").astype(np.int32)
        self._evaluator = None

    # -- PSD model ---------------------------------------------------------------------
    @property
    def psd_model(self):
        return self._kernel.get_psd

    @psd_model.setter
    def psd_model(self, new_psd_model):
        kernel = getattr(new_psd_model, "__self__", new_psd_model)
        if isinstance(kernel, PSDModel):
            # the reference's closed-form spectra (models/psd_models.py): every one that is the
            # spectrum of a celerite term is simulated through that term's coefficients
            kernel = kernel.to_term()
        if not isinstance(kernel, Term):
            raise ValueError("PSD model must be a Term, its get_psd method, or a models.psd_models spectrum "
                             "with a celerite equivalent")
        self._kernel = kernel

    def _engine_and_model(self):
        if self._evaluator is None:
            n = len(self._times)
            self._evaluator = LogProbEvaluator(self._times, np.zeros(n), np.ones(n), device=self.device)
        model = DeviceModel(self._kernel, ConstantModel(0.0), np.zeros(1, dtype=bool))
        if not model.device_terms:
            raise ValueError("the device simulator needs device-expandable terms")
        return self._evaluator._bind(model), model

    # -- simulation --------------------------------------------------------------------
    def simulate(self, thetas=None, noise=True, want_clean=False, make_resident=False, seed=None):
        """Light curves for S kernel parameter vectors ``thetas`` [S][P] (default: the
        kernel's current one) in one device call ->
        dict(rates[S][N], dy[S][N], means[S], clean[S][N] | None)."""
        eng, model = self._engine_and_model()
        if thetas is None:
            thetas = model.full[model.free_index][None, :]
        if seed is None:
            seed = int(self.random_state.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(self.random_state.randint(0, 2 ** 31 - 1))
        out = eng.simulate_tk95(thetas, seed, self.fftndatapoints, self.sim_dt, self.mean, self.seg_len,
                                self.win_lo, self.win_hi, noise_kind=self._noise_kind if noise else 0,
                                sigma_noise=self.sigma_noise, exposures=self._exposures, want_clean=want_clean,
                                make_resident=make_resident)
        if make_resident:
            eng.bound_to = None        # the engine now holds the simulated set, not this evaluator's dummy data
        return out

    def generate_lightcurve(self):
        """One noise-free realisation on the observing pattern (simulator.py:397-420)."""
        return self.simulate(noise=False)["rates"][0]

    def add_noise(self, rates):
        """Noisy rates and their uncertainties (simulator.py:300-338) for ONE light curve, on
        the host with this simulator's RandomState; the batched path adds noise on the device."""
        rates = np.asarray(rates, dtype=np.float64)
        if self._noise_kind == 1:
            return rates + self.random_state.normal(scale=self.sigma_noise, size=len(rates)), \
                self.sigma_noise * np.ones(len(rates))
        total_counts = rates * self._exposures
        poiss = self.random_state.poisson(total_counts)
        return poiss / self._exposures, np.sqrt((np.sqrt(poiss) / self._exposures) ** 2)
