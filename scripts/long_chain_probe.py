"""The tutorial's long chains as written (docs/notebooks/tutorial_ppp.ipynb cells 7, 9): derive_posteriors(max_steps=50000,
fit=True, cores=cpus) with the default 12 walkers on 1000 points, null (RealTerm) and alternative (ComplexTerm + RealTerm):
wall time, iterations run, where the time goes (cProfile).   python scripts/long_chain_probe.py"""
import cProfile, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mind_the_gaps_amd import terms
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models.psd_models import BendingPowerlaw
from mind_the_gaps_amd.simulator import Simulator

np.random.seed(10)
times = np.arange(0, 1000)
mean, variance_drw, w_bend = 100, 100.0, 2 * np.pi / 20
sim = Simulator(BendingPowerlaw(variance_drw, w_bend), times, np.ones(1000), mean, pdf="Gaussian", extension_factor=2, random_state=10)
rates = sim.generate_lightcurve()
noisy, dy = sim.add_noise(rates)
lc = GappyLightcurve(times, noisy, dy, exposures=1)
bounds_drw = dict(log_a=(-10, 50), log_c=(-10, 10))
null_kernel = terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend), bounds=bounds_drw)
w = 2 * np.pi / 10
alt_kernel = terms.ComplexTerm(log_a=np.log(variance_drw), log_c=np.log(0.5 * w / 80), log_d=np.log(w),
                               bounds=dict(log_a=(-10, 50), log_c=(-10, 10), log_d=(-5, 5))) \
    + terms.RealTerm(log_a=np.log(variance_drw), log_c=np.log(w_bend), bounds=bounds_drw)
GPModelling(lc, null_kernel).derive_posteriors(max_steps=600, fit=True, cores=15, progress=False)     # warm-up
for name, kernel in (("null", null_kernel), ("alternative", alt_kernel)):
    for converge in (True, False):
        m = GPModelling(lc, kernel)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pr.enable()
            m.derive_posteriors(max_steps=50000, fit=True, cores=15, converge=converge, progress=False)
            pr.disable()
        dt = time.perf_counter() - t0
        it = m.sampler.iteration
        print("%s, converge=%s: %d iterations in %.2f s = %.1f us per iteration; max lnL %.4f; samples %s" %
              (name, converge, it, dt, 1e6 * dt / it, m.max_loglikelihood, m.mcmc_samples.shape), flush=True)
        if not converge:
            st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(14)
