"""BASELINE configs[3] as a WORKFLOW on one MI355X: the Protassov posterior-predictive test with
2000 simulated light curves, N = 10 000 irregular epochs, null DRW+SHO against alternative
DRW+SHO+Lorentzian, 256 walkers x 500 steps per refit (ppp.protassov_test: observed chains ->
device TK95 simulation from the null posterior -> lock-step refits of both kernels -> p-value).
bench.py times the likelihood sweep this spends its time in; this script times the whole thing.

    python scripts/config3_probe.py [nsims] [N] [walkers] [steps]  ->  one JSON line
"""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import synthetic as synth, terms
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.ppp import protassov_test
from mind_the_gaps_amd.simulator import Simulator

AMP, OTHER = (-10, 50), (-10, 10)


def _kernels():
    th = synth.truth(synth.ALT_MODEL)

    def null_kernel():
        return DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4],
                                                                                   bounds=[AMP, OTHER, OTHER])

    def alt_kernel():
        return null_kernel() + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])
    return null_kernel, alt_kernel


def _observed(N, device):
    """The "observed" light curve: one realisation of the null process on the irregular sampling -> (lc, sim, seconds)."""
    null_kernel, _ = _kernels()
    rng = np.random.default_rng(20250704 + 3)
    times = synth.make_times(N, rng)
    exposure = 0.04                                   # below the shortest spacing (0.05 d)
    mean = 100.0
    t0 = time.perf_counter()
    sim = Simulator(null_kernel(), times, exposure, mean, "Gaussian", sigma_noise=1.0, extension_factor=2, random_state=3,
                    device=device)
    rates = sim.generate_lightcurve()
    noisy, dy = sim.add_noise(rates)
    return GappyLightcurve(times, noisy, dy, exposures=exposure), sim, time.perf_counter() - t0


def observed_chains_alone(N=10000, W=256, device=0):
    """Seconds of the observed light curve's chain of each model run ALONE (fit + up to 1000 steps, as protassov_test runs
    them): what ranks 0 and 1 of a sharded test each spend on step 1 (ppp.protassov_test, observed_split), where one GPU
    runs both side by side.  For the one-GPU projection of the 8-GPU time."""
    from mind_the_gaps_amd.gpmodelling import GPModelling
    lc, _, _ = _observed(N, device)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, make in zip(("null", "alt"), _kernels()):
            best = np.inf
            for rep in range(2):        # (the second run has its kernels and plans warm, as inside the workflow)
                g = GPModelling(lc, make(), device=device, random_state=np.random.RandomState(11 + rep))
                t0 = time.perf_counter()
                g.derive_posteriors(fit=True, max_steps=1000, walkers=W, progress=False, device_sampler=True)
                best = min(best, time.perf_counter() - t0)
            out[name] = best
    return out


def run(nsims=2000, N=10000, W=256, steps=500, sharded=False, device=0, concurrent_refits="auto", reproducible=True,
        keep_T_sim=False, pdf="Gaussian"):
    """-> dict (the JSON line of this script).  bench.py calls it for its `workflow_config3` entry; ``sharded``: inside
    a torch.distributed job, the simulated light curves split over the ranks (ppp.protassov_test(sharded=True)).
    ``reproducible`` (default): T_sim and the p-value do not depend on the number of ranks or the split -- the sharded
    run of `bench.py --gpus N` must print the p-value of the one-GPU run."""
    null_kernel, alt_kernel = _kernels()
    lc, sim, t_obs_sim = _observed(N, device)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        res = protassov_test(lc, null_kernel(), alt_kernel(), nsims=nsims, walkers=W, max_steps=1000, sim_walkers=W,
                             sim_steps=steps, sigma_noise=1.0, extension_factor=2, seed=1, device=device, sharded=sharded,
                             concurrent_refits=concurrent_refits, reproducible=reproducible, pdf=pdf)
        el = time.perf_counter() - t0
    evals = 2 * nsims * W * (steps + 1)
    extra = {"T_sim": [float(v) for v in res["T_sim"]]} if keep_T_sim else {}
    return {
        **extra,
        "workflow": "protassov_test, BASELINE configs[3]" + (", simulated light curves sharded over the ranks" if sharded else " on one GPU"),
        "nsims": nsims, "N": N, "walkers": W, "refit_steps": steps, "fft_points_per_simulation": sim.fftndatapoints,
        "flux_pdf": pdf, "segment_points_per_simulation": sim.seg_len,
        "observed_lightcurve_s": t_obs_sim, "whole_test_s": el,
        "seconds": {k: float(v) for k, v in res["seconds"].items()}, "split": res["split"],
        "refit_evaluations": evals, "refit_evaluations_per_s_end_to_end": evals / el,
        "T_obs": res["T_obs"], "p_value": res["p_value"], "reproducible": bool(res["reproducible"]),
        "T_sim_checksum": float(np.sum(res["T_sim"])),
        "T_sim_quantiles_50_90_99": [float(q) for q in np.quantile(res["T_sim"], [0.5, 0.9, 0.99])],
        # (a sharded test runs the observed chains on ranks 0 and 1 only: None elsewhere)
        "null_converged": None if res["null"] is None else bool(res["null"].converged),
        "alt_converged": None if res["alt"] is None else bool(res["alt"].converged),
    }


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:5]]
    mode = {"1": True, "0": False}.get(os.environ.get("MTG_C3_CONCURRENT_REFITS", ""), "auto")
    print(json.dumps(run(*args, concurrent_refits=mode)), flush=True)
