"""Where the refits of BASELINE configs[3] spend their time: the lock-step L-BFGS start (calls, rows per call,
seconds) against the 500 ensemble steps, for the null and the alternative kernel.

    python scripts/c3_phases.py [nsims] [N] [walkers] [steps]  ->  one JSON line
"""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mind_the_gaps_amd import ppp, synthetic as synth, terms
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.simulator import Simulator

AMP, OTHER = (-10, 50), (-10, 10)


def main(nsims=2000, N=10000, W=256, steps=500):
    th = synth.truth(synth.ALT_MODEL)
    rng = np.random.default_rng(20250704 + 3)
    times = synth.make_times(N, rng)

    def null_kernel():
        return DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])

    def alt_kernel():
        return null_kernel() + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])

    sim = Simulator(null_kernel(), times, 0.04, 100.0, "Gaussian", sigma_noise=1.0, extension_factor=2, random_state=3)
    theta = np.tile(null_kernel().get_parameter_vector(), (nsims, 1))
    out = sim.simulate(theta)
    res = {}
    inner = ppp.batched_minimize
    for name, kernel in (("null", null_kernel()), ("alt", alt_kernel())):
        calls = []

        def counted(fun, *a, **k):
            def f(x, lc):
                t0 = time.perf_counter()
                v = fun(x, lc)
                calls.append((len(x), time.perf_counter() - t0))
                return v
            t0 = time.perf_counter()
            r = inner(f, *a, **k)
            calls.append((-1, time.perf_counter() - t0))
            return r

        ppp.batched_minimize = counted
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            fit = ppp.derive_posteriors_batch(times, out["rates"], out["dy"], kernel, walkers=W, max_steps=steps, fit=True,
                                              seed=5, store_chain=False, quiet=True)
            total = time.perf_counter() - t0
        ppp.batched_minimize = inner
        fit_s = calls[-1][1]
        rows = np.array([c[0] for c in calls[:-1]])
        secs = np.array([c[1] for c in calls[:-1]])
        res[name] = {"total_s": total, "fit_s": fit_s, "fit_calls": len(rows), "fit_seconds_in_calls": float(secs.sum()),
                     "fit_rows_median": float(np.median(rows)), "fit_rows_total": int(rows.sum()),
                     "fit_call_ms_median": float(np.median(secs) * 1e3), "rest_s": total - fit_s, "seconds": fit.seconds,
                     "max_lnL_mean": float(np.mean(fit.max_loglikelihood)), "fit_lnL_mean": float(np.mean(fit.fit_loglikelihood))}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:5]])
