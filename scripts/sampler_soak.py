"""Soak of the device-resident sampler against its host replay with the ORACLE likelihood (tests/philox_replay.py), on
random models (rank <= 6, any term kind), light-curve lengths, ensembles and walker counts -- speculative iterations or not
as the library chooses.  A replay can only be asked to agree until an accept decision lands within rounding of its
threshold: a case counts as bad when the chains part ways by more than 1e-8 before that could explain it, i.e. when the FIRST
difference between device and replay is larger than 1e-9 in a coordinate.   python scripts/sampler_soak.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import philox_replay
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from oracle import celerite as oracle_c
import test_fuzz_gpu as F

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
eng = Engine(0)
bad = diverged = singular = 0
kernels = {}
for case in range(cases):
    kinds = F.random_model(rng, 6, 3)
    N = int(rng.choice([40, 256, 300, 1000, 4100, 10000]))
    E = int(rng.choice([1, 1, 2, 5]))
    P = sum(synth.NPARAMS[k] for k in kinds)
    W = int(2 * P + 2 * rng.integers(0, 12))
    steps = int(rng.choice([8, 15, 25]))
    seed = int(rng.integers(1, 1 << 40))
    t, y, dy = synth.make_lightcurves(N, E, seed=int(rng.integers(1 << 30)))
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    p0 = synth.truth(kinds) * (1 + 0.02 * rng.standard_normal((E, W, P)))

    def oracle_lnp(q, ens):
        fullv = np.hstack([q, y.mean(axis=1)[ens][:, None]])
        return oracle_c.logprob_batch(t, y, dy, kinds, fullv, bounds=bounds, lc_index=ens.astype(np.int32), add_prior=True, nthreads=8)[0]
    eng.ensemble_init(p0, seed=seed)
    lnp0 = oracle_lnp(p0.reshape(E * W, -1), np.repeat(np.arange(E), W)).reshape(E, W)
    chain, lnp_chain = eng.ensemble_run(steps, store_chain=True)
    kernels[eng.last_solver.split("<")[0]] = kernels.get(eng.last_solver.split("<")[0], 0) + 1
    ref_chain, ref_lnp, ref_acc = philox_replay.run(p0, lnp0, oracle_lnp, steps, seed)
    diff = np.max(np.abs(chain - ref_chain).reshape(steps, -1), axis=1)
    first = int(np.argmax(diff > 1e-9)) if np.any(diff > 1e-9) else -1
    if first >= 0:
        # the chains parted at iteration `first`: legitimate only if an accept decision there was within rounding of its threshold,
        # i.e. the two differ in WHICH walkers moved, by whole moves, not by small amounts
        moved = np.abs(chain[first] - ref_chain[first]).max(axis=-1)         # per walker
        parted = moved > 1e-9
        whole = np.all((np.abs(chain[first][parted] - (ref_chain[first - 1][parted] if first else p0[parted])).max(axis=-1) < 1e-9) |
                       (np.abs(ref_chain[first][parted] - (ref_chain[first - 1][parted] if first else p0[parted])).max(axis=-1) < 1e-9))
        w = np.argwhere(parted)
        prev = chain[first - 1] if first else p0
        stayed = [tuple(x) for x in w if np.abs(chain[first][x[0], x[1]] - prev[x[0], x[1]]).max() < 1e-12]   # device rejected, replay moved
        e0_, w0_ = stayed[0] if stayed else w[0]
        # a position where the evaluators do not even agree with each other -- the sweep, the time-parallel kernels and the C
        # oracle, all three of celerite's family -- is a numerically singular covariance (undamped cosines, Matern with a tiny
        # eps): lnP there is rounding noise and so is any accept decision taken on it
        spread = []
        for th in (chain[first][e0_, w0_], ref_chain[first][e0_, w0_]):
            vals = []
            for mode in (0, 1):
                eng.set_time_parallel(mode)
                vals.append(eng.loglike(th[None, :], np.array([e0_], dtype=np.int32), add_prior=True)[0][0])
            eng.set_time_parallel(2)
            vals.append(oracle_lnp(th[None, :], np.array([e0_]))[0])
            vals = np.array(vals)
            spread.append(np.inf if not np.all(np.isfinite(vals)) else float(np.ptp(vals) / np.abs(vals).max()))
        if whole and parted.sum() <= 2:
            diverged += 1
        elif max(spread) > 1e-7:
            singular += 1
            print("   case %d: chains part at iteration %d on a numerically singular covariance (evaluators spread %.1e): kinds %s" % (case, first, max(spread), kinds), flush=True)
        else:
            bad += 1
            for e_, w_ in w[:3]:
                print("   walker (%d, %d): device lnP %.15g -> %.15g, replay lnP %.15g -> %.15g" % (
                    e_, w_, (lnp_chain[first - 1] if first else lnp0)[e_, w_], lnp_chain[first][e_, w_],
                    (ref_lnp[first - 1] if first else lnp0)[e_, w_], ref_lnp[first][e_, w_]), flush=True)
            # the first walker's two positions through every evaluator: sweep, time-parallel, C oracle, 40-digit dense
            from oracle import dense
            e_, w_ = e0_, w0_
            for label, th in (("device's", chain[first][e_, w_]), ("replay's", ref_chain[first][e_, w_])):
                fullv = np.hstack([th, y.mean(axis=1)[e_]])
                co = dense.build_coeffs(kinds, fullv[:P])
                try:
                    d = float(dense.dense_loglike(t, y[e_], dy[e_], co, mean_kind=0, mean_params=(fullv[P],)))   # LAPACK float64
                except Exception as ex:
                    d = float("nan")
                vals = {}
                for mode in (0, 1):
                    eng.set_time_parallel(mode)
                    o, st_ = eng.loglike(th[None, :], np.array([e_], dtype=np.int32), add_prior=True)
                    vals[mode] = (o[0], int(st_[0]))
                eng.set_time_parallel(2)
                orc = oracle_c.logprob_batch(t, y, dy, kinds, fullv[None, :], bounds=bounds, lc_index=np.array([e_], dtype=np.int32), add_prior=True)
                print("   %s position of walker (%d, %d): sweep %.12g (status %d), time-parallel %.12g (status %d), oracle %.12g (status %d), dense %.12g"
                      % (label, e_, w_, vals[0][0], vals[0][1], vals[1][0], vals[1][1], orc[0][0], int(orc[1][0]), d), flush=True)
            print("SAMPLER MISMATCH case %d: kinds %s N %d E %d W %d steps %d first diff at %d (%.2e), kernel %s" % (case, kinds, N, E, W, steps, first, diff[first], eng.last_solver), flush=True)
print("sampler: %d cases, %d bad, %d parted by one borderline accept decision, %d on a numerically singular covariance; kernels %s"
      % (cases, bad, diverged, singular, kernels), flush=True)
raise SystemExit(1 if bad else 0)
