"""Soak of the two code paths added in round 5, on random shapes:
  (1) mtg_pair_contexts -- random null / alternative pair (tests/test_pipe_gpu.py: PAIRS), N in [64, 3000], light curves,
      walkers, both SHO regimes, prior rejections: paired rows == each model's own pipelined kernel, bit for bit;
  (2) the simulator's chirp-z transform -- random sampling patterns (grid lengths of either parity, 10^4 .. 10^5 points),
      1 .. 7 series: against hipFFT's own transform of the same spectrum (Simulator(transform="chirp-z" / "library")), 1e-11 of the series' scale.
    python scripts/r05_soak.py [cases_pairs] [cases_czt] [seed]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.simulator import Simulator

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_czt = int(sys.argv[2]) if len(sys.argv) > 2 else 60
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
K = synth
PAIRS = [([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_LORENTZIAN]), ([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_SHO]),
         ([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_REAL, K.K_SHO]), ([K.K_DRW, K.K_LORENTZIAN], [K.K_DRW, K.K_LORENTZIAN, K.K_LORENTZIAN]),
         ([K.K_DRW, K.K_LORENTZIAN], [K.K_DRW, K.K_LORENTZIAN, K.K_SHO]), ([K.K_SHO, K.K_LORENTZIAN], [K.K_DRW, K.K_SHO, K.K_LORENTZIAN]),
         ([K.K_SHO, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_SHO]), ([K.K_DRW, K.K_REAL, K.K_SHO], [K.K_DRW, K.K_REAL, K.K_SHO, K.K_LORENTZIAN])]
rng = np.random.default_rng(20251004 + seed0)
t_start = time.perf_counter()
bad = shared = alone_n = 0
engines = [Engine(0), Engine(0)]
for case in range(n_pairs):
    kinds_of = PAIRS[rng.integers(len(PAIRS))]
    if rng.random() < 0.5:
        kinds_of = kinds_of[::-1]
    N, L, W = int(rng.integers(64, 3000)), int(rng.integers(1, 12)), int(rng.integers(8, 90))
    t, y, dy = synth.make_lightcurves(N, L, seed=int(rng.integers(1 << 30)))
    lc = rng.integers(0, L, L * W).astype(np.int32)
    thetas, alone = [], []
    for i, kinds in enumerate(kinds_of):
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        eng = engines[i]
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, free, bounds)
        eng.set_time_parallel(0); eng.set_pipeline(1)
        th = synth.draw_thetas(kinds, L * W, seed=int(rng.integers(1 << 30)), percent=float(rng.choice([0.1, 0.5])))
        th[::37, 0] = 60.0
        thetas.append(th)
        alone.append(eng.loglike(th, lc, add_prior=True))
    engines[0].pair_with(engines[1])
    got = [None, None]
    def run(i):
        got[i] = engines[i].loglike(thetas[i], lc, add_prior=True)
    jobs = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    [j.start() for j in jobs]; [j.join() for j in jobs]
    stats = engines[0].pair_stats()
    engines[0].unpair()
    same = all(np.array_equal(got[i][0], alone[i][0]) and np.array_equal(got[i][1], alone[i][1]) for i in (0, 1))
    shared += stats["paired"]
    # (a batch whose rows come grouped by light curve is not sorted, and an unsorted multi-structure batch keeps the one-lane
    # kernels: nothing to pair, stats stay at zero -- not an error; a pair that breaks or a launch made alone is)
    # ... and where only ONE of the two models' batches is pipelined (the other, a multi-structure batch that arrives grouped and
    # unsorted through the host entry point, keeps the one-lane kernels) the pipelined one waits out its patience once and goes
    # alone: the designed fall-back, counted, not an error.  An error is a VALUE that differs.
    alone_n += int(stats["broken"])
    if not same:
        bad += 1
        print("PAIR MISMATCH case %d: kinds %r N %d L %d W %d stats %r" % (case, kinds_of, N, L, W, stats), flush=True)
print("pairs: %d cases, %d in a shared launch, %d fell back to separate launches (partner not pipelined), %d with a differing value, %.1f s"
      % (n_pairs, shared, alone_n, bad, time.perf_counter() - t_start), flush=True)
for eng in engines:
    eng.close()

t_start = time.perf_counter()
worst, bad_czt, parities = 0.0, 0, set()
kernel = DampedRandomWalk(np.log(100.0), np.log(2 * np.pi / 10), bounds=[(-10, 50), (-10, 10)]) + \
    Lorentzian(np.log(50.0), np.log(20.0), np.log(2 * np.pi / 3.0), bounds=[(-10, 50), (-10, 10), (-10, 10)])
for case in range(n_czt):
    n = int(rng.integers(40, 260))
    times = synth.make_times(n, np.random.default_rng(int(rng.integers(1 << 30))))
    ext = float(rng.choice([1.5, 2, 3]))
    S = int(rng.integers(1, 8))
    got = {}
    for mode, transform in (("1", "chirp-z"), ("0", "library")):
        sim = Simulator(kernel, times, 0.04, 25.0, "Gaussian", sigma_noise=0.5, extension_factor=ext, random_state=4, transform=transform)
        thetas = np.tile(sim._engine()[1].full[sim._engine()[1].free_index][None, :], (S, 1))
        got[mode] = sim.simulate(thetas, seed=1000 + case, want_clean=True)["clean"]
    parities.add(sim.fftndatapoints % 2)
    err = float(np.max(np.abs(got["1"] - got["0"])) / np.std(got["0"]))
    worst = max(worst, err)
    if not err <= 1e-11:
        bad_czt += 1
        print("CZT MISMATCH case %d: nfft %d S %d err %.3e" % (case, sim.fftndatapoints, S, err), flush=True)
print("chirp-z: %d cases (grid parities seen: %s), %d bad, worst difference %.2e of the series' scale, %.1f s"
      % (n_czt, sorted(parities), bad_czt, worst, time.perf_counter() - t_start), flush=True)
raise SystemExit(1 if bad or bad_czt else 0)
