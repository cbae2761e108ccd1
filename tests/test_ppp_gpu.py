"""GPU tests of the lock-step driver (many light curves in one launch) and of the
larger BASELINE configurations."""
import warnings

import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd import terms
from mind_the_gaps_amd.gpmodelling import GPModelling
from mind_the_gaps_amd.lightcurves import GappyLightcurve
from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
from mind_the_gaps_amd.ppp import derive_posteriors_batch
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu
AMP, OTHER = (-10, 50), (-10, 10)


def null_kernel():
    th = synth.truth(synth.NULL_MODEL)
    return DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER]) + terms.SHOTerm(th[2], th[3], th[4],
                                                                               bounds=[AMP, OTHER, OTHER])


def alt_kernel():
    th = synth.truth(synth.ALT_MODEL)
    return null_kernel() + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])


def test_batch_posteriors_match_oracle_and_single_lightcurve_runs():
    N, L, W = 400, 6, 12
    t, y, dy = synth.make_lightcurves(N, L, seed=31)
    y += 5.0 * np.arange(L)[:, None]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = derive_posteriors_batch(t, y, dy, null_kernel(), walkers=W, max_steps=80, fit=True, seed=4)
    assert res.max_loglikelihood.shape == (L,) and res.max_parameters.shape == (L, 5)
    assert res.tau.shape == (L, 5) and res.parameter_names[0] == "kernel:terms[0]:log_S0"
    # every reported maximum is the oracle's lnP of that sample on that light curve
    full = np.hstack([res.max_parameters, y.mean(axis=1)[:, None]])
    bounds = np.vstack([synth.bounds_for(synth.NULL_MODEL), [(-np.inf, np.inf)]])
    ref = oracle_c.logprob_batch(t, y, dy, synth.NULL_MODEL, full, bounds=bounds,
                                 lc_index=np.arange(L, dtype=np.int32), add_prior=True)[0]
    assert np.max(np.abs(res.max_loglikelihood - ref) / np.abs(ref)) < 1e-8
    # the lock-step fit lands where scipy's L-BFGS-B does light curve by light curve (the
    # surfaces are flat and both optimisers use noisy forward differences: a few units
    # of lnL either way), and the chains stay at that mode
    for l in (0, L - 1):
        g = GPModelling(GappyLightcurve(t, y[l], dy[l]), null_kernel())
        sol = g.fit()
        assert abs(res.fit_loglikelihood[l] + sol.fun) < 5.0
        assert res.max_loglikelihood[l] >= res.fit_loglikelihood[l] - 1.0
        ref_fit = oracle_c.logprob_batch(t, y[l], dy[l], synth.NULL_MODEL,
                                         np.append(res.fit_parameters[l], y[l].mean()))[0][0]
        assert abs(res.fit_loglikelihood[l] - ref_fit) / abs(ref_fit) < 1e-8


def test_lrt_statistic_null_vs_alternative():
    """BASELINE configs[2]: null vs alternative on the same data; T = -2 (lnL_null - lnL_alt) >= 0
    up to sampling noise (nested models)."""
    N, L, W = 500, 4, 16
    t, y, dy = synth.make_lightcurves(N, L, seed=41)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        null = derive_posteriors_batch(t, y, dy, null_kernel(), walkers=W, max_steps=200, seed=1, store_chain=False)
        alt = derive_posteriors_batch(t, y, dy, alt_kernel(), walkers=W, max_steps=200, seed=2, store_chain=False)
    T = -2.0 * (null.max_loglikelihood - alt.max_loglikelihood)
    assert T.shape == (L,) and np.all(np.isfinite(T)) and np.all(T > -3.0)


def test_config5_stress_parity(engine):
    """BASELINE configs[4]: N = 200 000 irregular samples, 5 SHO terms (J = 10)."""
    kinds = [synth.K_SHO] * 5
    N, B = 200000, 24
    t, y, dy = synth.make_lightcurves(N, 1, seed=20250709)
    th = synth.truth(kinds)
    for i in range(5):
        th[3 * i:3 * i + 3] = [np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
    rng = np.random.default_rng(5)
    theta = th + 0.05 * np.abs(th) * rng.standard_normal((B, len(th)))
    full = np.concatenate([th, [0.0]])
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, np.arange(15, dtype=np.int32), bounds)
    out, st = engine.loglike(theta, add_prior=True)
    ref, rst = oracle_c.logprob_batch(t, y[0], dy[0], kinds, np.hstack([theta, np.full((B, 1), y.mean())]),
                                      bounds=bounds, add_prior=True, nthreads=8)
    assert np.array_equal(st, rst) and np.all(st == 0)
    assert np.max(np.abs(out - ref) / np.abs(ref)) < 1e-8


def test_encode_properties_at_baseline_size(engine):
    """Size-independent properties at N = 1e4, J = 6 (no oracle): permutation of the batch
    permutes the output; duplicating light curves duplicates results; scaling y, dy and the
    amplitudes by s shifts lnL by -N ln s."""
    kinds = synth.ALT_MODEL
    N, L, W = 10000, 8, 64
    t, y, dy = synth.make_lightcurves(N, L, seed=99)
    y[L // 2:] = y[:L // 2]; dy[L // 2:] = dy[:L // 2]                   # duplicated light curves
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    theta = np.tile(synth.draw_thetas(kinds, W, seed=2), (L, 1))
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    out, st = engine.loglike(theta, lc)
    assert np.all(st == 0)
    o = out.reshape(L, W)
    assert np.array_equal(o[:L // 2], o[L // 2:])
    perm = np.random.default_rng(0).permutation(L * W)
    out_p, _ = engine.loglike(theta[perm], lc[perm])
    assert np.array_equal(out_p, out[perm])
    s = 3.0                                                               # K -> s^2 K, r -> s r
    engine.set_lightcurves(t, s * y, s * (dy + 1e-12), y_offset=s * y.mean(axis=1))
    theta_s = theta.copy()
    theta_s[:, [0, 2, 5]] += 2.0 * np.log(s)                              # log_S0 of DRW, SHO, Lorentzian
    out_s, st_s = engine.loglike(theta_s, lc)
    assert np.all(st_s == 0)
    assert np.max(np.abs(out_s - (out - N * np.log(s))) / np.abs(out)) < 1e-10


def test_protassov_test_end_to_end():
    """README workflow on the GPU: observed LRT, simulated light curves from the null posteriors,
    lock-step refits, p-value.  Data drawn from the null model must not look periodic."""
    from mind_the_gaps_amd.ppp import protassov_test
    from mind_the_gaps_amd.simulator import Simulator
    times = np.arange(0.5, 250.0, 1.0)
    th = synth.truth([synth.K_DRW])
    drw = lambda: DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER])
    lor = synth.truth([synth.K_LORENTZIAN])
    sim = Simulator(drw(), times, 0.2, 100.0, sigma_noise=2.0, extension_factor=3, random_state=9)
    obs = sim.simulate()                                           # one light curve from the null model
    lc = GappyLightcurve(times, obs["rates"][0], obs["dy"][0], exposures=0.2)
    alt = drw() + Lorentzian(lor[0], lor[1], lor[2], bounds=[AMP, OTHER, OTHER])
    res = protassov_test(lc, drw(), alt, nsims=24, walkers=16, max_steps=120, sim_steps=60, sigma_noise=2.0, seed=5)
    assert res["T_sim"].shape == (24,) and np.all(np.isfinite(res["T_sim"])) and np.isfinite(res["T_obs"])
    assert np.all(res["T_sim"] > -5.0)                             # nested models: alt never much worse
    assert 1 / 25 <= res["p_value"] <= 1.0 and res["p_value"] > 0.04
    assert abs(res["p_value_percentile"] - res["p_value"]) <= 1 / 24 and 0.0 <= res["p_value_percentile"] <= 1.0     # the tutorial's expression
    assert res["lightcurves"]["rates"].shape == (24, 250)
    # the two models' refits side by side on two contexts of their own: the same numbers as one after the other
    both = protassov_test(lc, drw(), alt, nsims=24, walkers=16, max_steps=120, sim_steps=60, sigma_noise=2.0, seed=5,
                          concurrent_refits=True)
    assert np.array_equal(both["T_sim"], res["T_sim"]) and both["T_obs"] == res["T_obs"]
    assert np.array_equal(both["sim_alt"].max_parameters, res["sim_alt"].max_parameters)


def test_an_engine_refuses_a_second_thread_while_a_call_is_in_flight(engine):
    """One mtg_ctx serves one thread at a time (a re-upload under another thread's running kernels is a GPU fault):
    the Python engine says so instead."""
    import threading
    from mind_the_gaps_amd.engine import EngineError
    t, y, dy = synth.make_lightcurves(50, 1, seed=3)
    engine.set_lightcurves(t, y, dy + 1e-12)
    held, release = threading.Event(), threading.Event()

    def holder():                      # stands for a thread that is inside a long library call
        with engine._busy:
            held.set()
            release.wait(30)

    th = threading.Thread(target=holder)
    th.start()
    assert held.wait(30)
    try:
        with pytest.raises(EngineError, match="another thread"):
            engine.set_lightcurves(t, y, dy + 1e-12)
    finally:
        release.set()
        th.join()
    engine.set_lightcurves(t, y, dy + 1e-12)      # free again


def test_invariances_at_the_bench_size(engine):
    """The bench workload itself (BASELINE configs[3] per GPU: 2000 light curves x 256 walkers,
    N = 1e4, J = 6 -- 512 000 evaluations per launch), checked through properties that need no
    oracle: splitting the batch changes nothing (bit for bit); stretching time by 2 while halving
    every frequency leaves the covariance -- and lnL -- unchanged; so does moving the time origin."""
    kinds = synth.ALT_MODEL
    N, L, W = 10000, 2000, 256
    t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    theta = synth.draw_thetas(kinds, L * W, seed=20250704 + 40)
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    off = y.mean(axis=1)
    engine.set_time_parallel(2)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=off)
    engine.set_model(kinds, full, free, bounds)
    out, st = engine.loglike(theta, lc, add_prior=True)
    assert np.all(st == 0) and np.all(np.isfinite(out))
    # (1) eight partial launches
    parts = [engine.loglike(theta[i::8], lc[i::8], add_prior=True)[0] for i in range(8)]
    for i in range(8):
        assert np.array_equal(parts[i], out[i::8])
    # (2) t -> 2 t with every rate halved: DRW (a, c) = (S0, w0); SHO a = S0 w0 Q, so S0 doubles;
    #     Lorentzian (a, c, d) = (S0, w0 / 2Q, w0)
    th2 = theta.copy()
    th2[:, [1, 4, 7]] -= np.log(2.0)
    th2[:, 2] += np.log(2.0)
    engine.set_lightcurves(2.0 * t, y, dy + 1e-12, y_offset=off)
    out2, st2 = engine.loglike(th2, lc, add_prior=True)
    assert np.all(st2 == 0)
    assert np.max(np.abs(out2 - out) / np.abs(out)) < 1e-10
    # (3) a new time origin (the kernels only ever see time differences)
    engine.set_lightcurves(t + 4096.0, y, dy + 1e-12, y_offset=off)
    out3, st3 = engine.loglike(theta, lc, add_prior=True)
    assert np.all(st3 == 0)
    assert np.max(np.abs(out3 - out) / np.abs(out)) < 1e-9


def test_refits_in_blocks_are_the_refits_in_one_call():
    """derive_posteriors_batch(index_base=...): seven light curves fitted in one call and in blocks of 4 + 3 (and
    2 + 5) -- other batch sizes in every launch of the fit and of the chains -- give every light curve the same
    starting fit, chain maximum and best sample to the last bit (walkers from a generator per light curve, Philox
    counters by global ensemble index, kernels whose results do not depend on the batch)."""
    N, L, W = 400, 7, 16
    t, y, dy = synth.make_lightcurves(N, L, seed=77)
    y += 3.0 * np.arange(L)[:, None]

    def run(lo, hi, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return derive_posteriors_batch(t, y[lo:hi], dy[lo:hi], alt_kernel(), walkers=W, max_steps=60, fit=True, seed=5,
                                           store_chain=False, quiet=True, **kw)
    whole = run(0, L, index_base=0)
    for cut in (4, 2):
        a, b = run(0, cut, index_base=0), run(cut, L, index_base=cut)
        for name in ("max_loglikelihood", "max_parameters", "fit_parameters", "fit_loglikelihood"):
            assert np.array_equal(np.concatenate([getattr(a, name), getattr(b, name)]), getattr(whole, name)), (cut, name)
    assert np.all(np.isfinite(whole.max_loglikelihood)) and np.all(whole.max_loglikelihood >= whole.fit_loglikelihood - 1.0)
    # without index_base the blocks draw from streams of their own: other numbers (the same statistics)
    plain = run(4, L)
    assert not np.array_equal(plain.max_loglikelihood, whole.max_loglikelihood[4:])
    with pytest.raises(ValueError):
        derive_posteriors_batch(t, y[:2], dy[:2], alt_kernel(), walkers=W, max_steps=5, index_base=0)      # needs a seed


def test_simulated_series_in_blocks_are_the_series_of_one_call():
    """Simulator.simulate(index_base=...): the noise and cut streams are keyed by the global series index."""
    from mind_the_gaps_amd.simulator import Simulator
    rng = np.random.default_rng(3)
    times = synth.make_times(300, rng)
    sim = Simulator(null_kernel(), times, 0.04, 100.0, "Gaussian", sigma_noise=1.0, extension_factor=2, random_state=1)
    thetas = synth.draw_thetas(synth.NULL_MODEL, 9, seed=8, percent=0.05)
    whole = sim.simulate(thetas, seed=12345, index_base=0)
    for lo, hi in ((0, 4), (4, 9), (7, 8), (3, 6)):
        part = sim.simulate(thetas[lo:hi], seed=12345, index_base=lo)
        for key in ("rates", "dy", "means"):
            assert np.array_equal(part[key], whole[key][lo:hi]), (lo, hi, key)
    assert not np.array_equal(sim.simulate(thetas[4:9], seed=12345)["rates"], whole["rates"][4:9])
    # without an index_base two series may share a transform (the chirp-z path packs them as real and imaginary part): the
    # same series to rounding, not to the last bit -- which is why blocks of a larger set are asked for with one
    plain = sim.simulate(thetas, seed=12345)
    assert np.max(np.abs(plain["rates"] - whole["rates"])) <= 1e-10 * np.std(whole["rates"])
    # ... or cut at EVEN global indices with the pairs kept: every series then has the partner it has in the whole set
    # (what protassov_test does, simulating up to two extra series per block), bit for bit again and at the paired speed
    paired = sim.simulate(thetas, seed=12345, index_base=0, pair_series=True)
    for lo, hi in ((0, 4), (4, 9), (2, 6), (8, 9)):
        part = sim.simulate(thetas[lo:hi], seed=12345, index_base=lo, pair_series=True)
        for key in ("rates", "dy", "means"):
            assert np.array_equal(part[key], paired[key][lo:hi]), (lo, hi, key)
    assert np.array_equal(paired["rates"], plain["rates"])          # (the same pairs as an ordinary call from index 0)
    with pytest.raises(ValueError):
        sim.simulate(thetas[3:6], seed=12345, index_base=3, pair_series=True)


@pytest.mark.parametrize("pdf,adjust_on", [("Gaussian", "device"), ("Lognormal", "host"), ("Lognormal", "device")])
def test_host_drawn_noise_in_blocks_is_the_noise_of_one_call(pdf, adjust_on):
    """A light curve with a background takes the Kraft noise model, drawn on the HOST; a non-Gaussian flux PDF the iterative
    adjustment -- on the device (its white series keyed by seed and global index, csrc/mtg_e13.hip) or, adjust_on="host",
    in numpy.  Given an index_base every series draws from a generator of its own (seed, global index), so blocks -- ranks
    of a sharded Protassov test -- reproduce the whole set; without it the host's draws share one stream and do not."""
    from mind_the_gaps_amd.simulator import Simulator
    rng = np.random.default_rng(4)
    times = synth.make_times(120, rng)
    sim = Simulator(null_kernel(), times, 0.04, 400.0, pdf, bkg_rate=30.0, bkg_rate_err=2.0, extension_factor=2,
                    random_state=1, max_iter=30, adjust_on=adjust_on)
    assert sim.noise_name == "Kraft"
    thetas = synth.draw_thetas(synth.NULL_MODEL, 6, seed=8, percent=0.05)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # (the PDF adjustment may stop at max_iter: not what is tested)
        whole = sim.simulate(thetas, seed=777, index_base=0)
        for lo, hi in ((0, 2), (2, 6), (5, 6)):
            part = sim.simulate(thetas[lo:hi], seed=777, index_base=lo)
            for key in ("rates", "dy", "means"):
                assert np.array_equal(part[key], whole[key][lo:hi]), (lo, hi, key)
        sim.random_state = np.random.RandomState(5)
        shared_a = sim.simulate(thetas[2:6], seed=777)
    assert np.all(np.isfinite(whole["rates"])) and np.all(whole["dy"] > 0)
    assert not np.array_equal(shared_a["rates"], whole["rates"][2:6])


def test_protassov_test_with_a_lognormal_flux_pdf_never_leaves_the_device(monkeypatch):
    """protassov_test(pdf="Lognormal"): every simulated light curve goes through the E13 adjustment ON THE DEVICE between the
    cut and the down-sampling (simulator.py:65-140; csrc/mtg_e13.hip), stays resident for the refits, and the numpy loop
    is never entered; same seed, same test."""
    from mind_the_gaps_amd.ppp import protassov_test
    from mind_the_gaps_amd.simulator import Simulator

    def never(self, *a, **k):
        raise AssertionError("the host adjustment ran")
    monkeypatch.setattr(Simulator, "_adjust_pdf", never)
    monkeypatch.setattr(Simulator, "_finish_on_host", never)
    times = np.arange(0.5, 200.0, 1.0)
    th = synth.truth([synth.K_DRW])
    drw = lambda: DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER])
    lor = synth.truth([synth.K_LORENTZIAN])
    obs = Simulator(drw(), times, 0.2, 100.0, "Lognormal", sigma_noise=2.0, extension_factor=3, random_state=9).simulate()
    assert np.all(obs["rates"] > 0)
    lc = GappyLightcurve(times, obs["rates"][0], obs["dy"][0], exposures=0.2)
    alt = drw() + Lorentzian(lor[0], lor[1], lor[2], bounds=[AMP, OTHER, OTHER])
    runs = [protassov_test(lc, drw(), alt, nsims=12, walkers=16, max_steps=100, sim_steps=40, sigma_noise=2.0, seed=5,
                           pdf="Lognormal") for _ in range(2)]
    res = runs[0]
    assert res["T_sim"].shape == (12,) and np.all(np.isfinite(res["T_sim"])) and np.isfinite(res["T_obs"])
    assert np.all(res["lightcurves"]["rates"] > 50.0) and 1 / 13 <= res["p_value"] <= 1.0
    assert np.array_equal(runs[1]["T_sim"], res["T_sim"]) and runs[1]["p_value"] == res["p_value"]
    # ... and it is another test than the Gaussian one of the same seed
    plain = protassov_test(lc, drw(), alt, nsims=12, walkers=16, max_steps=100, sim_steps=40, sigma_noise=2.0, seed=5)
    assert not np.array_equal(plain["T_sim"], res["T_sim"])
