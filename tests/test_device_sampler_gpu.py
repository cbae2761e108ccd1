"""GPU tests of the device-resident ensemble sampler (mtg_ensemble_*): exact replay of
its Philox-driven moves on the host with the oracle likelihood, and invariants."""
import numpy as np
import pytest

import philox_replay
from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu


def setup_problem(engine, kinds, N, L, W, seed):
    t, y, dy = synth.make_lightcurves(N, L, seed=seed)
    y += 7.0 * np.arange(L)[:, None]
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    P = len(free)
    rng = np.random.default_rng(seed)
    p0 = synth.truth(kinds) * (1 + 0.02 * rng.standard_normal((L, W, P)))

    def oracle_lnp(q, ens):
        fullv = np.hstack([q, y.mean(axis=1)[ens][:, None]])
        return oracle_c.logprob_batch(t, y, dy, kinds, fullv, bounds=bounds, lc_index=ens.astype(np.int32),
                                      add_prior=True, nthreads=8)[0]
    return p0, oracle_lnp


def test_device_sampler_replays_exactly_on_the_host(engine):
    kinds = synth.NULL_MODEL
    E, W, steps, seed = 5, 12, 40, 0x1234ABCD5678
    p0, oracle_lnp = setup_problem(engine, kinds, 200, E, W, seed=3)
    engine.ensemble_init(p0, seed=seed)
    st0 = engine.ensemble_state()
    lnp0 = oracle_lnp(p0.reshape(E * W, -1), np.repeat(np.arange(E), W)).reshape(E, W)
    assert np.max(np.abs(st0["log_prob"] - lnp0) / np.abs(lnp0)) < 1e-9 and st0["iteration"] == 0
    chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
    ref_chain, ref_lnp, ref_acc = philox_replay.run(p0, lnp0, oracle_lnp, steps, seed)
    assert chain.shape == (steps, E, W, 5)
    assert np.allclose(chain, ref_chain, rtol=0, atol=1e-11)          # same moves, same decisions
    assert np.allclose(lnp_chain, ref_lnp, rtol=1e-9)
    st = engine.ensemble_state()
    assert st["iteration"] == steps and st["n_not_pd"] == 0
    assert np.array_equal(st["naccept"], ref_acc)
    assert np.array_equal(st["coords"], chain[-1]) and np.array_equal(st["log_prob"], lnp_chain[-1])
    # the running best is the maximum over everything the chains visited
    visited = np.concatenate([lnp0[None], lnp_chain]).max(axis=(0, 2))
    assert np.allclose(st["best_log_prob"], visited, rtol=1e-12)
    # continuing is the same as one longer run (counter-based random numbers)
    more, _ = engine.ensemble_run(10, store_chain=True)
    ref2, _, _ = philox_replay.run(ref_chain[-1], ref_lnp[-1], oracle_lnp, 10, seed, start_iteration=steps)
    assert np.allclose(more, ref2, rtol=0, atol=1e-10)


def test_device_sampler_statistics_and_prior_box(engine):
    """Walkers stay inside the prior box, acceptance is healthy, one light curve can host
    many ensembles (lc_of_ensemble), and different seeds give different chains."""
    kinds = [synth.K_DRW]
    E, W = 8, 16
    p0, _ = setup_problem(engine, kinds, 150, 1, W, seed=9)
    p0 = np.repeat(p0, E, axis=0) * (1 + 0.01 * np.random.default_rng(1).standard_normal((E, W, 2)))
    engine.ensemble_init(p0, seed=1, lc_of_ensemble=np.zeros(E, dtype=np.int32))
    chain, lnp = engine.ensemble_run(300, store_chain=True)
    st = engine.ensemble_state()
    b = synth.bounds_for(kinds)
    assert np.all(chain >= b[:, 0]) and np.all(chain <= b[:, 1]) and np.all(np.isfinite(lnp))
    frac = st["naccept"] / 300.0
    assert 0.3 < frac.mean() < 0.95
    # all ensembles sample the same posterior: their means agree within the scatter
    means = chain[100:].reshape(200, E, W * 2).mean(axis=(0,)).reshape(E, W, 2).mean(axis=1)
    assert np.all(np.abs(means - means.mean(axis=0)) < 4 * chain[100:].std(axis=(0, 2)).mean(axis=0))
    engine.ensemble_init(p0, seed=2, lc_of_ensemble=np.zeros(E, dtype=np.int32))
    chain2, _ = engine.ensemble_run(5, store_chain=True)
    assert not np.array_equal(chain2, chain[:5])
    with pytest.raises(Exception):
        engine.ensemble_init(p0[:, :3], seed=1)          # fewer walkers than 2 * ndim (and odd)


def test_chain_autocorr_on_the_device_matches_the_host(engine):
    """mtg_chain_autocorr against sampler._mean_autocorr_function (emcee's function_1d per walker and dimension,
    averaged): correlated chains of several shapes, lengths on both sides of a power of two, and through
    integrated_time(acf=...)."""
    from mind_the_gaps_amd.sampler import _mean_autocorr_function, integrated_time
    rng = np.random.default_rng(3)
    for n_t, W, P in ((2, 2, 1), (37, 6, 3), (512, 32, 5), (1000, 128, 5), (1025, 12, 15)):
        x = rng.standard_normal((n_t, W, P))
        for t in range(1, n_t):                       # AR(1) with a different memory per dimension
            x[t] += (0.5 + 0.45 * np.arange(P) / max(P - 1, 1)) * x[t - 1]
        ref = _mean_autocorr_function(x)
        got = engine.chain_autocorr(x)
        assert got.shape == ref.shape == (n_t, P)
        assert np.max(np.abs(got - ref)) < 1e-11, (n_t, W, P, np.max(np.abs(got - ref)))
        if n_t >= 500:
            assert np.allclose(integrated_time(x, tol=0, acf=engine.chain_autocorr), integrated_time(x, tol=0),
                               rtol=1e-9, atol=0)
    many = rng.standard_normal((300, 7, 10, 3)).cumsum(axis=0)           # seven independent ensembles in one call
    got = engine.chain_autocorr(many)
    assert got.shape == (300, 7, 3)
    for e in range(7):
        assert np.max(np.abs(got[:, e] - _mean_autocorr_function(many[:, e]))) < 1e-11
    from mind_the_gaps_amd.device_sampler import _autocorr_time_where_it_is_cheapest
    engine.fft_ready = True
    engine.__dict__.pop("_acf_state", None)
    tau_host = _autocorr_time_where_it_is_cheapest(engine, many, dict(tol=0, quiet=True))        # rented: the host
    engine._acf_state["rented"][(512, 7, 10, 3)] = 10.0                                          # ... long enough
    tau = _autocorr_time_where_it_is_cheapest(engine, many, dict(tol=0, quiet=True))             # bought: the device
    assert engine._acf_state["planned"] == [(512, 7, 10, 3)] and np.allclose(tau, tau_host, rtol=1e-9)
    # the tutorial's loop checks two models' chains in turn: both shapes keep their plans (four slots, least recently used out)
    shapes = [(500, 30, 2), (500, 30, 5)]
    chains = [rng.standard_normal(sh).cumsum(axis=0) for sh in shapes]
    for c in chains:
        engine.chain_autocorr(c)                       # plans made
    built = engine.acf_plans_built
    for _ in range(10):
        for c in chains:
            got = engine.chain_autocorr(c)
    assert engine.acf_plans_built == built             # no plan rebuilt (9 ms a call while one slot was remade every time)
    assert np.max(np.abs(got - _mean_autocorr_function(chains[1]))) < 1e-11
    for i, c in enumerate(chains + [rng.standard_normal((500, 30, k)).cumsum(axis=0) for k in (3, 4, 6)]):   # a fifth shape evicts the oldest
        engine._acf_state["rented"][(512,) + c.shape[1:]] = 10.0
        _autocorr_time_where_it_is_cheapest(engine, c, dict(tol=0, quiet=True))
    assert len(engine._acf_state["planned"]) == 4 and engine._acf_state["planned"][-1] == (512, 30, 6) and (512, 30, 2) not in engine._acf_state["planned"]
    assert tau.shape == (7, 3) and np.allclose(tau, [integrated_time(many[:, e], tol=0) for e in range(7)], rtol=1e-9)
    still = rng.standard_normal((64, 4, 2))
    still[:, 1, 0] = 1.25                              # a walker that never moved: NaN in its dimension, as emcee
    got = engine.chain_autocorr(still)
    assert np.all(np.isnan(got[:, 0])) and np.all(np.isfinite(got[:, 1]))


def test_checkpoint_and_resume_continue_the_chain_bit_for_bit(engine, tmp_path):
    """DeviceEnsembleSampler.save / load: 12 iterations in one go against 5, a checkpoint, a NEW sampler that loads it,
    and 7 more -- chain, log-probabilities, acceptance counts and running best identical."""
    from mind_the_gaps_amd.device_sampler import DeviceEnsembleSampler
    kinds = synth.NULL_MODEL
    t, y, dy = synth.make_lightcurves(400, 1, seed=9)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)

    def bind():
        engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        engine.set_model(kinds, full, free, bounds)
        return engine
    P, W = len(free), 16
    p0 = synth.draw_thetas(kinds, W, seed=2, percent=0.02)[None]
    whole = DeviceEnsembleSampler(bind, W, P, seed=1234)
    whole.run_mcmc(p0, 12)
    first = DeviceEnsembleSampler(bind, W, P, seed=1234)
    first.run_mcmc(p0, 5)
    first.save(tmp_path / "ckpt")
    second = DeviceEnsembleSampler(bind, W, P, seed=999)        # (its own seed is replaced by the checkpoint's)
    second.load(tmp_path / "ckpt")
    state = second.run_mcmc(None, 7)
    assert second.iteration == 12 and np.array_equal(second.get_chain(), whole.get_chain())
    assert np.array_equal(second.get_log_prob(), whole.get_log_prob())
    for key in ("coords", "log_prob", "naccept", "best_log_prob", "best_coords"):
        assert np.array_equal(state[key], whole.state[key]), key


@pytest.mark.parametrize("case", ["four_waves", "rank10"])
def test_resume_is_bit_for_bit_where_the_batch_size_picks_the_kernel(engine, tmp_path, case):
    """The resume must not re-evaluate the saved state: `mtg_ensemble_init` evaluates E W rows in ONE batch, the run
    evaluated half-ensembles, and the row count picks kernel and summation order -- four waves per evaluation up to
    256 rows at N >= 4096 (one wave beyond), more chunks per evaluation for the rank-10 model's smaller batches.
    (i) E W = 512 walkers on N = 4100 samples: 256-row half-steps on the four-wave kernel, a 512-row restart on
    the one-wave kernel; (ii) five SHO terms, N = 2e5, 32 walkers: 16-row half-steps cut the series into 4096 chunks, the
    32-row restart into 2048.  `mtg_ensemble_restore` puts the saved log-probabilities back: identical continuation."""
    from mind_the_gaps_amd.device_sampler import DeviceEnsembleSampler
    if case == "four_waves":
        kinds, n, W, steps = synth.NULL_MODEL, 4100, 512, (4, 2, 2)
        p0 = synth.draw_thetas(kinds, W, seed=2, percent=0.02)[None]
    else:
        kinds, n, W, steps = [synth.K_SHO] * 5, 200000, 32, (4, 2, 2)
        base = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
                               for i in range(5)])
        p0 = (base * (1 + 0.01 * np.random.default_rng(5).standard_normal((W, 15))))[None]
    t, y, dy = synth.make_lightcurves(n, 1, seed=9)
    bnd = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    full = np.concatenate([p0[0, 0], [0.0]])
    free = np.arange(len(full) - 1, dtype=np.int32)

    def bind():
        engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        engine.set_model(kinds, full, free, bnd)
        return engine
    P = len(free)
    whole = DeviceEnsembleSampler(bind, W, P, seed=77)
    whole.run_mcmc(p0, steps[0])
    first = DeviceEnsembleSampler(bind, W, P, seed=77)
    first.run_mcmc(p0, steps[1])
    first.save(tmp_path / "ckpt")
    second = DeviceEnsembleSampler(bind, W, P, seed=5)
    restored = second.load(tmp_path / "ckpt")
    assert np.array_equal(restored["log_prob"], first.state["log_prob"])      # as saved, not as re-evaluated
    state = second.run_mcmc(None, steps[2])
    assert np.array_equal(second.get_chain(), whole.get_chain()) and np.array_equal(second.get_log_prob(), whole.get_log_prob())
    for key in ("coords", "log_prob", "naccept", "best_log_prob", "best_coords"):
        assert np.array_equal(state[key], whole.state[key]), key


@pytest.mark.parametrize("case", ["drw_n1000_w32", "null_n5000_w32", "alt_three_ensembles"])
def test_speculative_iterations_give_the_sequential_chain(engine, case):
    """Both half-steps of an iteration in one batch of 3 E W/2 rows (mtg_set_speculation, the default for small
    ensembles) against one solve per half-step: the same random numbers, the same rows through the same kernels,
    hence the same chain, acceptance counts and running best to the last bit."""
    kinds, N, E, W = {"drw_n1000_w32": ([synth.K_DRW], 1000, 1, 32), "null_n5000_w32": (synth.NULL_MODEL, 5000, 1, 32),
                      "alt_three_ensembles": (synth.ALT_MODEL, 600, 3, 16)}[case]
    steps, seed = 60, 0xFEEDBEEF
    p0, oracle_lnp = setup_problem(engine, kinds, N, E, W, seed=11)
    if case == "alt_three_ensembles":
        p0[:, ::3, 3] = np.log(0.3)                                  # some walkers start over-damped: two structures
    results = {}
    try:
        for mode in (0, 1, 2):      # 2: speculative with each split ranked in its sampler launch (long runs' fallback)
            engine.set_speculation(mode)
            engine.ensemble_init(p0, seed=seed)
            chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
            more, more_lnp = engine.ensemble_run(7, store_chain=True)       # continuing works the same
            results[mode] = (chain, lnp_chain, more, more_lnp, engine.ensemble_state())
    finally:
        engine.set_speculation(1)
    a, b = results[0], results[1]
    for other in (results[1], results[2]):
        for x, y in zip(a[:4], other[:4]):
            assert np.array_equal(x, y)
        for key in ("coords", "log_prob", "naccept", "best_log_prob", "best_coords"):
            assert np.array_equal(a[4][key], other[4][key]), key
    assert a[4]["iteration"] == b[4]["iteration"] == steps + 7 and a[4]["n_not_pd"] == b[4]["n_not_pd"]
    assert 0.1 < a[4]["naccept"].mean() / (steps + 7) < 0.9
    if case == "drw_n1000_w32":                                   # and it is the chain the host replay makes
        lnp0 = oracle_lnp(p0.reshape(E * W, -1), np.repeat(np.arange(E), W)).reshape(E, W)
        ref_chain, ref_lnp, ref_acc = philox_replay.run(p0, lnp0, oracle_lnp, steps, seed)
        assert np.allclose(b[0], ref_chain, rtol=0, atol=1e-10)


# ---- the speculative iteration at the shapes it ships on (BASELINE configs[1] and [2]) ---------------------------
# 3 E W/2 rows per iteration: 192 rows for the 128 walkers of configs[1] (four waves per evaluation, 1024-thread
# mtg_sampler_spec_kernel), 384 rows for the 256 walkers of configs[2] (two waves per evaluation at rank 5).  The
# sequential form evaluates W/2 = 64 / 128 rows per half-step, which the dispatch sends to the four-wave kernels: the
# two forms then run DIFFERENT kernels for the same row, whose sums differ in the last bits -- what holds between them is
# "the same decisions, log-probabilities to rounding", not "the same bits" (that holds where both forms dispatch the
# same kernel: test_speculative_iterations_give_the_sequential_chain above).
SHIPPED = {"configs1_null_w128": (synth.NULL_MODEL, 128, 192), "configs2_alt_w256": (synth.ALT_MODEL, 256, 384)}


def _shipped_problem(engine, case, overdamped=False):
    kinds, W, rows = SHIPPED[case]
    p0, oracle_lnp = setup_problem(engine, kinds, 10000, 1, W, seed=21)
    if overdamped:
        p0[:, ::5, 3] = np.log(0.3)               # every fifth walker starts with its SHO term over-damped (Q < 1/2)
    lnp0 = oracle_lnp(p0.reshape(W, -1), np.zeros(W, dtype=np.int64)).reshape(1, W)
    return kinds, W, rows, p0, lnp0, oracle_lnp


@pytest.mark.parametrize("case", sorted(SHIPPED))
def test_speculative_sampler_at_the_shipped_shapes_replays_on_the_host(engine, case):
    """(a) of the round-3 review: 40 default (speculative) iterations of the device sampler at N = 1e4, replayed on the
    host with the ORACLE likelihood: every accept decision equal, chain within 1e-10, log-probabilities within 1e-9."""
    kinds, W, rows, p0, lnp0, oracle_lnp = _shipped_problem(engine, case)
    steps, seed = 40, 0xC0FFEE123
    engine.set_speculation(1)
    engine.ensemble_init(p0, seed=seed)
    st0 = engine.ensemble_state()
    assert np.max(np.abs(st0["log_prob"] - lnp0) / np.abs(lnp0)) < 1e-9
    chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
    solver = engine.last_solver
    assert "mtg_tp_" in solver, solver                                # a time-parallel kernel took the 3 W/2 rows
    ref_chain, ref_lnp, ref_acc = philox_replay.run(p0, lnp0, oracle_lnp, steps, seed)
    st = engine.ensemble_state()
    assert np.array_equal(st["naccept"], ref_acc)                     # the same decisions, walker by walker
    moved = np.any(chain[1:] != chain[:-1], axis=-1)
    assert np.array_equal(moved, np.any(ref_chain[1:] != ref_chain[:-1], axis=-1))
    assert np.allclose(chain, ref_chain, rtol=0, atol=1e-10)
    assert np.allclose(lnp_chain, ref_lnp, rtol=1e-9, atol=0)
    assert st["iteration"] == steps and st["n_not_pd"] == 0
    assert 0.02 < st["naccept"].mean() / steps < 0.95


@pytest.mark.parametrize("case", sorted(SHIPPED))
def test_speculative_and_sequential_forms_at_the_shipped_shapes(engine, case):
    """(b): the same run with one solve per half-step.  The two forms send a row through different kernels at these
    sizes (rows per batch pick the kernel), so: identical accept decisions and coordinates that differ only through
    nothing at all -- a proposal is a function of coordinates and random numbers, not of log-probabilities -- while the
    stored log-probabilities agree to rounding (1e-12 relative; ~1e-14 observed)."""
    kinds, W, rows, p0, lnp0, oracle_lnp = _shipped_problem(engine, case)
    steps, seed = 40, 0xC0FFEE123
    out = {}
    try:
        for mode in (0, 1):
            engine.set_speculation(mode)
            engine.ensemble_init(p0, seed=seed)
            chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
            out[mode] = (chain, lnp_chain, engine.ensemble_state(), engine.last_solver)
    finally:
        engine.set_speculation(1)
    (c0, l0, s0, k0), (c1, l1, s1, k1) = out[0], out[1]
    assert np.array_equal(s0["naccept"], s1["naccept"])
    assert np.array_equal(c0, c1)                                     # same decisions => the very same coordinates
    assert np.max(np.abs(l0 - l1) / np.abs(l0)) < 1e-12, (k0, k1)
    assert np.allclose(s0["best_log_prob"], s1["best_log_prob"], rtol=1e-12) and s0["iteration"] == s1["iteration"] == steps


@pytest.mark.parametrize("case", sorted(SHIPPED))
def test_splits_ranked_up_front_or_inside_the_sampler_launch(engine, case):
    """The default run ranks every iteration's split in one launch before the first iteration; a run too long for that
    (steps * E * W * 4 bytes > 64 MiB) ranks each split inside its sampler launch (mtg_set_speculation mode 2 forces
    it).  Same counters, same kernels for the likelihood: the same chain to the last bit, also across a continuation."""
    kinds, W, rows, p0, lnp0, oracle_lnp = _shipped_problem(engine, case, overdamped=case == "configs2_alt_w256")
    steps, seed = 25, 0x5EED5EED
    out = {}
    try:
        for mode in (1, 2):
            engine.set_speculation(mode)
            engine.ensemble_init(p0, seed=seed)
            chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
            more, more_lnp = engine.ensemble_run(5, store_chain=True)
            out[mode] = (chain, lnp_chain, more, more_lnp, engine.ensemble_state())
    finally:
        engine.set_speculation(1)
    for x, y in zip(out[1][:4], out[2][:4]):
        assert np.array_equal(x, y)
    for key in ("coords", "log_prob", "naccept", "best_log_prob", "best_coords"):
        assert np.array_equal(out[1][4][key], out[2][4][key]), key
    assert out[1][4]["iteration"] == out[2][4]["iteration"] == steps + 5


def test_speculative_sampler_with_two_structures_in_a_384_row_batch(engine):
    """(c): a fifth of the 256 walkers start over-damped (SHO as two real terms): the 384 rows of an iteration hold both
    structures and go through the fused kernel (every structure in one launch).  Host replay with the oracle."""
    kinds, W, rows, p0, lnp0, oracle_lnp = _shipped_problem(engine, "configs2_alt_w256", overdamped=True)
    steps, seed = 30, 0xABCDEF01
    engine.set_speculation(1)
    engine.ensemble_init(p0, seed=seed)
    chain, lnp_chain = engine.ensemble_run(steps, store_chain=True)
    assert "mtg_tp_fused_kernel" in engine.last_solver, engine.last_solver
    over = chain[..., 3] < np.log(0.5)
    assert over.any() and (~over).any()                               # both structures were sampled throughout
    assert over[-1].any()
    ref_chain, ref_lnp, ref_acc = philox_replay.run(p0, lnp0, oracle_lnp, steps, seed)
    assert np.array_equal(engine.ensemble_state()["naccept"], ref_acc)
    assert np.allclose(chain, ref_chain, rtol=0, atol=1e-10)
    assert np.allclose(lnp_chain, ref_lnp, rtol=1e-9, atol=0)


def test_resume_across_a_speculative_run_at_256_walkers(engine, tmp_path):
    """(d): save / load in the middle of a speculative run of configs[2]'s shape: the continuation is the one-go chain
    bit for bit (the restart evaluates nothing: the saved log-probabilities come back through mtg_ensemble_restore)."""
    from mind_the_gaps_amd.device_sampler import DeviceEnsembleSampler
    kinds, W = synth.ALT_MODEL, 256
    t, y, dy = synth.make_lightcurves(10000, 1, seed=21)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)

    def bind():
        engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        engine.set_model(kinds, full, free, bounds)
        engine.set_speculation(1)
        return engine
    P = len(free)
    p0 = synth.draw_thetas(kinds, W, seed=2, percent=0.02)[None]
    whole = DeviceEnsembleSampler(bind, W, P, seed=4321)
    whole.run_mcmc(p0, 9)
    first = DeviceEnsembleSampler(bind, W, P, seed=4321)
    first.run_mcmc(p0, 4)
    first.save(tmp_path / "ckpt")
    second = DeviceEnsembleSampler(bind, W, P, seed=1)
    restored = second.load(tmp_path / "ckpt")
    assert np.array_equal(restored["log_prob"], first.state["log_prob"])
    state = second.run_mcmc(None, 5)
    assert second.iteration == 9
    assert np.array_equal(second.get_chain(), whole.get_chain()) and np.array_equal(second.get_log_prob(), whole.get_log_prob())
    for key in ("coords", "log_prob", "naccept", "best_log_prob", "best_coords"):
        assert np.array_equal(state[key], whole.state[key]), key

