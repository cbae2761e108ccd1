"""The throughput kernel's own order of evaluation (csrc/mtg_sort.hip): large batches are swept in the order of a
stable sort by (structure, light curve), whatever order the caller's rows have.  A row's result must not depend on
where it sits in the batch -- bit for bit --, rejected rows must stay rejected, and the library's calls must chain
correctly when they alternate between a caller's stream and the context's own."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu


@pytest.fixture
def sweep(engine):
    engine.set_time_parallel(0)          # the serial sweep is the kernel the order matters to
    yield engine
    engine.set_sort(2)
    engine.set_time_parallel(2)


@pytest.mark.parametrize("kinds", [synth.ALT_MODEL, [synth.K_SHO, synth.K_SHO]], ids=["alt", "2sho"])
def test_sorted_sweep_matches_the_callers_order_bit_for_bit(sweep, kinds):
    eng = sweep
    N, L, B = 300, 57, 5003                    # ragged: B is no multiple of the wave, 57 light curves
    t, y, dy = synth.make_lightcurves(N, L, seed=15)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    rng = np.random.default_rng(4)
    theta = synth.draw_thetas(kinds, B, seed=19, percent=0.6 if synth.K_SHO in kinds else 0.1)   # mixed SHO structures
    theta[rng.random(B) < 0.1, 0] = 60.0       # every tenth row outside the prior box
    lc = rng.integers(0, L, B).astype(np.int32)
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y.mean(axis=1)[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=4)
    results = {}
    for mode in (0, 1, 2):                     # caller's order, always sorted, automatic (random rows: sorted)
        eng.set_sort(mode)
        results[mode] = eng.loglike(theta, lc, add_prior=True)
        # one structure per launch, all of them in one, or (a batch of this size) the pipelined form of the same sweep
        assert eng.last_solver.startswith(("mtg_solve_kernel", "mtg_pipe_kernel"))
    out0, st0 = results[0]
    ok = st0 == 0
    assert np.array_equal(st0, rst) and 0.3 * B < ok.sum() < 0.95 * B
    assert np.max(np.abs(out0[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8
    assert np.all(np.isneginf(out0[st0 == 1]))
    for mode in (1, 2):
        out, st = results[mode]
        assert np.array_equal(st, st0), mode
        assert np.array_equal(out, out0), mode          # the same arithmetic per row, wherever the row sits
    # rows already grouped by light curve: the automatic mode keeps the caller's order (nothing to compare but the values)
    order = np.argsort(lc, kind="stable")
    eng.set_sort(2)
    out_g, st_g = eng.loglike(theta[order], lc[order], add_prior=True)
    assert np.array_equal(st_g, st0[order]) and np.array_equal(out_g, out0[order])


def test_calls_on_different_streams_are_chained(sweep):
    """mtg_loglike_batch_device on a caller's (non-default) stream, then at once the host-pointer entry point on the
    context's stream with other parameters, then the caller's stream again: the calls share the context's
    workspaces (coefficients, lists, sort buffers) and must run one after the other whatever their streams."""
    import torch
    eng = sweep
    kinds = synth.ALT_MODEL
    N, L, W = 4000, 24, 400
    B = L * W
    t, y, dy = synth.make_lightcurves(N, L, seed=23)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    rng = np.random.default_rng(8)
    theta_a, theta_b = synth.draw_thetas(kinds, B, seed=1), synth.draw_thetas(kinds, B, seed=2)
    lc = rng.integers(0, L, B).astype(np.int32)
    want_a, st_a = eng.loglike(theta_a, lc)
    want_b, st_b = eng.loglike(theta_b, lc)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(dev)
    d_lc = torch.from_numpy(lc).to(dev)
    d_a = torch.from_numpy(theta_a).to(dev)
    d_out = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(2)]
    d_st = [torch.full((B,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    torch.cuda.synchronize(dev)
    eng.loglike_device(B, d_a.data_ptr(), d_lc.data_ptr(), d_out[0].data_ptr(), d_st[0].data_ptr(), stream=side.cuda_stream)
    got_b, got_st_b = eng.loglike(theta_b, lc)                      # context's stream, host pointers: synchronous
    eng.loglike_device(B, d_a.data_ptr(), d_lc.data_ptr(), d_out[1].data_ptr(), d_st[1].data_ptr(), stream=side.cuda_stream)
    eng.synchronize()                                               # waits for the caller's stream as well
    assert np.array_equal(got_b, want_b) and np.array_equal(got_st_b, st_b)
    for k in range(2):
        assert np.array_equal(d_st[k].cpu().numpy(), st_a) and np.array_equal(d_out[k].cpu().numpy(), want_a)


def test_shard_info_of_an_unsharded_ensemble(engine):
    kinds = [synth.K_DRW]
    t, y, dy = synth.make_lightcurves(100, 1, seed=8)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    engine.ensemble_init(synth.draw_thetas(kinds, 8, seed=1)[None], seed=3)
    assert engine.ensemble_shard_info() == dict(kind="none", rank=0, world=1, comm_ranks=0)
    engine.shard_profile_begin(4)
    engine.ensemble_run(2)
    assert len(engine.shard_profile_read()) == 0        # no exchange without sharding


@pytest.mark.gpu
def test_two_structures_are_swept_reproducibly(engine):
    """DRW + SHO + Lorentzian with the SHO on both sides of Q = 1/2 and frequencies up to the top of the prior box:
    the structure lists are filled with atomics and a wave holding one row with a huge d dx takes the libm sincos for
    all its rows, so the values used to depend on which waves arrived first.  The sweep now runs in sorted order and
    repeats bit for bit -- also for a batch grouped by light curve, which needs no sort for locality."""
    kinds = synth.ALT_MODEL
    N, L, W = 400, 40, 512
    t, y, dy = synth.make_lightcurves(N, L, seed=77)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    rng = np.random.default_rng(78)
    lo, hi = bounds[free, 0], bounds[free, 1]
    theta = synth.draw_thetas(kinds, L * W, seed=79)
    wild = rng.random(L * W) < 0.3                      # a third of the rows anywhere in the box
    theta[wild] = rng.uniform(np.maximum(lo, -8.0), np.minimum(hi, 9.0), size=(int(wild.sum()), len(lo)))
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    engine.set_time_parallel(0)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    try:
        first, st = engine.loglike(theta, lc, add_prior=True)
        assert engine.last_solver.startswith(("mtg_solve_kernel", "mtg_pipe_kernel")) and (st == 0).sum() > L * W // 2
        for _ in range(4):
            again, st2 = engine.loglike(theta, lc, add_prior=True)
            assert np.array_equal(st, st2) and np.array_equal(first, again, equal_nan=True)
    finally:
        engine.set_time_parallel(2)
