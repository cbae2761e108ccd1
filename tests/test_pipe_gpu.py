"""The serial sweep as a two-wave pipeline (csrc/mtg_sweep_pipe.h, mtg_kernels_pipe.hip; mtg_set_pipeline): a producer
wave computes the generators of every sample, a consumer wave runs the recurrences.  Same expressions in the same
order as the one-lane-per-evaluation sweep, so a row must come out THE SAME TO THE LAST BIT as from
mtg_solve_kernel / mtg_solve_kernel_multi -- and within 1e-8 of the oracle (BASELINE.json north_star)."""
import numpy as np
import pytest

from mind_the_gaps_amd import synthetic as synth
from oracle import celerite as oracle_c

pytestmark = pytest.mark.gpu

K = synth
MODELS = {
    "null_drw_sho": [K.K_DRW, K.K_SHO],                      # (1, 1) / (3, 0): two structures
    "alt_drw_sho_lorentzian": K.ALT_MODEL,                   # (1, 2) / (3, 1), last complex term with b = 0
    "drw_lorentzian": [K.K_DRW, K.K_LORENTZIAN],             # (1, 1), one structure, b = 0
    "real_complex4": [K.K_REAL, K.K_COMPLEX4],               # (1, 1), one structure, b != 0
    "two_sho": [K.K_SHO, K.K_SHO],                           # (0, 2) / (2, 1) / (4, 0): three structures
    "drw_complex4_complex3_real": [K.K_DRW, K.K_COMPLEX4, K.K_COMPLEX3, K.K_REAL],   # (2, 2): rank 6
    "matern_sho": [K.K_MATERN32, K.K_SHO],                   # (0, 2) / (2, 1)
}


@pytest.fixture
def pipe(engine):
    engine.set_time_parallel(0)
    yield engine
    engine.set_pipeline(2)
    engine.set_time_parallel(2)


def both(eng, theta, lc, add_prior=True):
    eng.set_pipeline(0)
    want, wst = eng.loglike(theta, lc, add_prior=add_prior)
    serial = eng.last_solver
    eng.set_pipeline(1)
    got, gst = eng.loglike(theta, lc, add_prior=add_prior)
    assert "mtg_pipe_kernel" in eng.last_solver and "mtg_pipe_kernel" not in serial, (serial, eng.last_solver)
    return want, wst, got, gst


@pytest.mark.parametrize("name", sorted(MODELS))
@pytest.mark.parametrize("N", [64, 71, 77, 1000, 1002, 1003, 1009])
def test_pipeline_is_the_serial_sweep_bit_for_bit(pipe, name, N):
    """Every compiled shape, light-curve lengths on all sides of the chunking (four samples per hand-over, trips of three
    chunks), several light curves in scattered order, prior rejections and both SHO regimes in one batch."""
    kinds = MODELS[name]
    L, B = 7, 900
    t, y, dy = synth.make_lightcurves(N, L, seed=N)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    pipe.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    pipe.set_model(kinds, full, free, bounds)
    rng = np.random.default_rng(5)
    theta = synth.draw_thetas(kinds, B, seed=N + 1, percent=0.5 if K.K_SHO in kinds else 0.15)
    theta[::97, 0] = 60.0                                       # outside the prior box
    lc = rng.integers(0, L, B).astype(np.int32)
    want, wst, got, gst = both(pipe, theta, lc)
    assert np.array_equal(gst, wst)
    ok = wst == 0
    assert ok.sum() > B // 2 and (wst == 1).sum() >= 9
    assert np.array_equal(got, want)                            # -inf where rejected, bit for bit elsewhere
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y.mean(axis=1)[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=8)
    assert np.array_equal(rst, gst)
    assert np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8


def test_pipeline_with_a_fitted_mean_jitter_and_per_lightcurve_times(pipe):
    """The MEAN variant of the consumer (linear mean function fitted, JitterTerm) and per-light-curve sampling."""
    N, L, B = 333, 5, 640
    rng = np.random.default_rng(8)
    t = np.cumsum(0.05 + rng.exponential(1.0, (L, N)), axis=1)
    _, y, dy = synth.make_lightcurves(N, L, seed=12)
    kinds = [K.K_DRW, K.K_SHO, K.K_JITTER]
    from mind_the_gaps_amd.engine import MEAN_LINEAR
    full, free, bounds = synth.model_spec(kinds, y, mean_kind=MEAN_LINEAR, fit_mean=True)
    pipe.set_lightcurves(t, y, dy + 1e-12)
    pipe.set_model(kinds, full, free, bounds, mean_kind=MEAN_LINEAR)
    theta = np.tile(full, (B, 1)) + 0.05 * rng.standard_normal((B, len(full)))
    theta[:, -2] = 1e-3 * rng.standard_normal(B)                # slope
    theta[::4, 3] = np.log(0.2)                                 # over-damped rows
    lc = rng.integers(0, L, B).astype(np.int32)
    want, wst, got, gst = both(pipe, theta, lc, add_prior=False)
    assert np.array_equal(gst, wst) and np.all(gst == 0)
    assert np.array_equal(got, want)


def test_pipeline_takes_the_libm_sincos_with_the_serial_sweep(pipe):
    """A row whose phase step leaves the table's range sends its 64 rows through the libm sincos in both kernels
    (decided per 64 consecutive rows of the order in both): still bit for bit."""
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.exponential(0.5, 150))
    t[75:] += 2.0e8
    y, dy = rng.standard_normal(150), rng.uniform(0.2, 0.5, 150)
    kinds = [K.K_COMPLEX3, K.K_DRW]
    base = np.array([np.log(2.0), np.log(0.3), np.log(3.0), np.log(1.5), np.log(0.2)])
    full, free = np.concatenate([base, [0.0]]), np.arange(5, dtype=np.int32)
    bounds = np.tile([-np.inf, np.inf], (6, 1))
    theta = base + 0.01 * rng.standard_normal((200, 5))
    theta[70, 2] = np.log(9000.0)                               # d dx = 1.8e12 in the second group of 64 rows
    pipe.set_lightcurves(t, y, dy + 1e-12)
    pipe.set_model(kinds, full, free, bounds)
    want, wst, got, gst = both(pipe, theta, None, add_prior=False)
    assert np.array_equal(gst, wst) and np.array_equal(got, want)
    pipe.set_pipeline(0)
    calm, _ = pipe.loglike(np.delete(theta, 70, axis=0), None, add_prior=False)
    # without that row its wave-mates take the table sincos: the first 64 rows are untouched, the next 63 are not
    assert np.array_equal(calm[:64], want[:64]) and np.any(calm[64:127] != np.r_[want[64:70], want[71:128]])


def test_pipeline_default_dispatch_and_the_device_sampler(pipe):
    """Automatic mode: a batch beyond the time-parallel kernels' range that fits one workgroup per compute unit goes to
    the pipeline, a larger one to the one-lane sweep; the device sampler's half-steps take it too, same chain."""
    kinds = K.ALT_MODEL
    N, L, W = 300, 96, 128
    t, y, dy = synth.make_lightcurves(N, L, seed=2)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    pipe.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    pipe.set_model(kinds, full, free, bounds)
    pipe.set_time_parallel(2)
    pipe.set_pipeline(2)
    B = L * W
    theta = synth.draw_thetas(kinds, B, seed=4)
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    out, st = pipe.loglike(theta, lc)
    assert "mtg_pipe_kernel" in pipe.last_solver, pipe.last_solver         # 12 288 rows
    big = np.tile(theta, (4, 1))
    out4, st4 = pipe.loglike(big, np.tile(lc, 4))
    assert "mtg_solve_kernel" in pipe.last_solver, pipe.last_solver        # 49 152 rows: beyond one round
    assert np.array_equal(out4[:B], out) and np.array_equal(st4[:B], st)
    small, _ = pipe.loglike(theta[:2000], lc[:2000])
    assert "mtg_tp_" in pipe.last_solver, pipe.last_solver
    # the sampler: 96 ensembles x 64 proposals per half-step = 6144 rows -> time-parallel by default; force the serial
    # forms and compare the chains
    p0 = synth.truth(kinds) * (1 + 0.02 * np.random.default_rng(1).standard_normal((L, W, len(free))))
    chains = []
    pipe.set_time_parallel(0)
    for mode in (0, 1):
        pipe.set_pipeline(mode)
        pipe.ensemble_init(p0, seed=99)
        chains.append(pipe.ensemble_run(6, store_chain=True))
        assert ("mtg_pipe_kernel" in pipe.last_solver) == bool(mode), pipe.last_solver
    assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])


def test_models_without_a_pipeline_keep_the_one_lane_sweep(pipe):
    """Three complex terms hand over more than the ring holds, a real-only model next to nothing: mtg_set_pipeline(1)
    leaves them on the one-lane-per-evaluation sweep."""
    t, y, dy = synth.make_lightcurves(200, 2, seed=3)
    for kinds in ([K.K_COMPLEX4, K.K_COMPLEX3, K.K_LORENTZIAN], [K.K_DRW, K.K_REAL]):
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        pipe.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        pipe.set_model(kinds, full, free, bounds)
        pipe.set_pipeline(1)
        theta = synth.draw_thetas(kinds, 300, seed=1)
        out, st = pipe.loglike(theta, np.zeros(300, dtype=np.int32))
        assert "mtg_solve_kernel" in pipe.last_solver and np.all(st <= 1) and np.all(np.isfinite(out[st == 0])) and (st == 0).sum() > 100


def test_a_rows_bits_do_not_depend_on_its_batch_under_modes_0_and_3(engine):
    """What a job needs that cuts its rows over several GPUs and wants the one-GPU numbers (ppp.protassov_test(reproducible=
    True)): under mtg_set_time_parallel 0 (one-lane sweep or its pipeline) and 3 (the one-wave time-parallel kernel
    only) a row's value is the same whether it travels alone, with 200 or with 9000 others; under the automatic mode the
    batch size picks between one, two and four waves per evaluation, whose sums differ in the last bits."""
    kinds = K.ALT_MODEL
    N, B = 4500, 9000
    t, y, dy = synth.make_lightcurves(N, 1, seed=6)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    engine.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    engine.set_model(kinds, full, free, bounds)
    theta = synth.draw_thetas(kinds, B, seed=3, percent=0.05)
    try:
        seen = {}
        for mode in (0, 3, 2):
            engine.set_time_parallel(mode)
            outs = {}
            for rows in (1, 200, 600, B):
                outs[rows], st = engine.loglike(theta[:rows], None, add_prior=True)
                assert np.all(st == 0)
                seen[(mode, rows)] = engine.last_solver
            if mode in (0, 3):
                for rows in (1, 200, 600):
                    assert np.array_equal(outs[rows], outs[B][:rows]), (mode, rows, seen)
            else:
                assert any(not np.array_equal(outs[rows], outs[B][:rows]) for rows in (200, 600)), seen
            ref = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta[:200], np.full((200, 1), y.mean())]), bounds=bounds,
                                         add_prior=True, nthreads=8)[0]
            assert np.max(np.abs(outs[200] - ref) / np.abs(ref)) <= 1e-8
        assert "mtg_tp_kernel" in seen[(3, B)] or "mtg_tp_fused_kernel" in seen[(3, B)], seen      # one wave per evaluation, even for 9000 rows
    finally:
        engine.set_time_parallel(2)


def test_context_on_a_slice_of_the_compute_units():
    """mtg_create_on_slice: a context whose kernels keep to half of the GPU gives the numbers of an ordinary one (the pipeline
    sizes itself by the slice: half the rows per round), and two of them work side by side from two threads."""
    import threading
    from mind_the_gaps_amd.engine import Engine
    kinds = K.NULL_MODEL
    N, L, W = 300, 40, 128
    t, y, dy = synth.make_lightcurves(N, L, seed=4)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    theta = synth.draw_thetas(kinds, L * W, seed=5)
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    engines = [Engine(0), Engine(0, cu_slice=(0, 2)), Engine(0, cu_slice=(1, 2))]
    try:
        for eng in engines:
            eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
            eng.set_model(kinds, full, free, bounds)
        want, wst = engines[0].loglike(theta, lc)
        got = {}

        def work(i):
            got[i] = engines[i].loglike(theta, lc)
        threads = [threading.Thread(target=work, args=(i,)) for i in (1, 2)]
        [th.start() for th in threads]
        [th.join() for th in threads]
        for i in (1, 2):
            assert np.array_equal(got[i][1], wst) and np.array_equal(got[i][0], want)
        with pytest.raises(Exception):
            Engine(0, cu_slice=(2, 2))
    finally:
        for eng in engines:
            eng.close()


@pytest.mark.parametrize("name", ["null_drw_sho", "alt_drw_sho_lorentzian"])
def test_shipped_shape_against_the_oracle(pipe, name):
    """The shape the pipeline exists for -- one GPU's half-step of BASELINE configs[3] at 8 GPUs: 250 light curves x 128
    proposals = 32 000 rows, N = 1e4, both models -- compared with the ORACLE row by row (not only with the one-lane
    sweep, which the tests above show it equals), through the default dispatch (mtg_set_pipeline 2: no forcing)."""
    kinds = MODELS[name]
    N, L, W = 10000, 250, 128
    t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    pipe.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    pipe.set_model(kinds, full, free, bounds)
    pipe.set_pipeline(2)
    theta = synth.draw_thetas(kinds, L * W, seed=20250704 + 44)
    theta[::1013, 0] = 60.0                                     # a few rows outside the prior box
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    got, gst = pipe.loglike(theta, lc, add_prior=True)
    assert "mtg_pipe_kernel" in pipe.last_solver, pipe.last_solver
    ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y.mean(axis=1)[lc][:, None]]),
                                      bounds=bounds, lc_index=lc, add_prior=True, nthreads=16)
    assert np.array_equal(rst, gst)
    ok = gst == 0
    assert ok.sum() > 31000
    assert np.all(np.isneginf(got[gst == 1]))
    assert np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])) <= 1e-8


def _two_models(N=1000, L=9, W=40, seed=3, kinds_of=None):
    """Two contexts on the same light curves: DRW + SHO (null) and DRW + SHO + Lorentzian (alternative) by default."""
    from mind_the_gaps_amd.engine import Engine
    t, y, dy = synth.make_lightcurves(N, L, seed=seed)
    engines, thetas = [], []
    kinds_of = kinds_of or (MODELS["null_drw_sho"], MODELS["alt_drw_sho_lorentzian"])
    lc = np.repeat(np.arange(L, dtype=np.int32), W)
    for i, kinds in enumerate(kinds_of):
        eng = Engine(0)
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, free, bounds)
        eng.set_time_parallel(0)
        eng.set_pipeline(1)
        th = synth.draw_thetas(kinds, L * W, seed=seed + 10 + i, percent=0.5)     # both SHO regimes
        th[::53, 0] = 60.0                                                       # a few rows outside the prior box
        engines.append(eng)
        thetas.append(th)
    return engines, thetas, lc


def _both_at_once(engines, thetas, lc, rounds=1):
    """Every engine's batch from a thread of its own, `rounds` times -> [(lnP, status, solver) of the last round]."""
    import threading
    out = [None, None]

    def run(i):
        for _ in range(rounds):
            got = engines[i].loglike(thetas[i], lc, add_prior=True)
            out[i] = (got[0], got[1], engines[i].last_solver)
    threads = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return out


@pytest.mark.parametrize("N", [64, 70, 1001])
def test_paired_contexts_share_a_launch_and_keep_every_bit(N):
    """mtg_pair_contexts: the pipelined half-steps of two models in one launch of eight-wave workgroups
    (csrc/mtg_kernels_pipe_pair.hip).  Same rows, same bits as each model's own mtg_pipe_kernel -- odd and even numbers of
    two-sample chunks, both SHO regimes, prior rejections, grids of different sizes (the models' segments pad
    differently) --, in either order of pairing, repeatedly."""
    engines, thetas, lc = _two_models(N=N)
    try:
        alone = [engines[i].loglike(thetas[i], lc, add_prior=True) + (engines[i].last_solver,) for i in (0, 1)]
        assert all("mtg_pipe_kernel" in a[2] for a in alone)
        for first, second in ((0, 1), (1, 0)):
            engines[first].pair_with(engines[second])
            got = _both_at_once(engines, thetas, lc, rounds=3)
            stats = engines[0].pair_stats()
            engines[0].unpair()
            assert stats == {"paired": 3, "solo": 0, "broken": False}, stats
            for i in (0, 1):
                assert "mtg_pipe_pair_kernel" in got[i][2], got[i][2]
                assert np.array_equal(got[i][1], alone[i][1])
                assert np.array_equal(got[i][0], alone[i][0])          # -inf where rejected, bit for bit elsewhere
        assert engines[1].pair_stats() == {"paired": 0, "solo": 0, "broken": False}     # unpaired: nothing to report
    finally:
        for eng in engines:
            eng.close()


def test_a_partner_that_does_not_come_breaks_the_pair_not_the_run():
    """A paired context whose partner does not reach a pipelined half-step in time (mtg_set_pair_patience) launches alone
    THIS time -- a partner may be stalled between two C calls for a moment --, waits half as long the next time, and after
    four misses in a row never waits again; a pair of shapes without a compiled kernel goes alone at once.  Same results."""
    import time
    engines, thetas, lc = _two_models(N=300)
    try:
        alone = engines[1].loglike(thetas[1], lc, add_prior=True)
        engines[0].pair_with(engines[1])
        engines[1].set_pair_patience(40)
        t0 = time.perf_counter()
        got = engines[1].loglike(thetas[1], lc, add_prior=True)        # the partner never calls
        waited = time.perf_counter() - t0
        assert engines[1].pair_stats() == {"paired": 0, "solo": 1, "broken": False} and 0.03 < waited < 2.0
        for _ in range(3):
            again = engines[1].loglike(thetas[1], lc, add_prior=True)
        stats = engines[1].pair_stats()
        t0 = time.perf_counter()
        fifth = engines[1].loglike(thetas[1], lc, add_prior=True)      # broken: no wait at all
        assert time.perf_counter() - t0 < 1.0
        engines[1].unpair()
        assert "mtg_pipe_kernel" in engines[1].last_solver
        assert np.array_equal(got[0], alone[0]) and np.array_equal(again[0], alone[0]) and np.array_equal(fifth[0], alone[0])
        assert stats == {"paired": 0, "solo": 4, "broken": True}
        # the same model twice is not a pair the library has a kernel for: known at pairing time, both go alone at once
        full_kinds = MODELS["alt_drw_sho_lorentzian"]
        t, y, dy = synth.make_lightcurves(300, 9, seed=3)
        full, free, bounds = synth.model_spec(full_kinds, y, per_lc_mean=True)
        engines[0].set_model(full_kinds, full, free, bounds)
        engines[0].pair_with(engines[1])
        assert engines[0].pair_stats()["broken"]                       # (no waiting: the models are known already)
        t0 = time.perf_counter()
        both = _both_at_once(engines, [thetas[1], thetas[1]], lc)
        assert time.perf_counter() - t0 < 0.03 + 1.0                   # nobody sat out the patience
        stats = engines[0].pair_stats()
        engines[0].unpair()
        assert stats["paired"] == 0 and stats["broken"] and stats["solo"] == 2
        assert np.array_equal(both[0][0], alone[0]) and np.array_equal(both[1][0], alone[0])
    finally:
        for eng in engines:
            eng.close()


def test_a_context_may_be_closed_while_its_partner_waits_for_it():
    """mtg_destroy on one member of a pair while the partner's thread sits in the rendezvous (waiting for a half-step that
    will never come): the partner wakes, launches alone with the right numbers, and nothing is freed under it -- the
    rendezvous is reference-counted (csrc/mtg_capi.hip, MtgPair)."""
    import threading
    import time
    engines, thetas, lc = _two_models(N=300)
    try:
        alone = engines[1].loglike(thetas[1], lc, add_prior=True)
        for _ in range(5):
            engines[0].pair_with(engines[1])
            engines[1].set_pair_patience(2000)
            got = {}
            worker = threading.Thread(target=lambda: got.update(out=engines[1].loglike(thetas[1], lc, add_prior=True)))
            t0 = time.perf_counter()
            worker.start()
            time.sleep(0.05)                       # the worker is inside pair_launch, waiting for engine 0
            engines[0].unpair()                    # (what mtg_destroy does first)
            worker.join(10)
            assert not worker.is_alive() and time.perf_counter() - t0 < 1.5      # woken by the unpairing, not by the patience
            assert np.array_equal(got["out"][0], alone[0])
            assert engines[1].pair_stats() == {"paired": 0, "solo": 0, "broken": False}     # unpaired: nothing to report
        engines[0].pair_with(engines[1])
        engines[1].set_pair_patience(2000)
        worker = threading.Thread(target=lambda: got.update(out=engines[1].loglike(thetas[1], lc, add_prior=True)))
        worker.start()
        time.sleep(0.05)
        engines[0].close()                         # destroyed outright under the waiting partner
        worker.join(10)
        assert not worker.is_alive() and np.array_equal(got["out"][0], alone[0])
    finally:
        for eng in engines:
            eng.close()


# the pairs csrc/mtg_kernels_pipe_pair.hip is compiled for: a null model and the alternative that adds one term to it
PAIRS = {
    "drw_sho | + lorentzian": ([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_LORENTZIAN]),
    "drw_sho | + sho": ([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_SHO]),
    "drw_sho | + real": ([K.K_DRW, K.K_SHO], [K.K_DRW, K.K_REAL, K.K_SHO]),
    "drw_lorentzian | + lorentzian": ([K.K_DRW, K.K_LORENTZIAN], [K.K_DRW, K.K_LORENTZIAN, K.K_LORENTZIAN]),
    "drw_lorentzian | + sho": ([K.K_DRW, K.K_LORENTZIAN], [K.K_DRW, K.K_LORENTZIAN, K.K_SHO]),
    "sho_lorentzian | + drw": ([K.K_SHO, K.K_LORENTZIAN], [K.K_DRW, K.K_SHO, K.K_LORENTZIAN]),
    "sho_sho | + drw": ([K.K_SHO, K.K_SHO], [K.K_DRW, K.K_SHO, K.K_SHO]),
    "drw_real_sho | + lorentzian": ([K.K_DRW, K.K_REAL, K.K_SHO], [K.K_DRW, K.K_REAL, K.K_SHO, K.K_LORENTZIAN]),
}


@pytest.mark.parametrize("name", sorted(PAIRS))
def test_every_compiled_pair_of_models(name):
    """Each of the eight null / alternative pairs with a paired kernel: dispatched in one launch, in either order of the
    two contexts, every row the same to the last bit as from the model's own pipelined kernel."""
    engines, thetas, lc = _two_models(N=333, L=7, W=50, seed=21, kinds_of=PAIRS[name])
    try:
        alone = [engines[i].loglike(thetas[i], lc, add_prior=True) + (engines[i].last_solver,) for i in (0, 1)]
        assert all("mtg_pipe_kernel" in a[2] for a in alone), [a[2] for a in alone]
        for first, second in ((0, 1), (1, 0)):
            engines[first].pair_with(engines[second])
            got = _both_at_once(engines, thetas, lc, rounds=2)
            stats = engines[0].pair_stats()
            engines[0].unpair()
            assert stats == {"paired": 2, "solo": 0, "broken": False}, (name, stats)
            for i in (0, 1):
                assert "mtg_pipe_pair_kernel" in got[i][2]
                assert np.array_equal(got[i][1], alone[i][1]) and np.array_equal(got[i][0], alone[i][0])
                assert (alone[i][1] == 0).sum() > 300
    finally:
        for eng in engines:
            eng.close()


def test_light_curves_of_different_lengths_do_not_pair():
    """Two contexts whose light curves differ in length walk different numbers of hand-overs: no shared launch (the pair
    breaks, both go alone, same results)."""
    a, ta, lc = _two_models(N=300)
    b, tb, _ = _two_models(N=310)
    engines = [a[0], b[1]]
    try:
        alone = [engines[0].loglike(ta[0], lc, add_prior=True), engines[1].loglike(tb[1], lc, add_prior=True)]
        engines[0].pair_with(engines[1])
        got = _both_at_once(engines, [ta[0], tb[1]], lc)
        stats = engines[0].pair_stats()
        engines[0].unpair()
        assert stats["paired"] == 0 and stats["broken"]
        for i in (0, 1):
            assert np.array_equal(got[i][0], alone[i][0]) and "mtg_pipe_kernel" in got[i][2]
    finally:
        for eng in a + b:
            eng.close()
