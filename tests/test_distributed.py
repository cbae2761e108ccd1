"""CPU tests of the multi-GPU path: world_size-2 gloo processes stand in for two
MI355X ranks; the oracle stands in for the engine call (tests may use the oracle)."""
import os
import socket
import sys

import numpy as np
import pytest

from mind_the_gaps_amd import distributed as mdist
from mind_the_gaps_amd import synthetic as synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_bounds_balanced_and_contiguous():
    for n, w in ((2000, 8), (10, 3), (3, 8), (0, 2), (7, 1)):
        b = mdist.block_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and len(b) == w + 1
        sizes = np.diff(b)
        assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1
        assert [mdist.shard_rows(n, r, w) for r in range(w)] == [(int(b[r]), int(b[r + 1])) for r in range(w)]
    s = mdist.LightcurveShard(10, rank=1, world_size=3)
    assert (s.lo, s.hi, len(s)) == (4, 7, 3)
    assert list(s.to_local([4, 6])) == [0, 2] and list(s.owns([3, 4, 6, 7])) == [False, True, True, False]
    with pytest.raises(ValueError):
        s.to_local([7])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import celerite as oracle_c
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kinds = synth.NULL_MODEL
    N, L, W = 120, 5, 6
    t, y, dy = synth.make_lightcurves(N, L, seed=11)
    y_mean = y.mean(axis=1)
    theta = synth.draw_thetas(kinds, L * W, seed=12)
    lc = np.repeat(np.arange(L, dtype=np.int32), W)

    def evaluate_global(th, lcs):                     # the "engine" of a rank that holds every light curve
        full = np.hstack([th, y_mean[lcs][:, None]])
        return oracle_c.logprob_batch(t, y, dy, kinds, full, lc_index=lcs)

    # (1) walker sharding of one replicated data set + all-gather of lnP
    lnp, st = mdist.sharded_log_prob(evaluate_global, theta, lc)
    # (2) light-curve sharding: this rank only holds its block of light curves
    shard = mdist.LightcurveShard(L)
    mine = shard.owns(lc)

    def evaluate_local(th, lcs_local):
        yl, dyl, ml = y[shard.lo:shard.hi], dy[shard.lo:shard.hi], y_mean[shard.lo:shard.hi]
        full = np.hstack([th, ml[lcs_local][:, None]])
        return oracle_c.logprob_batch(t, yl, dyl, kinds, full, lc_index=lcs_local)

    lnp_local, _ = evaluate_local(theta[mine], shard.to_local(lc[mine]))
    best_local = lnp_local.reshape(len(shard), W).max(axis=1)      # max lnL per light curve
    best = shard.gather(best_local)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lnp=lnp, st=st, best=best,
             lo=shard.lo, hi=shard.hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    import torch.multiprocessing as mp
    from oracle import celerite as oracle_c
    oracle_c.lib()                                    # build the checker before forking workers
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world))
    # single-process truth
    kinds = synth.NULL_MODEL
    t, y, dy = synth.make_lightcurves(120, 5, seed=11)
    theta = synth.draw_thetas(kinds, 30, seed=12)
    lc = np.repeat(np.arange(5, dtype=np.int32), 6)
    want, wst = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, y.mean(axis=1)[lc][:, None]]), lc_index=lc)
    for r in (r0, r1):
        assert np.array_equal(r["lnp"], want) and np.array_equal(r["st"], wst)       # every rank has the full vector
        assert np.array_equal(r["best"], want.reshape(5, 6).max(axis=1))
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 3, 3, 5)


def _ppp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import celerite as oracle_c
    from mind_the_gaps_amd.models import DampedRandomWalk
    from mind_the_gaps_amd.ppp import derive_posteriors_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kinds = [synth.K_DRW]
    t, y, dy = synth.make_lightcurves(60, 5, seed=21)
    shard = mdist.LightcurveShard(5)
    yl, dyl = y[shard.lo:shard.hi], dy[shard.lo:shard.hi]

    def evaluate(theta, lc, add_prior):           # this rank's "engine": only its own light curves
        full = np.hstack([theta, yl.mean(axis=1)[lc][:, None]])
        b = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
        return oracle_c.logprob_batch(t, yl, dyl, kinds, full, bounds=b, lc_index=np.asarray(lc, np.int32),
                                      add_prior=add_prior)

    th = synth.truth(kinds)
    kernel = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)])
    best, local, sh = derive_posteriors_sharded(t, y, dy, kernel, walkers=8, max_steps=25, fit=False, seed=3,
                                                evaluate=evaluate, store_chain=False)
    np.savez(os.path.join(out_dir, "ppp%d.npz" % rank), best=best, local=local.max_loglikelihood, lo=sh.lo, hi=sh.hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_posteriors(tmp_path):
    """Light-curve sharded lock-step fit on 2 gloo ranks: every rank ends with the full
    max-lnL vector, made of each rank's own block."""
    import torch.multiprocessing as mp
    from oracle import celerite as oracle_c
    oracle_c.lib()
    world, port = 2, _free_port()
    mp.spawn(_ppp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / ("ppp%d.npz" % r)) for r in range(world))
    assert np.array_equal(r0["best"], r1["best"]) and r0["best"].shape == (5,)
    assert np.array_equal(r0["best"][:3], r0["local"]) and np.array_equal(r0["best"][3:], r1["local"])
    assert np.all(np.isfinite(r0["best"]))


# ---- walker-sharded chain of ONE light curve (BASELINE configs[1], [2], [4]) ----
def _chain_problem():
    kinds = synth.NULL_MODEL
    t, y, dy = synth.make_lightcurves(80, 1, seed=31)
    bounds = np.vstack([synth.bounds_for(kinds), [(-np.inf, np.inf)]])
    truth = synth.truth(kinds)
    return kinds, t, y, dy, bounds, truth


def _oracle_log_prob(coords):
    from oracle import celerite as oracle_c
    kinds, t, y, dy, bounds, _ = _chain_problem()
    full = np.hstack([coords, np.full((len(coords), 1), y.mean())])
    return oracle_c.logprob_batch(t, y, dy, kinds, full, bounds=bounds, add_prior=True)[0]


def _chain_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mind_the_gaps_amd.sampler import EnsembleSampler
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    truth = _chain_problem()[5]
    W, P = 12, len(truth)
    np.random.seed(100 + rank)                         # every rank starts from a different generator
    p0 = truth + 0.01 * np.random.randn(W, P)          # ... and a different ensemble
    calls = []

    def local(coords):
        calls.append(len(coords))
        return _oracle_log_prob(coords)

    sampler = EnsembleSampler(W, P, mdist.WalkerShardedLogProb(local))
    p0 = mdist.lockstep(sampler, p0)
    sampler.run_mcmc(p0, 15)
    np.savez(os.path.join(out_dir, "chain%d.npz" % rank), chain=sampler.get_chain(), lnp=sampler.get_log_prob(),
             calls=np.array(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_walker_sharded_chain(tmp_path):
    """Two gloo ranks, one light curve: each evaluates half of every half-ensemble, both end with
    the chain a single process produces from rank 0's generator and starting ensemble."""
    import torch.multiprocessing as mp
    from oracle import celerite as oracle_c
    from mind_the_gaps_amd.sampler import EnsembleSampler
    oracle_c.lib()
    world, port = 2, _free_port()
    mp.spawn(_chain_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / ("chain%d.npz" % r)) for r in range(world))
    assert np.array_equal(r0["chain"], r1["chain"]) and np.array_equal(r0["lnp"], r1["lnp"])
    # every half-step evaluated 6 proposals as 3 + 3 (the initial ensemble as 6 + 6)
    assert list(r0["calls"]) == [6] + [3] * 30 and list(r1["calls"]) == [6] + [3] * 30
    truth = _chain_problem()[5]
    np.random.seed(100)
    p0 = truth + 0.01 * np.random.randn(12, len(truth))
    single = EnsembleSampler(12, len(truth), _oracle_log_prob)
    single.run_mcmc(p0, 15)
    assert np.array_equal(single.get_chain(), r0["chain"]) and np.array_equal(single.get_log_prob(), r0["lnp"])
    assert len(np.unique(r0["chain"][:, 0, 0])) > 3          # the chain moved


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local(coords):
        if rank == 1:
            raise ArithmeticError("failed to factorize or solve matrix")
        return np.zeros(len(coords))

    try:
        mdist.WalkerShardedLogProb(local)(np.zeros((4, 2)))
        outcome = "returned"
    except ArithmeticError:
        outcome = "own error"
    except RuntimeError:
        outcome = "peer error"
    open(os.path.join(out_dir, "fail%d.txt" % rank), "w").write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_failure_on_one_rank_reaches_every_rank(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_failing_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "fail0.txt").read() == "peer error" and open(tmp_path / "fail1.txt").read() == "own error"


def _watchdog(out_dir, tag, seconds=150):
    """A worker that is still running after `seconds` writes every thread's stack to out_dir and exits: a
    deadlock between the ranks then fails the test with the place where each rank waited instead of hanging
    the suite."""
    import faulthandler
    fh = open(os.path.join(out_dir, "hang_%s.txt" % tag), "w")
    faulthandler.dump_traceback_later(seconds, exit=True, file=fh)
    return fh


def _spawn(fn, args, nprocs, out_dir):
    """mp.spawn with one retry when a worker was stopped by its watchdog (a rendezvous that never completed on a box
    still paging its libraries in); any other failure, or a second one, fails the test with the workers' stacks."""
    import glob
    import torch.multiprocessing as mp

    def dumps():
        files = sorted(glob.glob(os.path.join(str(out_dir), "hang_*.txt")))
        return "".join("\n--- %s ---\n%s" % (f, open(f).read()) for f in files if os.path.getsize(f))

    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=args, nprocs=nprocs, join=True)
            return
        except Exception as exc:
            stacks = dumps()
            if attempt == 1 or not stacks:
                raise AssertionError("worker failed: %s%s" % (exc, stacks))
            for f in glob.glob(os.path.join(str(out_dir), "hang_*.txt")):
                os.remove(f)
            args = (args[0], _free_port()) + tuple(args[2:])    # (world, port, ...): a fresh rendezvous


def _gpu_chain_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    guard = _watchdog(out_dir, "host%d_%d" % (world, rank))
    import warnings
    import torch.distributed as dist
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk
    from mind_the_gaps_amd import terms
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    th = synth.truth(synth.NULL_MODEL)
    t, y, dy = synth.make_lightcurves(300, 1, seed=41)
    kernel = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)]) + terms.SHOTerm(
        th[2], th[3], th[4], bounds=[(-10, 50), (-10, 10), (-10, 10)])
    g = GPModelling(GappyLightcurve(t, y[0], dy[0]), kernel)
    np.random.seed(7 if world == 1 else 7 + rank)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g.derive_posteriors(fit=False, max_steps=30, convergence_steps=30, walkers=16, progress=False,
                            device_sampler=False, shard_walkers=world > 1)
    np.savez(os.path.join(out_dir, "gpu%d_%d.npz" % (world, rank)), chain=g.sampler.get_chain(),
             lnp=g.sampler.get_log_prob(), best=g.max_loglikelihood)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    guard.close()


def _time_shard_worker(rank, world, port, out_dir):
    """One rank of the time-sharded rank-10 likelihood (proto/time_shard.py): its eighth -- here: half or third -- of
    the light curve for every row, one element per row, one all-gather, the combination on every rank."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import dense
    from proto.kalman_scan import model_matrices
    from proto import time_shard as ts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kinds = [synth.K_SHO] * 5
    N, rows = 1200, 3
    t, y, dy = synth.make_lightcurves(N, 1, seed=5)
    base = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
                           for i in range(5)])
    theta = base * (1 + 0.02 * np.random.default_rng(9).standard_normal((rows, 15)))     # the same rows on every rank
    r, var = y[0] - y[0].mean(), (dy[0] + 1e-12) ** 2
    bounds = ts.time_bounds(N, world)
    mine = []
    for th in theta:                                  # every row, this rank's stretch of time only
        blocks, jitter = model_matrices(dense.build_coeffs(kinds, th))
        mine.append(ts.pack(ts.shard_element(t, r, var, blocks, jitter, bounds[rank], bounds[rank + 1], nchunks=4)))
    mine = torch.from_numpy(np.stack(mine))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)                   # the one exchange: rows x 321 doubles per rank
    lnl = np.array([ts.combine_shards([ts.unpack(gathered[g][i].numpy(), 10) for g in range(world)]) for i in range(rows)])
    np.save(os.path.join(out_dir, "ts%d_%d.npy" % (world, rank)), lnl)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_time_sharded_likelihood_on_gloo_ranks(tmp_path, world):
    """Time sharding of the rank-10 model (what would give BASELINE configs[4] its eighth per GPU): every rank
    reduces its stretch of the light curve to one filtering element per row, one all-gather, every rank combines --
    against the C oracle (celerite's algorithm) on the whole light curve."""
    import torch.multiprocessing as mp
    from oracle import celerite as oracle_c
    mp.spawn(_time_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("ts%d_%d.npy" % (world, r))) for r in range(world)]
    assert all(np.array_equal(got[0], g) for g in got[1:])           # every rank ends with the same numbers
    kinds = [synth.K_SHO] * 5
    t, y, dy = synth.make_lightcurves(1200, 1, seed=5)
    base = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
                           for i in range(5)])
    theta = base * (1 + 0.02 * np.random.default_rng(9).standard_normal((3, 15)))
    ref, st = oracle_c.logprob_batch(t, y, dy, kinds, np.hstack([theta, np.full((3, 1), y[0].mean())]), add_prior=False)
    assert np.all(st == 0)
    assert np.max(np.abs(got[0] - ref) / np.abs(ref)) < 1e-10


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_derive_posteriors_shard_walkers_two_ranks_one_gpu(tmp_path):
    """GPModelling.derive_posteriors(shard_walkers=True) on two ranks (sharing the box's GPU, gloo
    for the gather): identical chains on both, equal to the single-process host-sampler chain."""
    world, port = 2, _free_port()
    _spawn(_gpu_chain_worker, (world, port, str(tmp_path)), world, tmp_path)
    _spawn(_gpu_chain_worker, (1, _free_port(), str(tmp_path)), 1, tmp_path)
    r0, r1, single = (np.load(tmp_path / name) for name in ("gpu2_0.npz", "gpu2_1.npz", "gpu1_0.npz"))
    assert np.array_equal(r0["chain"], r1["chain"]) and np.array_equal(r0["lnp"], r1["lnp"])
    assert np.allclose(single["chain"], r0["chain"], rtol=1e-9, atol=0) and np.allclose(single["lnp"], r0["lnp"], rtol=1e-9)
    assert np.isfinite(float(r0["best"]))


# ---- the Protassov test with its simulated light curves sharded (BASELINE configs[3] as a workflow) ---------

def _protassov_worker(rank, world, port, out_dir, split="lightcurves"):
    sys.path.insert(0, ROOT)
    guard = _watchdog(out_dir, "ppp%d_%d" % (world, rank))
    import warnings
    import torch.distributed as dist
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
    from mind_the_gaps_amd.ppp import protassov_test
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    th = synth.truth(synth.ALT_MODEL)
    t, y, dy = synth.make_lightcurves(400, 1, seed=43)     # (long enough for the time-parallel kernels to be an option)
    lc = GappyLightcurve(t, y[0] + 50.0, dy[0], exposures=0.5 * np.diff(t).min())
    null = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)])
    alt = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)]) + Lorentzian(
        th[5], th[6], th[7], bounds=[(-10, 50), (-10, 10), (-10, 10)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = protassov_test(lc, null, alt, nsims=5, walkers=16, max_steps=60, sim_steps=40, seed=11, sharded=world > 1,
                             split=split, reproducible=True)
    n_local = 0 if res["lightcurves"] is None else len(res["lightcurves"]["rates"])
    both = res["sim_null"] is not None and res["sim_alt"] is not None
    local = (-2.0 * (res["sim_null"].max_loglikelihood - res["sim_alt"].max_loglikelihood)) if both else np.empty(0)
    np.savez(os.path.join(out_dir, "pt%s%d_%d.npz" % (split[0], world, rank)), T_obs=res["T_obs"], T_sim=res["T_sim"],
             p=res["p_value"], n_local=n_local, local=local, has_null=res["sim_null"] is not None,
             has_alt=res["sim_alt"] is not None, obs_null=res["null"] is not None, obs_alt=res["alt"] is not None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    guard.close()


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_protassov_test_sharded_two_ranks_one_gpu(tmp_path):
    """ppp.protassov_test(sharded=True) on two ranks (sharing the box's GPU, gloo): both end with the same T_obs,
    T_sim and p-value; T_sim is rank 0's block of 3 followed by rank 1's block of 2, each the rank's own refits -- and
    (reproducible mode, the default of a sharded run) the numbers are those of ONE process holding all five light
    curves, to the last bit, for either split and for an odd number of ranks."""
    world = 2
    _spawn(_protassov_worker, (world, _free_port(), str(tmp_path)), world, tmp_path)
    r0, r1 = (np.load(tmp_path / ("ptl2_%d.npz" % r)) for r in range(world))
    assert float(r0["T_obs"]) == float(r1["T_obs"]) and float(r0["p"]) == float(r1["p"])
    assert np.array_equal(r0["T_sim"], r1["T_sim"]) and r0["T_sim"].shape == (5,)
    assert (int(r0["n_local"]), int(r1["n_local"])) == (3, 2)
    # the observed light curve's chains are split by model: rank 0 ran the null model's, rank 1 the alternative's
    assert (bool(r0["obs_null"]), bool(r0["obs_alt"]), bool(r1["obs_null"]), bool(r1["obs_alt"])) == (True, False, False, True)
    assert np.allclose(r0["T_sim"][:3], r0["local"], rtol=1e-12) and np.allclose(r0["T_sim"][3:], r1["local"], rtol=1e-12)
    assert np.all(np.isfinite(r0["T_sim"])) and 0.0 <= float(r0["p"]) <= 1.0
    # one rank alone holds everything and the unsharded call is untouched by the new arguments
    _spawn(_protassov_worker, (1, _free_port(), str(tmp_path)), 1, tmp_path)
    single = np.load(tmp_path / "ptl1_0.npz")
    assert int(single["n_local"]) == 5 and np.all(np.isfinite(single["T_sim"]))
    assert bool(single["obs_null"]) and bool(single["obs_alt"])
    assert float(single["T_obs"]) == float(r0["T_obs"])                            # same seed, same observed chains
    # split by light curve: blocks of 3 and 2 against all 5 in one process -- other batch sizes in every launch, the
    # same numbers (noise and Philox streams keyed by global light-curve index, batch-independent kernels)
    assert np.array_equal(r0["T_sim"], single["T_sim"]) and float(r0["p"]) == float(single["p"])
    # split by model: rank 0 refits the null model, rank 1 the alternative, each on all five light curves -- the same
    # light curves, seeds and batches as one process alone, hence its T_sim to the last bit
    _spawn(_protassov_worker, (world, _free_port(), str(tmp_path), "models"), world, tmp_path)
    m0, m1 = (np.load(tmp_path / ("ptm2_%d.npz" % r)) for r in range(world))
    assert (bool(m0["has_null"]), bool(m0["has_alt"]), bool(m1["has_null"]), bool(m1["has_alt"])) == (True, False, False, True)
    assert (int(m0["n_local"]), int(m1["n_local"])) == (5, 5)
    assert np.array_equal(m0["T_sim"], m1["T_sim"]) and float(m0["p"]) == float(m1["p"])
    assert np.array_equal(m0["T_sim"], single["T_sim"]) and float(m0["T_obs"]) == float(single["T_obs"])
    # three ranks, split by model: ranks 0 and 1 as above, the odd rank out takes no part in the refits but holds the result
    _spawn(_protassov_worker, (3, _free_port(), str(tmp_path), "models"), 3, tmp_path)
    o = [np.load(tmp_path / ("ptm3_%d.npz" % r)) for r in range(3)]
    assert [int(x["n_local"]) for x in o] == [5, 5, 0]
    assert [(bool(x["obs_null"]), bool(x["obs_alt"])) for x in o] == [(True, False), (False, True), (False, False)]
    assert all(np.array_equal(x["T_sim"], single["T_sim"]) and float(x["p"]) == float(single["p"]) for x in o)


def _protassov_observed_failure_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    guard = _watchdog(out_dir, "pof%d" % rank)
    import warnings
    import torch.distributed as dist
    from mind_the_gaps_amd import gpmodelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
    from mind_the_gaps_amd.ppp import protassov_test
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th = synth.truth(synth.ALT_MODEL)
    t, y, dy = synth.make_lightcurves(400, 1, seed=43)
    lc = GappyLightcurve(t, y[0] + 50.0, dy[0], exposures=0.5 * np.diff(t).min())
    null = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)])
    alt = DampedRandomWalk(th[0], th[1], bounds=[(-10, 50), (-10, 10)]) + Lorentzian(
        th[5], th[6], th[7], bounds=[(-10, 50), (-10, 10), (-10, 10)])
    if rank == 1:       # the alternative model's observed chain dies on the rank that runs it

        def broken(self, *a, **k):
            raise ArithmeticError("failed to factorize or solve matrix")
        gpmodelling.GPModelling.derive_posteriors = broken
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            protassov_test(lc, null, alt, nsims=4, walkers=16, max_steps=40, sim_steps=20, seed=11, sharded=True, reproducible=True)
        outcome = "returned"
    except ArithmeticError:
        outcome = "own error"
    except RuntimeError as exc:
        outcome = "peer error" if "rank(s) [1]" in str(exc) else "other: %s" % exc
    open(os.path.join(out_dir, "pof%d.txt" % rank), "w").write(outcome)
    dist.barrier()
    dist.destroy_process_group()
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    guard.close()


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_protassov_observed_chain_failure_reaches_every_rank(tmp_path):
    """Step 1 split by model (rank 0 the null chain, rank 1 the alternative's): a chain that raises on its rank raises on
    every rank before the first broadcast -- nobody waits for a maximum that will never be sent."""
    world = 3
    _spawn(_protassov_observed_failure_worker, (world, _free_port(), str(tmp_path)), world, tmp_path)
    assert [open(tmp_path / ("pof%d.txt" % r)).read() for r in range(world)] == ["peer error", "own error", "peer error"]


# ---- the device-resident sampler, walker-sharded (mtg_ensemble_shard_*) -------------------------------------

class _FakeShardEngine:
    """Stands in for Engine in the CPU test of the host exchange: keeps the callback the way the library would."""

    unsharded = 0

    def ensemble_shard_host(self, rank, world, exchange):
        self.rank, self.world, self.exchange = rank, world, exchange

    # the RCCL bring-up as the library exposes it; `fail_on` = the ranks whose ncclCommInitRank "fails"
    fail_on = ()

    def rccl_unique_id(self):
        return bytes(range(128))

    def ensemble_shard_rccl(self, unique_id, rank, world):
        assert unique_id == bytes(range(128))
        if rank in self.fail_on:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")
        self.comm = (rank, world)

    def ensemble_unshard(self):
        self.unsharded += 1
        self.comm = None


def _host_exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mind_the_gaps_amd.distributed import shard_device_ensemble, broadcast_start
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeShardEngine()
    assert shard_device_ensemble(eng) == "host" and (eng.rank, eng.world) == (rank, world)
    ok = True
    for count in (7, 8, 1, 64):          # ragged last block, even split, fewer rows than ranks
        chunk = -(-count // world)
        lo, hi = min(rank * chunk, count), min((rank + 1) * chunk, count)
        truth_lnp = -np.arange(count, dtype=np.float64) - 0.25
        truth_st = (np.arange(count) % 4).astype(np.int32)
        lnp, st = np.full(count, np.nan), np.full(count, -1, dtype=np.int32)
        lnp[lo:hi], st[lo:hi] = truth_lnp[lo:hi], truth_st[lo:hi]
        eng.exchange(lnp, st, lo, hi)
        ok = ok and np.array_equal(lnp, truth_lnp) and np.array_equal(st, truth_st)
    p0, seed = broadcast_start(np.full((4, 2), float(rank)), 100 + rank)
    ok = ok and np.all(p0 == 0.0) and seed == 100
    from mind_the_gaps_amd.distributed import broadcast_array
    tau = broadcast_array(np.array([[1.5 + rank, 2.0], [np.nan, 7.0 - rank]]))     # rank 0's autocorrelation times win
    ok = ok and tau.shape == (2, 2) and tau[0, 0] == 1.5 and tau[1, 1] == 7.0 and np.isnan(tau[1, 0])
    # a communicator that comes up on rank 0 only: EVERY rank ends on the host-staged exchange, rank 0 having dropped its
    # half-made communicator, and the returned transport says why (bench.py prints it as walker_sharded.*.transport)
    eng = _FakeShardEngine()
    eng.fail_on = (1,)
    with pytest.warns(UserWarning, match="RCCL communicator not available"):
        got = shard_device_ensemble(eng, transport="rccl")
    ok = ok and got.startswith("host (rccl failed: rank 1: RuntimeError: ncclCommInitRank failed") and eng.unsharded == (1 if rank == 0 else 0)
    lnp, st = np.full(5, np.nan), np.full(5, -1, dtype=np.int32)
    lo, hi = min(rank * 3, 5), min(rank * 3 + 3, 5)
    lnp[lo:hi], st[lo:hi] = -np.arange(5.0)[lo:hi], np.arange(5, dtype=np.int32)[lo:hi]
    eng.exchange(lnp, st, lo, hi)
    ok = ok and np.array_equal(lnp, -np.arange(5.0)) and np.array_equal(st, np.arange(5))
    # ... and one that comes up everywhere stays on RCCL
    eng = _FakeShardEngine()
    ok = ok and shard_device_ensemble(eng, transport="rccl") == "rccl" and eng.comm == (rank, world) and eng.unsharded == 0
    open(os.path.join(out_dir, "hx%d.txt" % rank), "w").write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()


def test_host_exchange_of_the_sharded_device_ensemble_two_ranks(tmp_path):
    """distributed.shard_device_ensemble(transport 'host') on two gloo ranks: the callback the library would
    call fills every other rank's rows (lnP and status) in the library's chunk layout; rank 0's start wins."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_host_exchange_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "hx0.txt").read() == "ok" and open(tmp_path / "hx1.txt").read() == "ok"


def _device_chain_worker(rank, world, port, out_dir, case, tp_mode, n, walkers, transport):
    sys.path.insert(0, ROOT)
    guard = _watchdog(out_dir, "dev_%s_%d_%d" % (case, world, rank))
    import faulthandler
    import warnings
    import torch.distributed as dist
    from mind_the_gaps_amd.gp import get_engine
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
    from mind_the_gaps_amd import terms
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device = 0
    if world > 1 and transport == "rccl":   # one GPU per rank, RCCL between them
        import torch
        device = rank
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU of the box
    th = synth.truth(synth.ALT_MODEL)
    t, y, dy = synth.make_lightcurves(n, 1, seed=41)
    amp, other = (-10, 50), (-10, 10)
    kernel = DampedRandomWalk(th[0], th[1], bounds=[amp, other]) + terms.SHOTerm(
        th[2], th[3], th[4], bounds=[amp, other, other]) + Lorentzian(th[5], th[6], th[7], bounds=[amp, other, other])
    g = GPModelling(GappyLightcurve(t, y[0], dy[0]), kernel, device=device)
    np.random.seed(7 if world == 1 else 7 + rank)
    eng = get_engine(device)
    eng.set_time_parallel(tp_mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if transport == "rccl1":   # the RCCL code path with a communicator of one rank
            from mind_the_gaps_amd.device_sampler import DeviceEnsembleSampler
            sampler = g._device_sampler(walkers)
            p0 = g.spread_walkers(walkers, g.initial_params, np.array(g.gp.get_parameter_bounds()))
            bound = sampler._bind()
            bound.ensemble_init(p0[None], seed=sampler.seed)
            bound.ensemble_shard_rccl(bound.rccl_unique_id(), 0, 1)
            chain, lnp = bound.ensemble_run(20, store_chain=True)
            np.savez(os.path.join(out_dir, "dev_%s_rccl.npz" % case), chain=chain[:, 0], lnp=lnp[:, 0])
            bound.ensemble_unshard()
            bound.ensemble_init(p0[None], seed=sampler.seed)
            chain, lnp = bound.ensemble_run(20, store_chain=True)
            np.savez(os.path.join(out_dir, "dev_%s_plain.npz" % case), chain=chain[:, 0], lnp=lnp[:, 0])
            faulthandler.cancel_dump_traceback_later()
            guard.close()
            return
        g.derive_posteriors(fit=False, max_steps=20, convergence_steps=20, walkers=walkers, progress=False,
                            device_sampler=True, shard_walkers=world > 1)
    np.savez(os.path.join(out_dir, "dev_%s_%d_%d.npz" % (case, world, rank)), chain=g.sampler.get_chain(),
             lnp=g.sampler.get_log_prob(), best=g.max_loglikelihood)
    if world > 1:
        assert g.sampler.transport == transport
        if transport == "rccl":
            info = eng.ensemble_shard_info()
            assert info == dict(kind="rccl", rank=rank, world=world, comm_ranks=world), info
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    guard.close()


def _visible_gpus():
    """GPUs of this box, counted without initialising one in the test process (the workers are spawned)."""
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_device_sampler_walker_sharded_rccl_two_gpus(tmp_path):
    """The real transport: two processes, one MI355X each, `mtg_ensemble_shard_rccl` -- a two-rank communicator
    made by the library from the ncclUniqueId that torch.distributed (backend nccl) broadcasts, the grouped in-place
    ncclAllGather pair over xGMI in every half-step.  Both ranks hold the same chain and it is the one-process
    chain BIT FOR BIT (serial sweep: the same kernel whatever the number of rows).  Needs two GPUs: skipped on the
    one-GPU boxes, run by the driver's 8-GPU tier."""
    if _visible_gpus() < 2:
        pytest.skip("needs two GPUs (the one-GPU box cannot put two RCCL ranks on its card)")
    world, port = 2, _free_port()
    _spawn(_device_chain_worker, (world, port, str(tmp_path), "rccl2", 0, 300, 16, "rccl"), world, tmp_path)
    _spawn(_device_chain_worker, (1, _free_port(), str(tmp_path), "rccl2", 0, 300, 16, "rccl"), 1, tmp_path)
    r0, r1, single = (np.load(tmp_path / ("dev_rccl2_%s.npz" % name)) for name in ("2_0", "2_1", "1_0"))
    assert np.array_equal(r0["chain"], r1["chain"]) and np.array_equal(r0["lnp"], r1["lnp"])
    assert np.array_equal(single["chain"], r0["chain"]) and np.array_equal(single["lnp"], r0["lnp"])
    assert len(np.unique(r0["chain"][:, :, 0])) > 16


@pytest.mark.gpu
@pytest.mark.timeout(400)
@pytest.mark.parametrize("case,tp_mode,n,walkers", [("sweep", 0, 300, 16), ("timeparallel", 1, 5000, 32),
                                                    ("ragged", 2, 300, 18)])
def test_device_sampler_walker_sharded_two_ranks_one_gpu(tmp_path, case, tp_mode, n, walkers):
    """derive_posteriors(device_sampler=True, shard_walkers=True) on two processes (sharing the box's GPU, so
    the exchange is the host callback over gloo: RCCL refuses two ranks on one device): both ranks hold the
    same chain, and it is the one-process device chain BIT FOR BIT -- serial sweep, time-parallel kernels
    (same kernel whatever the number of rows) and a half-ensemble that does not split evenly (9 rows)."""
    world, port = 2, _free_port()
    _spawn(_device_chain_worker, (world, port, str(tmp_path), case, tp_mode, n, walkers, "host"), world, tmp_path)
    _spawn(_device_chain_worker, (1, _free_port(), str(tmp_path), case, tp_mode, n, walkers, "host"), 1, tmp_path)
    r0, r1, single = (np.load(tmp_path / ("dev_%s_%s.npz" % (case, name))) for name in ("2_0", "2_1", "1_0"))
    assert np.array_equal(r0["chain"], r1["chain"]) and np.array_equal(r0["lnp"], r1["lnp"])
    assert np.array_equal(single["chain"], r0["chain"]) and np.array_equal(single["lnp"], r0["lnp"])
    assert len(np.unique(r0["chain"][:, :, 0])) > walkers     # the walkers moved
    assert np.isfinite(float(r0["best"]))


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_device_sampler_rccl_exchange_one_rank(tmp_path):
    """The RCCL transport end to end with a communicator of ONE rank (all a one-GPU box allows): library
    look-up, ncclGetUniqueId, ncclCommInitRank, the grouped in-place ncclAllGather pair on the engine's
    stream in every half-step, unshard -- and the chain equals the unsharded one bit for bit."""
    _spawn(_device_chain_worker, (1, _free_port(), str(tmp_path), "one", 2, 300, 16, "rccl1"), 1, tmp_path)
    a, b = np.load(tmp_path / "dev_one_rccl.npz"), np.load(tmp_path / "dev_one_plain.npz")
    assert np.array_equal(a["chain"], b["chain"]) and np.array_equal(a["lnp"], b["lnp"])
    assert len(np.unique(a["chain"][:, :, 0])) > 16
