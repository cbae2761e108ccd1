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
