"""Sharding the (theta x light-curve) batch across the GPUs of one node.

The reference's only cross-worker traffic is ``multiprocessing.Pool.map`` returning one
float per task (/root/reference/mind_the_gaps/gpmodelling.py:245-248) and the
user-level loop over simulated light curves
(docs/notebooks/tutorial_ppp.ipynb:326-343).  Both axes are embarrassingly
parallel, so the MI355X layout is: one process per GPU (``torch.distributed``,
backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU tests), every rank owns a
contiguous block of the work, no collective on the data path, and ONE all-gather
of the log-probabilities (8 bytes per evaluation) when every rank needs the full
vector -- e.g. to run the identical accept/reject step of a walker-sharded
ensemble.  Nothing here computes a likelihood: ``evaluate`` is the engine call.
"""
import numpy as np

__all__ = ["block_bounds", "shard_rows", "shard_lightcurves", "all_gather_rows",
           "sharded_log_prob", "LightcurveShard", "WalkerShardedLogProb", "lockstep", "shard_device_ensemble",
           "broadcast_start", "broadcast_array"]


def block_bounds(n_items, world_size):
    """Contiguous, balanced blocks: rank r owns [b[r], b[r+1])."""
    base, extra = divmod(int(n_items), int(world_size))
    sizes = np.full(world_size, base, dtype=np.int64)
    sizes[:extra] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


def shard_rows(n_rows, rank, world_size):
    """Row range of a theta batch owned by ``rank`` (configs with one replicated
    light curve: the half-ensemble is split evenly across GPUs)."""
    b = block_bounds(n_rows, world_size)
    return int(b[rank]), int(b[rank + 1])


def shard_lightcurves(n_lightcurves, rank, world_size):
    """Light-curve range resident on ``rank`` (the Protassov sweep: independent
    light curves, no communication until the final gather)."""
    return shard_rows(n_lightcurves, rank, world_size)


def all_gather_rows(local, counts, group=None, device=None):
    """All-gather variable-length 1-d float64 (or int32) shards into the full vector.

    ``counts[r]`` is the length of rank r's shard.  Shards are padded to the longest so
    that a single fixed-size collective moves them (RCCL all_gather needs equal sizes)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [int(c) for c in counts]
    assert len(counts) == world and len(local) == counts[dist.get_rank(group)]
    if device is None and dist.get_backend(group) == "nccl":
        # RCCL moves device buffers only: stage on this process's GPU (torch.cuda.set_device)
        device = torch.device("cuda", torch.cuda.current_device())
    width = max(counts) if counts else 0
    dtype = torch.float64 if np.asarray(local).dtype.kind == "f" else torch.int32
    buf = torch.zeros(width, dtype=dtype, device=device)
    if len(local):
        buf[:len(local)] = torch.as_tensor(np.ascontiguousarray(local), dtype=dtype, device=device)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    parts = [g[:c].cpu().numpy() for g, c in zip(gathered, counts)]
    return np.concatenate(parts) if parts else np.empty(0)


def sharded_log_prob(evaluate, theta, lc_index=None, group=None, device=None):
    """Row-sharded evaluation with one all-gather.

    Every rank calls this with the SAME ``theta[B, P]`` (and ``lc_index[B]``); rank r
    evaluates rows ``shard_rows(B, r, world)`` through ``evaluate(theta_rows, lc_rows) ->
    (lnP, status)`` and all ranks return the full ``(lnP[B], status[B])``."""
    import torch.distributed as dist
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    B = theta.shape[0]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return evaluate(theta, lc_index)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = block_bounds(B, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    lc_rows = None if lc_index is None else np.asarray(lc_index)[lo:hi]
    if hi > lo:
        lnp, status = evaluate(theta[lo:hi], lc_rows)
    else:
        lnp, status = np.empty(0), np.empty(0, dtype=np.int32)
    counts = np.diff(bounds)
    return (all_gather_rows(np.asarray(lnp, dtype=np.float64), counts, group, device),
            all_gather_rows(np.asarray(status, dtype=np.int32), counts, group, device).astype(np.int32))


class LightcurveShard:
    """This rank's block of a light-curve set (Protassov sweep, BASELINE configs[3]).

    ``lo, hi`` are the global indices resident here; ``to_local`` maps a global
    ``lc_index`` array of rows that belong to this rank to indices into the local
    ``Y[lo:hi]`` uploaded to this rank's engine; ``gather`` all-gathers a per-light-curve
    result (e.g. max lnL per light curve, gpmodelling.py:431-436) onto every rank."""

    def __init__(self, n_lightcurves, rank=None, world_size=None, group=None):
        import torch.distributed as dist
        if rank is None or world_size is None:
            if dist.is_available() and dist.is_initialized():
                rank, world_size = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world_size = 0, 1
        self.n, self.rank, self.world, self.group = int(n_lightcurves), int(rank), int(world_size), group
        self.bounds = block_bounds(self.n, self.world)
        self.lo, self.hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])

    def __len__(self):
        return self.hi - self.lo

    def owns(self, lc_index):
        lc_index = np.asarray(lc_index)
        return (lc_index >= self.lo) & (lc_index < self.hi)

    def to_local(self, lc_index):
        lc_index = np.asarray(lc_index)
        if np.any(~self.owns(lc_index)):
            raise ValueError("light curve outside this rank's shard [%d, %d)" % (self.lo, self.hi))
        return (lc_index - self.lo).astype(np.int32)

    def gather(self, per_lightcurve_values, device=None):
        values = np.asarray(per_lightcurve_values, dtype=np.float64)
        if values.shape != (len(self),):
            raise ValueError("expected one value per local light curve")
        if self.world == 1:
            return values
        return all_gather_rows(values, np.diff(self.bounds), self.group, device)


class WalkerShardedLogProb:
    """``log_prob_fn`` of a walker-sharded ensemble (one replicated light curve, BASELINE
    configs[1], [2], [4]): every rank holds the light curve, is handed
    the SAME half-ensemble ``coords[B, P]`` by its own copy of the sampler, evaluates rows
    ``shard_rows(B, rank, world)`` on its GPU and all-gathers the log-probabilities (8 bytes per
    walker over RCCL), so that every rank takes the identical accept/reject decisions.

    ``log_prob_fn(coords[b, P]) -> lnP[b]`` is the local engine call (e.g.
    ``GPModelling._log_probability``).  Exceptions it raises (a non positive-definite
    covariance in strict mode) are raised on EVERY rank after the collective, so that no rank
    is left waiting in the all-gather."""

    def __init__(self, log_prob_fn, group=None, device=None):
        self.log_prob_fn, self.group, self.device = log_prob_fn, group, device

    def __call__(self, coords):
        failure = []

        def evaluate(theta, lc_rows):
            try:
                lnp = np.asarray(self.log_prob_fn(theta), dtype=np.float64)
                return lnp, np.zeros(len(lnp), dtype=np.int32)
            except Exception as exc:  # re-raised below, after the collective
                failure.append(exc)
                return np.full(len(theta), np.nan), np.ones(len(theta), dtype=np.int32)

        lnp, status = sharded_log_prob(evaluate, coords, None, self.group, self.device)
        if failure:
            raise failure[0]
        if np.any(status != 0):
            raise RuntimeError("the log-probability failed on another rank of the walker-sharded ensemble")
        return lnp


def lockstep(sampler, p0, group=None):
    """Put the ranks' copies of a host-side ``EnsembleSampler`` in lock-step: rank 0's random
    state and starting ensemble go to every rank (the reference seeds neither; emcee copies
    numpy's global state, which differs between processes).  Returns the common ``p0``."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(p0, dtype=np.float64)
    objs = [sampler.random_state, np.asarray(p0, dtype=np.float64)] if dist.get_rank(group) == 0 else [None, None]
    dist.broadcast_object_list(objs, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                               device=_object_device(group))
    sampler.random_state = objs[0]
    return objs[1]


def broadcast_start(p0, seed, group=None):
    """Rank 0's starting ensemble and Philox seed to every rank (the reference seeds nothing; each process
    would draw its own).  Returns the common ``(p0, seed)``."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(p0, dtype=np.float64), int(seed)
    objs = [np.asarray(p0, dtype=np.float64), int(seed)] if dist.get_rank(group) == 0 else [None, None]
    dist.broadcast_object_list(objs, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                               device=_object_device(group))
    return objs[0], objs[1]


def broadcast_array(values, group=None, src=0):
    """Rank ``src``'s (default 0's) float64 array to every rank (same shape everywhere): decisions that must not differ
    between the ranks of a walker-sharded run -- the autocorrelation times behind the convergence test -- are taken from it."""
    import torch
    import torch.distributed as dist
    values = np.ascontiguousarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return values
    buf = torch.from_numpy(values.copy())
    dev = _object_device(group)
    if dev is not None:
        buf = buf.to(dev)
    dist.broadcast(buf, src=dist.get_global_rank(group, int(src)) if group is not None else int(src), group=group)
    return buf.cpu().numpy().reshape(values.shape)


def _object_device(group):
    """Where broadcast_object_list stages its pickles: RCCL moves device buffers only."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else None


def shard_device_ensemble(engine, group=None, transport=None):
    """Shard the engine's resident ensembles (``mtg_ensemble_init`` done, same coordinates and seed on every
    rank) across the ranks of ``group``: each evaluates its block of every half-step's proposals.

    ``transport`` "rccl": one grouped ``ncclAllGather`` per half-step on the engine's stream, no host round trip
    -- the library makes its own communicator from a ``ncclUniqueId`` that rank 0 creates and this function
    broadcasts over ``group``.  "host": the library stages this rank's rows on the host and the exchange runs
    over ``group``'s own backend (gloo in the CPU-side tests; also the way for processes that share one GPU,
    which RCCL refuses).  Default: "rccl" when ``group``'s backend is nccl, else "host"."""
    import os
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return "none"
    # MTG_SHARD_ONE_RANK=1 (rehearsals on a one-GPU box): a group of ONE rank still goes through the transport -- unique
    # id, broadcast, ncclCommInitRank, the all-gather pair of every half-step
    if dist.get_world_size(group) == 1 and os.environ.get("MTG_SHARD_ONE_RANK") != "1":
        return "none"
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if transport is None:
        transport = "rccl" if dist.get_backend(group) == "nccl" else "host"
    if transport == "rccl":
        # Bringing the library's own communicator up is the one step no one-GPU box can rehearse with more than one
        # rank.  A rank on which it fails (or does not come back within MTG_RCCL_INIT_TIMEOUT_S, default 120 s) says so
        # over ``group``; if ANY rank failed, EVERY rank drops the communicator and takes the host-staged exchange over
        # ``group`` instead -- same chains (the exchange moves the same bytes), and the returned string says why.
        why = _try_rccl(engine, group, rank, world)
        flags = [None] * world
        dist.all_gather_object(flags, why, group=group)
        failed = [(r, w) for r, w in enumerate(flags) if w]
        if not failed:
            return "rccl"
        if why is None:
            engine.ensemble_unshard()
        import warnings
        reason = "rank %d: %s" % failed[0]
        warnings.warn("walker sharding: RCCL communicator not available (%s); host-staged exchange instead" % reason)
        _host_transport(engine, group, rank, world)
        return "host (rccl failed: %s)" % reason
    if transport != "host":
        raise ValueError("transport must be 'rccl' or 'host'")

    _host_transport(engine, group, rank, world)
    return transport


def _try_rccl(engine, group, rank, world):
    """None when this rank's communicator is up, else the reason it is not (never raises: the verdict is agreed on by
    all ranks afterwards)."""
    import os
    import threading
    import torch.distributed as dist
    if os.environ.get("MTG_SHARD_FAIL_RCCL") in ("all", str(rank)):      # rehearsal of the fall-back (bench.py, tests)
        return "injected failure (MTG_SHARD_FAIL_RCCL)"
    try:
        objs = [engine.rccl_unique_id() if rank == 0 else None]
    except Exception as exc:        # (rank 0 could not even load librccl: the others must not wait for an id)
        objs = [exc]
    dist.broadcast_object_list(objs, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                               device=_object_device(group))
    if isinstance(objs[0], Exception) or objs[0] is None:
        return "no ncclUniqueId from rank 0 (%r)" % (objs[0],)
    verdict = []

    def init():
        try:
            engine.ensemble_shard_rccl(objs[0], rank, world)
            verdict.append(None)
        except Exception as exc:
            verdict.append("%s: %s" % (type(exc).__name__, exc))

    limit = float(os.environ.get("MTG_RCCL_INIT_TIMEOUT_S", "120"))
    worker = threading.Thread(target=init, daemon=True)
    worker.start()
    worker.join(limit)
    if worker.is_alive():
        return "ncclCommInitRank did not return within %.0f s" % limit
    return verdict[0]


def _host_transport(engine, group, rank, world):
    import torch
    import torch.distributed as dist

    def exchange(lnp, status, lo, hi):
        # the library's layout: rank r owns rows [r * chunk, (r + 1) * chunk), chunk = ceil(count / world)
        count = len(lnp)
        chunk = -(-count // world)
        mine = torch.zeros(2 * chunk, dtype=torch.float64)
        mine[:hi - lo] = torch.from_numpy(lnp[lo:hi])
        mine[chunk:chunk + hi - lo] = torch.from_numpy(status[lo:hi].astype(np.float64))
        dev = _object_device(group)      # RCCL moves device buffers only: stage through the card when the group is nccl
        if dev is not None:
            mine = mine.to(dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        for r, part in enumerate(parts):
            part = part.cpu()
            a, b = min(r * chunk, count), min((r + 1) * chunk, count)
            lnp[a:b] = part[:b - a].numpy()
            status[a:b] = part[chunk:chunk + b - a].numpy().astype(np.int32)

    engine.ensemble_shard_host(rank, world, exchange)
