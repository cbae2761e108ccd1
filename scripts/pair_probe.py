"""Both models' 32 000-row half-steps (250 light curves x 128 proposals, N = 1e4): each alone, both from two threads on two
contexts, and both PAIRED in one launch (mtg_pair_contexts).  Time per round of both models = wall time of `reps` rounds /
reps (the launches are asynchronous and back to back).   python scripts/pair_probe.py [L] [W] [reps]  ->  one JSON line
bench.py calls run() for its `mid_batch_half_step.both_models` entry."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np


def run(L=250, W=128, reps=40, N=10000, device=0):
    import torch
    from mind_the_gaps_amd import synthetic as synth
    from mind_the_gaps_amd.engine import Engine
    dev = torch.device("cuda", device)
    t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
    lc = torch.from_numpy(np.repeat(np.arange(L, dtype=np.int32), W)).to(dev)
    engines, thetas, outs, sts, streams = [], [], [], [], []
    for i, kinds in enumerate((synth.NULL_MODEL, synth.ALT_MODEL)):
        eng = Engine(device)
        full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
        eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
        eng.set_model(kinds, full, free, bounds)
        eng.set_time_parallel(0)
        engines.append(eng)
        thetas.append(torch.from_numpy(synth.draw_thetas(kinds, L * W, seed=20250704 + 40 + i)).to(dev))
        outs.append(torch.empty(L * W, dtype=torch.float64, device=dev))
        sts.append(torch.empty(L * W, dtype=torch.int32, device=dev))
        streams.append(torch.cuda.Stream(dev))
    B = L * W

    def loop(i, n):
        for _ in range(n):
            engines[i].loglike_device(B, thetas[i].data_ptr(), lc.data_ptr(), outs[i].data_ptr(), sts[i].data_ptr(), add_prior=True,
                                      stream=streams[i].cuda_stream)

    def timed(which, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        threads = [threading.Thread(target=loop, args=(i, n)) for i in which]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    try:
        for i in (0, 1):
            loop(i, 3)
        torch.cuda.synchronize(dev)
        ref = [o.clone() for o in outs]
        torch.cuda.synchronize(dev)
        res = {"rows_per_model": B, "N": N,
               "null_alone_ms": min(timed((0,), reps) for _ in range(3)),
               "alt_alone_ms": min(timed((1,), reps) for _ in range(3)),
               "two_contexts_ms": min(timed((0, 1), reps) for _ in range(3))}
        engines[0].pair_with(engines[1])
        timed((0, 1), 3)
        res["paired_ms"] = min(timed((0, 1), reps) for _ in range(3))
        res["paired_launches"] = engines[0].pair_stats()
        res["paired_kernel"] = engines[0].last_solver
        engines[0].unpair()
        torch.cuda.synchronize(dev)
        res["bitwise_equal_to_unpaired"] = bool(all(torch.equal(a, b) for a, b in zip(ref, outs)))
        res["what"] = ("prepare + sort + solve of one half-step of both models (%d light curves x %d proposals each), wall time per "
                       "round over %d back-to-back rounds: each model alone, both from two host threads on two contexts, and both with "
                       "the contexts paired (mtg_pair_contexts: one launch of eight-wave workgroups)" % (L, W, reps))
    finally:
        for eng in engines:
            eng.close()
    return res


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:4]]
    print(json.dumps(run(*args)))
