"""Both models' 32 000-row half-steps (250 light curves x 128 proposals, N = 1e4): each alone, both from two threads on two
contexts, and both PAIRED in one launch (mtg_pair_contexts).  Device time per round of both models = wall time of `reps`
rounds / reps (the launches are asynchronous and back to back).   python scripts/pair_probe.py [L] [W] [reps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
N = 10000
dev = torch.device("cuda", 0)
t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
lc = torch.from_numpy(np.repeat(np.arange(L, dtype=np.int32), W)).to(dev)
engines, thetas, outs, sts, streams = [], [], [], [], []
for i, kinds in enumerate((synth.NULL_MODEL, synth.ALT_MODEL)):
    eng = Engine(0)
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


for i in (0, 1):
    loop(i, 3)
ref = [o.clone() for o in outs]
res = {}
res["null alone"] = min(timed((0,), reps) for _ in range(3))
res["alt alone"] = min(timed((1,), reps) for _ in range(3))
res["both, two contexts"] = min(timed((0, 1), reps) for _ in range(3))
engines[0].pair_with(engines[1])
timed((0, 1), 3)
res["both, paired"] = min(timed((0, 1), reps) for _ in range(3))
stats = engines[0].pair_stats()
solver = engines[0].last_solver
engines[0].unpair()
same = all(torch.equal(a, b) for a, b in zip(ref, outs))
print("rows per model %d (L = %d x W = %d), N = %d; ms per round of launches" % (B, L, W, N))
for k, v in res.items():
    print("  %-22s %.3f ms" % (k, v))
print("  paired launches %s, solver %s, bitwise equal to the unpaired kernels: %s" % (stats, solver, same))
