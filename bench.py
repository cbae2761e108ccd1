#!/usr/bin/env python3
"""bench.py -- celerite log-likelihood evaluations/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: BASELINE.json configs[3], the Protassov posterior-predictive sweep -- L = 2000
simulated light curves sharing one irregular sampling (N = 10 000), W = 256 walkers each,
alternative model DRW + SHO + Lorentzian (celerite rank J = 6, P = 8 free parameters).  One step =
one full-ensemble sweep = L x W = 512 000 log-probability evaluations through
mtg_loglike_batch_device (prior + coefficients + fused Cholesky/solve), with t, y, sigma^2 and theta
already resident in HBM.

Scaling (--scaling, default "strong" for --gpus N > 1, SURVEY.md 8(d) Config 4): the 2000 light
curves are cut into contiguous blocks, one per rank (distributed.shard_lightcurves); every rank
sweeps its own block -- no collective on the data path -- and each step ends with the only exchange
the path has: the all-gather of the per-light-curve maxima of lnP (RCCL over xGMI), inside the timed
region.  "weak" gives every rank its own 2000 light curves instead.

The JSON line also carries
  roofline        : algorithmic bytes (24 N + 8 P + 12 per evaluation) / mean
                    duration of the dominant kernel (mtg_solve_kernel<1,2,1>), HIP events on the launch
                    stream; the binding resource is FP64 vector issue (bound: "fp64_valu");
  end_to_end      : the same sweep through the host-pointer entry point (H2D theta, kernels, D2H);
  strong_shard_8  : one GPU on the share it gets of the 2000 light curves at 8 GPUs (250);
  cpu_baseline    : oracle/celerite_ref.c (a plain-C port of celerite's algorithm, fused one-sweep
                    variant, built -O3 -march=native on this host) single thread and on all usable
                    cores, bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# FP64 operations per sample and lane of mtg_solve_kernel<1,2,1> (91 fma + 53 mul/add in kernel v8;
# `scripts/loop_stats.py 1 2 1` counts the compiled loop: 182 fma + 106 mul/add per two steps)
FLOP_PER_SAMPLE = 2 * 91 + 53


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=10000, help="samples per light curve")
    ap.add_argument("--lightcurves", type=int, default=2000)
    ap.add_argument("--walkers", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="time box of the CPU baseline sample (0 = skip)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="default: strong when --gpus > 1 (the light curves are split over the ranks)")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 hardware threads but grants a 16-CPU quota)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(t, y, dy, kinds, theta, y_mean, seconds, bounds=None, gpu_out=None, gpu_status=None):
    """oracle/celerite_ref.c on the host cores, same inputs, bounded sample: the fused one-sweep
    recurrence built -O3 -march=native for this host, on all usable cores (`value`)
    and on one thread.  The values it produces are the first evaluations of the timed GPU batch, so they
    double as a parity check of that batch (outside both timed regions): the worst relative difference
    goes into the line."""
    import tempfile
    from oracle import celerite as oracle_c
    cores = usable_cores()
    oracle_c.lib()
    native = oracle_c.build_native(tempfile.mkdtemp(prefix="mtg_bench_"))
    build = "gcc -O3 -march=native" if native is not None else "gcc -O2 (prebuilt; no compiler on this host)"
    L = y.shape[0]

    def run(idx, nthreads):
        lc = (idx // (theta.shape[0] // L)).astype(np.int32) % L
        full = np.hstack([theta[idx], y_mean[lc][:, None]])
        t0 = time.perf_counter()
        ref, rstatus = oracle_c.logprob_batch(t, y, dy, kinds, full, bounds=bounds, lc_index=lc, add_prior=bounds is not None,
                                              nthreads=nthreads, fused=True, handle=native)
        return ref, rstatus, time.perf_counter() - t0

    # one thread: a quarter of the time box
    done1, spent1 = 0, 0.0
    while spent1 < 0.25 * seconds:
        _, _, dt = run((np.arange(16) + done1) % theta.shape[0], 1)
        done1 += 16; spent1 += dt
    chunk = max(cores * 32, 64)
    done, spent, worst, compared = 0, 0.0, 0.0, 0
    while True:
        idx = (np.arange(chunk) + done) % theta.shape[0]
        ref, rstatus, dt = run(idx, cores)
        spent += dt
        if gpu_out is not None and done + chunk <= theta.shape[0]:
            ok = (rstatus == 0) & (gpu_status[idx] == 0)
            if not np.array_equal(rstatus == 0, gpu_status[idx] == 0):
                raise SystemExit("bench: GPU and CPU port disagree on which evaluations are valid")
            worst = max(worst, float(np.max(np.abs(gpu_out[idx][ok] - ref[ok]) / np.abs(ref[ok]))))
            compared += int(ok.sum())
        done += chunk
        if spent >= 0.75 * seconds:
            break
    if compared and worst > 1e-8:
        raise SystemExit("bench: GPU batch differs from the CPU port by %.3e relative (> 1e-8)" % worst)
    return {"value": done / spent, "unit": "evals/s", "cores": cores, "kind": "port",
            "single_thread": done1 / spent1,
            "sample": "%d evaluations of the same workload (N=%d) in %.1f s on %d threads + %d in %.1f s on one, "
                      "oracle/celerite_ref.c, fused one-sweep recurrence, %s; the port carries the Lorentzian's null "
                      "real term as celerite would (J = 6), the GPU sweep drops it (J = 5 of arithmetic)"
                      % (done, len(t), spent, cores, done1, spent1, build),
            "max_rel_diff_vs_gpu": worst if compared else None, "compared": compared}


def single_lightcurve_configs():
    """BASELINE configs[0], [1], [2] and [4] (ONE light curve each): stretch-move iterations/s through
    GPModelling.derive_posteriors with the device-resident sampler; such small batches take the
    time-parallel kernels.  Reported next to the headline, not part of `value`."""
    import warnings
    from mind_the_gaps_amd import synthetic as synth, terms
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
    amp, other = (-10, 50), (-10, 10)
    th = synth.truth(synth.ALT_MODEL)

    def drw():
        return DampedRandomWalk(th[0], th[1], bounds=[amp, other])

    def null_kernel():
        return drw() + terms.SHOTerm(th[2], th[3], th[4], bounds=[amp, other, other])

    def five_sho():
        k = None
        for i in range(5):
            term = terms.SHOTerm(np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]),
                                 np.log(2 * np.pi / (5.0 + 6 * i)), bounds=[amp, other, other])
            k = term if k is None else k + term
        return k

    out = {}
    cases = (("configs[0] DRW N=1e3 32 walkers", drw, 1000, 32, 8000, 8),
             ("configs[1] DRW+SHO N=1e4 128 walkers", null_kernel, 10000, 128, 4000, 11),
             ("configs[2] DRW+SHO+Lorentzian N=1e4 256 walkers",
              lambda: null_kernel() + Lorentzian(th[5], th[6], th[7], bounds=[amp, other, other]), 10000, 256, 4000, 14),
             ("configs[4] 5 x SHO (J=10) N=2e5 512 walkers", five_sho, 200000, 512, 40, 21))
    for name, make_kernel, n, walkers, steps, P in cases:
        t, y, dy = synth.make_lightcurves(n, 1, seed=20250704 + 2)
        g = GPModelling(GappyLightcurve(t, y[0], dy[0]), make_kernel())
        np.random.seed(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g.derive_posteriors(fit=False, max_steps=min(10, steps), convergence_steps=10, walkers=walkers,
                                progress=False, device_sampler=True)
            t0 = time.perf_counter()
            g.derive_posteriors(fit=False, max_steps=steps, convergence_steps=steps, walkers=walkers,
                                progress=False, device_sampler=True)
            el = time.perf_counter() - t0
        evals = steps * walkers / el
        out[name] = {"iterations_per_s": steps / el, "evals_per_s": evals, "iterations_timed": steps,
                     "algorithmic_hbm_frac": evals * (24 * n + 8 * (P - 6) + 12) / (HBM_PEAK_GBS * 1e9)}
        if n == 200000:
            # what walker sharding over 8 GPUs can gain on this chain: device time of one half-step's likelihoods for
            # the whole half-ensemble (256 rows) and for one rank's share of it (32 rows), same engine, same model
            from mind_the_gaps_amd.gp import get_engine
            eng = get_engine(0)
            theta = np.asarray(g.sampler.get_chain()[-1], dtype=np.float64)
            ms = {}
            for rows in (256, 32):
                best = np.inf
                for _ in range(3):
                    eng.loglike(theta[:rows])
                    best = min(best, eng.last_kernel_ms)
                ms[rows] = best
            out[name]["walker_shard_8"] = {
                "half_step_ms_256_rows": ms[256], "half_step_ms_32_rows": ms[32], "speedup": ms[256] / ms[32],
                "what": "likelihoods of one half-step on one MI355X: the whole half-ensemble against the 32 rows a rank "
                        "evaluates when the walkers are sharded over 8 GPUs (mtg_ensemble_shard_rccl); the exchange is one "
                        "all-gather of 256 doubles + status words per half-step"}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # Rehearsal only: more ranks than GPUs (e.g. two ranks on a one-GPU box) share the cards and agree on
    # the timing over gloo -- RCCL refuses two ranks on one device.  The line then says "oversubscribed".
    ndev = torch.cuda.device_count()
    oversubscribed = world > ndev
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # MTG_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box): a process group of ONE rank over RCCL, so that every
    # collective of the multi-GPU path -- initialisation, all-gather, barrier, all-reduce -- runs as it will on a node
    grouped = world > 1 or os.environ.get("MTG_BENCH_FORCE_DIST") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if oversubscribed:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    scaling = args.scaling or ("strong" if grouped else "weak")

    from mind_the_gaps_amd import synthetic as synth
    from mind_the_gaps_amd.distributed import block_bounds
    from mind_the_gaps_amd.engine import Engine

    N, L_total, W = args.n, args.lightcurves, args.walkers
    kinds = synth.ALT_MODEL
    P = len(synth.truth(kinds))
    # seed = 20250704 + config index.  strong: one set of light curves, rank r owns a
    # contiguous block of it; weak: every rank owns its own set of L_total
    if scaling == "strong":
        t, y_all, dy_all = synth.make_lightcurves(N, L_total, seed=20250704 + 4)
        theta_all = synth.draw_thetas(kinds, L_total * W, seed=20250704 + 40)
        blocks = block_bounds(L_total, world)
        l0, l1 = int(blocks[rank]), int(blocks[rank + 1])
        y, dy, theta = y_all[l0:l1], dy_all[l0:l1], theta_all[l0 * W:l1 * W]
        L_pad = int(np.max(np.diff(blocks)))
        del y_all, dy_all, theta_all
    else:
        t, y, dy = synth.make_lightcurves(N, L_total, seed=20250704 + 4 + 1000 * rank)
        theta = synth.draw_thetas(kinds, L_total * W, seed=20250704 + 40 + 1000 * rank)
        L_pad = L_total
    L = y.shape[0]
    B = L * W
    B_job = (L_total if scaling == "strong" else world * L_total) * W   # evaluations of the whole job per step
    # every light curve keeps its own frozen mean, as GPModelling does (gpmodelling.py:83-87)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    y_mean = y.mean(axis=1)
    lc = np.repeat(np.arange(L, dtype=np.int32), W)

    eng = Engine(local_dev)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean)   # uploaded once; resident for the whole run
    eng.set_model(kinds, full, free, bounds)
    d_theta = torch.from_numpy(theta).to(dev)
    d_lc = torch.from_numpy(lc).to(dev)
    d_out = torch.empty(B, dtype=torch.float64, device=dev)
    d_status = torch.empty(B, dtype=torch.int32, device=dev)
    # the exchange step of the sharded sweep: per-light-curve maxima of lnP, all-gathered
    d_best = torch.full((L_pad,), -np.inf, dtype=torch.float64, device=dev)
    gdev = "cpu" if oversubscribed else dev
    d_gather = torch.empty(world * L_pad, dtype=torch.float64, device=gdev) if grouped else None
    stream = torch.cuda.current_stream(dev)

    def sweep(n_eval=B):
        eng.loglike_device(n_eval, d_theta.data_ptr(), d_lc.data_ptr(), d_out.data_ptr(),
                           d_status.data_ptr(), add_prior=True, stream=stream.cuda_stream)

    def step():
        sweep()
        if grouped:
            torch.amax(d_out.view(L, W), dim=1, out=d_best[:L])
            dist.all_gather_into_tensor(d_gather, d_best.cpu() if oversubscribed else d_best)

    def fence():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prep_ms, solve_ms = eng.profile_read()

    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if scaling == "strong":  # the gathered maxima are those of the whole set, in order, on every rank
            got = d_gather.cpu().numpy().reshape(world, L_pad)
            mine = d_out.cpu().numpy().reshape(L, W).max(axis=1)
            if not np.array_equal(got[rank, :L], mine):
                raise SystemExit("bench: gathered maxima differ from the local ones")

    # sanity: every evaluation finite or prior-rejected, and a spot check of the values
    status = d_status.cpu().numpy()
    out = d_out.cpu().numpy()
    n_ok = int((status == 0).sum())
    if not np.all(np.isfinite(out[status == 0])) or n_ok < B // 2:
        raise SystemExit("bench: non-finite log-likelihoods in the timed batch")

    extras = {}
    if world == 1 and not args.no_extras:
        # (a) the same sweep through the host-pointer entry point: H2D theta + kernels + D2H lnP, status
        reps = max(3, min(args.steps, 10))
        eng.loglike(theta, lc, add_prior=True)
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.loglike(theta, lc, add_prior=True)
        e2e = (time.perf_counter() - t1) / reps
        extras["end_to_end"] = {"value": B / e2e, "unit": "evals/s", "ms_per_step": e2e * 1e3,
                                "what": "mtg_loglike_batch: pageable host theta [B][P] + light-curve index up, kernels, "
                                        "lnP + status down, every step (%d MB + %d MB over PCIe)"
                                        % (theta.nbytes // 2**20 + lc.nbytes // 2**20, (B * 12) // 2**20)}
        # (b) the share one GPU gets of the 2000 light curves at 8 GPUs: is a 1/8 batch still efficient?
        L8 = max(1, L // 8)
        B8 = L8 * W
        for _ in range(3):
            sweep(B8)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(4 * reps):
            sweep(B8)
        torch.cuda.synchronize(dev)
        dt8 = (time.perf_counter() - t1) / (4 * reps)
        extras["strong_shard_8"] = {"lightcurves": L8, "evals_per_step": B8, "ms_per_step": dt8 * 1e3,
                                    "evals_per_s": B8 / dt8,
                                    "per_gpu_factor": (B8 / dt8) / (B * args.steps / elapsed),
                                    "what": "one MI355X sweeping 1/8 of the light curves (its share at 8 GPUs, ~1 wave per "
                                            "SIMD); 8 x per_gpu_factor is the strong-scaling speed-up the kernels allow "
                                            "before the all-gather of 2000 doubles"}

    if rank == 0:
        bytes_eval = 24 * N + 8 * P + 12
        solve_s = float(np.mean(solve_ms)) * 1e-3
        achieved = n_ok * bytes_eval / solve_s / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("N") == N and rec.get("B") == B:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/bench_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command (%s), not measured in this run" % rec.get("round", "round 1")
            except Exception:
                traffic = None
        line = {
            "metric": "celerite log-likelihood evals/sec (N=1e4, J=6)",
            "value": B_job * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[3]: Protassov PPP sweep, %d light curves x %d walkers = %d evals/step%s, "
                            "alt model DRW+SHO+Lorentzian (J=6, P=%d), N=%d irregular samples"
                            % (L_total, W, L_total * W, " per GPU" if scaling == "weak" and world > 1 else "", P, N),
                "N": N, "J": 6, "J_arith": 5, "P": P, "lightcurves": L_total if scaling == "strong" else world * L_total,
                "lightcurves_per_gpu": L, "walkers": W, "evals_per_step": B_job, "evals_per_step_per_gpu": B,
                "sharding": ("light curves split over the ranks (contiguous blocks), no data-path collective; each step "
                             "ends with the all-gather of the per-light-curve maxima of lnP (%d doubles per rank)" % L_pad)
                            if world > 1 else "one GPU",
            },
            "roofline": {
                # nominal roofline of SURVEY.md 8(d): algorithmic bytes over HBM peak (achieved / peak / frac);
                # what binds the kernel is FP64 vector issue (fp64_valu below, profiles/): 256 walkers share a
                # light curve through L2 / MALL and the real HBM traffic is ~1 % of the algorithmic bytes
                "bound": "fp64_valu",
                "kernel": "mtg_solve_kernel<1,2,1>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "bytes_per_eval": bytes_eval,
                "evals_per_launch": n_ok,
                "kernel_ms": solve_s * 1e3,
                "prepare_kernel_ms": float(np.mean(prep_ms)),
                "J_arith": 5,   # the Lorentzian's null real term (a = 0, c = 0) never enters D_n or z_n: not expanded
                # FP64 vector work of the J = 6 sweep (scripts/loop_stats.py on the sweep loop) against the
                # 78.6 TFLOP/s FP64 vector peak
                "fp64_valu": {"flop_per_sample": FLOP_PER_SAMPLE,
                              "achieved_tflops": n_ok * N * FLOP_PER_SAMPLE / solve_s / 1e12,
                              "peak_tflops": 78.6,
                              "frac": n_ok * N * FLOP_PER_SAMPLE / solve_s / 1e12 / 78.6},
            },
        }
        line.update(extras)
        if oversubscribed:
            line["oversubscribed"] = "%d ranks on %d GPU(s): rehearsal of the multi-rank path, not a scaling number" % (world, ndev)
        if world == 1 and args.cpu_seconds > 0 and not args.no_extras:
            line["cpu_baseline"] = cpu_baseline(t, y, dy, kinds, theta, y_mean, args.cpu_seconds, bounds, out, status)
            try:
                line["other_configs"] = single_lightcurve_configs()
            except Exception as exc:  # never let the side measurements break the headline line
                line["other_configs"] = {"error": repr(exc)}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    eng.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
