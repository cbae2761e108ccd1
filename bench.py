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

Launching.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF:
the parent spawns N copies of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a free port
on 127.0.0.1) before it has touched a GPU, passes rank 0's JSON line through and exits with the
worst child's code.  Under torch.distributed.run the ranks are already there and nothing is spawned.
More ranks than GPUs (a one-GPU box: at most 6, the box's process guard) rehearse the multi-rank path
with the ranks sharing the card and gloo as the transport -- RCCL refuses two ranks on one device;
the line then says "oversubscribed".

Scaling (--scaling, default "strong" for --gpus N > 1, SURVEY.md 8(d) Config 4): the 2000 light
curves are cut into contiguous blocks, one per rank (distributed.shard_lightcurves); every rank
sweeps its own block -- no collective on the data path -- and each step ends with the only exchange
the path has: the all-gather of the per-light-curve maxima of lnP (RCCL over xGMI), inside the timed
region.  "weak" gives every rank its own 2000 light curves instead.

The JSON line also carries
  roofline        : algorithmic bytes (24 N + 8 P + 12 per evaluation) / mean
                    duration of the dominant kernel (named by the library: mtg_last_solver), HIP events
                    on the launch stream; "bound": "hbm" (the contract's nominal roofline), "binds": "fp64_valu" (FP64
                    vector issue, what really limits it; the sub-object of that name prices it against the FP64 peak;
                    fp64_issue_frac_at_clock = the sweep's issue floor at the sclk the card holds under this load /
                    its measured time: the fraction that says how much headroom the kernel has);
  walker_sharded  : (N > 1) configs[2] and configs[4] -- ONE light curve, 256 / 512 walkers -- through
                    GPModelling.derive_posteriors(device_sampler=True, shard_walkers=True): every
                    half-step's proposals split over the ranks, ncclAllGather pair on the launch stream
                    (mtg_ensemble_shard_rccl): iterations/s, rows per rank, half-step ms, exchange us;
  end_to_end      : (N = 1) the same sweep through the host-pointer entry point (H2D, kernels, D2H);
  clock_under_load: (N = 1) sclk and socket power from rocm-smi while sweeps are queued (the FP64 peak is quoted at 2.4 GHz);
  strong_shard_8  : (N = 1) one GPU on the share it gets of the 2000 light curves at 8 GPUs (250);
  null_model_sweep, config2_raw_kernel, other_configs (configs[0], [1], [2] + T_LRT, [4]),
  workflow_config3_share_of_8_lognormal: (N = 1) one GPU's share of the Protassov test with a LOGNORMAL flux PDF: every simulated
                    segment through the E13 amplitude / rank adjustment on the device (csrc/mtg_e13.hip) -- the loop never
                    leaves the GPU (its `seconds.simulate` against `simulate_s_gaussian`);
  workflow_config3: (N = 1) the other SURVEY 8(d) figures: the null model's sweep beside the alternative's,
                    the raw kernel at B = 65 536 for DRW+SHO (J = 3 and zero-padded J = 4), the
                    single-light-curve chains, configs[3] as a whole workflow (scripts/config3_probe.py);
  workflow_config3_sharded: (N > 1) ppp.protassov_test(sharded=True): every rank simulates and refits its block
                    of the 2000 light curves, one all-gather of the maxima per model; time = max over ranks;
                    (both N > 1 extras run after the timed region under --extras-timeout: if one of them fails or
                    hangs on some rank, rank 0 still prints the line -- the scaling number stands -- with
                    `multi_rank_extras_error` saying what happened (exit code 3 with --strict-extras, else 0: the
                    first multi-GPU bring-up costs a field, not the record); MTG_BENCH_FAIL_EXTRAS=<rank> rehearses that)
  cpu_baseline    : oracle/celerite_ref.c (a plain-C port of celerite's algorithm, fused one-sweep
                    variant, built -O3 -march=native on this host) single thread and on all usable
                    cores, bounded sample (rank 0; at N > 1 on rank 0's block while the other ranks wait at a barrier
                    after the timed region);
  per_rank, exchange_us: (N > 1) every rank's kernel_ms, busy time per step and the time of the all-gather (events on
                    the launch stream), min / max / mean and by rank; roofline is then the SLOWEST rank's kernel.
A side measurement that fails fails the bench (non-zero exit, after the line where there is a timed region to report).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before anything initialises HIP (torch.cuda.* below does): eight hardware queues instead of four, so that the contexts
# that work side by side in the workflow (two models' refits) do not end up sharing one (mind_the_gaps_amd/engine.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6
# FP64 operations per sample and lane of the serial sweep's inner loop, mean-free variant (`scripts/loop_stats.py NR NC
# NB0` on the compiled loop: 2 x fma + mul/add per step)
FLOP_PER_SAMPLE = {"mtg_solve_kernel<1,2,1>": 2 * 91 + 53, "mtg_solve_kernel<1,1,0>": 2 * 47 + 28,
                   "mtg_solve_kernel<2,1,0>": 2 * 66 + 39, "mtg_solve_kernel<0,3,0>": 2 * 123 + 68,
                   # every structure of the model in one launch (csrc/mtg_kernels_multi.hip): the bench's rows are all
                   # under-damped, i.e. the same sweep loop as the one-structure kernel's
                   "mtg_solve_kernel_multi<1,2,2,1>": 2 * 91 + 53, "mtg_solve_kernel_multi<1,1,2,0>": 2 * 47 + 28}
# vector instructions per sample and lane in the serial sweep's loop (scripts/loop_stats.py NR NC NB0: VALU of the two-step
# trip / 2): what the kernel's time really scales with -- every FP64 opcode takes a SIMD's issue slot for 4 cycles per wave64
VALU_PER_SAMPLE = {"mtg_solve_kernel<1,2,1>": 166, "mtg_solve_kernel_multi<1,2,2,1>": 166, "mtg_solve_kernel<1,1,0>": 91,
                   "mtg_solve_kernel_multi<1,1,2,0>": 91, "mtg_solve_kernel<0,3,0>": 215, "mtg_solve_kernel<2,1,0>": 125}
SIMDS = 1024            # 256 CUs x 4
MAX_RANKS_PER_GPU = 6   # the GPU box's process guard


def fp64_issue(kernel, waves, n_samples, kernel_s, clock):
    """The sweep against its own issue floor AT THE CLOCK THE CARD HOLDS: waves x samples x VALU instructions x 4 cycles,
    spread over every SIMD, at the sclk rocm-smi reports under this load -- the bound that matters (the nominal HBM
    roofline of the contract counts bytes that 256 walkers share through L2; the FP64 peak is quoted at 2.4 GHz, which
    the card does not hold at its power limit).  None when the kernel's loop has not been counted or no clock was read."""
    valu = VALU_PER_SAMPLE.get(kernel)
    if not valu or not clock or not clock.get("sclk_mhz"):
        return None
    floor_s = waves * n_samples * valu * 4.0 / (SIMDS * clock["sclk_mhz"] * 1e6)
    return {"frac": floor_s / kernel_s, "floor_ms": floor_s * 1e3, "valu_per_sample_and_lane": valu, "cycles_per_instruction": 4,
            "simds": SIMDS, "sclk_mhz": clock["sclk_mhz"],
            "what": "waves x N x VALU instructions of the sweep loop x 4 cycles / (1024 SIMDs x sclk under load) / kernel time"}


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
    ap.add_argument("--no-workflow", action="store_true", help="skip configs[3] as a workflow (~30 s)")
    ap.add_argument("--extras-timeout", type=int, default=300,
                    help="N > 1: seconds the walker-sharded configs and the sharded workflow may take after the timed "
                         "region before the line is printed without them")
    ap.add_argument("--strict-extras", action="store_true",
                    help="N > 1: exit with code 3 when the extras after the timed region fail (default: the failure is a "
                         "field of the line, `multi_rank_extras_error`, and a message on stderr; the exit code stays 0 so "
                         "that a measured scaling number is never discarded for a side measurement)")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: N children, one rank each.  The parent imports no torch and makes no
    HIP / HSA call at all -- `--gpus` is taken as given, the children are fresh processes (never a re-exec of one that
    touched a GPU) and each of them checks the devices it sees and refuses loudly before the process group exists."""
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    children = []
    for rank in range(args.gpus):
        renv = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=renv,
                                         stdout=None if rank == 0 else subprocess.DEVNULL))
    worst = 0
    pending = set(range(args.gpus))
    while pending:
        for rank in sorted(pending):
            rc = children[rank].poll()
            if rc is None:
                continue
            pending.discard(rank)
            if rc != 0:
                worst = worst or rc
                for other in pending:          # a rank that died leaves the others in a collective: stop them
                    children[other].terminate()
        time.sleep(0.05)
    raise SystemExit(worst)


def clock_under_load(queue_work, wait):
    """{"sclk_mhz", "socket_power_w", "fp64_peak_tflops_at_sclk"} read through rocm-smi while `queue_work()` keeps the GPU
    busy; None when rocm-smi is missing, slow or says something this does not understand (a side reading, never fatal)."""
    import re
    import shutil
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    queue_work()
    try:
        out = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        out = ""
    finally:
        wait()
    sclk = re.search(r"sclk clock level: *\d+: *\((\d+)Mhz\)", out)
    power = re.search(r"Power \(W\): *([0-9.]+)", out)
    if not sclk:
        return None
    mhz = float(sclk.group(1))
    return {"sclk_mhz": mhz, "socket_power_w": float(power.group(1)) if power else None,
            "fp64_peak_tflops_at_sclk": FP64_PEAK_TFLOPS * mhz / 2400.0,
            "what": "rocm-smi --showclocks --showpower while 40 sweeps are queued; the nominal FP64 vector peak is quoted at 2400 MHz"}


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


# ---- kernels of the single-light-curve configs (same synthetic data: seed 20250704 + 2) ----------------------------
AMP, OTHER = (-10, 50), (-10, 10)


def _kernels():
    from mind_the_gaps_amd import synthetic as synth, terms
    from mind_the_gaps_amd.models import DampedRandomWalk, Lorentzian
    th = synth.truth(synth.ALT_MODEL)

    def drw():
        return DampedRandomWalk(th[0], th[1], bounds=[AMP, OTHER])

    def null_kernel():
        return drw() + terms.SHOTerm(th[2], th[3], th[4], bounds=[AMP, OTHER, OTHER])

    def alt_kernel():
        return null_kernel() + Lorentzian(th[5], th[6], th[7], bounds=[AMP, OTHER, OTHER])

    def five_sho():
        k = None
        for i in range(5):
            term = terms.SHOTerm(np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]),
                                 np.log(2 * np.pi / (5.0 + 6 * i)), bounds=[AMP, OTHER, OTHER])
            k = term if k is None else k + term
        return k
    return drw, null_kernel, alt_kernel, five_sho


def _gpmodel(make_kernel, n, device=0):
    from mind_the_gaps_amd import synthetic as synth
    from mind_the_gaps_amd.gpmodelling import GPModelling
    from mind_the_gaps_amd.lightcurves import GappyLightcurve
    t, y, dy = synth.make_lightcurves(n, 1, seed=20250704 + 2)
    return GPModelling(GappyLightcurve(t, y[0], dy[0]), make_kernel(), device=device)


def single_lightcurve_configs():
    """BASELINE configs[0], [1], [2] and [4] (ONE light curve each): stretch-move iterations/s through
    GPModelling.derive_posteriors with the device-resident sampler; such small batches take the
    time-parallel kernels.  configs[2] also runs the null model on the same data and reports
    T_LRT = -2 (max lnL_null - max lnL_alt) (tutorial_ppp.ipynb:336-340).  Reported next to the headline,
    not part of `value`."""
    import warnings
    drw, null_kernel, alt_kernel, five_sho = _kernels()
    out = {}
    cases = (("configs[0] DRW N=1e3 32 walkers", drw, 1000, 32, 8000, 8),
             ("configs[1] DRW+SHO N=1e4 128 walkers", null_kernel, 10000, 128, 4000, 11),
             ("configs[2] DRW+SHO+Lorentzian N=1e4 256 walkers", alt_kernel, 10000, 256, 4000, 14),
             ("configs[2] null model DRW+SHO N=1e4 256 walkers", null_kernel, 10000, 256, 4000, 11),
             ("configs[4] 5 x SHO (J=10) N=2e5 512 walkers", five_sho, 200000, 512, 40, 21))
    best = {}
    for name, make_kernel, n, walkers, steps, P in cases:
        g = _gpmodel(make_kernel, n)
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
        from mind_the_gaps_amd.gp import get_engine
        out[name] = {"iterations_per_s": steps / el, "evals_per_s": evals, "iterations_timed": steps,
                     "kernel": get_engine(0).last_solver,
                     "algorithmic_hbm_frac": evals * (24 * n + 8 * (P - 6) + 12) / (HBM_PEAK_GBS * 1e9)}
        best[name] = float(g.max_loglikelihood)
        if n == 200000:
            # what walker sharding over 8 GPUs can gain on this chain: device time of one half-step's likelihoods for
            # the whole half-ensemble (256 rows) and for one rank's share of it (32 rows), same engine, same model
            eng = get_engine(0)
            theta = np.asarray(g.sampler.get_chain()[-1], dtype=np.float64)
            ms = {}
            for rows in (256, 32):
                fastest = np.inf
                for _ in range(3):
                    eng.loglike(theta[:rows])
                    fastest = min(fastest, eng.last_kernel_ms)
                ms[rows] = fastest
            out[name]["walker_shard_8"] = {
                "half_step_ms_256_rows": ms[256], "half_step_ms_32_rows": ms[32], "speedup": ms[256] / ms[32],
                "algorithmic_hbm_frac_256_rows": 256 * (24 * n + 8 * 15 + 12) / (ms[256] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                "what": "likelihoods of one half-step on one MI355X: the whole half-ensemble against the 32 rows a rank "
                        "evaluates when the walkers are sharded over 8 GPUs (mtg_ensemble_shard_rccl); the exchange is one "
                        "all-gather of 256 doubles + status words per half-step"}
    alt, null = (k for k in best if k.startswith("configs[2]"))
    out[alt]["T_LRT"] = {"value": -2.0 * (best[null] - best[alt]), "max_lnL_null": best[null], "max_lnL_alt": best[alt],
                         "what": "-2 (max lnL_null - max lnL_alt) over the burned-in, thinned chains of the two models on the same "
                                 "(pure-noise) light curve, 256 walkers x 4000 steps each (tutorial_ppp.ipynb:336-340)"}
    return out


def raw_kernel_config2(dev):
    """SURVEY 8(d) Config 2, second half: the raw throughput kernel at B = 65 536 evaluations of ONE light curve
    (N = 1e4), DRW + SHO -- J = 3 as the model expands (1 real + 1 complex term), and J = 4 with a zero-amplitude real
    term appended (BASELINE's label counts 4) through the raw-coefficient entry point."""
    from mind_the_gaps_amd import synthetic as synth
    from mind_the_gaps_amd.engine import Engine
    N, B = 10000, 65536
    kinds = synth.NULL_MODEL
    t, y, dy = synth.make_lightcurves(N, 1, seed=20250704 + 2)
    theta = synth.draw_thetas(kinds, B, seed=20250704 + 20)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng = Engine(dev)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    eng.set_time_parallel(0)   # the throughput kernel is what this entry is about
    out = {}
    fastest, name = np.inf, ""
    for _ in range(4):
        _, status = eng.loglike(theta, add_prior=True)
        fastest, name = min(fastest, eng.last_kernel_ms), eng.last_solver
    n_ok = int((status == 0).sum())

    def entry(kernel, ms, J, P):
        evals = n_ok / (ms * 1e-3)
        flop = FLOP_PER_SAMPLE.get(kernel)
        return {"kernel": kernel, "J": J, "evals": n_ok, "kernel_ms": ms, "evals_per_s": evals,
                "algorithmic_hbm_frac": evals * (24 * N + 8 * P + 12) / (HBM_PEAK_GBS * 1e9),
                "fp64_valu_frac": None if flop is None else evals * N * flop / 1e12 / FP64_PEAK_TFLOPS}
    out["J3"] = entry(name, fastest, 3, 5)
    # the same evaluations with raw coefficients and a second real term of amplitude 0 (decay rate 1): J = 4
    ok = status == 0
    S0, Q, w0 = (np.exp(theta[ok, i]) for i in (2, 3, 4))
    f = np.sqrt(4.0 * Q * Q - 1.0)
    a_real = np.stack([np.exp(theta[ok, 0]), np.zeros(n_ok)], axis=1)
    c_real = np.stack([np.exp(theta[ok, 1]), np.ones(n_ok)], axis=1)
    a_c, b_c, c_c, d_c = S0 * w0 * Q, S0 * w0 * Q / f, 0.5 * w0 / Q, 0.5 * w0 / Q * f
    fastest4 = np.inf
    for _ in range(3):
        lnl4, st4 = eng.loglike_coeffs(a_real, c_real, a_c[:, None], b_c[:, None], c_c[:, None], d_c[:, None])
        fastest4 = min(fastest4, eng.last_kernel_ms)
    lnl3, _ = eng.loglike(theta[ok], add_prior=False)
    if not np.all(st4 == 0) or float(np.max(np.abs(lnl4 - lnl3) / np.abs(lnl3))) > 1e-9:
        raise SystemExit("bench: the zero-padded J = 4 evaluation differs from the J = 3 one")
    out["J4_zero_padded"] = entry("mtg_solve_kernel<2,1,0>", fastest4, 4, 5)
    out["what"] = ("one light curve, N = 1e4, B = 65 536 evaluations = one wave per SIMD on the serial sweep (a launch this small "
                   "is latency bound; the device sampler never makes it -- it takes the time-parallel kernels, other_configs)")
    eng.close()
    return out


def workflow_probe():
    import importlib.util
    spec = importlib.util.spec_from_file_location("config3_probe", os.path.join(ROOT, "scripts", "config3_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    return probe


def walker_sharded_configs(rank, world, local_dev, oversubscribed):
    """configs[2] and configs[4] (ONE light curve each) with every half-step's proposals split over the ranks: the
    device-resident sampler of GPModelling.derive_posteriors(device_sampler=True, shard_walkers=True) --
    mtg_ensemble_shard_rccl, a grouped in-place ncclAllGather pair on the launch stream (host-staged exchange over
    gloo when the ranks share a card).  The chains are identical on every rank by construction (same Philox key)."""
    import warnings
    import torch.distributed as dist
    from mind_the_gaps_amd.gp import get_engine
    _, _, alt_kernel, five_sho = _kernels()
    out = {}
    cases = (("configs[2] DRW+SHO+Lorentzian N=1e4 256 walkers", alt_kernel, 10000, 256, 2000),
             ("configs[4] 5 x SHO (J=10) N=2e5 512 walkers", five_sho, 200000, 512, 60))
    for name, make_kernel, n, walkers, steps in cases:
        g = _gpmodel(make_kernel, n, device=local_dev)
        np.random.seed(1)
        eng = get_engine(local_dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # the first run makes the communicator and warms the kernels (and RCCL's first collective) up
            g.derive_posteriors(fit=False, max_steps=10, convergence_steps=10, walkers=walkers, progress=False,
                                device_sampler=True, shard_walkers=True)
        sampler = g.sampler
        info = eng.ensemble_shard_info()
        if info["kind"] == "rccl":
            eng.shard_profile_begin(2 * steps)
        dist.barrier()
        t0 = time.perf_counter()
        sampler.run_mcmc(None, steps)          # the same sharded ensembles, continued
        dist.barrier()
        el = time.perf_counter() - t0
        exchange_ms = eng.shard_profile_read() if info["kind"] == "rccl" else np.empty(0)
        chain_tail = np.ascontiguousarray(sampler.get_chain()[-1])
        import torch
        gdev = "cpu" if oversubscribed else torch.device("cuda", local_dev)
        mine = torch.from_numpy(chain_tail).to(gdev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if not all(torch.equal(gathered[0], other) for other in gathered[1:]):
            raise SystemExit("bench: the ranks of the walker-sharded chain (%s) hold different chains" % name)
        out[name] = {"iterations_per_s": steps / el, "evals_per_s": steps * walkers / el, "half_step_ms": el / steps / 2 * 1e3,
                     "iterations_timed": steps, "rows_per_rank": -(-(walkers // 2) // world), "kernel": eng.last_solver,
                     # ("host (rccl failed: ...)" when the library's communicator could not be brought up on some rank:
                     # distributed.shard_device_ensemble then takes the host-staged exchange on every rank)
                     "transport": getattr(sampler, "transport", None) or info["kind"], "rccl_ranks": info["comm_ranks"],
                     "exchange_us_median": float(np.median(exchange_ms) * 1e3) if len(exchange_ms) else None,
                     "exchange_us_min": float(np.min(exchange_ms) * 1e3) if len(exchange_ms) else None,
                     "chains_identical_on_all_ranks": True}
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)    # does not return
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
    if world > ndev * MAX_RANKS_PER_GPU:     # every rank sees the same and leaves before the process group exists
        raise SystemExit("--gpus %d on %d GPU(s): at most %d ranks may share a card (rehearsal only)"
                         % (world, ndev, MAX_RANKS_PER_GPU))
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
    scaling = args.scaling or ("strong" if grouped else "weak")   # how the light curves are laid out over the ranks
    scaling_label = scaling if world > 1 else "none"               # (one GPU: nothing scales)

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

    def sweep(n_eval=B, th=d_theta):
        eng.loglike_device(n_eval, th.data_ptr(), d_lc.data_ptr(), d_out.data_ptr(),
                           d_status.data_ptr(), add_prior=True, stream=stream.cuda_stream)

    # the exchange of every timed step between two events on the stream it is ordered on (torch's current stream: the
    # collective's own stream waits for it and is waited for by it); ranks sharing a card stage through the host: host clock
    ex_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] \
        if grouped and not oversubscribed else []
    ex_host_us = []

    def step(timed=None):
        sweep()
        if grouped:
            torch.amax(d_out.view(L, W), dim=1, out=d_best[:L])
            if timed is None:
                dist.all_gather_into_tensor(d_gather, d_best.cpu() if oversubscribed else d_best)
            elif oversubscribed:
                staged = d_best.cpu()           # (waits for the sweep: the clock below is the exchange alone)
                t_x = time.perf_counter()
                dist.all_gather_into_tensor(d_gather, staged)
                ex_host_us.append((time.perf_counter() - t_x) * 1e6)
            else:
                ex_events[timed][0].record()
                dist.all_gather_into_tensor(d_gather, d_best)
                ex_events[timed][1].record()

    local_done = [0.0]

    def fence():
        torch.cuda.synchronize(dev)
        local_done[0] = time.perf_counter()     # this rank's own work is done; what follows is waiting for the others
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    busy = local_done[0] - t0
    prep_ms, solve_ms = eng.profile_read()
    kernel_name = eng.last_solver
    exchange_us = np.asarray([a.elapsed_time(b) * 1e3 for a, b in ex_events] if ex_events else ex_host_us, dtype=np.float64)

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
    solve_s = float(np.mean(solve_ms)) * 1e-3
    n_ok_roof, roof_rank, per_rank = n_ok, 0, None
    if grouped:
        # every rank's own figures, so that a curve below the target can be taken apart: load imbalance (kernel_ms),
        # launch floor (busy_ms_per_step - kernel_ms), exchange (exchange_us).  The roofline entry is the SLOWEST rank's.
        mine = torch.tensor([float(np.mean(solve_ms)), float(np.max(solve_ms)), float(np.mean(prep_ms)), busy / args.steps * 1e3,
                             float(n_ok), float(L), float(np.median(exchange_us)) if len(exchange_us) else np.nan,
                             float(np.mean(exchange_us)) if len(exchange_us) else np.nan,
                             float(np.max(exchange_us)) if len(exchange_us) else np.nan], dtype=torch.float64, device=gdev)
        table = torch.empty(world * mine.numel(), dtype=torch.float64, device=gdev)
        dist.all_gather_into_tensor(table, mine)
        table = table.cpu().numpy().reshape(world, -1)

        def spread(col, digits=4):
            v = table[:, col]
            return {"min": round(float(np.min(v)), digits), "max": round(float(np.max(v)), digits),
                    "mean": round(float(np.mean(v)), digits), "by_rank": [round(float(x), digits) for x in v]}
        roof_rank = int(np.argmax(table[:, 0]))
        solve_s, n_ok_roof = float(table[roof_rank, 0]) * 1e-3, int(table[roof_rank, 4])
        per_rank = {"kernel_ms": spread(0), "kernel_ms_slowest_launch": spread(1), "prepare_kernel_ms": spread(2),
                    "busy_ms_per_step": spread(3), "evals_ok": [int(x) for x in table[:, 4]],
                    "lightcurves": [int(x) for x in table[:, 5]],
                    "exchange_us_median": spread(6, 1), "exchange_us_mean": spread(7, 1), "exchange_us_max": spread(8, 1),
                    "what": "kernel_ms: HIP events around the solver of every timed step (mean per rank); busy_ms_per_step: host clock "
                            "from the first launch to this rank's own synchronize, before the barrier; exchange_us: "
                            + ("host clock around the gloo all-gather of the staged maxima (ranks sharing a card)" if oversubscribed
                               else "events on the launch stream around all_gather_into_tensor of %d doubles per rank" % L_pad)}
        extras["per_rank"] = per_rank
        extras["exchange_us"] = float(np.mean(table[:, 6]))

    def headline_line():
        """The contract's JSON line from the timed region alone (+ whatever `extras` holds by now)."""
        bytes_eval = 24 * N + 8 * P + 12
        achieved = n_ok_roof * bytes_eval / solve_s / 1e9
        flop = FLOP_PER_SAMPLE.get(kernel_name)
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("N") == N and rec.get("B") == B:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/bench_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                      "command (%s, commit %s, kernel %s), not measured in this run"
                                      % (rec.get("round", "round 1"), str(rec.get("head", "unknown"))[:12],
                                         str(rec.get("kernel", "?")).split("::")[-1].split("(")[0]))
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
            "scaling": scaling_label,
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
                "bound": "hbm",              # the contract's vocabulary: achieved / peak / frac below are the HBM figures
                "binds": "fp64_valu",        # what actually limits the kernel (the sub-object of that name)
                "kernel": kernel_name,   # as dispatched by the library (mtg_last_solver)
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "bytes_per_eval": bytes_eval,
                "evals_per_launch": n_ok_roof,
                "kernel_ms": solve_s * 1e3,
                "rank": roof_rank,       # (N > 1: the rank whose kernel took longest; its evaluations, its duration)
                "prepare_kernel_ms": float(np.mean(prep_ms)),   # theta -> coefficients, and the sort of the sweep's order
                "J_arith": 5,   # the Lorentzian's null real term (a = 0, c = 0) never enters D_n or z_n: not expanded
                # FP64 vector work of the J = 6 sweep (scripts/loop_stats.py on the sweep loop) against the
                # 78.6 TFLOP/s FP64 vector peak
                "fp64_valu": None if flop is None else {
                    "flop_per_sample": flop, "achieved_tflops": n_ok_roof * N * flop / solve_s / 1e12,
                    "peak_tflops": FP64_PEAK_TFLOPS, "frac": n_ok_roof * N * flop / solve_s / 1e12 / FP64_PEAK_TFLOPS},
            },
        }
        line.update(extras)
        issue = fp64_issue(kernel_name, n_ok_roof / 64.0, N, solve_s, extras.get("clock_under_load"))
        if issue:     # first-class: THE fraction that says how much headroom the kernel has (nominal frac above: 0.57 != 43 % left)
            line["roofline"]["fp64_issue_frac_at_clock"] = issue["frac"]
            line["roofline"]["fp64_issue"] = issue
        if "hbm_copy_measured" in extras:    # the algorithmic rate against the copy bandwidth measured on this box
            copy = extras["hbm_copy_measured"]["GB_per_s"]
            line["roofline"]["hbm_copy_measured"] = copy
            line["roofline"]["frac_of_copy_measured"] = achieved / copy
        if oversubscribed:
            line["oversubscribed"] = "%d ranks on %d GPU(s): rehearsal of the multi-rank path, not a scaling number" % (world, ndev)
        line.setdefault("cpu_baseline", None)
        return line

    if world > 1 and args.cpu_seconds > 0 and not args.no_extras:
        # the CPU port beside the GPUs in the N > 1 line as well: rank 0 times it on its own block (the first evaluations
        # of its timed batch, which it checks on the way), after the timed region; the other ranks wait at the barrier
        if rank == 0:
            extras["cpu_baseline"] = cpu_baseline(t, y, dy, kinds, theta, y_mean, args.cpu_seconds, bounds, out, status)
        dist.barrier()
    if world > 1 and not args.no_extras:
        # the clock rank 0's card holds under this load (rocm-smi's first card = rank 0's), for roofline.fp64_issue_frac_at_clock
        clock = clock_under_load(lambda: [sweep() for _ in range(40)], lambda: torch.cuda.synchronize(dev)) if rank == 0 else None
        if clock:
            extras["clock_under_load"] = dict(clock, what=clock["what"] + " (rank 0's card)")
        dist.barrier()
    if grouped:
        extras["rccl_ranks"] = {"torch_distributed_world": dist.get_world_size(), "backend": dist.get_backend()}
        if not args.no_extras and (world > 1 or os.environ.get("MTG_SHARD_ONE_RANK") == "1"):
            # (MTG_BENCH_FORCE_DIST=1 MTG_SHARD_ONE_RANK=1 on a one-GPU box: the whole multi-rank path -- process group over
            # RCCL, broadcasts, the library's communicator, the all-gather pair of every half-step -- with one rank)
            # The timed region is over and agreed on by every rank.  What follows runs collectives that no one-GPU box
            # can rehearse with more than one RCCL rank: a failure or a hang in them must not take the scaling number
            # with it.  A rank that fails says so in the line ("error") and on stderr; a rank that hangs is ended by
            # the timer below, rank 0 printing the line with what it has.
            import threading
            import traceback
            # exit code of a run whose timed region is fine and printed, but whose extras are not
            EXTRAS_FAILED = 3 if args.strict_extras else 0

            def give_up(why):
                """The line, with what there is: the scaling number is on stdout, the failure in the line
                ("multi_rank_extras_error"), on stderr, and with --strict-extras in the exit code."""
                sys.stderr.write("bench: rank %d: multi-rank extras: %s\n" % (rank, why))
                if rank == 0:
                    extras["multi_rank_extras_error"] = why
                    print(json.dumps(headline_line()), flush=True)
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(EXTRAS_FAILED)

            # a rank that fails says so in the job's key-value store; a watcher thread on every rank polls it, so that
            # nobody waits out the whole --extras-timeout inside a collective the failed rank will never enter
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:
                store = None
            stop_watching = threading.Event()

            def watch():
                while store is not None and not stop_watching.wait(0.5):
                    try:
                        if store.check(["mtg_bench_extras_error"]):
                            give_up(store.get("mtg_bench_extras_error").decode())
                    except Exception:
                        return

            watcher = threading.Thread(target=watch, daemon=True)
            watcher.start()
            timer = threading.Timer(args.extras_timeout, lambda: give_up("timed out after %d s" % args.extras_timeout))
            timer.daemon = True
            timer.start()
            try:
                if os.environ.get("MTG_BENCH_FAIL_EXTRAS") == str(rank):     # rehearsal of this guard
                    raise RuntimeError("injected failure (MTG_BENCH_FAIL_EXTRAS)")
                if not args.no_workflow:
                    # configs[3] as a whole workflow, its simulated light curves cut into one block per rank (first: it
                    # needs torch.distributed's collectives only, the walker-sharded configs bring the library's own
                    # RCCL communicator up).  Reproducible mode: T_sim and the p-value are those of the one-GPU run
                    # (workflow_config3 of the N = 1 line), to the last bit.
                    dist.barrier()
                    t1 = time.perf_counter()
                    wf = workflow_probe().run(sharded=True, device=local_dev)
                    wall = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=gdev)
                    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
                    wf["whole_test_s_max_over_ranks"] = float(wall.item())
                    wf["ranks"] = world
                    # every rank's phases (a whole_test_s above the one-GPU share has to be findable: which rank, which phase)
                    phases = ("observed_chains", "simulate", "refit_null", "refit_alt", "gather")
                    flag = {None: -1.0, False: 0.0, True: 1.0}
                    mine = torch.tensor([wf["seconds"].get(k, 0.0) for k in phases] + [wf["whole_test_s"], flag[wf["null_converged"]],
                                         flag[wf["alt_converged"]]], dtype=torch.float64, device=gdev)
                    every = torch.empty(world * mine.numel(), dtype=torch.float64, device=gdev)
                    dist.all_gather_into_tensor(every, mine)
                    every = every.cpu().numpy().reshape(world, -1)
                    wf["seconds_by_rank"] = {k: [round(float(v), 4) for v in every[:, i]] for i, k in enumerate(phases + ("whole_test_s",))}
                    # the observed light curve's chains ran on rank 0 (null) and rank 1 (alternative): their verdicts
                    wf["null_converged"] = bool(every[0, -2] > 0)
                    wf["alt_converged"] = bool(every[min(1, world - 1), -1] > 0)
                    extras["workflow_config3_sharded"] = wf
                    # World-size invariance CHECKED on this node, not only printed: 16 simulated light curves through the same
                    # sharded path (two per rank at 8), then the same 16 by rank 0 alone -- T_obs, every T_sim and the
                    # p-value must agree to the last bit (ppp.protassov_test(reproducible=True)), or the extras fail.
                    sub = workflow_probe().run(nsims=16, sharded=True, device=local_dev, keep_T_sim=True)
                    if rank == 0:
                        alone = workflow_probe().run(nsims=16, sharded=False, device=local_dev, keep_T_sim=True)
                        same = (sub["T_sim"] == alone["T_sim"] and sub["T_obs"] == alone["T_obs"] and sub["p_value"] == alone["p_value"])
                        wf["world_size_invariance"] = {
                            "lightcurves": 16, "ranks": world, "identical_to_one_rank_alone": bool(same),
                            "T_sim_checksum_sharded": float(np.sum(sub["T_sim"])), "T_sim_checksum_alone": float(np.sum(alone["T_sim"])),
                            "p_value_sharded": sub["p_value"], "p_value_alone": alone["p_value"],
                            "sharded_s": sub["whole_test_s"], "alone_s": alone["whole_test_s"],
                            "what": "protassov_test on 16 simulated light curves, N = 1e4, 256 walkers x 500 steps, both models: over "
                                    "all ranks (split %r), then on rank 0 alone in the same process; compared value for value" % sub["split"]}
                        if not same:
                            raise RuntimeError("sharded Protassov test differs from the one-rank run of the same 16 light curves: "
                                               "T_sim checksum %.17g vs %.17g" % (np.sum(sub["T_sim"]), np.sum(alone["T_sim"])))
                    dist.barrier()
                extras["walker_sharded"] = walker_sharded_configs(rank, world, local_dev, oversubscribed)
                dist.barrier()        # every rank got through: nobody is left inside a collective
                timer.cancel()
                stop_watching.set()
            except BaseException as exc:     # this rank failed: tell the others, let rank 0 print, leave
                traceback.print_exc()
                why = "rank %d: %r" % (rank, exc)
                if store is not None:
                    try:
                        store.set("mtg_bench_extras_error", why)
                    except Exception:
                        pass
                if rank != 0:
                    time.sleep(3.0)          # rank 0's watcher prints the line before a launcher that sees this rank die stops it
                give_up(why)
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
        # (a') the clock the card holds under this load (the 78.6 TFLOP/s FP64 vector peak assumes 2.4 GHz; the sweep is
        # power limited): rocm-smi read while ~1 s of sweeps is queued.  A reading that fails is left out.
        clock = clock_under_load(lambda: [sweep() for _ in range(40)], lambda: torch.cuda.synchronize(dev))
        if clock:
            extras["clock_under_load"] = clock
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
        # (c) the NULL model's sweep over the same light curves (configs[3] fits both models; SURVEY 8(d) Config 4)
        nkinds = synth.NULL_MODEL
        nfull, nfree, nbounds = synth.model_spec(nkinds, y, per_lc_mean=True)
        eng.set_model(nkinds, nfull, nfree, nbounds)
        d_theta0 = torch.from_numpy(synth.draw_thetas(nkinds, B, seed=20250704 + 41)).to(dev)
        for _ in range(2):
            sweep(B, d_theta0)
        torch.cuda.synchronize(dev)
        eng.profile_begin(reps)
        t1 = time.perf_counter()
        for _ in range(reps):
            sweep(B, d_theta0)
        torch.cuda.synchronize(dev)
        dt0 = (time.perf_counter() - t1) / reps
        _, solve0 = eng.profile_read()
        n_ok0 = int((d_status.cpu().numpy() == 0).sum())
        k0 = eng.last_solver
        s0 = float(np.mean(solve0)) * 1e-3
        flop0 = FLOP_PER_SAMPLE.get(k0)
        extras["null_model_sweep"] = {
            "model": "DRW+SHO (J=3, P=5)", "evals_per_step": B, "ms_per_step": dt0 * 1e3, "evals_per_s": B / dt0,
            "kernel": k0, "kernel_ms": s0 * 1e3,
            "algorithmic_hbm_frac": n_ok0 * (24 * N + 8 * 5 + 12) / s0 / 1e9 / HBM_PEAK_GBS,
            "fp64_valu_frac": None if flop0 is None else n_ok0 * N * flop0 / s0 / 1e12 / FP64_PEAK_TFLOPS,
            "fp64_issue_frac_at_clock": (fp64_issue(k0, n_ok0 / 64.0, N, s0, clock) or {}).get("frac"),
            "both_models_evals_per_s": 2 * B / (dt0 + elapsed / args.steps)}
        eng.set_model(kinds, full, free, bounds)
        # (d) a model whose six ranks are all arithmetic -- three SHO terms, NR = 0, NC = 3 -- beside the headline, whose
        # "J = 6" is five ranks of arithmetic (the Lorentzian's null real term is dropped)
        skinds = [synth.K_SHO] * 3
        sfull, sfree, sbounds = synth.model_spec(skinds, y, per_lc_mean=True)
        sth = np.array([np.log(50.0), np.log(3.0), np.log(2 * np.pi / 7.0), np.log(30.0), np.log(8.0), np.log(2 * np.pi / 13.0),
                        np.log(20.0), np.log(1.5), np.log(2 * np.pi / 31.0)])
        sfull[:9] = sth
        eng.set_model(skinds, sfull, sfree, sbounds)
        d_theta6 = torch.from_numpy(sth * (1 + 0.03 * np.random.default_rng(20250704 + 42).standard_normal((B, 9)))).to(dev)
        for _ in range(2):
            sweep(B, d_theta6)
        torch.cuda.synchronize(dev)
        eng.profile_begin(reps)
        t1 = time.perf_counter()
        for _ in range(reps):
            sweep(B, d_theta6)
        torch.cuda.synchronize(dev)
        dt6 = (time.perf_counter() - t1) / reps
        _, solve6 = eng.profile_read()
        n_ok6 = int((d_status.cpu().numpy() == 0).sum())
        k6, s6 = eng.last_solver, float(np.mean(solve6)) * 1e-3
        flop6 = FLOP_PER_SAMPLE.get(k6)
        if n_ok6 < B // 2:
            raise SystemExit("bench: the three-SHO sweep rejected most of its rows")
        extras["true_J6_sweep"] = {
            "model": "3 x SHO, all under-damped (NR = 0, NC = 3: J = 6 of arithmetic, P = 9)", "evals_per_step": B,
            "ms_per_step": dt6 * 1e3, "evals_per_s": B / dt6, "kernel": k6, "kernel_ms": s6 * 1e3, "evals_ok": n_ok6,
            "algorithmic_hbm_frac": n_ok6 * (24 * N + 8 * 9 + 12) / s6 / 1e9 / HBM_PEAK_GBS,
            "fp64_valu_frac": None if flop6 is None else n_ok6 * N * flop6 / s6 / 1e12 / FP64_PEAK_TFLOPS,
            "fp64_issue_frac_at_clock": (fp64_issue(k6, n_ok6 / 64.0, N, s6, clock) or {}).get("frac")}
        eng.set_model(kinds, full, free, bounds)
        # (e) the mid-batch regime: one GPU's half-step at 8 GPUs -- 250 light curves x 128 proposals = 32 000 rows -- on
        # the one-lane sweep and on its two-wave pipeline (csrc/mtg_kernels_pipe.hip), both models, device time
        mid = {}
        Lm, Wm = max(1, L // 8), W // 2
        Bm = Lm * Wm
        d_lc_mid = torch.from_numpy(np.repeat(np.arange(Lm, dtype=np.int32), Wm)).to(dev)
        for label, mk, th_dev in (("alt_J5_arith", kinds, d_theta), ("null_J3", nkinds, d_theta0)):
            mfull, mfree, mbounds = (full, free, bounds) if mk is kinds else (nfull, nfree, nbounds)
            eng.set_model(mk, mfull, mfree, mbounds)
            ms, names, outs = {}, {}, {}
            for _ in range(4):
                for mode in (0, 2):
                    eng.set_pipeline(mode)
                    eng.loglike_device(Bm, th_dev.data_ptr(), d_lc_mid.data_ptr(), d_out.data_ptr(), d_status.data_ptr(),
                                       add_prior=True, stream=stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    ms[mode] = min(ms.get(mode, np.inf), eng.last_kernel_ms)
                    names[mode] = eng.last_solver
                    outs[mode] = d_out[:Bm].cpu().numpy()
            if not np.array_equal(outs[0], outs[2]):
                raise SystemExit("bench: the pipelined sweep and the one-lane sweep differ (%s)" % label)
            mid[label] = {"rows": Bm, "one_lane_ms": ms[0], "one_lane_kernel": names[0], "pipeline_ms": ms[2],
                          "pipeline_kernel": names[2], "bitwise_equal": True}
        eng.set_pipeline(2)
        eng.set_model(kinds, full, free, bounds)
        mid["what"] = ("prepare + sort + solve of one half-step of %d light curves x %d proposals, N = %d: what one GPU of 8 "
                       "runs 1002 times per model in the configs[3] workflow" % (Lm, Wm, N))
        # ... and both models' half-steps at once, as the side-by-side refits run them: from two contexts, and with the
        # contexts paired (one launch of eight-wave workgroups: csrc/mtg_kernels_pipe_pair.hip)
        import importlib.util
        spec_pp = importlib.util.spec_from_file_location("pair_probe", os.path.join(ROOT, "scripts", "pair_probe.py"))
        pair_probe = importlib.util.module_from_spec(spec_pp)
        spec_pp.loader.exec_module(pair_probe)
        mid["both_models"] = pair_probe.run(L=Lm, W=Wm, N=N, device=local_dev)
        if not mid["both_models"]["bitwise_equal_to_unpaired"]:
            raise SystemExit("bench: the paired launch of both models differs from the unpaired kernels")
        extras["mid_batch_half_step"] = mid
        # (f) what HBM can actually stream on this box: a device-to-device copy of 2 GiB (read + write bytes / time),
        # beside the vendor's 8 TB/s that `roofline.peak` quotes
        nbytes = 1 << 31
        src_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
        dst_buf = torch.empty_like(src_buf)
        for _ in range(2):
            dst_buf.copy_(src_buf)
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best_ms = np.inf
        for _ in range(5):
            ev_a.record()
            dst_buf.copy_(src_buf)
            ev_b.record()
            ev_b.synchronize()
            best_ms = min(best_ms, ev_a.elapsed_time(ev_b))
        del src_buf, dst_buf
        extras["hbm_copy_measured"] = {"GB_per_s": 2 * nbytes / (best_ms * 1e-3) / 1e9, "bytes_each_way": nbytes,
                                       "what": "hipMemcpyAsync device to device, read + write bytes over the fastest of 5"}

    if rank == 0:
        line = headline_line()
        if world == 1 and args.cpu_seconds > 0 and not args.no_extras:
            line["cpu_baseline"] = cpu_baseline(t, y, dy, kinds, theta, y_mean, args.cpu_seconds, bounds, out, status)
        if world == 1 and not args.no_extras:
            # the other SURVEY 8(d) figures; a failure here fails the bench
            line["config2_raw_kernel"] = raw_kernel_config2(local_dev)
            line["other_configs"] = single_lightcurve_configs()
            if not args.no_workflow:
                line["workflow_config3"] = workflow_probe().run()
                # one GPU on the share of the workflow it gets at 8 GPUs, split by light curve: 250 simulated light curves,
                # both models (the observed light curve's chains are not split: every rank runs them)
                share = workflow_probe().run(nsims=max(1, args.lightcurves // 8))
                line["workflow_config3_share_of_8"] = share
                # On 8 GPUs the observed light curve's two chains run on two ranks, one model each (ppp.protassov_test,
                # observed_split), where this one GPU ran both side by side: count that phase as the longer chain alone
                alone = workflow_probe().observed_chains_alone()
                share["observed_chains_alone_s"] = alone
                share["whole_test_s_with_observed_split"] = (share["whole_test_s"] - share["seconds"]["observed_chains"]
                                                             + max(alone.values()))
                share["projected_speedup_at_8_gpus"] = (line["workflow_config3"]["whole_test_s"]
                                                        / share["whole_test_s_with_observed_split"])
                # the same share with a LOGNORMAL flux PDF (simulator.py:65-140): every simulated segment through the E13
                # amplitude / rank adjustment on the device (csrc/mtg_e13.hip) before it is observed -- the loop stays on the GPU
                logn = workflow_probe().run(nsims=max(1, args.lightcurves // 8), pdf="Lognormal")
                line["workflow_config3_share_of_8_lognormal"] = {
                    k: logn[k] for k in ("nsims", "flux_pdf", "segment_points_per_simulation", "whole_test_s", "seconds", "p_value",
                                         "refit_evaluations_per_s_end_to_end")}
                line["workflow_config3_share_of_8_lognormal"]["simulate_s_gaussian"] = share["seconds"]["simulate"]
                share["projected_speedup_at_8_gpus_what"] = (
                    "whole test on one GPU / (this GPU's share of 250 light curves, its observed-chains phase replaced by the "
                    "longer of the two chains run alone); leaves out the two broadcasts (rank 0's samples, rank 1's maximum) "
                    "and the all-gather of 2 x 2000 maxima, which the N = 8 line measures (workflow_config3_sharded)")
        print(json.dumps(line), flush=True)

    eng.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
