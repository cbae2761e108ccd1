"""Burst against sustained rate of the headline sweep (2000 light curves x 256 walkers, N = 10 000, J = 6):
bench.py times 20 steps (~0.5 s); the refits of configs[3] keep the FP64 pipes busy for ~30 s.  Prints the time per
sweep in windows of a sustained run, the same for the half-ensemble batch of a sampler half-step (128 rows per light
curve), and what rocm-smi says about clocks and power in the middle of the run.

    python scripts/sustained_probe.py [sweeps]  ->  text
"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=30).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction", "edge"))]
        return "; ".join(keep) if keep else out.strip()[:400]
    except Exception as exc:      # measurement aid only
        return "rocm-smi unavailable: %s" % exc


def main(sweeps=1000, N=10000, L=2000, W=256):
    dev = torch.device("cuda", 0)
    kinds = synth.ALT_MODEL
    t, y, dy = synth.make_lightcurves(N, L, seed=20250704 + 4)
    theta = synth.draw_thetas(kinds, L * W, seed=20250704 + 40)
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True)
    eng = Engine(0)
    eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y.mean(axis=1))
    eng.set_model(kinds, full, free, bounds)
    P = theta.shape[1]
    half = np.ascontiguousarray(theta.reshape(L, W, P)[:, :W // 2].reshape(-1, P))
    stream = torch.cuda.current_stream(dev)
    out = torch.empty(L * W, dtype=torch.float64, device=dev)
    st = torch.empty(L * W, dtype=torch.int32, device=dev)
    print("idle: " + smi(), flush=True)
    for name, th, per in (("full ensemble, 256 rows per light curve", theta, W), ("half ensemble, 128 rows per light curve", half, W // 2)):
        d_th = torch.from_numpy(th).to(dev)
        d_lc = torch.from_numpy(np.repeat(np.arange(L, dtype=np.int32), per)).to(dev)
        B = len(th)

        def run(k):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(k):
                eng.loglike_device(B, d_th.data_ptr(), d_lc.data_ptr(), out.data_ptr(), st.data_ptr(), add_prior=True,
                                   stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / k * 1e3
        run(3)
        time.sleep(2.0)
        print("%s (%d evaluations): burst of 20 after 2 s idle: %.3f ms per sweep" % (name, B, run(20)), flush=True)
        win = 50
        line = []
        for w in range(max(1, sweeps // win)):
            line.append(run(win))
            if w == sweeps // win // 2:
                # queue the next window before asking, so the card is busy while rocm-smi reads it
                for _ in range(win):
                    eng.loglike_device(B, d_th.data_ptr(), d_lc.data_ptr(), out.data_ptr(), st.data_ptr(), add_prior=True,
                                       stream=stream.cuda_stream)
                print("  busy: " + smi(), flush=True)
        print("  sustained, ms per sweep in windows of %d: %s" % (win, " ".join("%.2f" % v for v in line)), flush=True)
        print("  evaluations/s: first window %.3e, last window %.3e" % (B / line[0] * 1e3, B / line[-1] * 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:2]])
