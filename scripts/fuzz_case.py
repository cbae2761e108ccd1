"""One case of tests/test_fuzz_gpu.py taken apart: the rows where the time-parallel kernel is furthest from the C oracle, with the
serial sweep, the fused oracle and the 40-digit dense likelihood beside them.   python scripts/fuzz_case.py CASE (= case + MTG_FUZZ_OFFSET)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np
from mind_the_gaps_amd import synthetic as synth
from mind_the_gaps_amd.engine import Engine
from oracle import celerite as oracle_c, dense
import test_fuzz_gpu as F
case = int(sys.argv[1])
rng = np.random.default_rng(9000 + case)
tp_mode = int(rng.integers(0, 2))
kinds = F.random_model(rng, *((6, 3) if tp_mode else (10, 5)))
linear_mean = bool(rng.integers(0, 3) == 0)
N = int(rng.choice([1, 2, 5, 37, 256, 257, 800, 2500, 4096, 5001])); L = int(rng.integers(1, 5)); B = int(rng.choice([1, 3, 64, 65, 200, 700]))
per_lc_t = bool(rng.integers(0, 2)) and L > 1; add_prior = bool(rng.integers(0, 2))
t, y, dy = synth.make_lightcurves(N, L, seed=1000 + case)
assert not per_lc_t
if linear_mean:
    y = y + 0.01 * (t - t.min()); full, free, bounds = synth.model_spec(kinds, y, mean_kind=1, fit_mean=True); y_mean=None
else:
    full, free, bounds = synth.model_spec(kinds, y, per_lc_mean=True); y_mean = y.mean(axis=1)
theta = synth.draw_thetas(kinds, B, seed=case, percent=0.25)
off=0
for k in kinds:
    if k == synth.K_SHO: theta[rng.random(B) < 0.4, off + 1] = np.log(rng.uniform(0.05, 0.45))
    off += synth.NPARAMS[k]
pushed = rng.random(B) < 0.1; theta[pushed, 0] = 60.0
lc = rng.integers(0, L, B).astype(np.int32)
if linear_mean: theta = np.hstack([theta, 0.01 + 0.002 * rng.standard_normal((B, 1)), 100.0 + rng.standard_normal((B, 1))])
print("kinds", kinds, "N", N, "L", L, "B", B, "tp", tp_mode, "linear", linear_mean, "prior", add_prior)
eng = Engine(0)
eng.set_lightcurves(t, y, dy + 1e-12, y_offset=y_mean); eng.set_model(kinds, full, free, bounds, mean_kind=1 if linear_mean else 0)
res = {}
for mode in (0, 1):
    eng.set_time_parallel(mode); res[mode] = eng.loglike(theta, lc, add_prior=add_prior); print("mode", mode, eng.last_solver)
full_b = theta if linear_mean else np.hstack([theta, y_mean[lc][:, None]])
okw = dict(bounds=bounds, add_prior=add_prior, nthreads=4, mean_kind=1 if linear_mean else 0)
ref, rst = oracle_c.logprob_batch(t, y, dy, kinds, full_b, lc_index=lc, **okw)
reff, _ = oracle_c.logprob_batch(t, y, dy, kinds, full_b, lc_index=lc, fused=True, **okw)
keep = ~pushed & (rst == 0)
e0 = np.abs(res[0][0] - ref) / np.abs(ref); e1 = np.abs(res[1][0] - ref) / np.abs(ref); ef = np.abs(reff - ref)/np.abs(ref)
worst = np.argsort(np.where(keep, e1, 0))[-3:]
for i in worst:
    p = full_b[i]
    # dense truth
    nk = dense.n_kernel_params(kinds)
    co = dense.build_coeffs(kinds, p[:nk])
    try:
        d = float(dense.dense_loglike_mp(t, y[lc[i]], dy[lc[i]], co, mean_kind=1 if linear_mean else 0, mean_params=tuple(p[nk:]), dps=40))
    except Exception as ex:
        d = float('nan'); print("dense failed", ex)
    print("row %d: theta %s\n   oracle %.15g fused %.15g sweep %.15g tp %.15g dense %.15g | rel vs dense: oracle %.2e sweep %.2e tp %.2e" % (
        i, np.array2string(p, precision=4), ref[i], reff[i], res[0][0][i], res[1][0][i], d, abs(ref[i]-d)/abs(d), abs(res[0][0][i]-d)/abs(d), abs(res[1][0][i]-d)/abs(d)))
# the worst row again: filter pass forced, and with its SHO less and less over-damped
i = int(worst[-1])
nk = dense.n_kernel_params(kinds)
eng.set_time_parallel(1)
eng.set_tp_direct(0)
f_out, f_st = eng.loglike(theta[i:i + 1], lc[i:i + 1], add_prior=add_prior)
eng.set_tp_direct(1)
print("row %d through the filter pass: %.15g (status %d)" % (i, f_out[0], f_st[0]))
off = 0
for k in kinds:
    if k == synth.K_SHO:
        for lq in (-2.43, -2.0, -1.5, -1.0, -0.8, -0.7):
            th = theta[i:i + 1].copy(); th[0, off + 1] = lq
            p = (th if linear_mean else np.hstack([th, y_mean[lc[i:i + 1]][:, None]]))[0]
            co = dense.build_coeffs(kinds, p[:nk])
            d = float(dense.dense_loglike_mp(t, y[lc[i]], dy[lc[i]], co, mean_kind=1 if linear_mean else 0, mean_params=tuple(p[nk:]), dps=40))
            eng.set_time_parallel(1); a1, _ = eng.loglike(th, lc[i:i + 1], add_prior=add_prior)
            eng.set_time_parallel(0); a0, _ = eng.loglike(th, lc[i:i + 1], add_prior=add_prior)
            print("   log Q %.2f: a_real %s c_real %s | tp rel err %.2e, sweep rel err %.2e" % (lq, np.array2string(np.asarray(co[0]), precision=3), np.array2string(np.asarray(co[1]), precision=3), abs(a1[0] - d) / abs(d), abs(a0[0] - d) / abs(d)))
    off += synth.NPARAMS[k]
# the row exactly as it was, alone: on whatever kernel a batch of one gets, and on the one-wave kernel (mode 3)
for mode in (1, 3):
    for direct in (1, 0):
        eng.set_time_parallel(mode); eng.set_tp_direct(direct)
        o, st_ = eng.loglike(theta[i:i + 1], lc[i:i + 1], add_prior=add_prior)
        print("alone, mode %d direct %d: %.15g  [%s]" % (mode, direct, o[0], eng.last_solver))
# ... and inside growing prefixes of the batch
eng.set_time_parallel(1); eng.set_tp_direct(1)
for lo in (i - 3, i - 63, 0):
    o, st_ = eng.loglike(theta[lo:i + 1], lc[lo:i + 1], add_prior=add_prior)
    print("rows %d..%d: row %d = %.15g [%s]" % (lo, i, i, o[-1], eng.last_solver))
