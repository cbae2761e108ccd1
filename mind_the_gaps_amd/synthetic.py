"""Synthetic inputs of SURVEY.md section 8(d) (shared by tests, smoke and bench).

times: dt = 0.05 + Exp(1) days with a 100-day gap after every N/5-th sample;
dy ~ U(0.5, 2); y = 100 + 10 N(0, 1); theta drawn 10 % around the tutorial truth
values (reference docs/notebooks/tutorial_ppp.ipynb:52-53,242-246).
"""
import numpy as np

K_REAL, K_COMPLEX3, K_COMPLEX4, K_SHO, K_MATERN32, K_JITTER, K_DRW, K_LORENTZIAN, K_COSINUS, K_BPL = range(10)
NPARAMS = {K_REAL: 2, K_COMPLEX3: 3, K_COMPLEX4: 4, K_SHO: 3, K_MATERN32: 2, K_JITTER: 1,
           K_DRW: 2, K_LORENTZIAN: 3, K_COSINUS: 2, K_BPL: 3}

TRUTH = {
    K_DRW: [np.log(100.0), np.log(2 * np.pi / 20.0)],
    K_SHO: [np.log(50.0), np.log(3.0), np.log(2 * np.pi / 7.0)],
    K_LORENTZIAN: [np.log(100.0), np.log(80.0), np.log(2 * np.pi / 10.0)],
    K_REAL: [np.log(30.0), np.log(0.2)],
    K_COMPLEX3: [np.log(20.0), np.log(0.05), np.log(0.7)],
    K_COMPLEX4: [np.log(20.0), np.log(1.0), np.log(0.05), np.log(0.7)],
    K_MATERN32: [np.log(5.0), np.log(8.0)],
    K_JITTER: [np.log(0.7)],
    K_COSINUS: [np.log(15.0), np.log(2 * np.pi / 13.0)],
    K_BPL: [np.log(60.0), np.log(4.0), np.log(2 * np.pi / 25.0)],
}
# bounds of the tutorial: (-10, 50) for amplitudes, (-10, 10) otherwise
AMP_FIRST = True

NULL_MODEL = [K_DRW, K_SHO]
ALT_MODEL = [K_DRW, K_SHO, K_LORENTZIAN]


def make_times(N, rng, offset=0.0):
    dt = 0.05 + rng.exponential(1.0, N)
    if N >= 5:
        for k in range(1, 5):
            dt[k * (N // 5)] += 100.0
    return offset + np.cumsum(dt)


def make_lightcurves(N, L, seed, offset=0.0):
    rng = np.random.default_rng(seed)
    t = make_times(N, rng, offset)
    dy = rng.uniform(0.5, 2.0, (L, N))
    y = 100.0 + 10.0 * rng.standard_normal((L, N))
    return t, y, dy


def truth(kinds):
    return np.concatenate([TRUTH[k] for k in kinds])


def bounds_for(kinds):
    b = []
    for k in kinds:
        for i in range(NPARAMS[k]):
            b.append((-10.0, 50.0) if i == 0 else (-10.0, 10.0))
    return np.array(b)


def draw_thetas(kinds, B, seed, percent=0.1):
    """spread_walkers law (gpmodelling.py:321-322): N(theta*, percent * |theta*|)."""
    rng = np.random.default_rng(seed)
    th = truth(kinds)
    return th + percent * np.abs(th) * rng.standard_normal((B, len(th)))


def model_spec(kinds, y, mean_kind=0, fit_mean=False, mean_values=None, per_lc_mean=False):
    """(full_values, free_index, bounds) for engine.set_model, mean frozen at mean(y)
    unless fit_mean (the reference default, gpmodelling.py:83-87).  With ``per_lc_mean``
    the frozen mean is 0 here and every light curve's own average goes to
    ``engine.set_lightcurves(..., y_offset=y.mean(axis=1))`` -- what GPModelling does."""
    th = truth(kinds)
    nk = len(th)
    if per_lc_mean:
        mean_values = [0.0]
    if mean_values is None:
        mean_values = [float(np.mean(y))] if mean_kind == 0 else [0.0, float(np.mean(y))]
    full = np.concatenate([th, mean_values])
    bounds = np.vstack([bounds_for(kinds),
                        [(-np.inf, np.inf)] * len(mean_values)])
    free = list(range(nk)) + (list(range(nk, len(full))) if fit_mean else [])
    return full, np.array(free, dtype=np.int32), bounds
