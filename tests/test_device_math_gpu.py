"""GPU accuracy test of the kernel's own exp / sincos / reciprocal (csrc/mtg_math.h)
against mpmath-grade references (numpy's libm is used where it is correctly rounded
to < 1 ulp; the large-argument trig reference is mpmath)."""
import mpmath as mp
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


LIMIT = 1.0e12      # MTG_TRIG_FAST_MAX (csrc/mtg_math.h)


def test_device_math_accuracy(engine):
    rng = np.random.default_rng(0)
    x = np.concatenate([
        [0.0, 1e-300, 1e-20, 1e-8, 0.5, 1.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 700.0, 745.0, 800.0, 1e5],
        rng.uniform(0, 1, 2000), rng.uniform(0, 50, 4000), 10 ** rng.uniform(-12, 5, 6000),
        np.arange(1, 2000) * (np.pi / 2),                 # worst cases of the reduction
        np.float64(1.0e5) - rng.uniform(0, 10, 200), 1.0e5 + 10 ** rng.uniform(0, 9, 500),
        10 ** rng.uniform(5, 12, 3000), np.float64(LIMIT) - 10 ** rng.uniform(-3, 6, 200),   # the top of the table path's range
        LIMIT * (1.0 + rng.uniform(0.001, 5.0, 100)),                                        # OCML fallback
    ])
    e, s, c, r = engine.math_probe(x)
    # exp(-x): the one-constant reduction costs |x| 2^-53 of relative accuracy, which is
    # at most 4e-17 ABSOLUTE (x e^-x <= 0.37) -- what matters for a propagator
    want = np.exp(-x)
    big = want > 1e-300
    assert np.max(np.abs(e[big] - want[big]) / want[big] / (1.0 + 0.5 * x[big])) < 4.5e-16
    assert np.max(np.abs(e - want)) < 2.3e-16
    assert np.all(e[~big] <= 1e-300) and np.all(e >= 0)
    # sin / cos: absolute error against 40-digit mpmath.  The table path reduces with ONE constant
    # C = fl(2 pi / 16 N) inside an fma (exact product), so what it evaluates is the phase
    # x (1 - eps) with the fixed eps = (C - 2 pi / 16 N) / C, |eps| < 2^-53: in the sweep that is the
    # frequency d moved by less than its own rounding, the same for every sample -- not an error
    # that accumulates.  The OCML fallback (x > 1e12) evaluates the phase x itself.
    mp.mp.dps = 40
    n_trig = 2048
    c_fl = mp.mpf(float.fromhex("0x1.921fb54442d18p-2") / n_trig)
    eps = (c_fl - 2 * mp.pi / (16 * n_trig)) / c_fl
    assert abs(eps) < mp.mpf(2) ** -53
    idx = np.concatenate([np.arange(0, len(x), 7), np.arange(len(x) - 2700, len(x))])
    table = bool(np.all(x <= LIMIT))  # the probe takes the fallback for the whole wave otherwise
    ph = [mp.mpf(float(v)) * (1 - eps) if (table or float(v) <= LIMIT) else mp.mpf(float(v)) for v in x[idx]]
    ws = np.array([float(mp.sin(v)) for v in ph])
    wc = np.array([float(mp.cos(v)) for v in ph])
    es, ec = np.abs(s[idx] - ws), np.abs(c[idx] - wc)
    # lanes of a wave that also holds an x > 1e12 take OCML (plain phase): accept either reading there
    ws0 = np.array([float(mp.sin(mp.mpf(float(v)))) for v in x[idx]])
    wc0 = np.array([float(mp.cos(mp.mpf(float(v)))) for v in x[idx]])
    es, ec = np.minimum(es, np.abs(s[idx] - ws0)), np.minimum(ec, np.abs(c[idx] - wc0))
    assert np.max(es) < 3e-16 and np.max(ec) < 3e-16
    # (absolute accuracy is the contract: the pair feeds bounded generators U, V)
    # reciprocal
    pos = x > 0
    err = np.max(np.abs(r[pos] * x[pos] - 1.0))
    print("rcp max error %.2e" % err)
    assert err < 4e-15      # hardware seed + one Newton step: ~9 ulp, see csrc/mtg_math.h
