"""oracle/dense.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent definition-level oracle for the celerite log-likelihood that
``GPModelling._log_probability`` evaluates
(/root/reference/mind_the_gaps/gpmodelling.py:127-152, GP set-up at :47-59):

    K_nm = k(|t_n - t_m|) + delta_nm (sigma_n^2 + jitter),  sigma_n = dy_n + 1e-12  (gpmodelling.py:54)
    k(tau) = sum_j a_j e^{-c_j tau} + sum_k e^{-c_k tau} [a_k cos(d_k tau) + b_k sin(d_k tau)]
    lnL = -1/2 ( r^T K^-1 r + ln det K + N ln 2 pi ),  r = y - mu(t)

(Foreman-Mackey et al. 2017 eqs. 7-9).  It builds the
dense N x N covariance and factorises it with LAPACK (float64) or mpmath
(arbitrary precision), i.e. it shares no code and no algorithm with the
semiseparable recurrences of oracle/celerite_ref.c or the HIP kernels.

PARITY STATUS: "parity unpinned" at the celerite boundary (the reference's tests
hold no lnL value; celerite is not installable here).  This file is the anchor
used instead; the coefficient builders are pinned separately by the
reference's own PSD known answers (tests/models_test.py:14-102).

Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

K_REAL, K_COMPLEX3, K_COMPLEX4, K_SHO, K_MATERN32, K_JITTER, K_DRW, K_LORENTZIAN, K_COSINUS, K_BPL = range(10)

NPARAMS = {K_REAL: 2, K_COMPLEX3: 3, K_COMPLEX4: 4, K_SHO: 3, K_MATERN32: 2, K_JITTER: 1,
           K_DRW: 2, K_LORENTZIAN: 3, K_COSINUS: 2, K_BPL: 3}


def build_coeffs(kinds, params, extra=None):
    """theta -> (a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter).

    Restates mind_the_gaps/models/celerite_models.py:7-90 and the celerite
    built-in terms.  Lorentzian keeps its (0, 0) real
    term (celerite_models.py:12-15).
    """
    ar, cr, ac, bc, cc, dc = [], [], [], [], [], []
    jitter = 0.0
    off = 0
    for i, kind in enumerate(kinds):
        p = np.asarray(params[off:off + NPARAMS[kind]], dtype=np.float64)
        off += NPARAMS[kind]
        if kind == K_REAL:
            ar.append(np.exp(p[0])); cr.append(np.exp(p[1]))
        elif kind == K_COMPLEX3:
            ac.append(np.exp(p[0])); bc.append(0.0); cc.append(np.exp(p[1])); dc.append(np.exp(p[2]))
        elif kind == K_COMPLEX4:
            ac.append(np.exp(p[0])); bc.append(np.exp(p[1])); cc.append(np.exp(p[2])); dc.append(np.exp(p[3]))
        elif kind == K_SHO:
            S0, Q, w0 = np.exp(p)
            if Q < 0.5:
                f = np.sqrt(1.0 - 4.0 * Q * Q)
                ar += [0.5 * S0 * w0 * Q * (1.0 + 1.0 / f), 0.5 * S0 * w0 * Q * (1.0 - 1.0 / f)]
                cr += [0.5 * w0 / Q * (1.0 - f), 0.5 * w0 / Q * (1.0 + f)]
            else:
                f = np.sqrt(4.0 * Q * Q - 1.0)
                ac.append(S0 * w0 * Q); bc.append(S0 * w0 * Q / f)
                cc.append(0.5 * w0 / Q); dc.append(0.5 * w0 / Q * f)
        elif kind == K_MATERN32:
            eps = 0.01 if extra is None else extra[i]
            w0 = np.sqrt(3.0) * np.exp(-p[1])
            S0 = np.exp(2.0 * p[0]) / w0
            ac.append(w0 * S0); bc.append(w0 * w0 * S0 / eps); cc.append(w0); dc.append(eps)
        elif kind == K_JITTER:
            jitter += np.exp(2.0 * p[0])
        elif kind == K_DRW:
            ar.append(np.exp(p[0])); cr.append(0.5 * np.exp(p[1]) / 0.5)
        elif kind == K_LORENTZIAN:
            ar.append(0.0); cr.append(0.0)
            ac.append(np.exp(p[0])); bc.append(0.0)
            cc.append(0.5 * np.exp(p[2]) / np.exp(p[1])); dc.append(np.exp(p[2]))
        elif kind == K_COSINUS:
            ac.append(np.exp(p[0])); bc.append(0.0); cc.append(0.0); dc.append(np.exp(p[1]))
        elif kind == K_BPL:
            ac.append(np.exp(p[0])); bc.append(np.exp(p[1])); cc.append(np.exp(p[2])); dc.append(np.exp(p[2]))
        else:
            raise ValueError("unknown term kind %r" % (kind,))
    f64 = lambda v: np.asarray(v, dtype=np.float64)
    return f64(ar), f64(cr), f64(ac), f64(bc), f64(cc), f64(dc), float(jitter)


def n_kernel_params(kinds):
    return sum(NPARAMS[k] for k in kinds)


def psd(coeffs, omega):
    """celerite ``Term.get_psd``, used to pin the
    builders against the closed forms of mind_the_gaps/models/psd_models.py."""
    ar, cr, ac, bc, cc, dc, _ = coeffs
    w2 = np.asarray(omega, dtype=np.float64) ** 2
    p = np.zeros_like(w2)
    for a, c in zip(ar, cr):
        p += a * c / (c * c + w2)
    for a, b, c, d in zip(ac, bc, cc, dc):
        w02 = c * c + d * d
        p += ((a * c + b * d) * w02 + (a * c - b * d) * w2) / (w2 * w2 + 2.0 * (c * c - d * d) * w2 + w02 * w02)
    return np.sqrt(2.0 / np.pi) * p


def kernel_value(coeffs, tau):
    ar, cr, ac, bc, cc, dc, _ = coeffs
    tau = np.abs(np.asarray(tau, dtype=np.float64))
    k = np.zeros_like(tau)
    for a, c in zip(ar, cr):
        k += a * np.exp(-c * tau)
    for a, b, c, d in zip(ac, bc, cc, dc):
        k += np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))
    return k


def mean_value(mean_kind, mean_params, t):
    if mean_kind == 1:  # mean_models.py:24-31 LinearModel(slope, intercept)
        return mean_params[0] * np.asarray(t) + mean_params[1]
    return np.full(len(t), float(mean_params[0]))


def dense_loglike(t, y, dy, coeffs, mean_kind=0, mean_params=(0.0,)):
    """float64 dense Cholesky lnL.  Returns -inf if K is not positive definite."""
    t = np.asarray(t, dtype=np.float64)
    r = np.asarray(y, dtype=np.float64) - mean_value(mean_kind, mean_params, t)
    K = kernel_value(coeffs, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += (np.asarray(dy, dtype=np.float64) + 1e-12) ** 2 + coeffs[6]
    try:
        Lc = np.linalg.cholesky(K)
    except np.linalg.LinAlgError:
        return -np.inf
    from scipy.linalg import solve_triangular
    z = solve_triangular(Lc, r, lower=True)
    logdet = 2.0 * np.sum(np.log(np.diag(Lc)))
    return float(-0.5 * (z @ z + logdet + len(t) * np.log(2.0 * np.pi)))


def dense_loglike_mp(t, y, dy, coeffs, mean_kind=0, mean_params=(0.0,), dps=50):
    """mpmath dense Cholesky lnL at ``dps`` digits (small N only)."""
    import mpmath as mp
    with mp.workdps(dps):
        ar, cr, ac, bc, cc, dc, jitter = coeffs
        tt = [mp.mpf(float(v)) for v in t]
        n = len(tt)
        mu = mean_value(mean_kind, mean_params, np.asarray(t, dtype=np.float64))
        if mean_kind == 1:
            mu = [mp.mpf(float(mean_params[0])) * tv + mp.mpf(float(mean_params[1])) for tv in tt]
        r = mp.matrix([mp.mpf(float(y[i])) - mp.mpf(mu[i]) for i in range(n)])
        K = mp.matrix(n, n)
        for i in range(n):
            for j in range(i + 1):
                tau = abs(tt[i] - tt[j])
                k = mp.mpf(0)
                for a, c in zip(ar, cr):
                    k += mp.mpf(float(a)) * mp.exp(-mp.mpf(float(c)) * tau)
                for a, b, c, d in zip(ac, bc, cc, dc):
                    k += mp.exp(-mp.mpf(float(c)) * tau) * (
                        mp.mpf(float(a)) * mp.cos(mp.mpf(float(d)) * tau)
                        + mp.mpf(float(b)) * mp.sin(mp.mpf(float(d)) * tau))
                K[i, j] = k
                K[j, i] = k
            # sigma = dy + 1e-12 evaluated in float64 exactly as gpmodelling.py:54 does
            K[i, i] += mp.mpf(float(np.float64(dy[i]) + np.float64(1e-12))) ** 2 + mp.mpf(float(jitter))
        Lc = mp.cholesky(K)
        z = _forward_sub(Lc, r, n)
        dot = sum(zv * zv for zv in z)
        logdet = 2 * sum(mp.log(Lc[i, i]) for i in range(n))
        return float(-(dot + logdet + n * mp.log(2 * mp.pi)) / 2)


def _forward_sub(Lc, r, n):
    z = []
    for i in range(n):
        s = r[i]
        for j in range(i):
            s -= Lc[i, j] * z[j]
        z.append(s / Lc[i, i])
    return z


def log_prior(kinds, params_full, bounds):
    """celerite box prior + BendingPowerlaw.log_prior (celerite_models.py:85-90)."""
    p = np.asarray(params_full, dtype=np.float64)
    b = np.asarray(bounds, dtype=np.float64)
    if not np.all((p >= b[:, 0]) & (p <= b[:, 1])):
        return -np.inf
    off = 0
    for kind in kinds:
        if kind == K_BPL and p[off] < p[off + 1]:
            return -np.inf
        if kind == K_COMPLEX4 and p[off] + p[off + 2] < p[off + 1] + p[off + 3]:
            return -np.inf  # celerite ComplexTerm.log_prior
        off += NPARAMS[kind]
    return 0.0


def dense_predict(t, y, dy, coeffs, mean_kind=0, mean_params=(0.0,)):
    """celerite.GP.predict(y, return_var=True) at the training times by dense algebra:
    mu = mean + K_s K^-1 r, var = k(0) - diag(K_s K^-1 K_s), K_s the kernel matrix without
    the noise diagonal (what gpmodelling.py:366 obtains from celerite)."""
    t = np.asarray(t, dtype=np.float64)
    mean = mean_value(mean_kind, mean_params, t)
    r = np.asarray(y, dtype=np.float64) - mean
    Ks = kernel_value(coeffs, t[:, None] - t[None, :])
    K = Ks.copy()
    K[np.diag_indices_from(K)] += (np.asarray(dy, dtype=np.float64) + 1e-12) ** 2 + coeffs[6]
    sol = np.linalg.solve(K, np.column_stack([r, Ks]))
    mu = mean + Ks @ sol[:, 0]
    var = kernel_value(coeffs, 0.0) - np.sum(Ks * sol[:, 1:], axis=0)
    return mu, var
