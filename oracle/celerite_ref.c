/*
 * oracle/celerite_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64) of the log-likelihood hot path of
 * andresgur/mind_the_gaps:
 *
 *   GPModelling.__init__        /root/reference/mind_the_gaps/gpmodelling.py:47-59
 *                               (yerr = dy + 1e-12, line 54)
 *   GPModelling._log_probability            gpmodelling.py:127-152
 *   GPModelling._neg_log_like               gpmodelling.py:155-169
 *   coefficient builders        mind_the_gaps/models/celerite_models.py:7-90
 *   linear mean                 mind_the_gaps/models/mean_models.py:24-31
 *
 * The arithmetic of that path lives in the third-party dependency
 * celerite (pin: celerite>=0.4.2, /root/reference/pyproject.toml:16), whose
 * source is NOT under /root/reference and is not installed in this image.  Its
 * published algorithm (Foreman-Mackey, Agol, Ambikasaran & Angus 2017,
 * AJ 154, 220, eqs. 44-50 and the celerite 0.4 Python/C++ API, restated in
 * SURVEY.md Appendix A) is restated here the way celerite runs it: one sweep
 * that materialises U, V, phi, W, D (`compute`), one sweep for the forward
 * solve (`dot_solve`), then lnL = -1/2 (r^T K^-1 r + sum ln D + N ln 2pi).
 *
 * PARITY STATUS: "parity unpinned" at the lnL boundary -- the reference's
 * tests assert no log-likelihood value and celerite cannot be run here.  This
 * file is pinned instead (tests/test_oracle.py) against
 *   (1) the dense-covariance definition lnL = -1/2(r^T K^-1 r + ln det K + N ln 2pi)
 *       in float64 (LAPACK) and 50-digit mpmath (oracle/dense.py),
 *   (2) the closed-form PSD known answers of the reference's own
 *       tests/models_test.py:14-102 for the coefficient builders.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* term kind tags -- numerically identical to include/mtg.h */
enum {
    K_REAL = 0,      /* celerite.terms.RealTerm(log_a, log_c)                       */
    K_COMPLEX3 = 1,  /* celerite.terms.ComplexTerm(log_a, log_c, log_d)  (b = 0)    */
    K_COMPLEX4 = 2,  /* celerite.terms.ComplexTerm(log_a, log_b, log_c, log_d)      */
    K_SHO = 3,       /* celerite.terms.SHOTerm(log_S0, log_Q, log_omega0)           */
    K_MATERN32 = 4,  /* celerite.terms.Matern32Term(log_sigma, log_rho; eps)        */
    K_JITTER = 5,    /* celerite.terms.JitterTerm(log_sigma)                        */
    K_DRW = 6,       /* celerite_models.py:55-68  DampedRandomWalk                  */
    K_LORENTZIAN = 7,/* celerite_models.py:7-34   Lorentzian                        */
    K_COSINUS = 8,   /* celerite_models.py:36-52  Cosinus                           */
    K_BPL = 9        /* celerite_models.py:71-90  BendingPowerlaw                   */
};

ORACLE_API int oracle_term_nparams(int kind)
{
    switch (kind) {
    case K_REAL: return 2;
    case K_COMPLEX3: return 3;
    case K_COMPLEX4: return 4;
    case K_SHO: return 3;
    case K_MATERN32: return 2;
    case K_JITTER: return 1;
    case K_DRW: return 2;
    case K_LORENTZIAN: return 3;
    case K_COSINUS: return 2;
    case K_BPL: return 3;
    default: return -1;
    }
}

/*
 * theta (log-parameters of every term, in `+` order) -> the six celerite
 * coefficient arrays in TermSum concatenation order, plus the summed jitter.
 * Arrays must have room for 2 entries per term (real) / 1 per term (complex).
 */
ORACLE_API int oracle_build_coeffs(int nterms, const int *kinds, const double *extra,
                                   const double *p, int *jr_out, double *ar, double *cr,
                                   int *jc_out, double *ac, double *bc, double *cc,
                                   double *dc, double *jitter_out)
{
    int jr = 0, jc = 0;
    double jitter = 0.0;
    for (int i = 0; i < nterms; ++i) {
        switch (kinds[i]) {
        case K_REAL:
            ar[jr] = exp(p[0]); cr[jr] = exp(p[1]); ++jr;
            break;
        case K_COMPLEX3:
            ac[jc] = exp(p[0]); bc[jc] = 0.0; cc[jc] = exp(p[1]); dc[jc] = exp(p[2]); ++jc;
            break;
        case K_COMPLEX4:
            ac[jc] = exp(p[0]); bc[jc] = exp(p[1]); cc[jc] = exp(p[2]); dc[jc] = exp(p[3]); ++jc;
            break;
        case K_SHO: { /* SURVEY.md Appendix A.2 (celerite SHOTerm) */
            double S0 = exp(p[0]), Q = exp(p[1]), w0 = exp(p[2]);
            if (Q < 0.5) {
                double f = sqrt(1.0 - 4.0 * Q * Q);
                ar[jr] = 0.5 * S0 * w0 * Q * (1.0 + 1.0 / f);
                cr[jr] = 0.5 * w0 / Q * (1.0 - f); ++jr;
                ar[jr] = 0.5 * S0 * w0 * Q * (1.0 - 1.0 / f);
                cr[jr] = 0.5 * w0 / Q * (1.0 + f); ++jr;
            } else {
                double f = sqrt(4.0 * Q * Q - 1.0);
                ac[jc] = S0 * w0 * Q; bc[jc] = S0 * w0 * Q / f;
                cc[jc] = 0.5 * w0 / Q; dc[jc] = 0.5 * w0 / Q * f; ++jc;
            }
            break;
        }
        case K_MATERN32: {
            double eps = extra ? extra[i] : 0.01;
            double w0 = sqrt(3.0) * exp(-p[1]);
            double S0 = exp(2.0 * p[0]) / w0;
            ac[jc] = w0 * S0; bc[jc] = w0 * w0 * S0 / eps; cc[jc] = w0; dc[jc] = eps; ++jc;
            break;
        }
        case K_JITTER:
            jitter += exp(2.0 * p[0]);
            break;
        case K_DRW: /* celerite_models.py:58-66: a = S0, c = 0.5*w0/Q, Q = 1/2 */
            ar[jr] = exp(p[0]); cr[jr] = 0.5 * exp(p[1]) / 0.5; ++jr;
            break;
        case K_LORENTZIAN: /* celerite_models.py:9-31: real (0,0) + complex (S0,0,w0/2Q,w0) */
            ar[jr] = 0.0; cr[jr] = 0.0; ++jr;
            ac[jc] = exp(p[0]); bc[jc] = 0.0;
            cc[jc] = 0.5 * exp(p[2]) / exp(p[1]); dc[jc] = exp(p[2]); ++jc;
            break;
        case K_COSINUS: /* celerite_models.py:39-52 */
            ac[jc] = exp(p[0]); bc[jc] = 0.0; cc[jc] = 0.0; dc[jc] = exp(p[1]); ++jc;
            break;
        case K_BPL: /* celerite_models.py:77-83 */
            ac[jc] = exp(p[0]); bc[jc] = exp(p[1]); cc[jc] = exp(p[2]); dc[jc] = exp(p[2]); ++jc;
            break;
        default:
            return -1;
        }
        p += oracle_term_nparams(kinds[i]);
    }
    *jr_out = jr; *jc_out = jc; *jitter_out = jitter;
    return 0;
}

/*
 * Box prior of celerite's Model.log_prior (0 inside [lo, hi] for every
 * parameter, -inf outside) plus BendingPowerlaw.log_prior
 * (celerite_models.py:85-90: -inf when log_S0 < log_Q) and celerite's
 * ComplexTerm.log_prior (log_a + log_c < log_b + log_d -> -inf).  bounds = [PF][2],
 * +-inf for an open side.  Returns 0.0 or -INFINITY.
 */
ORACLE_API double oracle_log_prior(int nterms, const int *kinds, int PF, const double *p,
                                   const double *bounds)
{
    for (int k = 0; k < PF; ++k)
        if (!(p[k] >= bounds[2 * k] && p[k] <= bounds[2 * k + 1])) return -INFINITY;
    int off = 0;
    for (int i = 0; i < nterms; ++i) {
        if (kinds[i] == K_BPL && p[off] < p[off + 1]) return -INFINITY;
        /* celerite ComplexTerm.log_prior, 4-parameter form: a c >= b d in log space */
        if (kinds[i] == K_COMPLEX4 && p[off] + p[off + 2] < p[off + 1] + p[off + 3]) return -INFINITY;
        off += oracle_term_nparams(kinds[i]);
    }
    return 0.0;
}

/*
 * celerite CholeskySolver.compute + log_determinant + dot_solve, restated
 * (SURVEY.md Appendix A.3).  work must hold (4*J + 1) * N doubles.
 * status: 0 ok, 2 non-positive pivot (celerite raises LinAlgError),
 * 3 non-finite result (celerite returns -inf).
 */
ORACLE_API double oracle_loglike_coeffs(long N, const double *t, const double *y, const double *dy,
                                        int jr, const double *ar, const double *cr, int jc,
                                        const double *ac, const double *bc, const double *cc,
                                        const double *dc, double jitter, int mean_kind,
                                        const double *mean_params, double *work, int *status)
{
    const int J = jr + 2 * jc;
    double *U = work, *V = U + (size_t)J * N, *phi = V + (size_t)J * N, *W = phi + (size_t)J * N,
           *D = W + (size_t)J * N;
    double S[32 * 32];
    double f[32];
    if (J > 32) { *status = -1; return NAN; }
    *status = 0;

    double asum = jitter;
    for (int j = 0; j < jr; ++j) asum += ar[j];
    for (int k = 0; k < jc; ++k) asum += ac[k];

    /* ---- compute(): generators --------------------------------------- */
    for (long n = 0; n < N; ++n) {
        double *Un = U + (size_t)n * J, *Vn = V + (size_t)n * J, *pn = phi + (size_t)n * J;
        double dx = n > 0 ? t[n] - t[n - 1] : 0.0;
        for (int j = 0; j < jr; ++j) {
            Un[j] = ar[j]; Vn[j] = 1.0; pn[j] = exp(-cr[j] * dx);
        }
        for (int k = 0; k < jc; ++k) {
            double cd = cos(dc[k] * t[n]), sd = sin(dc[k] * t[n]);
            Un[jr + 2 * k] = ac[k] * cd + bc[k] * sd;
            Un[jr + 2 * k + 1] = ac[k] * sd - bc[k] * cd;
            Vn[jr + 2 * k] = cd;
            Vn[jr + 2 * k + 1] = sd;
            pn[jr + 2 * k] = pn[jr + 2 * k + 1] = exp(-cc[k] * dx);
        }
    }
    /* ---- compute(): factorisation K = L D L^T ------------------------- */
    memset(S, 0, sizeof(double) * (size_t)J * J);
    double logdet = 0.0;
    for (long n = 0; n < N; ++n) {
        const double *Un = U + (size_t)n * J, *Vn = V + (size_t)n * J, *pn = phi + (size_t)n * J;
        double *Wn = W + (size_t)n * J;
        double yerr = dy[n] + 1e-12; /* gpmodelling.py:54 */
        double Dn = yerr * yerr + asum;
        if (n > 0) {
            const double *Wp = W + (size_t)(n - 1) * J;
            double Dp = D[n - 1];
            for (int i = 0; i < J; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s = pn[i] * pn[j] * (S[i * J + j] + Dp * Wp[i] * Wp[j]);
                    S[i * J + j] = s; S[j * J + i] = s;
                }
        }
        for (int i = 0; i < J; ++i) {
            double q = 0.0;
            for (int j = 0; j < J; ++j) q += S[i * J + j] * Un[j];
            Wn[i] = Vn[i] - q;
            Dn -= Un[i] * q;
        }
        if (!(Dn > 0.0)) { *status = 2; return -INFINITY; }
        D[n] = Dn;
        for (int i = 0; i < J; ++i) Wn[i] /= Dn;
        logdet += log(Dn);
    }
    /* ---- dot_solve(): z = L^-1 r, r^T K^-1 r = sum z^2 / D ------------- */
    memset(f, 0, sizeof(double) * J);
    double dot = 0.0, zprev = 0.0;
    for (long n = 0; n < N; ++n) {
        const double *Un = U + (size_t)n * J, *pn = phi + (size_t)n * J;
        double mu = mean_kind == 1 ? mean_params[0] * t[n] + mean_params[1] : mean_params[0];
        double z = y[n] - mu;
        if (n > 0) {
            const double *Wp = W + (size_t)(n - 1) * J;
            for (int i = 0; i < J; ++i) {
                f[i] = pn[i] * (f[i] + Wp[i] * zprev);
                z -= Un[i] * f[i];
            }
        }
        dot += z * z / D[n];
        zprev = z;
    }
    double ll = -0.5 * (dot + logdet + (double)N * log(2.0 * M_PI));
    if (!isfinite(ll)) { *status = 3; return -INFINITY; }
    return ll;
}

/*
 * The same recurrences in ONE sweep with nothing materialised ("fused variant"):
 * per sample the generators, the S update, D, W, the forward-solve step and the two running sums.  This is
 * what a CPU implementation written for the likelihood alone would do, and the CPU baseline bench.py times
 * (celerite itself stores U, V, phi, W, D and sweeps twice: oracle_loglike_coeffs above).  J is a
 * compile-time constant in the specialised copies below so that the small loops unroll.
 */
static inline __attribute__((always_inline)) double
fused_impl(const int J, long N, const double *t, const double *y, const double *dy, int jr, const double *ar,
           const double *cr, int jc, const double *ac, const double *bc, const double *cc, const double *dc,
           double jitter, int mean_kind, const double *mean_params, int *status)
{
    double S[32 * 32], f[32], W[32], U[32], V[32], ph[32];
    *status = 0;
    double asum = jitter;
    for (int j = 0; j < jr; ++j) asum += ar[j];
    for (int k = 0; k < jc; ++k) asum += ac[k];
    memset(S, 0, sizeof(double) * (size_t)J * J);
    memset(f, 0, sizeof(double) * J);
    memset(W, 0, sizeof(double) * J);
    double logdet = 0.0, dot = 0.0, zprev = 0.0, Dp = 1.0;
    for (long n = 0; n < N; ++n) {
        const double dx = n > 0 ? t[n] - t[n - 1] : 0.0;
        for (int j = 0; j < jr; ++j) { U[j] = ar[j]; V[j] = 1.0; ph[j] = exp(-cr[j] * dx); }
        for (int k = 0; k < jc; ++k) {
            const double cd = cos(dc[k] * t[n]), sd = sin(dc[k] * t[n]), e = exp(-cc[k] * dx);
            U[jr + 2 * k] = ac[k] * cd + bc[k] * sd;
            U[jr + 2 * k + 1] = ac[k] * sd - bc[k] * cd;
            V[jr + 2 * k] = cd;
            V[jr + 2 * k + 1] = sd;
            ph[jr + 2 * k] = ph[jr + 2 * k + 1] = e;
        }
        const double yerr = dy[n] + 1e-12; /* gpmodelling.py:54 */
        const double mu = mean_kind == 1 ? mean_params[0] * t[n] + mean_params[1] : mean_params[0];
        double Dn = yerr * yerr + asum, z = y[n] - mu;
        if (n > 0) {
            for (int i = 0; i < J; ++i) {
                for (int j = 0; j <= i; ++j) {
                    const double s = ph[i] * ph[j] * (S[i * J + j] + Dp * W[i] * W[j]);
                    S[i * J + j] = s; S[j * J + i] = s;
                }
                f[i] = ph[i] * (f[i] + W[i] * zprev);
                z -= U[i] * f[i];
            }
        }
        for (int i = 0; i < J; ++i) {
            double q = 0.0;
            for (int j = 0; j < J; ++j) q += S[i * J + j] * U[j];
            W[i] = V[i] - q;
            Dn -= U[i] * q;
        }
        if (!(Dn > 0.0)) { *status = 2; return -INFINITY; }
        for (int i = 0; i < J; ++i) W[i] /= Dn;
        logdet += log(Dn);
        dot += z * z / Dn;
        zprev = z;
        Dp = Dn;
    }
    const double ll = -0.5 * (dot + logdet + (double)N * log(2.0 * M_PI));
    if (!isfinite(ll)) { *status = 3; return -INFINITY; }
    return ll;
}

#define FUSED_ARGS N, t, y, dy, jr, ar, cr, jc, ac, bc, cc, dc, jitter, mean_kind, mean_params, status
ORACLE_API double oracle_loglike_coeffs_fused(long N, const double *t, const double *y, const double *dy,
                                              int jr, const double *ar, const double *cr, int jc,
                                              const double *ac, const double *bc, const double *cc,
                                              const double *dc, double jitter, int mean_kind,
                                              const double *mean_params, int *status)
{
    const int J = jr + 2 * jc;
    switch (J) {
    case 1: return fused_impl(1, FUSED_ARGS);
    case 2: return fused_impl(2, FUSED_ARGS);
    case 3: return fused_impl(3, FUSED_ARGS);
    case 4: return fused_impl(4, FUSED_ARGS);
    case 5: return fused_impl(5, FUSED_ARGS);
    case 6: return fused_impl(6, FUSED_ARGS);
    case 8: return fused_impl(8, FUSED_ARGS);
    case 10: return fused_impl(10, FUSED_ARGS);
    default:
        if (J > 32) { *status = -1; return NAN; }
        return fused_impl(J, FUSED_ARGS);
    }
}

/*
 * _log_probability / log_likelihood for a batch of parameter vectors.
 * params: [B][PF] FULL parameter vectors (kernel terms in `+` order, then the
 * mean parameters: 1 for constant, 2 = (slope, intercept) for linear).
 * t: [N] shared; y, dy: [L][N]; lc_index: [B] or NULL (=0).
 * add_prior != 0 mirrors _log_probability (gpmodelling.py:147-152): prior
 * first, likelihood skipped when the prior is -inf (status 1).
 * nthreads <= 1 runs serially.
 */
static int logprob_batch_impl(int fused, long N, long L, const double *t, const double *y,
                              const double *dy, int nterms, const int *kinds,
                              const double *extra, int mean_kind, int PF,
                              const double *bounds, long B, const double *params,
                              const int *lc_index, int add_prior, int nthreads,
                              double *out, int *status)
{
    int bad = 0;
    (void)L;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        int jmax = 0; /* widest expansion: 2 slots per SHO/complex, 3 for Lorentzian */
        for (int i = 0; i < nterms; ++i) jmax += kinds[i] == K_LORENTZIAN ? 3 : 2;
        if (jmax > 32) jmax = 32;
        double *work = (double *)malloc(sizeof(double) * (size_t)(4 * jmax + 1) * (size_t)N);
        double ar[24], cr[24], ac[12], bc[12], cc[12], dc[12];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long b = 0; b < B; ++b) {
            const double *p = params + (size_t)b * PF;
            long lc = lc_index ? lc_index[b] : 0;
            int jr, jc, st = 0;
            double jitter;
            if (add_prior && oracle_log_prior(nterms, kinds, PF, p, bounds) != 0.0) {
                out[b] = -INFINITY; status[b] = 1; continue;
            }
            if (oracle_build_coeffs(nterms, kinds, extra, p, &jr, ar, cr, &jc, ac, bc, cc, dc,
                                    &jitter) != 0) {
                bad = 1; continue;
            }
            int nk = 0;
            for (int i = 0; i < nterms; ++i) nk += oracle_term_nparams(kinds[i]);
            out[b] = fused
                         ? oracle_loglike_coeffs_fused(N, t, y + (size_t)lc * N, dy + (size_t)lc * N, jr, ar, cr, jc,
                                                       ac, bc, cc, dc, jitter, mean_kind, p + nk, &st)
                         : oracle_loglike_coeffs(N, t, y + (size_t)lc * N, dy + (size_t)lc * N, jr, ar, cr, jc,
                                                 ac, bc, cc, dc, jitter, mean_kind, p + nk, work, &st);
            status[b] = st;
        }
        free(work);
    }
    return bad ? -1 : 0;
}

ORACLE_API int oracle_logprob_batch(long N, long L, const double *t, const double *y,
                                    const double *dy, int nterms, const int *kinds,
                                    const double *extra, int mean_kind, int PF,
                                    const double *bounds, long B, const double *params,
                                    const int *lc_index, int add_prior, int nthreads,
                                    double *out, int *status)
{
    return logprob_batch_impl(0, N, L, t, y, dy, nterms, kinds, extra, mean_kind, PF, bounds, B, params, lc_index,
                              add_prior, nthreads, out, status);
}

/* the same with the fused one-sweep recurrence (oracle_loglike_coeffs_fused) */
ORACLE_API int oracle_logprob_batch_fused(long N, long L, const double *t, const double *y,
                                          const double *dy, int nterms, const int *kinds,
                                          const double *extra, int mean_kind, int PF,
                                          const double *bounds, long B, const double *params,
                                          const int *lc_index, int add_prior, int nthreads,
                                          double *out, int *status)
{
    return logprob_batch_impl(1, N, L, t, y, dy, nterms, kinds, extra, mean_kind, PF, bounds, B, params, lc_index,
                              add_prior, nthreads, out, status);
}

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
