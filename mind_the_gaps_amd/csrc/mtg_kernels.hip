// mtg_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels of the celerite
// log-likelihood hot path.
//
//   mtg_lc_setup   : celerite.GP.compute(t, yerr) (called with yerr = dy + 1e-12
//                    at reference gpmodelling.py:54): sigma^2 = yerr^2, dx_n.
//   mtg_prepare    : set_parameter_vector + log_prior + Term.coefficients
//                    (gpmodelling.py:147-151, celerite_models.py:7-90,
//                    celerite built-in terms): theta -> prior verdict and the
//                    (a, c | a, b, c, d) coefficient columns, grouped by
//                    structure signature (SHOTerm: 1 complex or 2 real terms).
//   mtg_solve<NR,NC>: celerite CholeskySolver.compute + log_determinant +
//                    dot_solve fused into ONE sweep over the N samples with the
//                    whole semiseparable state (S: J(J+1)/2, W, f) in VGPRs;
//                    one lane = one (theta, light curve) evaluation, one 8-byte
//                    store per evaluation.  Nothing per-sample is written.
//
// Roofline: nominally HBM (24 N bytes of t, y, sigma^2 per evaluation), in
// practice FP64 VALU issue; there is no dense contraction here, so no MFMA.
#include "mtg_device.h"

#include <math.h>

#define MTG_LN_2PI 1.8378770664093454835606594728112

// ---------------------------------------------------------------------------
// light-curve set-up
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mtg_lc_setup_kernel(int64_t N, int64_t L, int64_t t_rows, const double *__restrict__ t,
                    const double *__restrict__ yerr, double *__restrict__ dx,
                    double *__restrict__ var)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < L * N; i += stride) {
        const double s = yerr[i];  // celerite squares the yerr handed to compute()
        var[i] = s * s;
    }
    for (int64_t i = i0; i < t_rows * N; i += stride) {
        const int64_t n = i % N;
        dx[i] = n == 0 ? 0.0 : t[i] - t[i - 1];
    }
}

void mtg_launch_lc_setup(int64_t N, int64_t L, int64_t t_rows, const double *t, const double *yerr,
                         double *dx, double *var, hipStream_t stream)
{
    int64_t blocks = (L * N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mtg_lc_setup_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, N, L,
                       t_rows, t, yerr, dx, var);
}

// ---------------------------------------------------------------------------
// theta -> prior + coefficients
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mtg_prepare_kernel(MtgPrepArgs a)
{
    const MtgModel &m = a.model;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < a.B;
    const double *th = a.theta + (live ? e : 0) * m.P;
    auto par = [&](int k) -> double {
        const int s = m.src[k];
        return s >= 0 ? th[s] : m.defaults[k];
    };

    bool ok = live;
    if (live && a.add_prior) {
        // celerite Model.log_prior: every parameter, frozen ones included
        for (int k = 0; k < m.PF; ++k) {
            const double v = par(k);
            ok = ok && (v >= m.lo[k]) && (v <= m.hi[k]);
        }
        // BendingPowerlaw.log_prior, celerite_models.py:85-90
        // celerite ComplexTerm.log_prior (4-parameter form): log_a + log_c >= log_b + log_d
        for (int i = 0; i < m.nterms; ++i) {
            const int o = m.poff[i];
            if (m.kinds[i] == MTG_TERM_BPL) ok = ok && !(par(o) < par(o + 1));
            if (m.kinds[i] == MTG_TERM_COMPLEX4)
                ok = ok && !(par(o) + par(o + 2) < par(o + 1) + par(o + 3));
        }
    }
    if (live) {
        a.status[e] = ok ? MTG_ST_OK : MTG_ST_PRIOR;
        if (!ok) a.out[e] = -INFINITY;
    }

    int nover = 0;
    if (ok) {
        MtgCoefLayout lay{m.nr_max, m.nc_max};
        double *c = a.coef + e;
        const int64_t cs = a.cstride;
        int ir = 0, ic = 0;
        double asum = 0.0;
        for (int i = 0; i < m.nterms; ++i) {
            const int o = m.poff[i];
            switch (m.kinds[i]) {
            case MTG_TERM_REAL: {
                const double av = exp(par(o));
                c[lay.ar(ir) * cs] = av; c[lay.cr(ir) * cs] = exp(par(o + 1)); ++ir; asum += av;
                break;
            }
            case MTG_TERM_COMPLEX3: {
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = exp(par(o + 1)); c[lay.dc(ic) * cs] = exp(par(o + 2)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_COMPLEX4: {
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = exp(par(o + 1));
                c[lay.cc(ic) * cs] = exp(par(o + 2)); c[lay.dc(ic) * cs] = exp(par(o + 3)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_SHO: {
                const double S0 = exp(par(o)), Q = exp(par(o + 1)), w0 = exp(par(o + 2));
                if (Q < 0.5) {  // over-damped: two real terms
                    const double f = sqrt(1.0 - 4.0 * Q * Q);
                    const double a1 = 0.5 * S0 * w0 * Q * (1.0 + 1.0 / f);
                    const double a2 = 0.5 * S0 * w0 * Q * (1.0 - 1.0 / f);
                    c[lay.ar(ir) * cs] = a1; c[lay.cr(ir) * cs] = 0.5 * w0 / Q * (1.0 - f); ++ir;
                    c[lay.ar(ir) * cs] = a2; c[lay.cr(ir) * cs] = 0.5 * w0 / Q * (1.0 + f); ++ir;
                    asum += a1; asum += a2;
                    ++nover;
                } else {
                    const double f = sqrt(4.0 * Q * Q - 1.0);
                    const double av = S0 * w0 * Q;
                    c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = av / f;
                    c[lay.cc(ic) * cs] = 0.5 * w0 / Q; c[lay.dc(ic) * cs] = 0.5 * w0 / Q * f; ++ic;
                    asum += av;
                }
                break;
            }
            case MTG_TERM_MATERN32: {
                const double eps = m.extra[i];
                const double w0 = sqrt(3.0) * exp(-par(o + 1));
                const double S0 = exp(2.0 * par(o)) / w0;
                const double av = w0 * S0;
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = w0 * w0 * S0 / eps;
                c[lay.cc(ic) * cs] = w0; c[lay.dc(ic) * cs] = eps; ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_JITTER:
                asum += exp(2.0 * par(o));
                break;
            case MTG_TERM_DRW: {  // celerite_models.py:58-66, Q = 1/2
                const double av = exp(par(o));
                c[lay.ar(ir) * cs] = av; c[lay.cr(ir) * cs] = 0.5 * exp(par(o + 1)) / 0.5; ++ir;
                asum += av;
                break;
            }
            case MTG_TERM_LORENTZIAN: {
                // celerite_models.py:9-31.  The (a=0, c=0) real term the reference
                // returns has U = 0, so it never enters D_n or z_n: it is not
                // expanded (identical lnL, one rank less work).
                const double av = exp(par(o));
                const double w0 = exp(par(o + 2));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = 0.5 * w0 / exp(par(o + 1)); c[lay.dc(ic) * cs] = w0; ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_COSINUS: {  // celerite_models.py:39-52
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = 0.0; c[lay.dc(ic) * cs] = exp(par(o + 1)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_BPL: {  // celerite_models.py:77-83
                const double av = exp(par(o));
                const double w0 = exp(par(o + 2));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = exp(par(o + 1));
                c[lay.cc(ic) * cs] = w0; c[lay.dc(ic) * cs] = w0; ++ic;
                asum += av;
                break;
            }
            default:
                break;
            }
        }
        c[lay.asum() * cs] = asum;
        c[lay.mean(0) * cs] = par(m.nk);
        c[lay.mean(1) * cs] = m.mean_kind == MTG_MEAN_LINEAR ? par(m.nk + 1) : 0.0;
    }

    if (a.nsig > 1) {
        // wave-aggregated append of the evaluation index to its signature list
        const int lane = threadIdx.x & 63;
        for (int k = 0; k < a.nsig; ++k) {
            const bool mine = ok && nover == k;
            const unsigned long long mask = __ballot(mine);
            if (mask == 0ull) continue;
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(a.counts + k, __popcll(mask));
            base = __shfl(base, leader);
            if (mine)
                a.lists[(int64_t)k * a.cstride + base + __popcll(mask & ((1ull << lane) - 1ull))] =
                    (int)e;
        }
    }
}

void mtg_launch_prepare(const MtgPrepArgs &a, hipStream_t stream)
{
    const int64_t blocks = (a.B + 255) / 256;
    hipLaunchKernelGGL(mtg_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
}

// ---------------------------------------------------------------------------
// fused factorisation + forward solve, one lane per evaluation
// ---------------------------------------------------------------------------
__device__ __forceinline__ double mtg_exp(double x) { return exp(x); }
__device__ __forceinline__ void mtg_sincos(double x, double *s, double *c) { sincos(x, s, c); }

template <int NR, int NC>
__global__ void __launch_bounds__(64) mtg_solve_kernel(MtgSolveArgs a)
{
    constexpr int J = NR + 2 * NC;   // celerite rank
    constexpr int NT = NR + NC;      // distinct exp(-c dx) factors
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if (gid >= count) return;
    const int64_t e = a.list ? (int64_t)a.list[gid] : gid;
    if (!a.list && a.status[e] != MTG_ST_OK) return;  // prior said -inf

    // ---- coefficients of this evaluation -------------------------------
    const double *cf = a.coef + e;
    const int64_t cs = a.cstride;
    double ar[NR > 0 ? NR : 1], cr[NR > 0 ? NR : 1];
    double ac[NC > 0 ? NC : 1], bc[NC > 0 ? NC : 1], cc[NC > 0 ? NC : 1], dc[NC > 0 ? NC : 1];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        ar[j] = cf[a.lay.ar(j) * cs];
        cr[j] = cf[a.lay.cr(j) * cs];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        ac[k] = cf[a.lay.ac(k) * cs];
        bc[k] = cf[a.lay.bc(k) * cs];
        cc[k] = cf[a.lay.cc(k) * cs];
        dc[k] = cf[a.lay.dc(k) * cs];
    }
    const double asum = cf[a.lay.asum() * cs];
    const double mean0 = cf[a.lay.mean(0) * cs];
    const double mean1 = cf[a.lay.mean(1) * cs];
    const bool linear = a.mean_kind == MTG_MEAN_LINEAR;

    const int64_t lc = a.lc_index ? (int64_t)a.lc_index[e] : 0;
    const int64_t N = a.N;
    const double *yp = a.y + lc * N;
    const double *vp = a.var + lc * N;
    const double *dxp = a.dx + lc * a.t_stride;
    const double *tp = a.t + lc * a.t_stride;

    // ---- recurrence state (all statically indexed -> VGPRs) -------------
    double S[J * (J + 1) / 2];
    double Wt[J];  // V_n - S U_n  (W_n = Wt / D_n)
    double f[J];
    double cs_[NC > 0 ? NC : 1], sn_[NC > 0 ? NC : 1];  // cos/sin d_k (t_n - t_0)
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) S[i] = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) { Wt[i] = 0.0; f[i] = 0.0; }
#pragma unroll
    for (int k = 0; k < NC; ++k) { cs_[k] = 1.0; sn_[k] = 0.0; }

    double invD = 0.0, z = 0.0, dot = 0.0;
    double dprod = 1.0;  // running product of pivots, exponent kept in dexp
    int dexp = 0;
    bool bad = false;

    double dx_n = dxp[0], y_n = yp[0], v_n = vp[0], t_n = linear ? tp[0] : 0.0;
    for (int64_t n = 0; n < N; ++n) {
        const double dxc = dx_n, yc = y_n, vc = v_n, tc = t_n;
        if (n + 1 < N) {  // prefetch the next sample under this step's arithmetic
            dx_n = dxp[n + 1]; y_n = yp[n + 1]; v_n = vp[n + 1];
            if (linear) t_n = tp[n + 1];
        }
        // -- per-term propagators and generators (celerite U, V, phi) -----
        double ph[NT > 0 ? NT : 1];
        double U[J], V[J];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            ph[j] = mtg_exp(-cr[j] * dxc);
            U[j] = ar[j];
            V[j] = 1.0;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            ph[NR + k] = mtg_exp(-cc[k] * dxc);
            double sd, cd;
            mtg_sincos(dc[k] * dxc, &sd, &cd);
            // rotate (cos, sin) d_k (t - t_0) by d_k dx: the kernel depends on
            // time differences only, so the phase origin is free.
            const double cn = cs_[k] * cd - sn_[k] * sd;
            const double sn = sn_[k] * cd + cs_[k] * sd;
            cs_[k] = cn; sn_[k] = sn;
            U[NR + 2 * k] = ac[k] * cn + bc[k] * sn;
            U[NR + 2 * k + 1] = ac[k] * sn - bc[k] * cn;
            V[NR + 2 * k] = cn;
            V[NR + 2 * k + 1] = sn;
        }
        // -- S <- (phi phi^T) o (S + D W W^T) ;  f <- phi o (f + W z) ------
        const double zs = z * invD;
        double wd[J];
#pragma unroll
        for (int i = 0; i < J; ++i) wd[i] = Wt[i] * invD;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            const int ti = i < NR ? i : NR + (i - NR) / 2;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const int tj = j < NR ? j : NR + (j - NR) / 2;
                const double pp = ph[ti] * ph[tj];
                S[i * (i + 1) / 2 + j] = pp * fma(Wt[i], wd[j], S[i * (i + 1) / 2 + j]);
            }
            f[i] = ph[ti] * fma(Wt[i], zs, f[i]);
        }
        // -- D_n = A_n - U^T S U ; Wt = V - S U ; z_n = r_n - U^T f ---------
        double D = vc + asum;
        double zn = yc - (linear ? fma(mean0, tc, mean1) : mean0);
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                q = fma(S[hi * (hi + 1) / 2 + lo], U[j], q);
            }
            Wt[i] = V[i] - q;
            D = fma(-U[i], q, D);
            zn = fma(-U[i], f[i], zn);
        }
        bad = bad || !(D > 0.0);
        invD = 1.0 / D;
        z = zn;
        dot = fma(zn * zn, invD, dot);
        dprod *= D;
        if ((n & 3) == 3) {  // keep the pivot product in range: D in (1e-24, 1e22)
            int ex;
            dprod = frexp(dprod, &ex);
            dexp += ex;
        }
    }
    const double logdet = log(dprod) + (double)dexp * 0.69314718055994530942;
    double ll = -0.5 * (dot + logdet + (double)N * MTG_LN_2PI);
    int st = MTG_ST_OK;
    if (bad) { st = MTG_ST_NOTPD; ll = -INFINITY; }
    else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
    a.out[e] = ll;
    a.status[e] = st;
}

template <int NR, int NC>
static void mtg_launch_solve(const MtgSolveArgs &a, int64_t nlanes, hipStream_t stream)
{
    const int64_t blocks = (nlanes + 63) / 64;
    if (blocks <= 0) return;
    hipLaunchKernelGGL((mtg_solve_kernel<NR, NC>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
}

// Compiled structures: NR real + NC complex terms, J = NR + 2 NC <= MTG_MAX_J.
#define MTG_MAX_NR 10
#define MTG_MAX_NC 5
template <int NR, int NC, bool OK = (NR + NC > 0 && NR + 2 * NC <= MTG_MAX_J)>
struct MtgSel { static constexpr mtg_solve_launcher fn = mtg_launch_solve<NR, NC>; };
template <int NR, int NC>
struct MtgSel<NR, NC, false> { static constexpr mtg_solve_launcher fn = nullptr; };
#define MTG_ROW(nr)                                                                      \
    { MtgSel<(nr), 0>::fn, MtgSel<(nr), 1>::fn, MtgSel<(nr), 2>::fn, MtgSel<(nr), 3>::fn, \
      MtgSel<(nr), 4>::fn, MtgSel<(nr), 5>::fn }

static const mtg_solve_launcher mtg_solver_table[MTG_MAX_NR + 1][MTG_MAX_NC + 1] = {
    MTG_ROW(0), MTG_ROW(1), MTG_ROW(2), MTG_ROW(3), MTG_ROW(4), MTG_ROW(5),
    MTG_ROW(6), MTG_ROW(7), MTG_ROW(8), MTG_ROW(9), MTG_ROW(10)};

mtg_solve_launcher mtg_find_solver(int nr, int nc)
{
    if (nr < 0 || nc < 0 || nr > MTG_MAX_NR || nc > MTG_MAX_NC) return nullptr;
    return mtg_solver_table[nr][nc];
}
