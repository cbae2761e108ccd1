// mtg_timeparallel.h -- the log-likelihood of ONE evaluation spread over a whole
// wave: parallel in time, for small batches (a single light curve with a few
// hundred walkers, BASELINE configs[1], [2], [4]) where one lane per evaluation
// leaves the GPU idle and a half-step costs N serial recurrence steps.
//
// Formulation (prototype + derivation: proto/kalman_scan.py).  The celerite model is the
// state-space model of its stochastic differential equation: per real term a scalar
// Ornstein-Uhlenbeck state (F = e^{-c dx}, P_inf = a), per complex term a 2-d state
// with F = e^{-c dx} R(d dx) and stationary covariance P_inf = [[a, -b], [-b, p]];
// observation h picks the first component of every term; process noise
// Q = P_inf - F P_inf F^T.  Its Kalman filter yields exactly celerite's pivots
// D_n = h^T C_n h + sigma_n^2 and residuals z_n (lnL = -1/2 sum(ln 2 pi D_n + z_n^2/D_n)).
// The filter recursion is made parallel with the associative filtering elements of
// Sarkka & Garcia-Fernandez (2021): the N samples are cut into 64 chunks, one per lane;
//   pass 1  every lane composes the element (A, b, C, eta, J) of its chunk: a Kalman filter
//           started from (0, 0) that also carries A, eta and J along, O(J^2) per step;
//           started from (0, 0) that also carries A, eta and J along, O(J^2) per step, and
//           with it the chunk's likelihood given x_in = 0 (the filter's own pivots and residuals);
//   pass 2  an inclusive scan of the chunk elements across the lanes (Hillis-Steele through
//           LDS, log2(lanes) rounds of the general element combination, one J x J inverse each).
//           The combination carries the likelihood along (the later chunk's Gaussian likelihood in
//           x_in integrated against the earlier chunk's N(b, C)), so the last lane's element,
//           integrated against the state after sample 0, IS lnL: the samples are read once.
//   pass 3  only when that number is suspect (a pivot or determinant not positive, non-finite,
//           or cancellation beyond 1e3): every lane applies the prefix of the earlier chunks to
//           the state after sample 0 -- its chunk's start state -- runs the ordinary Kalman
//           filter over its chunk and accumulates ln prod D and sum z^2 / D; a reduction
//           finishes.  This pass also decides "not positive definite".
// Work is ~2x the serial sweep, depth ~N/lanes + log2(lanes) combinations instead of N.
#pragma once
#include "mtg_device.h"

// the chunk elements of a workgroup sit in LDS (160 KiB per CU)
// (rank 6: two buffers of them, see tp_combine_lds)
#define MTG_TP_PINGPONG(J) ((J) >= 6)
#define MTG_TP_LDS_DOUBLES(J, LANES) ((MTG_TP_PINGPONG(J) ? 2 : 1) * (LANES) * MTG_TP_ELEM(J))
#define MTG_TP_IN_LDS(J, LANES) (MTG_TP_LDS_DOUBLES(J, LANES) * 8 <= 150 * 1024)

// 256-entry tables here (2 KiB + 4 KiB): LDS is needed for the chunk elements (the rank-10 path,
// mtg_tp_big.h, keeps its elements in registers / global memory and asks for the 2048-entry tables)
#ifndef MTG_EXP_BITS
#define MTG_EXP_BITS 8
#define MTG_TRIG_BITS 8
#endif
#include "mtg_math.h"

#include <math.h>

#define MTG_LN_2PI 1.8378770664093454835606594728112

namespace {

template <int J> struct Sym {  // symmetric J x J, lower triangle
    double v[J * (J + 1) / 2];
    __device__ __forceinline__ double &operator()(int i, int j) { return i >= j ? v[i * (i + 1) / 2 + j] : v[j * (j + 1) / 2 + i]; }
    __device__ __forceinline__ double operator()(int i, int j) const { return i >= j ? v[i * (i + 1) / 2 + j] : v[j * (j + 1) / 2 + i]; }
};

// per-lane description of the state-space model
template <int NR, int NC> struct TpModel {
    double ar[NR > 0 ? NR : 1], cr[NR > 0 ? NR : 1];
    double ac[NC > 0 ? NC : 1], bc[NC > 0 ? NC : 1], cc[NC > 0 ? NC : 1], dc[NC > 0 ? NC : 1], pc[NC > 0 ? NC : 1];
};

// transition of one step: real terms phi; complex terms e * [[cs, -sn], [sn, cs]]
template <int NR, int NC> struct TpTrans {
    double phi[NR > 0 ? NR : 1];
    double ec[NC > 0 ? NC : 1], es[NC > 0 ? NC : 1];  // e cos(d dx), e sin(d dx)
};

// FAST: table exp / sincos (every d_k * dx of this wave is inside the table reduction's range)
template <int NR, int NC, bool FAST>
__device__ __forceinline__ void tp_transition(const TpModel<NR, NC> &M, double dx, TpTrans<NR, NC> &T,
                                              const MtgMathTables *tab)
{
#pragma unroll
    for (int j = 0; j < NR; ++j)
        T.phi[j] = FAST ? mtg_exp_cdx(-M.cr[j], M.cr[j] * -MTG_EXP_CSCALE, dx, tab) : exp(-M.cr[j] * dx);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        double e, s, c;
        if (FAST) {
            e = mtg_exp_cdx(-M.cc[k], M.cc[k] * -MTG_EXP_CSCALE, dx, tab);
            double r0 = 0.0;
            int m0 = 0;
            mtg_phase_step(M.dc[k], dx, r0, m0, &s, &c, tab);
        } else {
            e = exp(-M.cc[k] * dx);
            sincos(M.dc[k] * dx, &s, &c);
        }
        T.ec[k] = e * c;
        T.es[k] = e * s;
    }
}

// y <- F x  (x, y vectors of length J; in place allowed)
template <int NR, int NC>
__device__ __forceinline__ void tp_apply_F(const TpTrans<NR, NC> &T, double *x)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) x[j] *= T.phi[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double x0 = x[NR + 2 * k], x1 = x[NR + 2 * k + 1];
        x[NR + 2 * k] = T.ec[k] * x0 - T.es[k] * x1;
        x[NR + 2 * k + 1] = T.es[k] * x0 + T.ec[k] * x1;
    }
}

// X <- F X (X general J x J, row-major): F acts on the rows of X
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_left_F(const TpTrans<NR, NC> &T, double (&X)[J][J])
{
#pragma unroll
    for (int c = 0; c < J; ++c) {
        double col[J];
#pragma unroll
        for (int i = 0; i < J; ++i) col[i] = X[i][c];
        tp_apply_F<NR, NC>(T, col);
#pragma unroll
        for (int i = 0; i < J; ++i) X[i][c] = col[i];
    }
}

// Covariances are propagated as their deviation from the stationary one, Dv = C - P_inf: the
// prediction C <- F C F^T + Q with Q = P_inf - F P_inf F^T is then Dv <- F Dv F^T, done block by
// block of the (real | 2-d complex) structure of F on the lower triangle only.
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_sub_pinf(const TpModel<NR, NC> &M, Sym<J> &C)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) C(j, j) -= M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
        C(o, o) -= M.ac[k]; C(o + 1, o) += M.bc[k]; C(o + 1, o + 1) -= M.pc[k];
    }
}

template <int NR, int NC, int J>
__device__ __forceinline__ void tp_add_pinf(const TpModel<NR, NC> &M, Sym<J> &C)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) C(j, j) += M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
        C(o, o) += M.ac[k]; C(o + 1, o) -= M.bc[k]; C(o + 1, o + 1) += M.pc[k];
    }
}

template <int NR, int NC, int J>
__device__ __forceinline__ void tp_predict_dev(const TpTrans<NR, NC> &T, Sym<J> &Dv)
{
    // real x real
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Dv(i, j) *= T.phi[i] * T.phi[j];
    // complex x real: a 2-vector rotated by F_k and scaled by phi_j
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const double ecj = T.ec[k] * T.phi[j], esj = T.es[k] * T.phi[j];
            const double x0 = Dv(o, j), x1 = Dv(o + 1, j);
            Dv(o, j) = ecj * x0 - esj * x1;
            Dv(o + 1, j) = esj * x0 + ecj * x1;
        }
    }
    // complex k x complex l < k: full 2 x 2 block B <- F_k B F_l^T
#pragma unroll
    for (int k = 1; k < NC; ++k) {
        const int ok = NR + 2 * k;
#pragma unroll
        for (int l = 0; l < k; ++l) {
            const int ol = NR + 2 * l;
            const double b00 = Dv(ok, ol), b01 = Dv(ok, ol + 1), b10 = Dv(ok + 1, ol), b11 = Dv(ok + 1, ol + 1);
            const double y00 = T.ec[k] * b00 - T.es[k] * b10, y01 = T.ec[k] * b01 - T.es[k] * b11;
            const double y10 = T.es[k] * b00 + T.ec[k] * b10, y11 = T.es[k] * b01 + T.ec[k] * b11;
            Dv(ok, ol) = y00 * T.ec[l] - y01 * T.es[l];
            Dv(ok, ol + 1) = y00 * T.es[l] + y01 * T.ec[l];
            Dv(ok + 1, ol) = y10 * T.ec[l] - y11 * T.es[l];
            Dv(ok + 1, ol + 1) = y10 * T.es[l] + y11 * T.ec[l];
        }
    }
    // complex k x complex k: symmetric 2 x 2 block
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
        const double d00 = Dv(o, o), d10 = Dv(o + 1, o), d11 = Dv(o + 1, o + 1);
        const double y00 = T.ec[k] * d00 - T.es[k] * d10, y01 = T.ec[k] * d10 - T.es[k] * d11;
        const double y10 = T.es[k] * d00 + T.ec[k] * d10, y11 = T.es[k] * d10 + T.ec[k] * d11;
        Dv(o, o) = y00 * T.ec[k] - y01 * T.es[k];
        Dv(o + 1, o) = y10 * T.ec[k] - y11 * T.es[k];
        Dv(o + 1, o + 1) = y10 * T.es[k] + y11 * T.ec[k];
    }
}

// h^T x and C h for the 0/1 observation vector h (first component of every term)
template <int NR, int NC>
__device__ __forceinline__ double tp_h_dot(const double *x)
{
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) s += x[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) s += x[NR + 2 * k];
    return s;
}

template <int NR, int NC, int J>
__device__ __forceinline__ void tp_C_h(const Sym<J> &C, double *out)
{
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) s += C(i, j);
#pragma unroll
        for (int k = 0; k < NC; ++k) s += C(i, NR + 2 * k);
        out[i] = s;
    }
}

// One Kalman filter step from the FILTERED state of the previous sample (mean m, covariance
// P_inf + Dv): predict with T, update with (r, R).  Returns the pivot D, its reciprocal, the
// residual z and the gain kd = (P_pred h) / D.
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_filter_step(const TpModel<NR, NC> &M, const TpTrans<NR, NC> &T, double r, double R,
                                               double *m, Sym<J> &Dv, double &D, double &inv, double &z, double *kd)
{
    tp_apply_F<NR, NC>(T, m);
    tp_predict_dev<NR, NC, J>(T, Dv);
    double ch[J];  // (P_inf + Dv) h
    tp_C_h<NR, NC, J>(Dv, ch);
#pragma unroll
    for (int j = 0; j < NR; ++j) ch[j] += M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) { ch[NR + 2 * k] += M.ac[k]; ch[NR + 2 * k + 1] -= M.bc[k]; }
    D = tp_h_dot<NR, NC>(ch) + R;
    z = r - tp_h_dot<NR, NC>(m);
    inv = mtg_rcp(D);
#pragma unroll
    for (int i = 0; i < J; ++i) kd[i] = ch[i] * inv;
#pragma unroll
    for (int i = 0; i < J; ++i) m[i] = fma(kd[i], z, m[i]);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) Dv(i, j) = fma(-kd[i], ch[j], Dv(i, j));
}

// The samples a lane walks through are 16 bytes apart in two arrays, each lane in its own cache lines: a
// load issued where its value is needed stalls the step for a memory round trip (~1 us against ~0.4 us of
// arithmetic at rank 3).  This keeps the next one or two samples in flight in registers.
// (rank 6 is at the register limit: one sample ahead there)
#define MTG_TP_AHEAD_OF(J) ((J) >= 6 ? 1 : 2)
#ifndef MTG_TP_TREE
#define MTG_TP_TREE 1
#endif
template <int MTG_TP_AHEAD> struct TpSamples {
    double2 y[MTG_TP_AHEAD], x[MTG_TP_AHEAD];
    __device__ __forceinline__ TpSamples(const double2 *yv, const double2 *dxt, int64_t lo, int64_t N)
    {
#pragma unroll
        for (int k = 0; k < MTG_TP_AHEAD; ++k) {
            const int64_t i = lo + k < N ? lo + k : N - 1;  // (clamped: an empty chunk reads the last sample, unused)
            y[k] = yv[i];
            x[k] = dxt[i];
        }
    }
    // after sample n was taken from slot 0
    __device__ __forceinline__ void advance(const double2 *yv, const double2 *dxt, int64_t n, int64_t N)
    {
#pragma unroll
        for (int k = 0; k + 1 < MTG_TP_AHEAD; ++k) { y[k] = y[k + 1]; x[k] = x[k + 1]; }
        const int64_t i = n + MTG_TP_AHEAD < N ? n + MTG_TP_AHEAD : N - 1;
        y[MTG_TP_AHEAD - 1] = yv[i];
        x[MTG_TP_AHEAD - 1] = dxt[i];
    }
};

// filtering element of a chunk: x_out | x_in ~ N(A x_in + b, C), p(y_chunk | x_in) ~ N_I(eta, Jm)
// and the likelihood of the chunk given x_in = 0, ln p(y_chunk | 0) = kq - 1/2 (ln kdm + kde ln 2) (without
// the 2 pi terms): the combination below carries it along, so the scanned element of the whole series IS
// the likelihood and no second pass over the samples is needed.  kmin = smallest pivot met (not positive,
// or NaN: the evaluation goes through the filter pass), kmag = sum of the magnitudes that went into kq.
template <int J> struct TpElem {
    double A[J][J];
    double b[J], eta[J];
    Sym<J> C, Jm;
    double kq, kdm, kde, kmin, kmag;
};

template <int J>
__device__ __forceinline__ void tp_identity(TpElem<J> &e)
{
#pragma unroll
    for (int i = 0; i < J; ++i) {
        e.b[i] = 0.0; e.eta[i] = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = i == j ? 1.0 : 0.0;
    }
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) { e.C.v[i] = 0.0; e.Jm.v[i] = 0.0; }
    e.kq = 0.0; e.kdm = 1.0; e.kde = 0.0; e.kmin = INFINITY; e.kmag = 0.0;
}

// e <- e o step(T, r, R).  Combining an element with a single-step element is the Kalman filter
// itself: (b, C) of the chunk are the mean and covariance of the filter started from (0, 0), and
// with K its gain, g = (h F A)^T, D its pivot and z its residual,
//   A <- (I - K h) F A,   eta <- eta + g z / D,   Jm <- Jm + g g^T / D
// (the general combination's (I + C J)^-1 with the step's rank-one J is that filter update).
// Dv = C - P_inf is carried by the caller across the steps of a chunk.
// kap: (sum z^2/D, prod D [renormalised by the caller], min D) of this recursion, i.e. the likelihood of
// the chunk given x_in = 0.
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_compose_step(const TpModel<NR, NC> &M, const TpTrans<NR, NC> &T, double y, double R,
                                                TpElem<J> &e, Sym<J> &Dv, double (&kap)[3])
{
    tp_left_F<NR, NC, J>(T, e.A);
    double g[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) s += e.A[r][j];
#pragma unroll
        for (int k = 0; k < NC; ++k) s += e.A[NR + 2 * k][j];
        g[j] = s;
    }
    double D, inv, z, kd[J];
    tp_filter_step<NR, NC, J>(M, T, y, R, e.b, Dv, D, inv, z, kd);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = fma(-kd[i], g[j], e.A[i][j]);
    const double zi = z * inv;
    kap[0] = fma(z, zi, kap[0]);
    kap[1] *= D;
    kap[2] = fmin(kap[2], D);
#pragma unroll
    for (int j = 0; j < J; ++j) e.eta[j] = fma(g[j], zi, e.eta[j]);
#pragma unroll
    for (int i = 0; i < J; ++i) {
        const double gi = g[i] * inv;
#pragma unroll
        for (int j = 0; j <= i; ++j) e.Jm(i, j) = fma(gi, g[j], e.Jm(i, j));
    }
}

// Likelihood of a chunk given its start state N(m, C), relative to its likelihood given x_in = 0:
//   ln p(y_chunk | m, C) - kappa = -1/2 ln det G + eta^T m - 1/2 m^T Jm m + 1/2 t^T G^-1 C t,
//   G = I + C Jm,  t = eta - Jm m
// (p(y_chunk | x_in) = exp(kappa + eta^T x_in - 1/2 x_in^T Jm x_in), integrated against N(m, C)).
// NaN when det G is not positive.
template <int J>
__device__ __forceinline__ double tp_chunk_correction(const double *eta, const Sym<J> &Jm, const double *m, const Sym<J> &C)
{
    double t[J], G[J][J], x[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = eta[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(-Jm(i, k), m[k], s);
        t[i] = s;
    }
    double quad = 0.0;   // eta^T m - 1/2 m^T Jm m = 1/2 m^T (eta + t)
#pragma unroll
    for (int i = 0; i < J; ++i) quad = fma(0.5 * m[i], eta[i] + t[i], quad);
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(C(i, k), t[k], s);
        x[i] = s;                                              // C t
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double g = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) g = fma(C(i, k), Jm(k, j), g);
            G[i][j] = g;
        }
    }
    // x <- G^-1 (C t) by Gauss-Jordan without pivoting; det G = product of the pivots
    double dm = 1.0;
    int de = 0;
#pragma unroll
    for (int p = 0; p < J; ++p) {
        const double ip = 1.0 / G[p][p];
        const double pr = dm * G[p][p];
        dm = __builtin_amdgcn_frexp_mant(pr);
        de += __builtin_amdgcn_frexp_exp(pr);
#pragma unroll
        for (int j = p + 1; j < J; ++j) G[p][j] *= ip;
        x[p] *= ip;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            if (i == p) continue;
            const double f = G[i][p];
#pragma unroll
            for (int j = p + 1; j < J; ++j) G[i][j] = fma(-f, G[p][j], G[i][j]);
            x[i] = fma(-f, x[p], x[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < J; ++i) quad = fma(0.5 * t[i], x[i], quad);
    return dm > 0.0 ? quad - 0.5 * (log(dm) + (double)de * 0.69314718055994530942) : __builtin_nan("");
}

// (m, C) <- element applied to the filtered state (m, C):
//   m' = A (I + C Jm)^-1 (m + C eta) + b ;  C' = A (I + C Jm)^-1 C A^T + Cc
template <int J>
__device__ __forceinline__ void tp_apply_elem(const TpElem<J> &e, double *m, Sym<J> &C)
{
    // G = I + C Jm ; right-hand sides [m + C eta | C]
    double G[J][J], Rh[J][J + 1];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = m[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(C(i, k), e.eta[k], s);
        Rh[i][0] = s;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double g = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) g = fma(C(i, k), e.Jm(k, j), g);
            G[i][j] = g;
            Rh[i][j + 1] = C(i, j);
        }
    }
    // Gaussian elimination without pivoting (I + C Jm is similar to a symmetric positive definite matrix)
#pragma unroll
    for (int p = 0; p < J; ++p) {
        const double ip = 1.0 / G[p][p];
#pragma unroll
        for (int j = p + 1; j < J; ++j) G[p][j] *= ip;
#pragma unroll
        for (int j = 0; j < J + 1; ++j) Rh[p][j] *= ip;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            if (i == p) continue;
            const double f = G[i][p];
#pragma unroll
            for (int j = p + 1; j < J; ++j) G[i][j] = fma(-f, G[p][j], G[i][j]);
#pragma unroll
            for (int j = 0; j < J + 1; ++j) Rh[i][j] = fma(-f, Rh[p][j], Rh[i][j]);
        }
    }
    // m' = A x_m + b ; C' = A X_C A^T + Cc
    double Y[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e.b[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(e.A[i][k], Rh[k][0], s);
        m[i] = s;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) t = fma(e.A[i][k], Rh[k][j + 1], t);
            Y[i][j] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = 0.0, t = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) { s = fma(Y[i][k], e.A[j][k], s); t = fma(Y[j][k], e.A[i][k], t); }
            C(i, j) = 0.5 * (s + t) + e.C(i, j);
        }
}

// View of an element stored by tp_store (A | b | eta | C | Jm | kq kdm kde kmin kmag)
template <int J> struct TpSlot {
    double *p;
    static constexpr int OB = J * J, OE = OB + J, OC = OE + J, OJ = OC + J * (J + 1) / 2, OK = OJ + J * (J + 1) / 2;
    __device__ __forceinline__ double &k(int i) const { return p[OK + i]; }
    __device__ __forceinline__ static int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
    __device__ __forceinline__ double &A(int i, int j) const { return p[i * J + j]; }
    __device__ __forceinline__ double &b(int i) const { return p[OB + i]; }
    __device__ __forceinline__ double &eta(int i) const { return p[OE + i]; }
    __device__ __forceinline__ double &C(int i, int j) const { return p[OC + tri(i, j)]; }
    __device__ __forceinline__ double &Jm(int i, int j) const { return p[OJ + tri(i, j)]; }
};

// e2 <- e1 o e2 (e1 earlier in time): the general combination of two filtering elements
//   G = I + C1 J2;  A = A2 G^-1 A1;  b = A2 G^-1 (b1 + C1 eta2) + b2;  C = A2 G^-1 C1 A2^T + C2
//   eta = A1^T G^-T (eta2 - J2 b1) + eta1;  J = A1^T G^-T J2 A1 + J1
//   ln p(y_12 | 0) = ln p(y_1 | 0) + ln p(y_2 | 0) - 1/2 ln det G + 1/2 b1^T (eta2 + t) + 1/2 t^T G^-1 C1 t,
//   t = eta2 - J2 b1   (the second chunk's likelihood integrated against N(b1, C1), see tp_chunk_correction)
// e1 is in registers (it had to be read before the barrier); e2 stays in this lane's own LDS
// slot and is read and overwritten piece by piece: both in registers would be 4 J (J + 1.5)
// VGPRs plus the temporaries and spill to scratch from J = 5 on.
// G^-T J2 = (J2^-1 + C1)^-1 and G^-1 C1 are symmetric, so the two congruences fill one triangle.
template <int J>
__device__ __forceinline__ void tp_combine(const TpElem<J> &e1, double *slot2)
{
    const TpSlot<J> e2{slot2};
    // Gi = (I + C1 J2)^-1 by Gauss-Jordan (similar to a symmetric positive definite matrix: no pivoting)
    double G[J][J], Gi[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double g = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) g = fma(e1.C(i, k), e2.Jm(k, j), g);
            G[i][j] = g;
            Gi[i][j] = i == j ? 1.0 : 0.0;
        }
    double dm = e1.kdm * e2.k(1), de = e1.kde + e2.k(2);  // det G = product of the pivots
    de += (double)__builtin_amdgcn_frexp_exp(dm);
    dm = __builtin_amdgcn_frexp_mant(dm);
#pragma unroll
    for (int p = 0; p < J; ++p) {
        const double ip = 1.0 / G[p][p];
        const double pr = dm * G[p][p];
        dm = __builtin_amdgcn_frexp_mant(pr);
        de += (double)__builtin_amdgcn_frexp_exp(pr);
#pragma unroll
        for (int j = 0; j < J; ++j) { G[p][j] *= ip; Gi[p][j] *= ip; }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            if (i == p) continue;
            const double f = G[i][p];
#pragma unroll
            for (int j = 0; j < J; ++j) { G[i][j] = fma(-f, G[p][j], G[i][j]); Gi[i][j] = fma(-f, Gi[p][j], Gi[i][j]); }
        }
    }
    // (single pivots of the non-symmetric G may be negative; only the sign of det G matters)
    e2.k(3) = dm > 0.0 ? fmin(e1.kmin, e2.k(3)) : __builtin_nan("");
    e2.k(1) = dm;
    e2.k(2) = de;
    // w = b1 + C1 eta2 (state part, needs the OLD eta2)
    double w[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e1.b[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(e1.C(i, k), e2.eta(k), s);
        w[i] = s;
    }
    // ---- information part: eta, Jm (uses A1, b1, eta1, J1 and the OLD eta2, J2) ------------
    {
        double t[J], yeta[J];
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e2.eta(i);
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(-e2.Jm(i, k), e1.b[k], s);
            t[i] = s;
        }
#pragma unroll
        for (int i = 0; i < J; ++i) {  // G^-T t
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(Gi[k][i], t[k], s);
            yeta[i] = s;
        }
        {   // likelihood part
            double lin = 0.0, quad = 0.0;
#pragma unroll
            for (int i = 0; i < J; ++i) {
                double c1t = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) c1t = fma(e1.C(i, k), t[k], c1t);
                quad = fma(0.5 * yeta[i], c1t, quad);
                lin = fma(0.5 * e1.b[i], e2.eta(i) + t[i], lin);
            }
            e2.k(0) += e1.kq + lin + quad;
            e2.k(4) += e1.kmag + fabs(lin) + fabs(quad);
        }
        double YJ[J][J], Z[J][J];  // YJ = G^-T J2 ; Z = YJ A1
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) s = fma(Gi[k][i], e2.Jm(k, j), s);
                YJ[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) s = fma(YJ[i][k], e1.A[k][j], s);
                Z[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e1.eta[i];
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(e1.A[k][i], yeta[k], s);
            e2.eta(i) = s;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double u = e1.Jm(i, j);
#pragma unroll
                for (int k = 0; k < J; ++k) u = fma(e1.A[k][i], Z[k][j], u);
                e2.Jm(i, j) = u;
            }
        }
    }
    // ---- state part: XA = Gi A1, Xb = Gi w, XC = Gi C1 ---------------------------------------
    double Xb[J], XA[J][J], XC[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(Gi[i][k], w[k], s);
        Xb[i] = s;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) { a = fma(Gi[i][k], e1.A[k][j], a); c = fma(Gi[i][k], e1.C(k, j), c); }
            XA[i][j] = a;
            XC[i][j] = c;
        }
    }
    // C <- A2 XC A2^T + C2 (needs the OLD A2), then A <- A2 XA and b <- A2 Xb + b2 row by row
    double A2[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) A2[i][j] = e2.A(i, j);
    {
        double Y[J][J];
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double c = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) c = fma(A2[i][k], XC[k][j], c);
                Y[i][j] = c;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double u = e2.C(i, j);
#pragma unroll
                for (int k = 0; k < J; ++k) u = fma(Y[i][k], A2[j][k], u);
                e2.C(i, j) = u;
            }
    }
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e2.b(i);
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(A2[i][k], Xb[k], s);
        e2.b(i) = s;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) a = fma(A2[i][k], XA[k][j], a);
            e2.A(i, j) = a;
        }
    }
}

// dst <- e1 o e2 with all three in LDS (dst a slot of the OTHER buffer of a ping-pong pair): nothing of e1
// is held in registers across the barrier and the pieces are read where they are used.  For the rank
// where tp_combine's register version no longer fits the register file (J = 6: G, G^-1, e1 and the
// temporaries are ~560 VGPRs; its scratch spills are slow and were miscompiled, scripts/agpr_lint.py).
template <int J>
__device__ __forceinline__ void tp_combine_lds(const double *slot1, const double *slot2, double *slotd)
{
    const TpSlot<J> e1{const_cast<double *>(slot1)}, e2{const_cast<double *>(slot2)}, d{slotd};
    double Gi[J][J];
    {
        double G[J][J];
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double g = i == j ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) g = fma(e1.C(i, k), e2.Jm(k, j), g);
                G[i][j] = g;
                Gi[i][j] = i == j ? 1.0 : 0.0;
            }
        double dm = e1.k(1) * e2.k(1), de = e1.k(2) + e2.k(2);
        de += (double)__builtin_amdgcn_frexp_exp(dm);
        dm = __builtin_amdgcn_frexp_mant(dm);
#pragma unroll
        for (int p = 0; p < J; ++p) {
            const double ip = 1.0 / G[p][p];
            const double pr = dm * G[p][p];
            dm = __builtin_amdgcn_frexp_mant(pr);
            de += (double)__builtin_amdgcn_frexp_exp(pr);
#pragma unroll
            for (int j = 0; j < J; ++j) { G[p][j] *= ip; Gi[p][j] *= ip; }
#pragma unroll
            for (int i = 0; i < J; ++i) {
                if (i == p) continue;
                const double f = G[i][p];
#pragma unroll
                for (int j = 0; j < J; ++j) { G[i][j] = fma(-f, G[p][j], G[i][j]); Gi[i][j] = fma(-f, Gi[p][j], Gi[i][j]); }
            }
        }
        d.k(3) = dm > 0.0 ? fmin(e1.k(3), e2.k(3)) : __builtin_nan("");
        d.k(1) = dm;
        d.k(2) = de;
    }
    // ---- information part and likelihood ------------------------------------------------------
    {
        double t[J], yeta[J];
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e2.eta(i);
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(-e2.Jm(i, k), e1.b(k), s);
            t[i] = s;
        }
#pragma unroll
        for (int i = 0; i < J; ++i) {  // G^-T t
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(Gi[k][i], t[k], s);
            yeta[i] = s;
        }
        double lin = 0.0, quad = 0.0;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double c1t = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) c1t = fma(e1.C(i, k), t[k], c1t);
            quad = fma(0.5 * yeta[i], c1t, quad);
            lin = fma(0.5 * e1.b(i), e2.eta(i) + t[i], lin);
        }
        d.k(0) = e1.k(0) + e2.k(0) + lin + quad;
        d.k(4) = e1.k(4) + e2.k(4) + fabs(lin) + fabs(quad);
        double YJ[J][J], Z[J][J];  // YJ = G^-T J2 ; Z = YJ A1
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) s = fma(Gi[k][i], e2.Jm(k, j), s);
                YJ[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) s = fma(YJ[i][k], e1.A(k, j), s);
                Z[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e1.eta(i);
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(e1.A(k, i), yeta[k], s);
            d.eta(i) = s;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double u = e1.Jm(i, j);
#pragma unroll
                for (int k = 0; k < J; ++k) u = fma(e1.A(k, i), Z[k][j], u);
                d.Jm(i, j) = u;
            }
        }
    }
    // ---- state part --------------------------------------------------------------------------
    {
        double w[J], Xb[J];  // w = b1 + C1 eta2 ; Xb = Gi w ; b = A2 Xb + b2
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e1.b(i);
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(e1.C(i, k), e2.eta(k), s);
            w[i] = s;
        }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(Gi[i][k], w[k], s);
            Xb[i] = s;
        }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double s = e2.b(i);
#pragma unroll
            for (int k = 0; k < J; ++k) s = fma(e2.A(i, k), Xb[k], s);
            d.b(i) = s;
        }
    }
    {
        double X[J][J];  // XA = Gi A1 ; A = A2 XA
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) a = fma(Gi[i][k], e1.A(k, j), a);
                X[i][j] = a;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) a = fma(e2.A(i, k), X[k][j], a);
                d.A(i, j) = a;
            }
    }
    {
        double X[J][J], Y[J][J];  // XC = Gi C1 ; Y = A2 XC ; C = Y A2^T + C2
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double c = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) c = fma(Gi[i][k], e1.C(k, j), c);
                X[i][j] = c;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double c = 0.0;
#pragma unroll
                for (int k = 0; k < J; ++k) c = fma(e2.A(i, k), X[k][j], c);
                Y[i][j] = c;
            }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double u = e2.C(i, j);
#pragma unroll
                for (int k = 0; k < J; ++k) u = fma(Y[i][k], e2.A(j, k), u);
                d.C(i, j) = u;
            }
    }
}

// ---- the combination split into parts that different WAVES compute (tp_reduce_tree) ------------------
// R <- G^-1 R by Gauss-Jordan without pivoting (G is destroyed); (dm, de): det G as mantissa and exponent
template <int J, int NRHS>
__device__ __forceinline__ void tp_eliminate(double (&G)[J][J], double (&R)[J][NRHS], double &dm, double &de)
{
#pragma unroll
    for (int p = 0; p < J; ++p) {
        const double ip = mtg_rcp(G[p][p]);
        const double pr = dm * G[p][p];
        dm = __builtin_amdgcn_frexp_mant(pr);
        de += (double)__builtin_amdgcn_frexp_exp(pr);
#pragma unroll
        for (int j = p + 1; j < J; ++j) G[p][j] *= ip;
#pragma unroll
        for (int j = 0; j < NRHS; ++j) R[p][j] *= ip;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            if (i == p) continue;
            const double f = G[i][p];
#pragma unroll
            for (int j = p + 1; j < J; ++j) G[i][j] = fma(-f, G[p][j], G[i][j]);
#pragma unroll
            for (int j = 0; j < NRHS; ++j) R[i][j] = fma(-f, R[p][j], R[i][j]);
        }
    }
}

template <int J>
__device__ __forceinline__ void tp_form_G(const TpSlot<J> &e1, const TpSlot<J> &e2, double (&G)[J][J])
{
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double g = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) g = fma(e1.C(i, k), e2.Jm(k, j), g);
            G[i][j] = g;
        }
}

// Information part and likelihood of e1 o e2:  with XA = G^-1 A1 and t = eta2 - J2 b1,
//   eta = XA^T t + eta1,  Jm = A1^T (J2 XA) + J1  (J2 G^-1 = (J2^-1 + C1)^-1 is symmetric),
//   kq += 1/2 b1^T (eta2 + t) + 1/2 t^T G^-1 C1 t,  det G into (kdm, kde)
template <int J> struct TpInfoPart {
    double eta[J];
    Sym<J> Jm;
    double k[5];
    __device__ __forceinline__ void compute(const double *slot1, const double *slot2)
    {
        const TpSlot<J> e1{const_cast<double *>(slot1)}, e2{const_cast<double *>(slot2)};
        double G[J][J], R[J][J + 1], t[J];  // R = [A1 | C1 t]
        tp_form_G<J>(e1, e2, G);
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double u = e2.eta(i);
#pragma unroll
            for (int q = 0; q < J; ++q) u = fma(-e2.Jm(i, q), e1.b(q), u);
            t[i] = u;
        }
        double lin = 0.0;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double c1t = 0.0;
#pragma unroll
            for (int q = 0; q < J; ++q) c1t = fma(e1.C(i, q), t[q], c1t);
            R[i][J] = c1t;
#pragma unroll
            for (int j = 0; j < J; ++j) R[i][j] = e1.A(i, j);
            lin = fma(0.5 * e1.b(i), e2.eta(i) + t[i], lin);
        }
        double dm = e1.k(1) * e2.k(1), de = e1.k(2) + e2.k(2);
        de += (double)__builtin_amdgcn_frexp_exp(dm);
        dm = __builtin_amdgcn_frexp_mant(dm);
        tp_eliminate<J, J + 1>(G, R, dm, de);
        double quad = 0.0;
#pragma unroll
        for (int i = 0; i < J; ++i) quad = fma(0.5 * t[i], R[i][J], quad);
        k[0] = e1.k(0) + e2.k(0) + lin + quad;
        k[1] = dm;
        k[2] = de;
        k[3] = dm > 0.0 ? fmin(e1.k(3), e2.k(3)) : __builtin_nan("");
        k[4] = e1.k(4) + e2.k(4) + fabs(lin) + fabs(quad);
        double Z[J][J];  // J2 XA
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                double u = 0.0;
#pragma unroll
                for (int q = 0; q < J; ++q) u = fma(e2.Jm(i, q), R[q][j], u);
                Z[i][j] = u;
            }
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double u = e1.eta(i);
#pragma unroll
            for (int q = 0; q < J; ++q) u = fma(R[q][i], t[q], u);
            eta[i] = u;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double v = e1.Jm(i, j);
#pragma unroll
                for (int q = 0; q < J; ++q) v = fma(e1.A(q, i), Z[q][j], v);
                Jm(i, j) = v;
            }
        }
    }
    __device__ __forceinline__ void store(double *slot) const
    {
        const TpSlot<J> d{slot};
#pragma unroll
        for (int i = 0; i < J; ++i) {
            d.eta(i) = eta[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) d.Jm(i, j) = Jm(i, j);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) d.k(i) = k[i];
    }
};

// State part of e1 o e2:  A = A2 G^-1 A1,  b = A2 G^-1 (b1 + C1 eta2) + b2  (WHAT & 1),
//                         C = A2 G^-1 C1 A2^T + C2  (WHAT & 2)
template <int J, int WHAT> struct TpStatePart {
    static constexpr bool AB = (WHAT & 1) != 0, CC = (WHAT & 2) != 0;
    static constexpr int NA = AB ? J + 1 : 0, NRHS = NA + (CC ? J : 0);
    double A[AB ? J : 1][AB ? J : 1], b[AB ? J : 1];
    Sym<CC ? J : 1> C;
    __device__ __forceinline__ void compute(const double *slot1, const double *slot2)
    {
        const TpSlot<J> e1{const_cast<double *>(slot1)}, e2{const_cast<double *>(slot2)};
        double G[J][J], R[J][NRHS];  // R = [A1 | b1 + C1 eta2 | C1]
        tp_form_G<J>(e1, e2, G);
#pragma unroll
        for (int i = 0; i < J; ++i) {
            if (AB) {
                double w = e1.b(i);
#pragma unroll
                for (int q = 0; q < J; ++q) w = fma(e1.C(i, q), e2.eta(q), w);
                R[i][J] = w;
#pragma unroll
                for (int j = 0; j < J; ++j) R[i][j] = e1.A(i, j);
            }
            if (CC) {
#pragma unroll
                for (int j = 0; j < J; ++j) R[i][NA + j] = e1.C(i, j);
            }
        }
        double dm = 1.0, de = 0.0;
        tp_eliminate<J, NRHS>(G, R, dm, de);
        if (AB) {
#pragma unroll
            for (int i = 0; i < J; ++i) {
                double u = e2.b(i);
#pragma unroll
                for (int q = 0; q < J; ++q) u = fma(e2.A(i, q), R[q][J], u);
                b[i] = u;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < J; ++q) v = fma(e2.A(i, q), R[q][j], v);
                    A[i][j] = v;
                }
            }
        }
        if (CC) {
            double Y[J][J];
#pragma unroll
            for (int i = 0; i < J; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < J; ++q) v = fma(e2.A(i, q), R[q][NA + j], v);
                    Y[i][j] = v;
                }
#pragma unroll
            for (int i = 0; i < J; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double v = e2.C(i, j);
#pragma unroll
                    for (int q = 0; q < J; ++q) v = fma(Y[i][q], e2.A(j, q), v);
                    C(i, j) = v;
                }
        }
    }
    __device__ __forceinline__ void store(double *slot) const
    {
        const TpSlot<J> d{slot};
        if (AB) {
#pragma unroll
            for (int i = 0; i < J; ++i) {
                d.b(i) = b[i];
#pragma unroll
                for (int j = 0; j < J; ++j) d.A(i, j) = A[i][j];
            }
        }
        if (CC) {
#pragma unroll
            for (int i = 0; i < J; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) d.C(i, j) = C(i, j);
        }
    }
};

// Reduction of the 256 chunk elements of a four-wave workgroup to the element of the whole series, in
// slot 0.  Only the likelihood is wanted, so a tree of 255 combinations does instead of a scan, and the
// lanes it leaves idle share each combination: the parts of e1 o e2 are independent once G = I + C1 J2 is
// eliminated, so one WAVE computes the information part of all the level's combinations, another (A, b),
// a third C -- each eliminating G for itself, ~4.5 J^3 multiply-adds instead of 10 J^3 per level (roles
// must be wave-uniform: lanes of one wave on different parts would run one after the other).  Level 0 has
// 128 combinations: two waves take the information parts, two the whole state part.
// Layout: a level with n elements keeps element i in slot (i >> 1) + (i & 1) * n/2 -- left operands in
// the first half, right operands in the second -- so that the lanes of a wave, working on consecutive
// combinations, read consecutive slots (odd stride in doubles: no bank conflicts; slots 2c and 2c + 1 would
// put four lanes on every bank, and the strided survivors of an in-place tree up to all of them).
// Results are written after a barrier (the other waves read the same operands).
__device__ __forceinline__ int tp_tree_slot(int i, int n) { return (i >> 1) + (i & 1) * (n >> 1); }

template <int J>
__device__ __forceinline__ void tp_reduce_tree(double *buf)
{
    constexpr int ELEM = MTG_TP_ELEM(J);
    const int wave = threadIdx.x >> 6, sub = threadIdx.x & 63;
    TpInfoPart<J> info;
    {   // level 0: 256 elements, combinations c < 128
        const int c = (wave & 1) * 64 + sub;
        const double *L = buf + c * ELEM, *R = buf + (128 + c) * ELEM;
        double *out = buf + tp_tree_slot(c, 128) * ELEM;
        TpStatePart<J, 3> state;
        if (wave < 2) info.compute(L, R);
        else state.compute(L, R);
        __syncthreads();  // every wave has read its operands
        if (wave < 2) info.store(out);
        else state.store(out);
        __syncthreads();
    }
#pragma unroll 1
    for (int n = 128; n > 1; n >>= 1) {  // n elements, n / 2 <= 64 combinations
        const bool active = sub < (n >> 1);
        const int c = active ? sub : 0;
        const double *L = buf + c * ELEM, *R = buf + ((n >> 1) + c) * ELEM;
        double *out = buf + tp_tree_slot(c, n >> 1) * ELEM;
        TpStatePart<J, 1> ab;
        TpStatePart<J, 2> cc;
        if (active) {
            if (wave == 0) info.compute(L, R);
            else if (wave == 1) ab.compute(L, R);
            else if (wave == 2) cc.compute(L, R);
        }
        __syncthreads();
        if (active) {
            if (wave == 0) info.store(out);
            else if (wave == 1) ab.store(out);
            else if (wave == 2) cc.store(out);
        }
        __syncthreads();
    }
}

// Element <-> its LDS slot (A | b | eta | C | Jm | kq kdm kde kmin kmag)
template <int J>
__device__ __forceinline__ void tp_store(const TpElem<J> &e, double *slot)
{
    int o = 0;
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) slot[o++] = e.A[i][j];
#pragma unroll
    for (int i = 0; i < J; ++i) slot[o++] = e.b[i];
#pragma unroll
    for (int i = 0; i < J; ++i) slot[o++] = e.eta[i];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) slot[o++] = e.C.v[i];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) slot[o++] = e.Jm.v[i];
    slot[o] = e.kq; slot[o + 1] = e.kdm; slot[o + 2] = e.kde; slot[o + 3] = e.kmin; slot[o + 4] = e.kmag;
}

template <int J>
__device__ __forceinline__ void tp_load(TpElem<J> &e, const double *slot)
{
    int o = 0;
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = slot[o++];
#pragma unroll
    for (int i = 0; i < J; ++i) e.b[i] = slot[o++];
#pragma unroll
    for (int i = 0; i < J; ++i) e.eta[i] = slot[o++];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) e.C.v[i] = slot[o++];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) e.Jm.v[i] = slot[o++];
    e.kq = slot[o]; e.kdm = slot[o + 1]; e.kde = slot[o + 2]; e.kmin = slot[o + 3]; e.kmag = slot[o + 4];
}

}  // namespace

// one wave per evaluation; lane = chunk
// LANES = chunks per evaluation = workgroup size: 64 (one wave) or 256 (four waves, for the
// smallest batches: the scan then crosses waves through LDS and workgroup barriers)
// `elems`: LANES element slots in LDS; `red`: a few doubles of LDS for the cross-wave reductions.
template <int NR, int NC, bool FAST, int LANES>
__device__ __forceinline__ void mtg_tp_body(const MtgSolveArgs &a, const TpModel<NR, NC> &M, double jitter, double slope,
                                            double icpt, int64_t ev, int64_t lc, const MtgMathTables *tab, double *elems,
                                            double *red, bool direct)
{
    constexpr int J = NR + 2 * NC;
    constexpr int ELEM = MTG_TP_ELEM(J);  // doubles per element
    const int lane = threadIdx.x;
    const int64_t N = a.N;
    const double2 *yv = a.yv + lc * N, *dxt = a.dxt + lc * a.t_stride;

    // chunk of this lane: samples [lo, hi); sample 0 is the prior update below
    const int64_t per = (N + LANES - 1) / LANES;
    int64_t lo = (int64_t)lane * per, hi = lo + per;
    if (lo > N) lo = N;
    if (hi > N) hi = N;
    if (lo == 0) lo = 1;

    // filtered state after sample 0 (update of the stationary prior), identical on every lane; computed where
    // it is used, not kept across the passes
    auto head_state = [&](double *m, Sym<J> &C, double &D0, double &z0) {
#pragma unroll
        for (int i = 0; i < J * (J + 1) / 2; ++i) C.v[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) C(j, j) = M.ar[j];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int o = NR + 2 * k;
            C(o, o) = M.ac[k]; C(o + 1, o) = -M.bc[k]; C(o + 1, o + 1) = M.pc[k];
        }
        double ch[J];
        tp_C_h<NR, NC, J>(C, ch);
        D0 = tp_h_dot<NR, NC>(ch) + yv[0].y + jitter;
        z0 = yv[0].x - fma(slope, dxt[0].y, icpt);
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = ch[i] * z0 / D0;
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) C(i, j) -= ch[i] * ch[j] / D0;
    };
    int *flag = (int *)(red + 3 * (LANES / 64));
    double *buf = elems;
    // ---- pass 1: element of the chunk, with the chunk's likelihood given x_in = 0, into the lane's slot ----
    auto compose_chunk = [&](int slot) {
        TpElem<J> e;
        tp_identity<J>(e);
        tp_sub_pinf<NR, NC, J>(M, e.C);  // e.C holds C - P_inf inside the loop
        double kap[3] = {0.0, 1.0, INFINITY};
        int kexp = 0;
        TpSamples<MTG_TP_AHEAD_OF(J)> q(yv, dxt, lo, N);
        for (int64_t n = lo; n < hi; ++n) {
            const double2 sy = q.y[0], sx = q.x[0];
            q.advance(yv, dxt, n, N);
            TpTrans<NR, NC> T;
            tp_transition<NR, NC, FAST>(M, sx.x, T, tab);
            const double r = sy.x - fma(slope, sx.y, icpt);
            tp_compose_step<NR, NC, J>(M, T, r, sy.y + jitter, e, e.C, kap);
            kexp += __builtin_amdgcn_frexp_exp(kap[1]);
            kap[1] = __builtin_amdgcn_frexp_mant(kap[1]);
        }
        e.kq = -0.5 * kap[0]; e.kdm = kap[1]; e.kde = (double)kexp; e.kmin = kap[2]; e.kmag = 0.5 * kap[0];
        tp_add_pinf<NR, NC, J>(M, e.C);
        tp_store<J>(e, elems + slot * ELEM);
        __syncthreads();
    };
    // The element of samples 1 .. N-1 (in `slot`) integrated against the state after sample 0 is the
    // likelihood.  Anything suspicious sends the evaluation through pass 3.  Returns "done" (workgroup-uniform).
    auto finish = [&](const double *slot, int owner) -> bool {
        if (lane == owner) {
            const TpSlot<J> tot{const_cast<double *>(slot)};
            double m[J], D0, z0;
            Sym<J> C;
            head_state(m, C, D0, z0);
            double eta[J];
            Sym<J> Jm;
#pragma unroll
            for (int i = 0; i < J; ++i) eta[i] = tot.eta(i);
#pragma unroll
            for (int i = 0; i < J; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) Jm(i, j) = tot.Jm(i, j);
            const double corr = tp_chunk_correction<J>(eta, Jm, m, C);
            const double ld = log(tot.k(1)) + tot.k(2) * 0.69314718055994530942 + log(D0);
            const double q0 = 0.5 * z0 * z0 / D0;
            const double ll = tot.k(0) + corr - q0 - 0.5 * (ld + (double)N * MTG_LN_2PI);
            const double mag = tot.k(4) + fabs(corr) + q0;
            const bool good = direct && fmin(tot.k(3), D0) > 0.0 && isfinite(ll) && mag <= 1.0e3 * fabs(ll);
            if (good) { a.out[ev] = ll; a.status[ev] = MTG_ST_OK; }
            *flag = good ? 0 : 1;
        }
        __syncthreads();
        return *flag == 0;
    };

    constexpr bool TREE = LANES == 256 && MTG_TP_TREE;
    compose_chunk(TREE && direct ? tp_tree_slot(lane, LANES) : lane);
    // ---- pass 2 ------------------------------------------------------------------------------------------
    // Four waves: a tree reduces the chunk elements to the element of the whole series (tp_reduce_tree) -- no
    // prefixes, so an evaluation that turns out to need pass 3 composes its chunks a second time and scans them.
    if (TREE && direct) {
        if constexpr (TREE) tp_reduce_tree<J>(buf);
        if (finish(buf, 0)) return;
        compose_chunk(lane);
    }
    // inclusive scan of the chunk elements (Hillis-Steele through LDS): lane l gets the element of chunks 0 .. l
    if (MTG_TP_PINGPONG(J)) {  // two buffers: read the round's inputs from one, write its outputs to the other
        double *nxt = elems + LANES * ELEM;
        for (int off = 1; off < LANES; off <<= 1) {
            if (lane >= off) tp_combine_lds<J>(buf + (lane - off) * ELEM, buf + lane * ELEM, nxt + lane * ELEM);
            else
                for (int i = 0; i < ELEM; ++i) nxt[lane * ELEM + i] = buf[lane * ELEM + i];
            __syncthreads();
            double *sw = buf; buf = nxt; nxt = sw;
        }
    } else {
        for (int off = 1; off < LANES; off <<= 1) {
            TpElem<J> prev;
            if (lane >= off) tp_load<J>(prev, buf + (lane - off) * ELEM);
            __syncthreads();  // everybody has read its partner before anybody overwrites
            if (lane >= off) tp_combine<J>(prev, buf + lane * ELEM);  // own element stays in its LDS slot
            __syncthreads();
        }
    }
    if (!(TREE && direct) && finish(buf + (LANES - 1) * ELEM, LANES - 1)) return;

    // ---- pass 3: ordinary Kalman filter over the chunk from its start state -----------------
    // start state of this lane's chunk: the prefix of the earlier chunks applied to the state after sample 0
    double m[J], D0, z0;
    Sym<J> C;
    head_state(m, C, D0, z0);
    if (lane > 0) {
        TpElem<J> pre;
        tp_load<J>(pre, buf + (lane - 1) * ELEM);
        tp_apply_elem<J>(pre, m, C);
    }
    double dot = lane == 0 ? z0 * z0 / D0 : 0.0;
    double dmin = lane == 0 ? D0 : INFINITY;
    double dprod = 1.0;
    int dexp = 0;
    tp_sub_pinf<NR, NC, J>(M, C);  // deviation form from here on
    TpSamples<MTG_TP_AHEAD_OF(J)> q(yv, dxt, lo, N);
    for (int64_t n = lo; n < hi; ++n) {
        const double2 sy = q.y[0], sx = q.x[0];
        q.advance(yv, dxt, n, N);
        TpTrans<NR, NC> T;
        tp_transition<NR, NC, FAST>(M, sx.x, T, tab);
        const double r = sy.x - fma(slope, sx.y, icpt);
        double D, inv, z, kd[J];
        tp_filter_step<NR, NC, J>(M, T, r, sy.y + jitter, m, C, D, inv, z, kd);
        dot = fma(z * z, inv, dot);
        dmin = fmin(dmin, D);
        const double pr = dprod * D;
        dprod = __builtin_amdgcn_frexp_mant(pr);
        dexp += __builtin_amdgcn_frexp_exp(pr);
    }
    double ld = (lane == 0 ? log(D0) : 0.0) + log(dprod) + (double)dexp * 0.69314718055994530942;
    for (int off = 32; off > 0; off >>= 1) {
        dot += __shfl_down(dot, off);
        ld += __shfl_down(ld, off);
        dmin = fmin(dmin, __shfl_down(dmin, off));
    }
    if (LANES > 64) {  // combine the waves' partial sums through LDS
        if ((lane & 63) == 0) {
            double *w = red + 3 * (lane >> 6);
            w[0] = dot; w[1] = ld; w[2] = dmin;
        }
        __syncthreads();
        if (lane == 0)
            for (int k = 1; k < LANES / 64; ++k) {
                const double *w = red + 3 * k;
                dot += w[0]; ld += w[1]; dmin = fmin(dmin, w[2]);
            }
    }
    if (lane == 0) {
        double ll = -0.5 * (dot + ld + (double)N * MTG_LN_2PI);
        int st = MTG_ST_OK;
        if (!(dmin > 0.0)) { st = MTG_ST_NOTPD; ll = -INFINITY; }
        else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
        a.out[ev] = ll;
        a.status[ev] = st;
    }
}

// One evaluation `ev` of structure <NR, NC> by the whole workgroup: load its model, pick the
// trigonometric path, run the three passes.
template <int NR, int NC, int LANES>
__device__ __forceinline__ void mtg_tp_eval(const MtgSolveArgs &a, int64_t ev, const MtgMathTables *tab, double *elems,
                                            double *red)
{
    // ---- model of this evaluation (same on every lane) ------------------------------------
    TpModel<NR, NC> M;
    const double *cf = a.coef + ev;
    const int64_t cs = a.cstride;
    double dmax = 0.0;
    bool direct = a.tp_direct != 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) { M.ar[j] = cf[a.lay.ar(j) * cs]; M.cr[j] = cf[a.lay.cr(j) * cs]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double aa = cf[a.lay.ac(k) * cs], bb = cf[a.lay.bc(k) * cs], c = cf[a.lay.cc(k) * cs], d = cf[a.lay.dc(k) * cs];
        M.ac[k] = aa; M.bc[k] = bb; M.cc[k] = c; M.dc[k] = d;
        // free entry of P_inf: the value maximising det(noise covariance) (proto/kalman_scan.py)
        M.pc[k] = d != 0.0 ? (2.0 * d * (2.0 * c * bb + d * aa) + 4.0 * c * (c * aa - d * bb)) / (2.0 * d * d) : aa;
        dmax = fmax(dmax, fabs(d));
        // A complex term's power spectrum is non-negative iff b d <= a c.  Terms with a free b (ComplexTerm,
        // BendingPowerlaw outside their prior) can break that, the covariance matrix then need not be
        // positive definite, and the scanned likelihood cannot tell: it sees det and quadratic form, not the
        // sign of every pivot (two negative pivots cancel).  Those evaluations take the filter pass.
        // (1e-12: SHOTerm and Matern32Term sit ON the boundary, b d = a c up to rounding.)
        if (!(fabs(bb * d) <= aa * c * (1.0 + 1.0e-12))) direct = false;
    }
    const double jitter = cf[a.lay.jit() * cs];
    const double slope = cf[a.lay.mean(0) * cs], icpt = cf[a.lay.mean(1) * cs];
    const int64_t lc = a.lc_index ? (int64_t)a.lc_index[ev] : 0;
    // a device-side lc_index cannot be validated by the host: this kernel reads through raw
    // pointers (the throughput kernel's buffer loads are bounds-checked by the hardware)
    if (lc < 0 || (uint64_t)(lc + 1) * (uint64_t)a.N * 16u > (uint64_t)a.yv_bytes) {
        if (threadIdx.x == 0) { a.out[ev] = -INFINITY; a.status[ev] = MTG_ST_NONFINITE; }
        return;
    }
    if (dmax * *a.dxmax <= MTG_TRIG_FAST_MAX)
        mtg_tp_body<NR, NC, true, LANES>(a, M, jitter, slope, icpt, ev, lc, tab, elems, red, direct);
    else
        mtg_tp_body<NR, NC, false, LANES>(a, M, jitter, slope, icpt, ev, lc, tab, elems, red, direct);
}

template <int NR, int NC, int LANES>
__global__ void __launch_bounds__(LANES, 1) mtg_tp_kernel(MtgSolveArgs a)
{
    constexpr int J = NR + 2 * NC;
    static_assert(MTG_TP_IN_LDS(J, LANES), "the chunk elements of a workgroup live in LDS (rank 10: mtg_tp_big.h)");
    __shared__ double sh[MTG_TP_LDS_DOUBLES(J, LANES)];
    __shared__ double red[3 * (LANES / 64) + 1];
    __shared__ MtgMathTables tab;
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if ((int64_t)blockIdx.x >= count) return;
    const int64_t ev = a.list ? (int64_t)a.list[blockIdx.x] : (int64_t)blockIdx.x;
    if (!a.list && a.status[ev] != MTG_ST_OK) return;
    mtg_load_tables<(NC > 0)>(&tab, static_cast<const MtgMathTables *>(a.tables), threadIdx.x, LANES);
    __syncthreads();
    mtg_tp_eval<NR, NC, LANES>(a, ev, &tab, sh, red);
}

template <int NR, int NC, int LANES = 64>
static void mtg_launch_tp(const MtgSolveArgs &a, int64_t nevals, hipStream_t stream)
{
    if (nevals <= 0) return;
    hipLaunchKernelGGL((mtg_tp_kernel<NR, NC, LANES>), dim3((unsigned)nevals), dim3(LANES), 0, stream, a);
}

// ---- every signature of a model with SHO terms in ONE launch -------------------------------
// A model with S SHOTerms has S + 1 structures (NR0 + 2k, NC0 - k), all of rank J; the prepare
// kernel sorts the evaluations into one list per structure.  Launching the structures one after
// the other leaves the GPU to a few dozen workgroups S + 1 times; here workgroup b walks the
// list lengths to find its (structure, evaluation) -- a.list is the base of the lists (stride
// a.cstride), a.count_ptr the base of the counts -- and the launch has B workgroups.
template <int NR0, int NC0, int K, int NSIG, int LANES>
__device__ __forceinline__ void mtg_tp_dispatch(int k, const MtgSolveArgs &a, int64_t ev, const MtgMathTables *tab,
                                                double *elems, double *red)
{
    if (k == K) mtg_tp_eval<NR0 + 2 * K, NC0 - K, LANES>(a, ev, tab, elems, red);
    else if constexpr (K + 1 < NSIG) mtg_tp_dispatch<NR0, NC0, K + 1, NSIG, LANES>(k, a, ev, tab, elems, red);
}

template <int NR0, int NC0, int NSIG, int LANES>
__global__ void __launch_bounds__(LANES, 1) mtg_tp_fused_kernel(MtgSolveArgs a)
{
    constexpr int J = NR0 + 2 * NC0;
    static_assert(NSIG >= 2 && NSIG - 1 <= NC0, "one structure per number of over-damped SHO terms");
    __shared__ double sh[MTG_TP_LDS_DOUBLES(J, LANES)];
    __shared__ double red[3 * (LANES / 64) + 1];
    __shared__ MtgMathTables tab;
    int64_t r = blockIdx.x;
    int k = 0;
    for (; k < NSIG; ++k) {
        const int64_t c = a.count_ptr[k];
        if (r < c) break;
        r -= c;
    }
    if (k == NSIG) return;  // beyond the evaluations that passed the prior
    const int64_t ev = a.list[(int64_t)k * a.cstride + r];
    mtg_load_tables<(NC0 > 0)>(&tab, static_cast<const MtgMathTables *>(a.tables), threadIdx.x, LANES);
    __syncthreads();
    mtg_tp_dispatch<NR0, NC0, 0, NSIG, LANES>(k, a, ev, &tab, sh, red);
}

template <int NR0, int NC0, int NSIG, int LANES>
static void mtg_launch_tp_fused(const MtgSolveArgs &a, int64_t nevals, hipStream_t stream)
{
    if (nevals <= 0) return;
    hipLaunchKernelGGL((mtg_tp_fused_kernel<NR0, NC0, NSIG, LANES>), dim3((unsigned)nevals), dim3(LANES), 0, stream, a);
}

