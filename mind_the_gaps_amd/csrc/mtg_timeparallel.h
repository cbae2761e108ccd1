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
//   pass 1  every lane composes the elements (A, b, C, eta, J) of its chunk -- a
//           rank-one (Sherman-Morrison) composition per step, O(J^2);
//   pass 2  an inclusive scan of the 64 chunk elements across the lanes (Hillis-Steele through
//           LDS, six rounds of the general element combination, one J x J inverse each);
//           every lane then applies the prefix of the earlier chunks to the state after
//           sample 0: its chunk's start state;
//   pass 3  every lane runs the ordinary Kalman filter over its chunk from its start
//           state and accumulates ln prod D and sum z^2 / D; a wave reduction finishes.
// Work is ~3x the serial sweep, depth ~2 N/64 + 6 combinations instead of N.
#pragma once
#include "mtg_device.h"

// 256-entry tables here (2 KiB + 4 KiB): LDS is needed for the chunk elements
#define MTG_EXP_BITS 8
#define MTG_TRIG_BITS 8
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

// y <- F^T x
template <int NR, int NC>
__device__ __forceinline__ void tp_apply_Ft(const TpTrans<NR, NC> &T, double *x)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) x[j] *= T.phi[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double x0 = x[NR + 2 * k], x1 = x[NR + 2 * k + 1];
        x[NR + 2 * k] = T.ec[k] * x0 + T.es[k] * x1;
        x[NR + 2 * k + 1] = -T.es[k] * x0 + T.ec[k] * x1;
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

// C <- F C F^T + Q with Q = P_inf - F P_inf F^T, i.e. C <- P_inf + F (C - P_inf) F^T
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_predict_cov(const TpModel<NR, NC> &M, const TpTrans<NR, NC> &T, Sym<J> &C)
{
    double X[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) X[i][j] = C(i, j);
    // subtract P_inf (block diagonal)
#pragma unroll
    for (int j = 0; j < NR; ++j) X[j][j] -= M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
        X[o][o] -= M.ac[k]; X[o][o + 1] += M.bc[k]; X[o + 1][o] += M.bc[k]; X[o + 1][o + 1] -= M.pc[k];
    }
    tp_left_F<NR, NC, J>(T, X);  // F X
    // (F X) F^T: F acts on the columns -> apply F to every row
#pragma unroll
    for (int i = 0; i < J; ++i) tp_apply_F<NR, NC>(T, X[i]);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) C(i, j) = 0.5 * (X[i][j] + X[j][i]);
#pragma unroll
    for (int j = 0; j < NR; ++j) C(j, j) += M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
        C(o, o) += M.ac[k]; C(o + 1, o) -= M.bc[k]; C(o + 1, o + 1) += M.pc[k];
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

// one Kalman filter step from the FILTERED state of the previous sample:
// predict with T, update with (r, R); returns the pivot D and the residual z
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_filter_step(const TpModel<NR, NC> &M, const TpTrans<NR, NC> &T, double r, double R,
                                               double *m, Sym<J> &C, double &D, double &z)
{
    tp_apply_F<NR, NC>(T, m);
    tp_predict_cov<NR, NC, J>(M, T, C);
    double ch[J];
    tp_C_h<NR, NC, J>(C, ch);
    D = tp_h_dot<NR, NC>(ch) + R;
    z = r - tp_h_dot<NR, NC>(m);
    const double inv = 1.0 / D;
#pragma unroll
    for (int i = 0; i < J; ++i) m[i] = fma(ch[i], z * inv, m[i]);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) C(i, j) = fma(-ch[i] * inv, ch[j], C(i, j));
}

// filtering element of a chunk: x_out | x_in ~ N(A x_in + b, C), p(y_chunk | x_in) ~ N_I(eta, Jm)
template <int J> struct TpElem {
    double A[J][J];
    double b[J], eta[J];
    Sym<J> C, Jm;
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
}

// e <- e o step(T, r, R): the single-step element has a rank-one information part, so the
// composition is Sherman-Morrison algebra, O(J^2) (see the header of this file)
template <int NR, int NC, int J>
__device__ __forceinline__ void tp_compose_step(const TpModel<NR, NC> &M, const TpTrans<NR, NC> &T, double y, double R,
                                                TpElem<J> &e)
{
    // single-step quantities: Q = P_inf - F P_inf F^T, S2 = h^T Q h + R, K2 = Q h / S2, v = F^T h
    Sym<J> Q;
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) Q.v[i] = 0.0;
    tp_predict_cov<NR, NC, J>(M, T, Q);  // with C = 0: P_inf - F P_inf F^T
    double qh[J];
    tp_C_h<NR, NC, J>(Q, qh);
    const double S2 = tp_h_dot<NR, NC>(qh) + R;
    double K2[J];
#pragma unroll
    for (int i = 0; i < J; ++i) K2[i] = qh[i] / S2;
    double v[J];
#pragma unroll
    for (int i = 0; i < J; ++i) v[i] = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) v[j] = 1.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) v[NR + 2 * k] = 1.0;
    tp_apply_Ft<NR, NC>(T, v);
    // u = C1 v, gamma = S2 + v^T u, a = A1^T v, va = v^T A1 (row vector)
    double u[J], a[J];
    double gamma = S2;
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) s = fma(e.C(i, j), v[j], s);
        u[i] = s;
        gamma = fma(v[i], s, gamma);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < J; ++i) s = fma(e.A[i][j], v[i], s);
        a[j] = s;
    }
    const double ig = 1.0 / gamma;
    double vb = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) vb = fma(v[i], e.b[i], vb);
    // information part first (uses the OLD A1, b1)
    const double kappa = (y - vb) * ig;
#pragma unroll
    for (int i = 0; i < J; ++i) e.eta[i] = fma(a[i], kappa, e.eta[i]);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) e.Jm(i, j) = fma(a[i] * ig, a[j], e.Jm(i, j));
    // XA = A1 - u a^T / gamma ; Xb = w - u (v^T w) / gamma with w = b1 + u y / S2 ; XC = C1 - u u^T / gamma
    double w[J], vw = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) { w[i] = fma(u[i], y / S2, e.b[i]); vw = fma(v[i], w[i], vw); }
#pragma unroll
    for (int i = 0; i < J; ++i) {
        w[i] = fma(-u[i], vw * ig, w[i]);
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = fma(-u[i] * ig, a[j], e.A[i][j]);
    }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) e.C(i, j) = fma(-u[i] * ig, u[j], e.C(i, j));
    // A <- (I - K2 h^T) F XA ; b <- (I - K2 h^T) F Xb + K2 y
    tp_left_F<NR, NC, J>(T, e.A);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        double hg = 0.0;  // h^T (F XA)[:, j]
#pragma unroll
        for (int r = 0; r < NR; ++r) hg += e.A[r][j];
#pragma unroll
        for (int k = 0; k < NC; ++k) hg += e.A[NR + 2 * k][j];
#pragma unroll
        for (int i = 0; i < J; ++i) e.A[i][j] = fma(-K2[i], hg, e.A[i][j]);
    }
    tp_apply_F<NR, NC>(T, w);
    const double hw = tp_h_dot<NR, NC>(w);
#pragma unroll
    for (int i = 0; i < J; ++i) e.b[i] = fma(K2[i], y - hw, w[i]);
    // C <- (I - K2 h^T) F XC F^T (I - h K2^T) + C2,  C2 = Q - K2 K2^T S2
    double X[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) X[i][j] = e.C(i, j);
    tp_left_F<NR, NC, J>(T, X);
#pragma unroll
    for (int i = 0; i < J; ++i) tp_apply_F<NR, NC>(T, X[i]);
    double th[J], hth = 0.0;  // T h and h^T T h for T = F XC F^T
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) s += X[i][r];
#pragma unroll
        for (int k = 0; k < NC; ++k) s += X[i][NR + 2 * k];
        th[i] = s;
    }
    hth = tp_h_dot<NR, NC>(th);
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            e.C(i, j) = 0.5 * (X[i][j] + X[j][i]) - K2[i] * th[j] - th[i] * K2[j] + K2[i] * K2[j] * (hth - S2) + Q(i, j);
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

// e2 <- e1 o e2 (e1 earlier in time): the general combination of two filtering elements
//   G = I + C1 J2;  A = A2 G^-1 A1;  b = A2 G^-1 (b1 + C1 eta2) + b2;  C = A2 G^-1 C1 A2^T + C2
//   eta = A1^T G^-T (eta2 - J2 b1) + eta1;  J = A1^T G^-T J2 A1 + J1
template <int J>
__device__ __forceinline__ void tp_combine(const TpElem<J> &e1, TpElem<J> &e2)
{
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
#pragma unroll
    for (int p = 0; p < J; ++p) {
        const double ip = 1.0 / G[p][p];
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
    // information part (uses A1, b1, eta1, J1 and the OLD eta2, J2)
    double t[J], yeta[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e2.eta[i];
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
    double eta_new[J];
    Sym<J> J_new;
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e1.eta[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(e1.A[k][i], yeta[k], s);
        eta_new[i] = s;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double u = 0.0, v = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) { u = fma(e1.A[k][i], Z[k][j], u); v = fma(e1.A[k][j], Z[k][i], v); }
            J_new(i, j) = 0.5 * (u + v) + e1.Jm(i, j);
        }
    }
    // state part: XA = Gi A1, Xb = Gi (b1 + C1 eta2), XC = Gi C1
    double w[J], Xb[J], XA[J][J], XC[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e1.b[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(e1.C(i, k), e2.eta[k], s);
        w[i] = s;
    }
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
    double A_new[J][J], b_new[J], Y[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        double s = e2.b[i];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(e2.A[i][k], Xb[k], s);
        b_new[i] = s;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) { a = fma(e2.A[i][k], XA[k][j], a); c = fma(e2.A[i][k], XC[k][j], c); }
            A_new[i][j] = a;
            Y[i][j] = c;
        }
    }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double u = 0.0, v = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) { u = fma(Y[i][k], e2.A[j][k], u); v = fma(Y[j][k], e2.A[i][k], v); }
            e2.C(i, j) = 0.5 * (u + v) + e2.C(i, j);
        }
#pragma unroll
    for (int i = 0; i < J; ++i) {
        e2.b[i] = b_new[i];
        e2.eta[i] = eta_new[i];
#pragma unroll
        for (int j = 0; j < J; ++j) e2.A[i][j] = A_new[i][j];
    }
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) e2.Jm.v[i] = J_new.v[i];
}

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
}

}  // namespace

// one wave per evaluation; lane = chunk
// LANES = chunks per evaluation = workgroup size: 64 (one wave) or 256 (four waves, for the
// smallest batches: the scan then crosses waves through LDS and workgroup barriers)
template <int NR, int NC, bool FAST, int LANES>
__device__ __forceinline__ void mtg_tp_body(const MtgSolveArgs &a, const TpModel<NR, NC> &M, double jitter, double slope,
                                            double icpt, int64_t ev, int64_t lc, const MtgMathTables *tab, double *sh)
{
    constexpr int MTG_TP_LANES = LANES;
    constexpr int J = NR + 2 * NC;
    constexpr int ELEM = J * J + 2 * J + J * (J + 1);  // doubles per element
    const int lane = threadIdx.x;
    const int64_t N = a.N;
    const double2 *yv = a.yv + lc * N, *dxt = a.dxt + lc * a.t_stride;

    // chunk of this lane: samples [lo, hi); sample 0 is the prior update below
    const int64_t per = (N + MTG_TP_LANES - 1) / MTG_TP_LANES;
    int64_t lo = (int64_t)lane * per, hi = lo + per;
    if (lo > N) lo = N;
    if (hi > N) hi = N;
    if (lo == 0) lo = 1;

    // ---- pass 1: element of the chunk ---------------------------------------------------
    TpElem<J> e;
    tp_identity<J>(e);
    for (int64_t n = lo; n < hi; ++n) {
        TpTrans<NR, NC> T;
        tp_transition<NR, NC, FAST>(M, dxt[n].x, T, tab);
        const double r = yv[n].x - fma(slope, dxt[n].y, icpt);
        tp_compose_step<NR, NC, J>(M, T, r, yv[n].y + jitter, e);
    }

    // ---- pass 2: inclusive scan of the 64 chunk elements (Hillis-Steele through LDS) ------
    double *buf = sh;
    tp_store<J>(e, buf + lane * ELEM);
    __syncthreads();
    for (int off = 1; off < MTG_TP_LANES; off <<= 1) {
        TpElem<J> prev;
        if (lane >= off) tp_load<J>(prev, buf + (lane - off) * ELEM);
        __syncthreads();  // everybody has read its partner before anybody overwrites
        if (lane >= off) {
            tp_combine<J>(prev, e);
            tp_store<J>(e, buf + lane * ELEM);
        }
        __syncthreads();
    }
    double *cur = buf;
    // filtered state after sample 0 (update of the stationary prior), identical on every lane
    double m[J];
    Sym<J> C;
#pragma unroll
    for (int i = 0; i < J; ++i) m[i] = 0.0;
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
    const double D0 = tp_h_dot<NR, NC>(ch) + yv[0].y + jitter;
    const double z0 = yv[0].x - fma(slope, dxt[0].y, icpt);
#pragma unroll
    for (int i = 0; i < J; ++i) m[i] = ch[i] * z0 / D0;
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) C(i, j) -= ch[i] * ch[j] / D0;
    // start state of this lane's chunk: the prefix of the earlier chunks applied to it
    if (lane > 0) {
        TpElem<J> pre;
        tp_load<J>(pre, cur + (lane - 1) * ELEM);
        tp_apply_elem<J>(pre, m, C);
    }

    // ---- pass 3: ordinary Kalman filter over the chunk from its start state -------------
    double dot = lane == 0 ? z0 * z0 / D0 : 0.0, dprod = 1.0, dmin = lane == 0 ? D0 : INFINITY;
    int dexp = 0;
    for (int64_t n = lo; n < hi; ++n) {
        TpTrans<NR, NC> T;
        tp_transition<NR, NC, FAST>(M, dxt[n].x, T, tab);
        const double r = yv[n].x - fma(slope, dxt[n].y, icpt);
        double D, z;
        tp_filter_step<NR, NC, J>(M, T, r, yv[n].y + jitter, m, C, D, z);
        dot = fma(z * z, 1.0 / D, dot);
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
    if (LANES > 64) {  // combine the waves' partial sums through LDS (the element buffer is free now)
        __syncthreads();
        if ((lane & 63) == 0) { sh[3 * (lane >> 6)] = dot; sh[3 * (lane >> 6) + 1] = ld; sh[3 * (lane >> 6) + 2] = dmin; }
        __syncthreads();
        if (lane == 0)
            for (int w = 1; w < LANES / 64; ++w) { dot += sh[3 * w]; ld += sh[3 * w + 1]; dmin = fmin(dmin, sh[3 * w + 2]); }
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
__device__ __forceinline__ void mtg_tp_eval(const MtgSolveArgs &a, int64_t ev, const MtgMathTables *tab, double *sh)
{
    // ---- model of this evaluation (same on every lane) ------------------------------------
    TpModel<NR, NC> M;
    const double *cf = a.coef + ev;
    const int64_t cs = a.cstride;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) { M.ar[j] = cf[a.lay.ar(j) * cs]; M.cr[j] = cf[a.lay.cr(j) * cs]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double aa = cf[a.lay.ac(k) * cs], bb = cf[a.lay.bc(k) * cs], c = cf[a.lay.cc(k) * cs], d = cf[a.lay.dc(k) * cs];
        M.ac[k] = aa; M.bc[k] = bb; M.cc[k] = c; M.dc[k] = d;
        // free entry of P_inf: the value maximising det(noise covariance) (proto/kalman_scan.py)
        M.pc[k] = d != 0.0 ? (2.0 * d * (2.0 * c * bb + d * aa) + 4.0 * c * (c * aa - d * bb)) / (2.0 * d * d) : aa;
        dmax = fmax(dmax, fabs(d));
    }
    double ksum = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) ksum += M.ar[j];
#pragma unroll
    for (int k = 0; k < NC; ++k) ksum += M.ac[k];
    const double jitter = cf[a.lay.asum() * cs] - ksum;
    const double slope = cf[a.lay.mean(0) * cs], icpt = cf[a.lay.mean(1) * cs];
    const int64_t lc = a.lc_index ? (int64_t)a.lc_index[ev] : 0;
    // a device-side lc_index cannot be validated by the host: this kernel reads through raw
    // pointers (the throughput kernel's buffer loads are bounds-checked by the hardware)
    if (lc < 0 || (uint64_t)(lc + 1) * (uint64_t)a.N * 16u > (uint64_t)a.yv_bytes) {
        if (threadIdx.x == 0) { a.out[ev] = -INFINITY; a.status[ev] = MTG_ST_NONFINITE; }
        return;
    }
    if (dmax * *a.dxmax <= MTG_TRIG_FAST_MAX)
        mtg_tp_body<NR, NC, true, LANES>(a, M, jitter, slope, icpt, ev, lc, tab, sh);
    else
        mtg_tp_body<NR, NC, false, LANES>(a, M, jitter, slope, icpt, ev, lc, tab, sh);
}

template <int NR, int NC, int LANES>
__global__ void __launch_bounds__(LANES, 1) mtg_tp_kernel(MtgSolveArgs a)
{
    constexpr int J = NR + 2 * NC;
    __shared__ double sh[LANES * (J * J + 2 * J + J * (J + 1))];
    __shared__ MtgMathTables tab;
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if ((int64_t)blockIdx.x >= count) return;
    const int64_t ev = a.list ? (int64_t)a.list[blockIdx.x] : (int64_t)blockIdx.x;
    if (!a.list && a.status[ev] != MTG_ST_OK) return;
    mtg_fill_tables(&tab, threadIdx.x, LANES);
    __syncthreads();
    mtg_tp_eval<NR, NC, LANES>(a, ev, &tab, sh);
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
                                                double *sh)
{
    if (k == K) mtg_tp_eval<NR0 + 2 * K, NC0 - K, LANES>(a, ev, tab, sh);
    else if constexpr (K + 1 < NSIG) mtg_tp_dispatch<NR0, NC0, K + 1, NSIG, LANES>(k, a, ev, tab, sh);
}

template <int NR0, int NC0, int NSIG, int LANES>
__global__ void __launch_bounds__(LANES, 1) mtg_tp_fused_kernel(MtgSolveArgs a)
{
    constexpr int J = NR0 + 2 * NC0;
    static_assert(NSIG >= 2 && NSIG - 1 <= NC0, "one structure per number of over-damped SHO terms");
    __shared__ double sh[LANES * (J * J + 2 * J + J * (J + 1))];
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
    mtg_fill_tables(&tab, threadIdx.x, LANES);
    __syncthreads();
    mtg_tp_dispatch<NR0, NC0, 0, NSIG, LANES>(k, a, ev, &tab, sh);
}

template <int NR0, int NC0, int NSIG, int LANES>
static void mtg_launch_tp_fused(const MtgSolveArgs &a, int64_t nevals, hipStream_t stream)
{
    if (nevals <= 0) return;
    hipLaunchKernelGGL((mtg_tp_fused_kernel<NR0, NC0, NSIG, LANES>), dim3((unsigned)nevals), dim3(LANES), 0, stream, a);
}
