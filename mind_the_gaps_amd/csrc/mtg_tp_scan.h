// mtg_tp_scan.h -- the scan over the chunk elements of the big-J time-parallel path
// (mtg_tp_big.h), with every J x J operation of a combination spread over a group of 16 lanes.
//
// A filtering element of rank J = 10 is 230 doubles; combining two of them in one lane needs
// eight 10 x 10 products and an inverse -- ~800 doubles of temporaries, i.e. scratch memory (round 1:
// 9.8 KB per lane, 10 ms per launch).  Here no lane ever holds a matrix: lane r of a group owns ROW r
// of whatever is being computed (10 doubles), the operands sit in the group's LDS region and are
// read as broadcasts (every lane of the group reads the same row of the right-hand operand), so a
// product costs each lane J (J + J/2) issue slots and ~60 VGPRs; four groups share a wave.
//
// Structure of the scan (per evaluation, C chunk elements e_0 .. e_{C-1} in global memory):
//   up-sweep    mtg_tpb_reduce_kernel: groups of g (mtg_tp_big_gsize) consecutive elements are composed into
//               one (g - 1 sequential combinations per lane group), level after level until four
//               elements remain.  With the likelihood record (below) carried along, those four applied to
//               the state after sample 0 give lnL (mtg_tpb_top_direct_kernel): no down-sweep, no filter pass;
//   down-sweep  mtg_tpb_down_kernel (only for evaluations that need the filter pass): from the state after
//               sample 0 the start state of every top-level element by sequential application, then level
//               by level down to the chunks: the start state of element g k + i follows from that of group
//               k by applying i elements.
// Work: ~C g/(g-1) combinations (+ as many applications for the down-sweep) per evaluation (a
// Hillis-Steele scan needs C log2 C combinations); depth: g - 1 combinations per level.
//
// Likelihood record of an element, 4 doubles (dot, ld, dmin, mag): ln p(y_element | x_in = 0) =
// -1/2 (dot + ld) without the 2 pi terms.  For a chunk the composition pass leaves its filter's sum z^2/D,
// ln prod D and min D (parts[chunk][0..2]); a combination e1 o e2 adds (mtg_timeparallel.h has the derivation)
//   dot += -2 (lin + quad),  ld += ln det G,   lin = 1/2 b1^T (eta2 + t),  quad = 1/2 t^T G^-1 C1 t,
// G = I + C1 J2, t = eta2 - J2 b1; dmin becomes -1 (and ld NaN) when det G is not positive; mag sums the magnitudes.
//
// Global layouts (full matrices, so that loading is a plain copy):
//   element: A[J][J] | b[J] | eta[J] | C[J][J] | Jm[J][J]     MTG_TPB_ELEM(J) doubles
//   state:   m[J] | P[J][J]                                   MTG_TPB_STATE(J) doubles
#pragma once
#include "mtg_device.h"

#include <stdlib.h>

#define MTG_TPB_ELEM(J) (3 * (J) * (J) + 2 * (J))
#define MTG_TPB_STATE(J) ((J) * (J) + (J))
#define MTG_TPB_LANES 16    /* lanes per lane group */
#define MTG_TPB_MAX_LEVELS 8

// Workspace of the big-J path, in doubles from a.tp_ws: element and state arrays per scan level,
// per-chunk partial sums of the final filter pass, per-evaluation head (sample 0).
#define MTG_TPB_TOP 4        /* elements per evaluation at the top level */
struct MtgTpBigPlan {
    int C;                         // chunks per evaluation (a power of two >= 64)
    int g;                         // elements per scan group (4 or 16), fewer where a level has less than 4 g
    int nlev;                      // scan levels; level 0 = the chunks, level nlev - 1 has MTG_TPB_TOP elements
    int n[MTG_TPB_MAX_LEVELS];     // elements per evaluation at each level
    int gl[MTG_TPB_MAX_LEVELS];    // group size that takes level l to level l + 1
    int64_t elem_off[MTG_TPB_MAX_LEVELS], state_off[MTG_TPB_MAX_LEVELS];
    int64_t rec_off[MTG_TPB_MAX_LEVELS];   // likelihood records of the levels >= 1 (level 0: the parts array)
    int64_t part_off, head_off, redo_off, total;
};

static inline MtgTpBigPlan mtg_tp_big_plan(int J, int64_t B, int C, int g)
{
    MtgTpBigPlan p;
    p.C = C;
    p.g = g;
    p.nlev = 0;
    int64_t off = 0;
    for (int n = C;;) {
        const int l = p.nlev++;
        p.n[l] = n;
        p.elem_off[l] = off; off += B * n * MTG_TPB_ELEM(J);
        p.state_off[l] = off; off += B * n * MTG_TPB_STATE(J);
        p.rec_off[l] = off; off += l > 0 ? B * n * 4 : 0;
        p.gl[l] = 0;
        if (n <= MTG_TPB_TOP || p.nlev == MTG_TPB_MAX_LEVELS) break;
        p.gl[l] = g < n / MTG_TPB_TOP ? g : n / MTG_TPB_TOP;   // (C is a power of two >= 64: ends on exactly four)
        n /= p.gl[l];
    }
    p.part_off = off; off += B * C * 4;
    p.head_off = off; off += B * 4;
    // evaluations sent back through the filter pass (mtg_tp_big.h): int list + counter
    p.redo_off = off; off += (B + 16) / 2 + 1;
    p.total = off;
    return p;
}

// chunks per evaluation: enough (chunk, evaluation) pairs to fill the GPU ONCE -- the composition kernel gives 64
// chunks to a workgroup of two waves, one per SIMD, two workgroups to a CU: 512 x 64 = 32 768 chunks in flight --, at
// least 64, at most 4096, and no chunk shorter than ~24 samples.  (Round 2 asked for 65 536: two rounds of
// workgroups with chunks half as long take the composition exactly as long, and leave the scan twice the elements.)
static inline int mtg_tp_big_chunks(int64_t N, int64_t B)
{
    // chunks over the whole batch: two waves per SIMD in the composition (32 768), one for the smallest batches, whose up-sweep
    // is then half as long (scripts/c5_chunk_target.sh, N = 2e5, ms per half-step at 32 768 / 16 384: 8 rows 0.311 / 0.265,
    // 16 rows 0.385 / 0.349, 32 rows 0.540 / 0.534, 64 rows 0.843 / 0.876, 256 rows 2.66 / 2.87)
    int64_t target = B <= 16 ? 16384 : 32768;
    if (const char *env = mtg_measure_env("MTG_TP_CHUNK_TARGET")) {  // MTG_MEASURE builds only: the chunk count decides a result's bits
        const long v = atol(env);
        if (v >= 64) target = v;
    }
    int C = 64;
    while (C < 4096 && (int64_t)C * B < target && (int64_t)C * 2 * 24 <= N) C *= 2;
    return C;
}

// elements per scan group.  A group is one wave's chain of g - 1 dependent combinations (6.5 us each when the wave
// has a SIMD to itself, ~1 us of LDS traffic per CU when the GPU is full): the number of combinations of the whole
// scan is the number of elements whatever g is, so the short chains of g = 4 cost nothing but a launch per level
// (~5 us) and cut the depth from 15 to 3 combinations per level.
static inline int mtg_tp_big_gsize(int64_t B, int C)
{
    (void)B; (void)C;
    return 4;
}

// up-sweep; kappa != 0: every group (the last one's total too) and the likelihood records; zero_me: an int the first
// level's first workgroup clears (the redo counter), or NULL
void mtg_launch_tpb_up(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, int kappa, int *zero_me,
                       hipStream_t stream);
// lnL from the top-level elements and their records; suspects are appended to the redo list
void mtg_launch_tpb_top_direct(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, int *redo_list,
                               int *redo_count, hipStream_t stream);
// down-sweep: the start state of every chunk (for the filter pass)
void mtg_launch_tpb_down(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, hipStream_t stream);
// the whole path (mtg_tp_big_filter.hip): every prepared evaluation of a rank-10 model, whatever its structure
void mtg_launch_tp_big(const MtgSolveArgs &a, int64_t nevals, hipStream_t stream);

#ifdef __HIPCC__
namespace tpg {

// wave-level ordering of LDS traffic between the lanes of a group (they run in lock-step: one
// wave; LDS operations of a wave complete in order)
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int J> struct Lds {  // one lane group's region
    static constexpr int M = J * J;
    double A1[M], eta1[J], b1[J], C1[M], J1[M];   // running element (reduce) / state (b1 | C1, contiguous) (down)
    double A2[M], b2[J], eta2[J], C2[M], J2[M];   // the next element, in the order of the global layout
    double T1[M], T2[M], T3[M];
    double v1[J], v2[J], v3[J];
    double piv[2 * J];
};

template <int J> __device__ __forceinline__ void row(const double *X, int r, double (&x)[J])
{
#pragma unroll
    for (int k = 0; k < J; ++k) x[k] = X[r * J + k];
}
template <int J> __device__ __forceinline__ void col(const double *X, int r, double (&x)[J])
{
#pragma unroll
    for (int k = 0; k < J; ++k) x[k] = X[k * J + r];
}
template <int J> __device__ __forceinline__ void put(double *X, int r, const double (&x)[J])
{
#pragma unroll
    for (int k = 0; k < J; ++k) X[r * J + k] = x[k];
}
// o += x Y (x a row vector, Y in LDS, its rows read by every lane of the group)
template <int J> __device__ __forceinline__ void mm(const double (&x)[J], const double *Y, double (&o)[J])
{
#pragma unroll
    for (int k = 0; k < J; ++k) {
#pragma unroll
        for (int j = 0; j < J; ++j) o[j] = fma(x[k], Y[k * J + j], o[j]);
    }
}
// o += x Y^T
template <int J> __device__ __forceinline__ void mmT(const double (&x)[J], const double *Y, double (&o)[J])
{
#pragma unroll
    for (int j = 0; j < J; ++j) {
        double s = o[j];
#pragma unroll
        for (int k = 0; k < J; ++k) s = fma(x[k], Y[j * J + k], s);
        o[j] = s;
    }
}
template <int J> __device__ __forceinline__ double dot(const double (&x)[J], const double *v)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < J; ++k) s = fma(x[k], v[k], s);
    return s;
}

// Lanes J .. 15 of a group have no row of their own: they are given r = J - 1 and duplicate that row's
// work, identical stores included (predicating the stores instead makes the compiler sink the arithmetic
// into the predicated block while the LDS reads stay where they are -- every operand then lives in
// scratch memory in between).
//
// Row r of (I + X Y)^-1 by Gauss-Jordan without pivoting (I + C J is similar to a symmetric positive
// definite matrix); the pivot row travels through L.piv.  Returns the row of the inverse in gi and the
// reciprocal of the product of the pivots (1 / det) as dm * 2^de.
template <int J>
__device__ __forceinline__ void inv_ipxy(Lds<J> &L, const double *X, const double *Y, int r, double (&gi)[J],
                                         double &dm, int &de)
{
    double g[J], x[J];
    row<J>(X, r, x);
#pragma unroll
    for (int j = 0; j < J; ++j) { g[j] = j == r ? 1.0 : 0.0; gi[j] = g[j]; }
    mm<J>(x, Y, g);
    dm = 1.0; de = 0;
#pragma unroll
    for (int p = 0; p < J; ++p) {
        if (r == p) {
            const double ip = 1.0 / g[p];
#pragma unroll
            for (int j = 0; j < J; ++j) { g[j] *= ip; gi[j] *= ip; }
#pragma unroll
            for (int j = 0; j < J; ++j) { L.piv[j] = g[j]; L.piv[J + j] = gi[j]; }
            L.v3[0] = ip;
        }
        wsync();
        const double f = r == p ? 0.0 : g[p];
        {
            // (single pivots of I + X Y may be negative -- it is not symmetric -- only their product is a sign test)
            const double pr = dm * L.v3[0];
            dm = __builtin_amdgcn_frexp_mant(pr);
            de += __builtin_amdgcn_frexp_exp(pr);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) { g[j] = fma(-f, L.piv[j], g[j]); gi[j] = fma(-f, L.piv[J + j], gi[j]); }
        wsync();
    }
}

// running element (A1, b1, eta1, C1, J1) <- (running) o (A2, b2, eta2, C2, J2), the running one earlier in
// time:  G = I + C1 J2;  A = A2 G^-1 A1;  b = A2 G^-1 (b1 + C1 eta2) + b2;  C = A2 G^-1 C1 A2^T + C2;
//        eta = A1^T G^-T (eta2 - J2 b1) + eta1;  J = A1^T G^-T J2 A1 + J1,
// with G^-T J2 = J2 G^-1 (push-through identity).
// KAPPA: kap = (lin, quad) of the likelihood record's update and 1 / det G = dm 2^de (same on every lane of
// the group; the caller takes the logarithm of the product of a whole group's determinants once).
template <int J, bool KAPPA = false>
__device__ __forceinline__ void combine(Lds<J> &L, int r, double (&kap)[2], double &dm, int &de)
{
    double gi[J];
    inv_ipxy<J>(L, L.C1, L.J2, r, gi, dm, de);
    // w = b1 + C1 eta2 -> v1 ; t = eta2 - J2 b1 -> v2 ; Gi -> T1
    {
        double x[J];
        row<J>(L.C1, r, x);
        const double w = L.b1[r] + dot<J>(x, L.eta2);
        row<J>(L.J2, r, x);
        const double t = L.eta2[r] - dot<J>(x, L.b1);
        L.v1[r] = w; L.v2[r] = t;
        put<J>(L.T1, r, gi);
    }
    wsync();
    // XA = Gi A1 -> T2 ; XC = Gi C1 -> T3 ; xb = Gi w -> v3
    {
        double xa[J], xc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { xa[j] = 0.0; xc[j] = 0.0; }
        mm<J>(gi, L.A1, xa);
        mm<J>(gi, L.C1, xc);
        const double xb = dot<J>(gi, L.v1);
        put<J>(L.T2, r, xa);
        put<J>(L.T3, r, xc);
        L.v3[r] = xb;
        if (KAPPA) {  // this row's share of lin and of quad (t in v2; piv is free after the inverse)
            const double tr = L.v2[r];
            L.piv[r] = 0.5 * L.b1[r] * (L.eta2[r] + tr);
            L.piv[J + r] = 0.5 * tr * dot<J>(xc, L.v2);
        }
    }
    // yeta = G^-T t (column r of Gi; stays in this lane until A1's columns are read);
    // YJ = G^-T J2 = J2 Gi ; Z = YJ A1
    double yeta, z[J];
    {
        double x[J], yj[J];
        col<J>(L.T1, r, x);
        yeta = dot<J>(x, L.v2);
        row<J>(L.J2, r, x);
#pragma unroll
        for (int j = 0; j < J; ++j) { yj[j] = 0.0; z[j] = 0.0; }
        mm<J>(x, L.T1, yj);
        mm<J>(yj, L.A1, z);
    }
    wsync();                       // everybody has read Gi (T1) and v2
    if (KAPPA) {
        double lin = 0.0, quad = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) { lin += L.piv[k]; quad += L.piv[J + k]; }
        kap[0] = lin; kap[1] = quad;
    }
    put<J>(L.T1, r, z);       // Z -> T1
    L.v2[r] = yeta;
    wsync();
    // new information part: eta = eta1 + A1^T yeta ; J = J1 + A1^T Z   (column r of A1)
    double eta_new, j_new[J];
    {
        double a1c[J];
        col<J>(L.A1, r, a1c);
        eta_new = L.eta1[r] + dot<J>(a1c, L.v2);
        row<J>(L.J1, r, j_new);
        mm<J>(a1c, L.T1, j_new);
    }
    // new state part: A = A2 XA ; Y = A2 XC ; C = C2 + Y A2^T ; b = b2 + A2 xb
    double a_new[J], c_new[J], b_new;
    {
        double a2[J], y[J];
        row<J>(L.A2, r, a2);
#pragma unroll
        for (int j = 0; j < J; ++j) { a_new[j] = 0.0; y[j] = 0.0; }
        mm<J>(a2, L.T2, a_new);
        mm<J>(a2, L.T3, y);
        row<J>(L.C2, r, c_new);
        mmT<J>(y, L.A2, c_new);
        b_new = L.b2[r] + dot<J>(a2, L.v3);
    }
    wsync();                       // all reads of the old running element are done
    put<J>(L.A1, r, a_new);
    put<J>(L.C1, r, c_new);
    put<J>(L.J1, r, j_new);
    L.b1[r] = b_new; L.eta1[r] = eta_new;
    wsync();
}

// state (m in b1, P in C1) <- element (A2, b2, eta2, C2, J2) applied to it:
//   G = I + P J2;  m' = A2 G^-1 (m + P eta2) + b2;  P' = A2 G^-1 P A2^T + C2
// CORR: also returns the element's chunk likelihood given the state it is applied to, relative to
// the chunk likelihood given x_in = 0 (kappa, which the composition pass accumulates):
//   ln p(y_chunk | m, P) - kappa = -1/2 ln det G + eta^T m - 1/2 m^T J m + 1/2 t^T (G^-1 P) t,  t = eta - J m
// (p(y_chunk | x_in) = exp(kappa + eta^T x_in - 1/2 x_in^T J x_in), integrated against N(m, P)).
// NaN when det G is not positive.  UPDATE = false: only that number, the state is left alone.
template <int J, bool CORR = false, bool UPDATE = true> __device__ __forceinline__ double apply(Lds<J> &L, int r)
{
    double gi[J], dm;
    int de;
    inv_ipxy<J>(L, L.C1, L.J2, r, gi, dm, de);
    double t = 0.0, mr = 0.0, er = 0.0;
    {
        double x[J];
        row<J>(L.C1, r, x);
        const double w = L.b1[r] + dot<J>(x, L.eta2);
        L.v1[r] = w;
        if (CORR) {
            row<J>(L.J2, r, x);
            mr = L.b1[r]; er = L.eta2[r];
            t = er - dot<J>(x, L.b1);
            L.v2[r] = t;
        }
    }
    wsync();
    double corr = 0.0;
    {
        double xc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xc[j] = 0.0;
        mm<J>(gi, L.C1, xc);
        const double xb = dot<J>(gi, L.v1);
        put<J>(L.T3, r, xc);
        L.v3[r] = xb;
        if (CORR) {
            const double xt = dot<J>(xc, L.v2);
            // this row's share of eta^T m - 1/2 m^T J m + 1/2 t^T X t   (J m = eta - t)
            L.piv[r] = fma(er, mr, 0.5 * fma(t, xt, -mr * (er - t)));
            wsync();
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < J; ++k) sum += L.piv[k];
            // 1 / det G = dm 2^de > 0, or some pivot was not positive
            corr = dm > 0.0 ? sum + 0.5 * (log(dm) + (double)de * 0.69314718055994530942) : __builtin_nan("");
        }
    }
    if (!UPDATE) { wsync(); return corr; }
    wsync();
    double c_new[J], b_new;
    {
        double a2[J], y[J];
        row<J>(L.A2, r, a2);
#pragma unroll
        for (int j = 0; j < J; ++j) y[j] = 0.0;
        mm<J>(a2, L.T3, y);
        row<J>(L.C2, r, c_new);
        mmT<J>(y, L.A2, c_new);
        b_new = L.b2[r] + dot<J>(a2, L.v3);
    }
    wsync();
    put<J>(L.C1, r, c_new);
    L.b1[r] = b_new;
    wsync();
    return corr;
}

// copy n doubles global <-> LDS by the 16 lanes of a group
__device__ __forceinline__ void gcopy(double *dst, const double *src, int n, int l16)
{
    for (int i = l16; i < n; i += MTG_TPB_LANES) dst[i] = src[i];
}

// The next element travels global memory -> registers (issued before the current combination, whose
// arithmetic hides the latency) -> LDS (after it): MTG_TPB_ELEM(J) / 16 doubles per lane.
template <int J> struct Pre { double v[(MTG_TPB_ELEM(J) + MTG_TPB_LANES - 1) / MTG_TPB_LANES]; };
template <int J> __device__ __forceinline__ void fetch(Pre<J> &p, const double *e, int l16)
{
    constexpr int N = MTG_TPB_ELEM(J), Q = (N + MTG_TPB_LANES - 1) / MTG_TPB_LANES;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = l16 + MTG_TPB_LANES * q;
        p.v[q] = e[i < N ? i : N - 1];
    }
}
template <int J> __device__ __forceinline__ void put_second(Lds<J> &L, const Pre<J> &p, int l16)
{
    constexpr int N = MTG_TPB_ELEM(J), Q = (N + MTG_TPB_LANES - 1) / MTG_TPB_LANES;
    double *dst = L.A2;  // A2 | b2 | eta2 | C2 | J2 are contiguous and in the global order
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = l16 + MTG_TPB_LANES * q;
        if (i < N) dst[i] = p.v[q];
    }
}
template <int J> __device__ __forceinline__ void load_first(Lds<J> &L, const double *e, int l16)
{
    constexpr int M = J * J;
    gcopy(L.A1, e, M, l16);
    gcopy(L.b1, e + M, J, l16);
    gcopy(L.eta1, e + M + J, J, l16);
    gcopy(L.C1, e + M + 2 * J, M, l16);
    gcopy(L.J1, e + 2 * M + 2 * J, M, l16);
}
template <int J> __device__ __forceinline__ void store_first(const Lds<J> &L, double *e, int l16)
{
    constexpr int M = J * J;
    gcopy(e, L.A1, M, l16);
    gcopy(e + M, L.b1, J, l16);
    gcopy(e + M + J, L.eta1, J, l16);
    gcopy(e + M + 2 * J, L.C1, M, l16);
    gcopy(e + 2 * M + 2 * J, L.J1, M, l16);
}

}  // namespace tpg

// ---------------------------------------------------------------------------------------------------------------
// One WAVE per combination (round 3).  The 16-lane groups above leave a combination 13 us long -- a chain of ~1100
// dependent FP64 instructions per lane and 7.9 KB of LDS per group, four groups to a wave, five waves to a CU -- and
// the up-sweep is a chain of such combinations: 0.37 ms of a 0.85 ms half-step at 32 evaluations.  Here lane (r, q) of
// a wave owns the column PAIR (2q, 2q + 1) of row r of whatever J x J matrix is being computed (J = 10: 50 lanes at
// work, the other 14 repeat lane 49), so that a product costs a lane 2 J multiply-adds instead of J^2, the elimination
// J steps of 4 instead of 2 J, and a wave's 8.8 KB of LDS let eighteen of them share a CU.  Operands are read from
// LDS as before -- row r of the left factor (the same address for the five lanes of a row), pairs of the right factor's
// rows (the same address for the ten lanes of a column pair).  There is one wave, so every "barrier" is the wave's own
// LDS ordering (tpg::wsync).
namespace tpw {

using tpg::wsync;

template <int J> struct alignas(16) Lds {
    static_assert(J % 2 == 0, "column pairs");
    static constexpr int M = J * J;
    double A1[M], eta1[J], b1[J], C1[M], J1[M];   // running element (b1 | C1 contiguous: the state of `apply`)
    double A2[M], b2[J], eta2[J], C2[M], J2[M];   // the next element, in the order of the global layout
    double T1[M], T2[M], T3[M], T4[M];
    double v1[J], v2[J], v3[J], v4[J], v5[J];
    double kq[J], kl[J];
};

struct Lane {
    int r, c0, q;
};
template <int J> __device__ __forceinline__ Lane lane_of(int l64)
{
    constexpr int NQ = J / 2, LAST = J * NQ - 1;
    const int l = l64 < LAST ? l64 : LAST;
    Lane w;
    w.r = l / NQ;
    w.q = l - w.r * NQ;
    w.c0 = 2 * w.q;
    return w;
}

// Pairs travel as 16-byte LDS accesses (rows are J = even doubles long and every array of Lds starts on a 16-byte
// boundary): ds_read_b128 / ds_write_b128 take a 16-bit immediate offset, where the compiler's ds_read2_b64 of two
// neighbouring doubles reaches 2 KB only and pays a v_add for every other base address.
__device__ __forceinline__ double2 ld2(const double *p) { return *(const double2 *)p; }
__device__ __forceinline__ void st2(double *p, double a, double b) { *(double2 *)p = make_double2(a, b); }

template <int J> __device__ __forceinline__ void row(const double *X, int r, double (&x)[J])
{
#pragma unroll
    for (int k = 0; k < J; k += 2) {
        const double2 v = ld2(X + r * J + k);
        x[k] = v.x; x[k + 1] = v.y;
    }
}
template <int J> __device__ __forceinline__ void col(const double *X, int r, double (&x)[J])
{
#pragma unroll
    for (int k = 0; k < J; ++k) x[k] = X[k * J + r];
}
// o += x Y, columns c0 and c0 + 1
template <int J> __device__ __forceinline__ void mm2(const double (&x)[J], const double *Y, int c0, double (&o)[2])
{
#pragma unroll
    for (int k = 0; k < J; ++k) {
        const double2 y = ld2(Y + k * J + c0);
        o[0] = fma(x[k], y.x, o[0]);
        o[1] = fma(x[k], y.y, o[1]);
    }
}
// o += x Y^T, columns c0 and c0 + 1
template <int J> __device__ __forceinline__ void mmT2(const double (&x)[J], const double *Y, int c0, double (&o)[2])
{
#pragma unroll
    for (int k = 0; k < J; k += 2) {
        const double2 ya = ld2(Y + c0 * J + k), yb = ld2(Y + (c0 + 1) * J + k);
        o[0] = fma(x[k], ya.x, o[0]); o[0] = fma(x[k + 1], ya.y, o[0]);
        o[1] = fma(x[k], yb.x, o[1]); o[1] = fma(x[k + 1], yb.y, o[1]);
    }
}
template <int J> __device__ __forceinline__ void put2(double *X, int r, int c0, const double (&o)[2])
{
    st2(X + r * J + c0, o[0], o[1]);
}
template <int J> __device__ __forceinline__ double dot(const double (&x)[J], const double *v)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < J; k += 2) {
        const double2 y = ld2(v + k);
        s = fma(x[k], y.x, s); s = fma(x[k + 1], y.y, s);
    }
    return s;
}
template <int J> __device__ __forceinline__ double sum(const double *v)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < J; k += 2) {
        const double2 y = ld2(v + k);
        s += y.x; s += y.y;
    }
    return s;
}
// 1 / d to the last bit or so: v_rcp_f64 and two Newton steps (the IEEE division sequence is ~30 instructions)
__device__ __forceinline__ double rcp2(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-d, x, 1.0);
    return __builtin_fma(x, e, x);
}

// Columns (c0, c0 + 1) of row r of (I + X Y)^-1 by Gauss-Jordan without pivoting (I + C J is similar to a symmetric
// positive definite matrix); xr = row r of X.  Both halves of the augmented matrix [G | Gi] live in registers, two
// entries each per lane; at every step all lanes publish theirs (T2, T3: free during the elimination -- no
// predicated store, no branch), read back the pivot, their row's entry of the pivot column and their column pair of
// the pivot row, scale that row themselves and eliminate -- row p included, with the multiplier (pivot - 1), which
// leaves it the scaled pivot row.  One LDS round trip per step.  1 / det = dm 2^de (single pivots may be negative,
// only their product is a sign test).
template <int J>
__device__ __forceinline__ void inv_ipxy(Lds<J> &L, const double (&xr)[J], const double *Y, const Lane w, double (&gi)[2],
                                         double &dm, int &de)
{
    double g[2];
    g[0] = w.c0 == w.r ? 1.0 : 0.0;
    g[1] = w.c0 + 1 == w.r ? 1.0 : 0.0;
    gi[0] = g[0]; gi[1] = g[1];
    mm2<J>(xr, Y, w.c0, g);
    dm = 1.0; de = 0;
    double *Gm = L.T2, *GIm = L.T3;
#pragma unroll
    for (int p = 0; p < J; ++p) {
        put2<J>(Gm, w.r, w.c0, g);
        put2<J>(GIm, w.r, w.c0, gi);
        wsync();
        const double pv = Gm[p * J + p];
        const double fr = Gm[w.r * J + p];
        const double2 pg = ld2(Gm + p * J + w.c0), pgi = ld2(GIm + p * J + w.c0);
        const double ip = rcp2(pv);
        const double fs = (w.r == p ? pv - 1.0 : fr) * ip;
        g[0] = fma(-fs, pg.x, g[0]);
        g[1] = fma(-fs, pg.y, g[1]);
        gi[0] = fma(-fs, pgi.x, gi[0]);
        gi[1] = fma(-fs, pgi.y, gi[1]);
        const double pr = dm * ip;
        dm = __builtin_amdgcn_frexp_mant(pr);
        de += __builtin_amdgcn_frexp_exp(pr);
    }
}

// running element (A1, b1, eta1, C1, J1) <- (running) o (A2, b2, eta2, C2, J2), the running one earlier in time
// (the formulas of tpg::combine; J2 G^-1 A1 is taken as J2 (G^-1 A1), so that G^-T J2 is never formed).
template <int J, bool KAPPA = false>
__device__ __forceinline__ void combine(Lds<J> &L, const Lane w, double (&kap)[2], double &dm, int &de)
{
    const int r = w.r, c0 = w.c0;
    double gi[2];
    // w = b1 + C1 eta2 -> v1 ; t = eta2 - J2 b1 -> v2 ; u = C1 t needs t of every row: after the elimination
    double c1r[J];
    row<J>(L.C1, r, c1r);
    {
        double x[J];
        L.v1[r] = L.b1[r] + dot<J>(c1r, L.eta2);
        row<J>(L.J2, r, x);
        L.v2[r] = L.eta2[r] - dot<J>(x, L.b1);
    }
    inv_ipxy<J>(L, c1r, L.J2, w, gi, dm, de);
    put2<J>(L.T1, r, c0, gi);   // Gi -> T1
    if (KAPPA) L.v5[r] = dot<J>(c1r, L.v2);   // C1 t (t of every row has been visible since the elimination's first step)
    wsync();
    // XA = Gi A1 -> T2 ; XC = Gi C1 -> T3 ; xb = Gi w -> v3 ; yeta = Gi^T t -> v4 ; quad = 1/2 t^T XC t, lin = 1/2 b1^T (eta2 + t)
    {
        double gr[J], o[2];
        row<J>(L.T1, r, gr);
        o[0] = 0.0; o[1] = 0.0;
        mm2<J>(gr, L.A1, c0, o);
        put2<J>(L.T2, r, c0, o);
        o[0] = 0.0; o[1] = 0.0;
        mm2<J>(gr, L.C1, c0, o);
        put2<J>(L.T3, r, c0, o);
        L.v3[r] = dot<J>(gr, L.v1);
        if (KAPPA) {
            const double ct = dot<J>(gr, L.v5);   // (XC t)_r = Gi_r . (C1 t)
            const double tr = L.v2[r];
            L.kq[r] = 0.5 * tr * ct;
            L.kl[r] = 0.5 * L.b1[r] * (L.eta2[r] + tr);
        }
        double gc[J];
        col<J>(L.T1, r, gc);
        L.v4[r] = dot<J>(gc, L.v2);
    }
    wsync();
    // Z = J2 XA -> T1 ; A = A2 XA ; Y = A2 XC -> T4 ; b = b2 + A2 xb
    double a_new[2], b_new;
    {
        double x[J], o[2];
        row<J>(L.J2, r, x);
        o[0] = 0.0; o[1] = 0.0;
        mm2<J>(x, L.T2, c0, o);
        row<J>(L.A2, r, x);
        a_new[0] = 0.0; a_new[1] = 0.0;
        mm2<J>(x, L.T2, c0, a_new);
        double y[2] = {0.0, 0.0};
        mm2<J>(x, L.T3, c0, y);
        b_new = L.b2[r] + dot<J>(x, L.v3);
        if (KAPPA) { kap[0] = sum<J>(L.kl); kap[1] = sum<J>(L.kq); }
        put2<J>(L.T1, r, c0, o);    // (every lane has read its row and column of Gi in the phase before)
        put2<J>(L.T4, r, c0, y);
    }
    wsync();
    // eta = eta1 + A1^T yeta ; J = J1 + A1^T Z ; C = C2 + Y A2^T
    double eta_new, j_new[2], c_new[2];
    {
        double x[J];
        col<J>(L.A1, r, x);
        eta_new = L.eta1[r] + dot<J>(x, L.v4);
        { const double2 v = ld2(L.J1 + r * J + c0); j_new[0] = v.x; j_new[1] = v.y; }
        mm2<J>(x, L.T1, c0, j_new);
        row<J>(L.T4, r, x);
        { const double2 v = ld2(L.C2 + r * J + c0); c_new[0] = v.x; c_new[1] = v.y; }
        mmT2<J>(x, L.A2, c0, c_new);
    }
    wsync();                       // all reads of the old running element are done
    put2<J>(L.A1, r, c0, a_new);
    put2<J>(L.C1, r, c0, c_new);
    put2<J>(L.J1, r, c0, j_new);
    L.b1[r] = b_new; L.eta1[r] = eta_new;
    wsync();
}

// state (m in b1, P in C1) <- element (A2, b2, eta2, C2, J2) applied to it (tpg::apply): CORR also returns the element's
// chunk likelihood given that state relative to kappa; UPDATE = false: only that number.
template <int J, bool CORR = false, bool UPDATE = true> __device__ __forceinline__ double apply(Lds<J> &L, const Lane w)
{
    const int r = w.r, c0 = w.c0;
    double gi[2], dm;
    int de;
    double t = 0.0, mr = 0.0, er = 0.0;
    double c1r[J];
    row<J>(L.C1, r, c1r);
    L.v1[r] = L.b1[r] + dot<J>(c1r, L.eta2);
    if (CORR) {
        double x[J];
        row<J>(L.J2, r, x);
        mr = L.b1[r]; er = L.eta2[r];
        t = er - dot<J>(x, L.b1);
        L.v2[r] = t;
    }
    inv_ipxy<J>(L, c1r, L.J2, w, gi, dm, de);
    put2<J>(L.T1, r, c0, gi);
    if (CORR) L.v5[r] = dot<J>(c1r, L.v2);
    wsync();
    double corr = 0.0;
    {
        double gr[J], o[2] = {0.0, 0.0};
        row<J>(L.T1, r, gr);
        mm2<J>(gr, L.C1, c0, o);
        put2<J>(L.T3, r, c0, o);
        L.v3[r] = dot<J>(gr, L.v1);
        if (CORR) {
            const double ct = dot<J>(gr, L.v5);
            // this row's share of eta^T m - 1/2 m^T J m + 1/2 t^T X t   (J m = eta - t)
            L.kq[r] = fma(er, mr, 0.5 * fma(t, ct, -mr * (er - t)));
            wsync();
            const double s = sum<J>(L.kq);
            corr = dm > 0.0 ? s + 0.5 * (log(dm) + (double)de * 0.69314718055994530942) : __builtin_nan("");
        }
    }
    if (!UPDATE) { wsync(); return corr; }
    wsync();
    double b_new;
    {
        double x[J], y[2] = {0.0, 0.0};
        row<J>(L.A2, r, x);
        mm2<J>(x, L.T3, c0, y);
        b_new = L.b2[r] + dot<J>(x, L.v3);
        put2<J>(L.T4, r, c0, y);
    }
    wsync();
    double c_new[2];
    {
        double x[J];
        row<J>(L.T4, r, x);
        { const double2 v = ld2(L.C2 + r * J + c0); c_new[0] = v.x; c_new[1] = v.y; }
        mmT2<J>(x, L.A2, c0, c_new);
    }
    wsync();
    put2<J>(L.C1, r, c0, c_new);
    L.b1[r] = b_new;
    wsync();
    return corr;
}

// copy n doubles global <-> LDS by the 64 lanes of the wave
__device__ __forceinline__ void gcopy(double *dst, const double *src, int n, int l64)
{
    for (int i = l64; i < n; i += 64) dst[i] = src[i];
}

// The next element travels global memory -> registers (issued before the current combination, whose arithmetic
// hides the latency) -> LDS (after it): MTG_TPB_ELEM(J) / 64 doubles per lane.
template <int J> struct Pre { double v[(MTG_TPB_ELEM(J) + 63) / 64]; };
template <int J> __device__ __forceinline__ void fetch(Pre<J> &p, const double *e, int l64)
{
    constexpr int N = MTG_TPB_ELEM(J), Q = (N + 63) / 64;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = l64 + 64 * q;
        p.v[q] = e[i < N ? i : N - 1];
    }
}
template <int J> __device__ __forceinline__ void put_second(Lds<J> &L, const Pre<J> &p, int l64)
{
    constexpr int N = MTG_TPB_ELEM(J), Q = (N + 63) / 64;
    double *dst = L.A2;  // A2 | b2 | eta2 | C2 | J2 are contiguous and in the global order
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = l64 + 64 * q;
        if (i < N) dst[i] = p.v[q];
    }
}
template <int J> __device__ __forceinline__ void load_first(Lds<J> &L, const double *e, int l64)
{
    constexpr int M = J * J;
    gcopy(L.A1, e, M, l64);
    gcopy(L.b1, e + M, J, l64);
    gcopy(L.eta1, e + M + J, J, l64);
    gcopy(L.C1, e + M + 2 * J, M, l64);
    gcopy(L.J1, e + 2 * M + 2 * J, M, l64);
}
template <int J> __device__ __forceinline__ void store_first(const Lds<J> &L, double *e, int l64)
{
    constexpr int M = J * J;
    gcopy(e, L.A1, M, l64);
    gcopy(e + M, L.b1, J, l64);
    gcopy(e + M + J, L.eta1, J, l64);
    gcopy(e + M + 2 * J, L.C1, M, l64);
    gcopy(e + 2 * M + 2 * J, L.J1, M, l64);
}

}  // namespace tpw
#endif  // __HIPCC__
