// mtg_tp_big.h -- time-parallel log-likelihood for the rank-10 structures (five SHOTerms, BASELINE
// configs[4]: N = 2e5 samples, 512 walkers), where neither the filtering element (230 doubles) nor
// the combination of two of them fits the registers of one lane.
//
// Round 1 ran the whole algorithm of mtg_timeparallel.h in one kernel with one lane per chunk: 256
// chunks per evaluation, the scan's combinations in scratch memory (9.8 KB per lane), ~10-14 ms per
// launch whatever the batch.  Here the passes are separate kernels, each with the geometry that suits
// it, none with scratch memory, and the number of chunks follows the batch so that 32 evaluations fill
// the GPU as well as 256 do (mtg_tp_big_chunks):
//   compose  mtg_tpb_compose4q_kernel: the element of every chunk by the filter-from-zero recursion, FOUR
//            waves per 64 chunks (two keep A's columns, two the symmetric matrices Dv and Jm; see below);
//            also the chunk's likelihood given x_in = 0 (kappa);
//   up-sweep mtg_tp_scan.h: combinations level by level down to four elements per evaluation, each J x J
//            operation spread over 16 lanes with the operands in LDS -- no lane holds a matrix; the
//            likelihood records (kappa and the combinations' contributions) travel along;
//   top      mtg_tpb_top_direct_kernel: those four elements applied to the state after sample 0 give lnL --
//            no further pass over the data.  Evaluations whose terms cancel badly, or that met a pivot that
//            is not positive, go through
//   down-sweep + filter   mtg_tpb_down_kernel: every chunk's start state; mtg_tpb_filter_kernel<NR, NC>:
//            lane = chunk, the ordinary Kalman filter over the chunk from its start state: celerite's own
//            pivots and residuals (mtg_tpb_finish_kernel sums them).
// The light curve is cut into C chunks of `per` samples after sample 0, whose update of the
// stationary prior is the scan's initial state (mtg_tpb_down_kernel).
#pragma once
#include "mtg_timeparallel.h"
#include "mtg_tp_scan.h"

// byte masks of the table look-ups of the composition kernels.  -DMTG_DBG_TABLE_BROADCAST (measurements only, wrong
// results): every lane reads entry 0 -- a broadcast, no bank conflict -- which tells the conflicts of the look-ups (64
// random addresses per wave-instruction) from those of the rings (lane-strided doubles: conflict-free by construction)
#ifdef MTG_DBG_TABLE_BROADCAST
#define TPB_EXP_MASK 0
#define TPB_TRIG_MASK 0
#else
#define TPB_EXP_MASK ((MTG_EXP_N - 1) * 8)
#define TPB_TRIG_MASK ((MTG_TRIG_N - 1) * 16)
#endif

namespace {

// tp_predict_dev with the real x real part scaled row by row (phi_i phi_j formed inside the entry's
// own product chain instead of as 55 simultaneous temporaries) and a scheduling barrier per block row
template <int NR, int NC, int J>
__device__ __forceinline__ void tpb_predict_dev(const TpTrans<NR, NC> &T, Sym<J> &Dv)
{
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) Dv(i, j) = (Dv(i, j) * T.phi[i]) * T.phi[j];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int o = NR + 2 * k;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const double x0 = Dv(o, j) * T.phi[j], x1 = Dv(o + 1, j) * T.phi[j];
            Dv(o, j) = T.ec[k] * x0 - T.es[k] * x1;
            Dv(o + 1, j) = T.es[k] * x0 + T.ec[k] * x1;
        }
#pragma unroll
        for (int l = 0; l < k; ++l) {
            const int ol = NR + 2 * l;
            const double b00 = Dv(o, ol), b01 = Dv(o, ol + 1), b10 = Dv(o + 1, ol), b11 = Dv(o + 1, ol + 1);
            const double y00 = T.ec[k] * b00 - T.es[k] * b10, y01 = T.ec[k] * b01 - T.es[k] * b11;
            const double y10 = T.es[k] * b00 + T.ec[k] * b10, y11 = T.es[k] * b01 + T.ec[k] * b11;
            Dv(o, ol) = y00 * T.ec[l] - y01 * T.es[l];
            Dv(o, ol + 1) = y00 * T.es[l] + y01 * T.ec[l];
            Dv(o + 1, ol) = y10 * T.ec[l] - y11 * T.es[l];
            Dv(o + 1, ol + 1) = y10 * T.es[l] + y11 * T.ec[l];
        }
        {
            const double d00 = Dv(o, o), d10 = Dv(o + 1, o), d11 = Dv(o + 1, o + 1);
            const double y00 = T.ec[k] * d00 - T.es[k] * d10, y01 = T.ec[k] * d10 - T.es[k] * d11;
            const double y10 = T.es[k] * d00 + T.ec[k] * d10, y11 = T.es[k] * d10 + T.ec[k] * d11;
            Dv(o, o) = y00 * T.ec[k] - y01 * T.es[k];
            Dv(o + 1, o) = y10 * T.ec[k] - y11 * T.es[k];
            Dv(o + 1, o + 1) = y10 * T.es[k] + y11 * T.ec[k];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Transition of one step, in two phases so that the LDS latency of the table look-ups is paid once per
// step, not once per term (one wave per SIMD: nothing else hides it): phase 1 reduces every argument and
// issues every look-up, phase 2 runs the polynomials.
//   exp(-c dx) as mtg_exp_cdx (mtg_math.h) with the per-term constants -c and -c 8N/ln2 replaced by one
//   product per step, dxs = dx 8N/ln2: the model of an evaluation is uniform over the workgroup and sits
//   in SGPRs, but anything COMPUTED from it is a vector value (there is no scalar FP64 unit) -- two
//   hoisted doubles per term are 40 registers at ten real terms, which the compose kernel does not have.
//   fast = false (some d_k * max dx beyond the exact range of the table reduction): the phase increment
//   is first reduced modulo 2 pi with a two-part constant -- n = rint(x / 2 pi) is exact in a double, the
//   fused multiply-adds form x - n C1 - n C2 with one rounding each, and the rounding error of the
//   product d * dx itself is carried along -- and the remainder, |r| <= pi, goes through the same table
//   path.  (OCML's sincos / exp here cost a second copy of the loop and ~150 bytes of scratch per lane.)
template <int NR, int NC, class Tab>
__device__ __forceinline__ void tpb_transition(const TpModel<NR, NC> &M, double dx, TpTrans<NR, NC> &T, const Tab *tab,
                                               bool fast)
{
    constexpr int NT = NR + NC;
    const double dxs = dx * MTG_EXP_CSCALE;
    const double magic = 0x1.8p+55;                                                 // 1.5 * 2^(52+3)
    double er[NT > 0 ? NT : 1], et[NT > 0 ? NT : 1];   // remainder and table value of every exp
    int ek[NT > 0 ? NT : 1];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const double c = i < NR ? M.cr[i < NR ? i : 0] : M.cc[i < NR ? 0 : i - NR];
        const double w = __builtin_fma(-c, dxs, magic);
        const double q8 = w - magic;                                                // 8 rint(-c dx N / ln2)
        const int i8 = (int)q8;
        et[i] = *(const double *)((const char *)tab->exp2_frac + (i8 & TPB_EXP_MASK));
        er[i] = __builtin_fma(-c, dx, q8 * -MTG_EXP_C1);
        ek[i] = i8 >> (3 + MTG_EXP_BITS);
    }
    double pr[NC > 0 ? NC : 1];
    double2 pj[NC > 0 ? NC : 1];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        double dk = M.dc[k], xk = dx;
        if (!fast) {  // uniform over the workgroup (one evaluation)
            const double x = dk * dx, xl = fma(dk, dx, -x);
            const double n = rint(x * 0x1.45f306dc9c883p-3);       // x / 2 pi
            const double r = fma(-n, 0x1.921fb54442d18p+2, x);      // 2 pi, head
            xk = fma(-n, 0x1.1a62633145c07p-52, r) + xl;            // 2 pi, tail
            dk = 1.0;
        }
        // mtg_phase_step from phase 0 (mtg_math.h): reduction, table entry
        const double tm = 0x1.8p+56;                                                // 1.5 * 2^(52+4)
        const double x = dk * xk;
        const double w = __builtin_fma(x, 0x1.45f306dc9c883p+1 * MTG_TRIG_N, tm);   // x 16 N / 2 pi
        const double md16 = w - tm;
        pr[k] = __builtin_fma(md16, -(0x1.921fb54442d18p-2 / MTG_TRIG_N), x);       // 2 pi / 16 N
        pj[k] = *(const double2 *)((const char *)tab->cis + ((__double2loint(w) << 4) & TPB_TRIG_MASK));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const double p = mtg_expm1_small(er[i]);
        const double e = __builtin_ldexp(__builtin_fma(et[i], p, et[i]), ek[i]);
        if (i < NR) {
            T.phi[i < NR ? i : 0] = e;
        } else {
            const int k = i < NR ? 0 : i - NR;
            double sn, cs;
            mtg_sincos_small(pr[k], &sn, &cs);
            const double s = __builtin_fma(pj[k].x, sn, pj[k].y * cs), c = __builtin_fma(-pj[k].y, sn, pj[k].x * cs);
            T.ec[k] = e * c;
            T.es[k] = e * s;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// model of evaluation `ev` (uniform over the workgroup: scalar loads); false = light-curve index
// outside the resident set
template <int NR, int NC>
__device__ __forceinline__ bool tpb_load_model(const MtgSolveArgs &a, int64_t ev, TpModel<NR, NC> &M, double &jitter,
                                               double &slope, double &icpt, int64_t &lc, bool &fast)
{
    const double *cf = a.coef + ev;
    const int64_t cs = a.cstride;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) { M.ar[j] = cf[a.lay.ar(j) * cs]; M.cr[j] = cf[a.lay.cr(j) * cs]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double aa = cf[a.lay.ac(k) * cs], bb = cf[a.lay.bc(k) * cs], c = cf[a.lay.cc(k) * cs], d = cf[a.lay.dc(k) * cs];
        M.ac[k] = aa; M.bc[k] = bb; M.cc[k] = c; M.dc[k] = d;
        M.pc[k] = d != 0.0 ? (2.0 * d * (2.0 * c * bb + d * aa) + 4.0 * c * (c * aa - d * bb)) / (2.0 * d * d) : aa;
        dmax = fmax(dmax, fabs(d));
    }
    jitter = cf[a.lay.jit() * cs];
    slope = cf[a.lay.mean(0) * cs];
    icpt = cf[a.lay.mean(1) * cs];
    lc = a.lc_index ? (int64_t)a.lc_index[ev] : 0;
    // (1e5, not MTG_TRIG_FAST_MAX: beyond it tpb_transition's two-part reduction carries the rounding error of
    // d * dx along, which is more accurate than the plain product and costs four instructions)
    fast = dmax * *a.dxmax <= 1.0e5;
    return !(lc < 0 || (uint64_t)(lc + 1) * (uint64_t)a.N * 16u > (uint64_t)a.yv_bytes);
}

// workgroup -> evaluation through the structure's list; -1 = nothing to do
__device__ __forceinline__ int64_t tpb_evaluation(const MtgSolveArgs &a, int64_t i)
{
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if (i >= count) return -1;
    const int64_t ev = a.list ? (int64_t)a.list[i] : i;
    if (!a.list && a.status[ev] != MTG_ST_OK) return -1;
    return ev;
}

// samples [lo, hi) of chunk c: C chunks of `per` samples after sample 0.  32-bit indices: a light curve
// is at most 2^28 samples (its 16-byte records stay below the 4 GiB window, mtg_set_lightcurves), and
// the loops address samples as (uniform 64-bit base) + (32-bit byte offset per lane).
__device__ __forceinline__ void tpb_chunk_range(int64_t N, int C, uint32_t c, uint32_t &lo, uint32_t &hi)
{
    const uint32_t n = (uint32_t)N, per = (n - 1u + (uint32_t)C - 1u) / (uint32_t)C;
    const uint64_t l = 1u + (uint64_t)c * per, h = l + per;
    lo = l > n ? n : (uint32_t)l;
    hi = h > n ? n : (uint32_t)h;
}

__device__ __forceinline__ double2 tpb_sample(const double2 *base, uint32_t byte_off)
{
    return *(const double2 *)((const char *)base + byte_off);
}

// ---------------------------------------------------------------------------------------------
// Composition by FOUR waves per 64 chunks (mtg_tpb_compose4q_kernel).
//
// One lane per chunk cannot hold the element (A, b, Dv, eta, Jm: 330 doubles) in directly addressable registers, and
// a lone wave per SIMD is handed one instruction of ANY kind per issue slot: its register copies, LDS and scalar
// instructions cost as much as its FP64 ones (rounds 1-2: one and two waves per 64 chunks, docs/HISTORY.md).  Here
// the element of a chunk is shared by the same lane of FOUR waves, each below 256 registers, two workgroups = eight
// waves per CU, two waves per SIMD: nothing lives in accumulation registers and one wave's LDS / scalar / wait
// instructions issue under the other's arithmetic.  Roles:
//   0 "columns, low"   columns 0..4 of A, eta[0..4], the transition of the first terms
//   1 "columns, high"  columns 5..9 of A, eta[5..9], the transition of the other terms, the mean b and the residual z
//   2, 3 "filter"      the symmetric matrices Dv and Jm, cut block by block of the term structure into two halves of
//                      equal work (tpb_owner): prediction, gain and update of the own blocks.  The gain needs
//                      (P_inf + Dv) h, a sum over all blocks: each wave sums its own, they exchange the partial sums.
// One workgroup barrier per tick; a step travels through the waves in five ticks (s = step, t = tick):
//   t = s      waves 0, 1   transition F(s)                                        -> ring F[s & 3]
//   t = s + 1  waves 2, 3   part A: predict own blocks with F(s), partial (Dv h)    -> ring part[wave][s & 1]
//   t = s + 2  waves 2, 3   part B: sum the partials, pivot D, 1 / D, update own blocks;  wave 2 -> ring ch[s & 1]
//   t = s + 3  waves 0, 1   A <- (I - K h) F A for the own columns, g = h F A      -> ring g[s & 1];  wave 1: b, z,
//                           z / D -> ring zi[s & 1], eta[5..9]
//   t = s + 4  waves 2, 3   Jm += g g^T / D on the own blocks;  wave 0: eta[0..4]
// A workgroup is one such quartet; two of them share a CU with independent barriers, and a wave takes its role from
// the SIMD it runs on (mtg_tp_big_compose4q.hip), so every SIMD runs one "columns" and one "filter" wave: whatever the
// imbalance between the two kinds, the four SIMDs carry the same load, and a wave deep in LDS traffic shares its SIMD
// with one deep in arithmetic.
// Every hand-over crosses exactly one barrier and part A of step s + 1 follows part B of step s in the same
// tick of the same wave (the only true recurrence, Dv, never waits for another wave inside a tick).  Chunks shorter
// than `per` steps run the rest as no-ops: dx = 0 makes the transition the identity exactly, and a measurement variance
// of 1e300 makes the gain vanish below rounding.
// (measurements only: -DMTG_TPB4_NOSYNC times the arithmetic without its barriers -- the results are then wrong)
#ifdef MTG_TPB4_NOSYNC
#define TPB4_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define TPB4_SYNC() __syncthreads()
#endif
// THE ROTATING FRAME (round 5).  In the SDE basis a complex term's propagator is e R(d dx), a rotation: predicting the
// deviation, Dv <- F Dv F^T, costs 16 operations per 2 x 2 block of the triangle (230 of the step's ~1070 FP64
// operations at five complex terms) and F A another 20 per column.  The composition therefore runs in the frame that
// turns with every term, x~_n = R(-Theta_n) x_n with Theta_n = d (t_n - t_start) accumulated from the chunk's start:
//   propagator   Phi = diag(e^{-c dx}) -- the same factor for both components of a complex term, no rotation;
//   observation  h~_n = R(-Theta_n) h = (cos Theta_n, -sin Theta_n) per complex term, 1 per real term;
//   stationary covariance  P~_n = R(-Theta_n) P_inf R(Theta_n), so P~_n h~_n = R(-Theta_n) P_inf h = (a C - b S, -a S - b C);
//   deviation form as before: Dv~ <- Phi Dv~ Phi (one product per entry), ch~ = (P~_n + Dv~) h~_n, D = h~^T ch~ + R.
// These are celerite's own generators (U, V rotate, the propagator is diagonal), seen from the state-space side.  The
// frame coincides with the fixed one at the chunk's start (Theta = 0), so eta and J -- which refer to the incoming state
// -- are the fixed-frame ones; A, b and C leave the chunk in the turned frame and are rotated back ONCE, after the last
// step: A = R(Theta_e) A~, b = R(Theta_e) b~, C = R(Theta_e) Dv~ R(Theta_e)^T + P_inf.  Pivots D and residuals z do not
// depend on the frame, so neither does kappa.  The ring hands over, per step, the propagators and (cos, sin) of the
// ACCUMULATED phase (mtg_phase_step's exact one-constant reduction, mtg_math.h): NR + 3 NC doubles.
template <int J> struct TpbRing4 {
    double F[4][J + J / 2][64];   // [0, NR): phi; complex k: e at NR + 3 k, cos Theta at + 1, sin Theta at + 2
    double ch[2][J + 1][64];
    double g[2][J][64];
    double zi[2][64];
    double part[2][2][J][64];
};

// term of state row i, first row of term a, rows of term a
template <int NR> __host__ __device__ constexpr int tpb_term(int i) { return i < NR ? i : NR + (i - NR) / 2; }
template <int NR> __host__ __device__ constexpr int tpb_row0(int a) { return a < NR ? a : NR + 2 * (a - NR); }
// which filter wave owns block (a, b), a >= b, of the symmetric matrices: alternating along the triangle -- within 6 %
// of an even split of the work for all six structures
__host__ __device__ constexpr int tpb_owner(int a, int b) { return (a * (a + 1) / 2 + b) & 1; }
template <int NR> __host__ __device__ constexpr bool tpb_owns(int f, int i, int j)
{
    return tpb_owner(tpb_term<NR>(i > j ? i : j), tpb_term<NR>(i > j ? j : i)) == f;
}
// terms whose generators wave 0 computes (the others: wave 1, which also has b and z): real ~14 instructions, complex ~37
template <int NR, int NC> __host__ __device__ constexpr int tpb_trans_split()
{
    const int total = 14 * NR + 37 * NC, target = (total + 30) / 2;
    int n = 0, c = 0;
    while (n < NR + NC && c < target) { c += n < NR ? 14 : 37; ++n; }
    return n;
}

// generators of one step as the waves read them back from a ring slot
template <int NR, int NC> struct TpbGen {
    double phi[NR > 0 ? NR : 1];
    double e[NC > 0 ? NC : 1], c[NC > 0 ? NC : 1], s[NC > 0 ? NC : 1];
};
template <int NR, int NC> __device__ __forceinline__ void tpb4_read_gen(const double (*F)[64], int lane, TpbGen<NR, NC> &G)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) G.phi[j] = F[j][lane];
#pragma unroll
    for (int q = 0; q < NC; ++q) { G.e[q] = F[NR + 3 * q][lane]; G.c[q] = F[NR + 3 * q + 1][lane]; G.s[q] = F[NR + 3 * q + 2][lane]; }
}
// (cos, sin) of the accumulated phase alone (part B of the filter waves, the rotation back at the chunk's end)
template <int NR, int NC> __device__ __forceinline__ void tpb4_read_cis(const double (*F)[64], int lane, double *c, double *s)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) { c[q] = F[NR + 3 * q + 1][lane]; s[q] = F[NR + 3 * q + 2][lane]; }
}
// x <- Phi x (rows of a state vector or of a column of A)
template <int NR, int NC> __device__ __forceinline__ void tpb4_scale(const TpbGen<NR, NC> &G, double *x)
{
#pragma unroll
    for (int j = 0; j < NR; ++j) x[j] *= G.phi[j];
#pragma unroll
    for (int q = 0; q < NC; ++q) { x[NR + 2 * q] *= G.e[q]; x[NR + 2 * q + 1] *= G.e[q]; }
}
// h~^T x
template <int NR, int NC> __device__ __forceinline__ double tpb4_hdot(const double *c, const double *s, const double *x)
{
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) sum += x[j];
#pragma unroll
    for (int q = 0; q < NC; ++q) sum = fma(c[q], x[NR + 2 * q], fma(-s[q], x[NR + 2 * q + 1], sum));
    return sum;
}
// x <- R(Theta) x on the rows of every complex term (back to the fixed frame)
template <int NR, int NC> __device__ __forceinline__ void tpb4_rotate_back(const double *c, const double *s, double *x)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const double x0 = x[NR + 2 * q], x1 = x[NR + 2 * q + 1];
        x[NR + 2 * q] = c[q] * x0 - s[q] * x1;
        x[NR + 2 * q + 1] = s[q] * x0 + c[q] * x1;
    }
}

// generators of terms [T0, T1) of one step straight into a ring slot; pr / pm: the accumulated phase of the wave's
// complex terms (remainder and table byte offset, mtg_phase_step), carried from step to step.  Two phases so that the
// LDS latency of the table look-ups is paid once per step: phase 1 reduces every argument and issues every look-up,
// phase 2 runs the polynomials.  fast = false (some d_k * max dx beyond 1e5): the phase increment is first reduced
// modulo 2 pi with a two-part constant, the rounding error of the product d * dx itself carried along.
template <int NR, int NC, int T0, int T1, class Tab>
__device__ __forceinline__ void tpb_generators_part(const TpModel<NR, NC> &M, double dx, double (*F)[64], int lane, const Tab *tab,
                                                    bool fast, double *pr, int *pm)
{
    constexpr int NTP = T1 - T0 > 0 ? T1 - T0 : 1;
    constexpr int C0 = T0 > NR ? T0 - NR : 0, C1 = T1 > NR ? T1 - NR : 0, NCP = C1 - C0 > 0 ? C1 - C0 : 1;
    const double dxs = dx * MTG_EXP_CSCALE;
    const double magic = 0x1.8p+55;
    double er[NTP], et[NTP];
    int ek[NTP];
#pragma unroll
    for (int i = T0; i < T1; ++i) {
        const double c = i < NR ? M.cr[i < NR ? i : 0] : M.cc[i < NR ? 0 : i - NR];
        const double w = __builtin_fma(-c, dxs, magic);
        const double q8 = w - magic;
        const int i8 = (int)q8;
        et[i - T0] = *(const double *)((const char *)tab->exp2_frac + (i8 & TPB_EXP_MASK));
        er[i - T0] = __builtin_fma(-c, dx, q8 * -MTG_EXP_C1);
        ek[i - T0] = i8 >> (3 + MTG_EXP_BITS);
    }
    double2 pj[NCP];
#pragma unroll
    for (int k = C0; k < C1; ++k) {
        double dk = M.dc[k], xk = dx;
        if (!fast) {  // uniform over the workgroup (one evaluation)
            const double x = dk * dx, xl = fma(dk, dx, -x);
            const double n = rint(x * 0x1.45f306dc9c883p-3);
            const double r = fma(-n, 0x1.921fb54442d18p+2, x);
            xk = fma(-n, 0x1.1a62633145c07p-52, r) + xl;
            dk = 1.0;
        }
        // mtg_phase_step (mtg_math.h): add the increment to the remainder, re-reduce, accumulate the table offset
        const double tm = 0x1.8p+56;
        const double x = __builtin_fma(dk, xk, pr[k - C0]);
        const double w = __builtin_fma(x, 0x1.45f306dc9c883p+1 * MTG_TRIG_N, tm);
        const double md16 = w - tm;
        pr[k - C0] = __builtin_fma(md16, -(0x1.921fb54442d18p-2 / MTG_TRIG_N), x);
        asm("v_lshl_add_u32 %0, %1, 4, %0" : "+v"(pm[k - C0]) : "v"(__double2loint(w)));
        pj[k - C0] = *(const double2 *)((const char *)tab->cis + (pm[k - C0] & TPB_TRIG_MASK));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = T0; i < T1; ++i) {
        const double p = mtg_expm1_small(er[i - T0]);
        const double e = __builtin_ldexp(__builtin_fma(et[i - T0], p, et[i - T0]), ek[i - T0]);
        if (i < NR) {
            F[i][lane] = e;
        } else {
            const int k = i < NR ? 0 : i - NR;
            double sn, cs;
            mtg_sincos_small(pr[k - C0], &sn, &cs);
            const double2 t = pj[k - C0];
            F[NR + 3 * k][lane] = e;
            F[NR + 3 * k + 1][lane] = __builtin_fma(-t.y, sn, t.x * cs);   // cos Theta
            F[NR + 3 * k + 2][lane] = __builtin_fma(t.x, sn, t.y * cs);    // sin Theta
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// waves 0 and 1: HALF = 0 / 1
template <int NR, int NC, int HALF, class Tab>
__device__ __forceinline__ void tpb4_columns(const MtgSolveArgs &a, const TpModel<NR, NC> &M, double slope, double icpt, int64_t lc,
                                             const Tab *tab, bool fast, TpbRing4<NR + 2 * NC> &ring, double *slot, double *part,
                                             uint32_t lo, uint32_t hi, uint32_t per)
{
    constexpr int J = NR + 2 * NC, H = J / 2, C0 = HALF * H, NT = NR + NC, NT0 = tpb_trans_split<NR, NC>();
    constexpr int T0 = HALF ? NT0 : 0, T1 = HALF ? NT : NT0;
    constexpr int PC0 = T0 > NR ? T0 - NR : 0, PC1 = T1 > NR ? T1 - NR : 0, NPC = PC1 - PC0 > 0 ? PC1 - PC0 : 1;
    const int lane = threadIdx.x & 63;
    const double2 *dxt = a.dxt + lc * a.t_stride, *yv = a.yv + lc * a.N;
    double A[J][H];   // columns C0 .. C0 + H - 1
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < H; ++j) A[i][j] = i == C0 + j ? 1.0 : 0.0;
    double eta[H], gk[H], b[J];
#pragma unroll
    for (int j = 0; j < H; ++j) { eta[j] = 0.0; gk[j] = 0.0; }
#pragma unroll
    for (int i = 0; i < J; ++i) b[i] = 0.0;
    double pr[NPC];   // accumulated phase of the complex terms whose generators this wave computes
    int pm[NPC];
#pragma unroll
    for (int k = 0; k < NPC; ++k) { pr[k] = 0.0; pm[k] = 0; }
    double dot = 0.0;
    const uint32_t last = ((uint32_t)a.N - 1u) * 16u;
    // sample of the generators (step t) and, wave 1, of the residual (step t - 3), each with its prefetch
    uint32_t off = lo * 16u, offz = lo * 16u;
    double dxn = tpb_sample(dxt, off < last ? off : last).x;
    double2 yn = tpb_sample(yv, offz < last ? offz : last);
    double tn = tpb_sample(dxt, offz < last ? offz : last).y;
    for (uint32_t t = 0; t < per + 4u; ++t) {
        if (t < per) {
            const double dx = off < hi * 16u ? dxn : 0.0;
            off += 16u;
            dxn = tpb_sample(dxt, off < last ? off : last).x;
            tpb_generators_part<NR, NC, T0, T1>(M, dx, ring.F[t & 3u], lane, tab, fast, pr, pm);
        }
        if (HALF == 0 && t >= 4u && t < per + 4u) {   // eta of step t - 4 with last tick's g
            const double zi = ring.zi[(t - 4u) & 1u][lane];
#pragma unroll
            for (int j = 0; j < H; ++j) eta[j] = fma(gk[j], zi, eta[j]);
        }
        if (t >= 3u && t < per + 3u) {
            const uint32_t s = t - 3u;
            TpbGen<NR, NC> G;
            tpb4_read_gen<NR, NC>(ring.F[s & 3u], lane, G);
            const double(*Gc)[64] = ring.ch[s & 1u];
            double ch[J];
#pragma unroll
            for (int i = 0; i < J; ++i) ch[i] = Gc[i][lane];
            const double inv = Gc[J][lane];
            double zi = 0.0;
            if (HALF == 1) {   // mean of the filter-from-zero and residual of step s
                const bool valid = offz < hi * 16u;
                const double r = valid ? fma(-slope, tn, yn.x - icpt) : 0.0;
                offz += 16u;
                yn = tpb_sample(yv, offz < last ? offz : last);
                tn = tpb_sample(dxt, offz < last ? offz : last).y;
                tpb4_scale<NR, NC>(G, b);
                const double z = r - tpb4_hdot<NR, NC>(G.c, G.s, b);
                zi = z * inv;
                ring.zi[s & 1u][lane] = zi;
                dot = fma(z, zi, dot);
#pragma unroll
                for (int i = 0; i < J; ++i) b[i] = fma(ch[i], zi, b[i]);
            }
            double(*gout)[64] = ring.g[s & 1u];
#pragma unroll
            for (int j = 0; j < H; ++j) {
                double col[J];
#pragma unroll
                for (int i = 0; i < J; ++i) col[i] = A[i][j];
                tpb4_scale<NR, NC>(G, col);
                const double gj = tpb4_hdot<NR, NC>(G.c, G.s, col);
                const double gs = gj * inv;
#pragma unroll
                for (int i = 0; i < J; ++i) A[i][j] = fma(-ch[i], gs, col[i]);
                gout[C0 + j][lane] = gj;
                if (HALF == 1) eta[j] = fma(gj, zi, eta[j]);
                else gk[j] = gj;
            }
        }
        TPB4_SYNC();
    }
    // back to the fixed frame: the accumulated (cos, sin) of the chunk's last step are still in their ring slot
    double ce[NC > 0 ? NC : 1], se[NC > 0 ? NC : 1];
    tpb4_read_cis<NR, NC>(ring.F[(per - 1u) & 3u], lane, ce, se);
    constexpr int MM = J * J;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        double col[J];
#pragma unroll
        for (int i = 0; i < J; ++i) col[i] = A[i][j];
        tpb4_rotate_back<NR, NC>(ce, se, col);
#pragma unroll
        for (int i = 0; i < J; ++i) slot[i * J + C0 + j] = col[i];
    }
#pragma unroll
    for (int j = 0; j < H; ++j) slot[MM + J + C0 + j] = eta[j];
    if (HALF == 1) {
        tpb4_rotate_back<NR, NC>(ce, se, b);
#pragma unroll
        for (int i = 0; i < J; ++i) slot[MM + i] = b[i];
        part[0] = dot;
    }
}

// waves 2 and 3: FW = 0 / 1 (FW = 0 also publishes ch, 1 / D and keeps the pivot statistics)
template <int NR, int NC, int FW>
__device__ __forceinline__ void tpb4_filter(const MtgSolveArgs &a, const TpModel<NR, NC> &M, double jitter, int64_t lc,
                                            TpbRing4<NR + 2 * NC> &ring, double *slot, double *part, uint32_t lo, uint32_t hi,
                                            uint32_t per)
{
    constexpr int J = NR + 2 * NC, MM = J * J, NT = NR + NC;
    const int lane = threadIdx.x & 63;
    const double2 *yv = a.yv + lc * a.N;
    Sym<J> Dv, Jm;   // only the entries of the own blocks are ever touched (the others never become registers)
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) { Dv.v[i] = 0.0; Jm.v[i] = 0.0; }
    tp_sub_pinf<NR, NC, J>(M, Dv);  // C - P_inf, C = 0 (Theta = 0 at the chunk's start: both frames agree)
    double chp[J];                  // own partial sums of Dv h~ of the step in flight
    double kap1 = 1.0, kap2 = INFINITY;
    int kexp = 0;
    double inv_b = 0.0, inv_1 = 0.0, inv_2 = 0.0;   // 1 / D of the steps t - 2 (part B of this tick), t - 3, t - 4
    const uint32_t last = ((uint32_t)a.N - 1u) * 16u;
    uint32_t off = lo * 16u;
    double vn = tpb_sample(yv, off < last ? off : last).y;
    for (uint32_t t = 0; t < per + 4u; ++t) {
        inv_2 = inv_1; inv_1 = inv_b;
        if (t >= 4u && t < per + 4u) {   // Jm of step t - 4
            const double(*G)[64] = ring.g[(t - 4u) & 1u];
            double g[J];
#pragma unroll
            for (int j = 0; j < J; ++j) g[j] = G[j][lane];
#pragma unroll
            for (int i = 0; i < J; ++i) {
                const double gi = g[i] * inv_2;
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if (tpb_owns<NR>(FW, i, j)) Jm(i, j) = fma(gi, g[j], Jm(i, j));
                if (NR > 6 && (i & 1)) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);   // (the three parts of a tick one after the other: their operands do not pile up)
        }
        if (t >= 2u && t < per + 2u) {   // part B of step t - 2
            const uint32_t s = t - 2u;
            const bool valid = off < hi * 16u;
            const double R = valid ? vn + jitter : 1.0e300;
            off += 16u;
            vn = tpb_sample(yv, off < last ? off : last).y;
            const double(*P)[64] = ring.part[1 - FW][s & 1u];
            double cs[NC > 0 ? NC : 1], sn[NC > 0 ? NC : 1];
            tpb4_read_cis<NR, NC>(ring.F[s & 3u], lane, cs, sn);
            double ch[J];
#pragma unroll
            for (int i = 0; i < J; ++i) {
                // P~_n h~_n = R(-Theta) P_inf h: a per real term, (a C - b S, -a S - b C) per complex term
                double ph;
                if (i < NR) ph = M.ar[i < NR ? i : 0];
                else {
                    const int q = i < NR ? 0 : (i - NR) / 2;
                    ph = (i - NR) & 1 ? -fma(M.ac[q], sn[q], M.bc[q] * cs[q]) : fma(M.ac[q], cs[q], -(M.bc[q] * sn[q]));
                }
                ch[i] = (chp[i] + P[i][lane]) + ph;
            }
            const double D = tpb4_hdot<NR, NC>(cs, sn, ch) + R;
            const double inv = mtg_rcp(D);
            inv_b = inv;
            if (FW == 0) {
                double(*G)[64] = ring.ch[s & 1u];
#pragma unroll
                for (int i = 0; i < J; ++i) G[i][lane] = ch[i];
                G[J][lane] = inv;
                const double pr = kap1 * (valid ? D : 1.0);
                kexp += __builtin_amdgcn_frexp_exp(pr);
                kap1 = __builtin_amdgcn_frexp_mant(pr);
                kap2 = fmin(kap2, D);
            }
#pragma unroll
            for (int i = 0; i < J; ++i) {
                const double ki = ch[i] * inv;
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if (tpb_owns<NR>(FW, i, j)) Dv(i, j) = fma(-ki, ch[j], Dv(i, j));
                if (NR > 6 && (i & 1)) __builtin_amdgcn_sched_barrier(0);   // (real-heavy structures: row pairs one after the other, or the scaled rows pile up)
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            inv_b = 0.0;
        }
        if (t >= 1u && t < per + 1u) {   // part A of step t - 1
            const uint32_t s = t - 1u;
            TpbGen<NR, NC> G;
            tpb4_read_gen<NR, NC>(ring.F[s & 3u], lane, G);
            // Dv <- Phi Dv Phi on the own blocks: one product of propagators per block, one multiplication per entry
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) {
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) {
                    if (tpb_owner(ta, tb) != FW) continue;
                    const int oa = tpb_row0<NR>(ta), ob = tpb_row0<NR>(tb);
                    const bool ca = ta >= NR, cb = tb >= NR;
                    const double pa = ca ? G.e[ca ? ta - NR : 0] : G.phi[ca ? 0 : ta], pb = cb ? G.e[cb ? tb - NR : 0] : G.phi[cb ? 0 : tb];
                    if (!ca) {          // real x real: one entry, scaled inside its own product chain (no temporary)
                        Dv(oa, ob) = (Dv(oa, ob) * pa) * pb;
                    } else {
                        const double pp = pa * pb;
                        Dv(oa, ob) *= pp;
                        Dv(oa + 1, ob) *= pp;
                        if (cb) {
                            Dv(oa + 1, ob + 1) *= pp;
                            if (ta != tb) Dv(oa, ob + 1) *= pp;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // own share of Dv h~: entry (i, j), i >= j, feeds row i with h~_j and -- off the diagonal -- row j with h~_i
            // (h~ = 1 for a real row, cos Theta for the first row of a complex term, -sin Theta for the second: the sign
            // rides on the multiply-add)
#pragma unroll
            for (int i = 0; i < J; ++i) chp[i] = 0.0;
            auto feed = [&](int row, int with, double v) __attribute__((always_inline)) {
                if (with < NR) chp[row] += v;
                else if ((with - NR) & 1) chp[row] = fma(-v, G.s[with < NR ? 0 : (with - NR) / 2], chp[row]);
                else chp[row] = fma(v, G.c[with < NR ? 0 : (with - NR) / 2], chp[row]);
            };
#pragma unroll
            for (int i = 0; i < J; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    if (!tpb_owns<NR>(FW, i, j)) continue;
                    feed(i, j, Dv(i, j));
                    if (i != j) feed(j, i, Dv(i, j));
                }
            double(*P)[64] = ring.part[FW][s & 1u];
#pragma unroll
            for (int i = 0; i < J; ++i) P[i][lane] = chp[i];
        }
        TPB4_SYNC();
    }
    // back to the fixed frame, Dv <- R(Theta_e) Dv R(Theta_e)^T block by block, then the own entries of C = Dv + P_inf
    // and of Jm, both triangles of the full matrices
    double ce[NC > 0 ? NC : 1], se[NC > 0 ? NC : 1];
    tpb4_read_cis<NR, NC>(ring.F[(per - 1u) & 3u], lane, ce, se);
#pragma unroll
    for (int ta = NR; ta < NT; ++ta) {
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
            if (tpb_owner(ta, tb) != FW) continue;
            const int oa = tpb_row0<NR>(ta), ob = tpb_row0<NR>(tb);
            const int ka = ta - NR, kb = tb >= NR ? tb - NR : 0;
            if (tb < NR) {          // complex x real
                const double x0 = Dv(oa, ob), x1 = Dv(oa + 1, ob);
                Dv(oa, ob) = ce[ka] * x0 - se[ka] * x1;
                Dv(oa + 1, ob) = se[ka] * x0 + ce[ka] * x1;
            } else if (ta != tb) {
                const double b00 = Dv(oa, ob), b01 = Dv(oa, ob + 1), b10 = Dv(oa + 1, ob), b11 = Dv(oa + 1, ob + 1);
                const double y00 = ce[ka] * b00 - se[ka] * b10, y01 = ce[ka] * b01 - se[ka] * b11;
                const double y10 = se[ka] * b00 + ce[ka] * b10, y11 = se[ka] * b01 + ce[ka] * b11;
                Dv(oa, ob) = y00 * ce[kb] - y01 * se[kb];
                Dv(oa, ob + 1) = y00 * se[kb] + y01 * ce[kb];
                Dv(oa + 1, ob) = y10 * ce[kb] - y11 * se[kb];
                Dv(oa + 1, ob + 1) = y10 * se[kb] + y11 * ce[kb];
            } else {
                const double d00 = Dv(oa, oa), d10 = Dv(oa + 1, oa), d11 = Dv(oa + 1, oa + 1);
                const double y00 = ce[ka] * d00 - se[ka] * d10, y01 = ce[ka] * d10 - se[ka] * d11;
                const double y10 = se[ka] * d00 + ce[ka] * d10, y11 = se[ka] * d10 + ce[ka] * d11;
                Dv(oa, oa) = y00 * ce[ka] - y01 * se[ka];
                Dv(oa + 1, oa) = y10 * ce[ka] - y11 * se[ka];
                Dv(oa + 1, oa + 1) = y10 * se[ka] + y11 * ce[ka];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            if (!tpb_owns<NR>(FW, i, j)) continue;
            double c = Dv(i, j);
            if (i == j) c += i < NR ? M.ar[i < NR ? i : 0] : ((i - NR) & 1 ? M.pc[i < NR ? 0 : (i - NR) / 2] : M.ac[i < NR ? 0 : (i - NR) / 2]);
            if (i >= NR && ((i - NR) & 1) && j == i - 1) c -= M.bc[i < NR ? 0 : (i - NR) / 2];
            slot[MM + 2 * J + i * J + j] = c;
            slot[MM + 2 * J + j * J + i] = c;
            slot[2 * MM + 2 * J + i * J + j] = Jm(i, j);
            slot[2 * MM + 2 * J + j * J + i] = Jm(i, j);
        }
    if (FW == 0) {
        part[1] = log(kap1) + (double)kexp * 0.69314718055994530942;
        part[2] = kap2;
    }
}

// 64 chunks (block cb) of one evaluation of structure <NR, NC> by four waves: lane & 63 = chunk, `role` uniform per wave
template <int NR, int NC>
__device__ __forceinline__ void tpb4_compose_eval(const MtgSolveArgs &a, int64_t ev, double *elems, double *parts, int C,
                                                  TpbRing4<NR + 2 * NC> &ring, const MtgMathTables *tab, uint32_t cb, int role)
{
    constexpr int J = NR + 2 * NC;
    const uint32_t c = cb * 64u + (threadIdx.x & 63u);
    uint32_t lo, hi;
    tpb_chunk_range(a.N, C, c, lo, hi);
    const uint32_t per = ((uint32_t)a.N - 1u + (uint32_t)C - 1u) / (uint32_t)C;
    double *slot = elems + (ev * C + c) * MTG_TPB_ELEM(J), *part = parts + (ev * C + c) * 4;
#ifdef MTG_TPB_ONLY_ROLE   // (resource accounting of one role inside the whole kernel)
    if (role != MTG_TPB_ONLY_ROLE) return;
#endif
    // The model is loaded INSIDE every role's branch: a role uses half of it (the column waves the decay rates and
    // frequencies, the filter waves the amplitudes), and loaded before the switch all of it -- 56 scalar registers at five
    // complex terms -- stays live into four bodies that are each within a few registers of the limit.
    TpModel<NR, NC> M;
    double jitter, slope, icpt;
    int64_t lc;
    bool fast;
    switch (role) {
    case 0:
        if (!tpb_load_model<NR, NC>(a, ev, M, jitter, slope, icpt, lc, fast)) return;  // the finish kernel reports it
        tpb4_columns<NR, NC, 0>(a, M, slope, icpt, lc, tab, fast, ring, slot, part, lo, hi, per);
        break;
    case 1:
        if (!tpb_load_model<NR, NC>(a, ev, M, jitter, slope, icpt, lc, fast)) return;
        tpb4_columns<NR, NC, 1>(a, M, slope, icpt, lc, tab, fast, ring, slot, part, lo, hi, per);
        break;
    case 2:
        if (!tpb_load_model<NR, NC>(a, ev, M, jitter, slope, icpt, lc, fast)) return;
        tpb4_filter<NR, NC, 0>(a, M, jitter, lc, ring, slot, part, lo, hi, per);
        break;
    default:
        if (!tpb_load_model<NR, NC>(a, ev, M, jitter, slope, icpt, lc, fast)) return;
        tpb4_filter<NR, NC, 1>(a, M, jitter, lc, ring, slot, part, lo, hi, per);
        break;
    }
}

template <int NR, int NC>
__device__ __forceinline__ void tpb_filter_body(const MtgSolveArgs &a, const TpModel<NR, NC> &M, double jitter, double slope,
                                                double icpt, int64_t lc, const MtgMathTables *tab, bool fast, const double *st,
                                                double *part, uint32_t lo, uint32_t hi)
{
    constexpr int J = NR + 2 * NC;
    const double2 *yv = a.yv + lc * a.N, *dxt = a.dxt + lc * a.t_stride;
    double m[J];
    Sym<J> C;
#pragma unroll
    for (int i = 0; i < J; ++i) {
        m[i] = st[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) C(i, j) = st[J + i * J + j];
    }
    tp_sub_pinf<NR, NC, J>(M, C);  // deviation form
    double dot = 0.0, dprod = 1.0, dmin = INFINITY;
    int dexp = 0;
    const uint32_t last = ((uint32_t)a.N - 1u) * 16u, end = hi * 16u;
    uint32_t off = lo * 16u;
    double2 dn = tpb_sample(dxt, off < last ? off : last), yn = tpb_sample(yv, off < last ? off : last);
    for (; off < end; off += 16u) {
        const double2 dc = dn, yc = yn;
        const uint32_t nn = off + 16u < last ? off + 16u : last;  // next sample, loaded under this one's arithmetic
        dn = tpb_sample(dxt, nn); yn = tpb_sample(yv, nn);
        TpTrans<NR, NC> T;
        tpb_transition<NR, NC>(M, dc.x, T, tab, fast);
        const double r = fma(-slope, dc.y, yc.x - icpt);
        double D, inv, z, kd[J];
        tp_filter_step<NR, NC, J>(M, T, r, yc.y + jitter, m, C, D, inv, z, kd);
        dot = fma(z * z, inv, dot);
        dmin = fmin(dmin, D);
        const double pr = dprod * D;
        dprod = __builtin_amdgcn_frexp_mant(pr);
        dexp += __builtin_amdgcn_frexp_exp(pr);
    }
    part[0] = dot;
    part[1] = log(dprod) + (double)dexp * 0.69314718055994530942;
    part[2] = dmin;
}

// One evaluation of structure <NR, NC> by a workgroup of 256 lanes (lane = chunk)
template <int NR, int NC>
__device__ __forceinline__ void tpb_filter_eval(const MtgSolveArgs &a, int64_t ev, const double *states, double *parts, int C,
                                                const MtgMathTables *tab)
{
    constexpr int J = NR + 2 * NC;
    TpModel<NR, NC> M;
    double jitter, slope, icpt;
    int64_t lc;
    bool fast;
    if (!tpb_load_model<NR, NC>(a, ev, M, jitter, slope, icpt, lc, fast)) return;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= (uint32_t)C) return;
    uint32_t lo, hi;
    tpb_chunk_range(a.N, C, c, lo, hi);
    const double *st = states + (ev * C + c) * MTG_TPB_STATE(J);
    double *part = parts + (ev * C + c) * 4;
    tpb_filter_body<NR, NC>(a, M, jitter, slope, icpt, lc, tab, fast, st, part, lo, hi);
}

// Every rank-10 structure of a model -- (nr0 + 2 k, nc0 - k), k = number of over-damped SHO terms of the
// evaluation (a.sig) -- is one of (0,5) (2,4) (4,3) (6,2) (8,1) (10,0); one kernel holds all six and each
// workgroup branches, uniformly, into the structure of its evaluation.  (Round 1 and the first version
// of this path launched every kernel once per structure on its own stream: 6 x 10 launches per
// half-step, most of them for empty lists whose workgroups still queued behind the busy ones.)
template <template <int, int> class F, class... Args>
__device__ __forceinline__ void tpb_dispatch(int nr, Args &&...args)
{
#ifdef MTG_TPB_ONLY_NR   // (resource accounting of one structure inside the whole kernel)
    if (nr != MTG_TPB_ONLY_NR) return;
#endif
    switch (nr) {
    case 0: F<0, 5>::run(args...); break;
    case 2: F<2, 4>::run(args...); break;
    case 4: F<4, 3>::run(args...); break;
    case 6: F<6, 2>::run(args...); break;
    case 8: F<8, 1>::run(args...); break;
    case 10: F<10, 0>::run(args...); break;
    default: break;
    }
}
template <int NR, int NC> struct TpbCompose4F {
    static __device__ __forceinline__ void run(const MtgSolveArgs &a, int64_t ev, double *elems, double *parts, int C,
                                               TpbRing4<10> &ring, const MtgMathTables *tab, uint32_t cb, int role)
    {
        tpb4_compose_eval<NR, NC>(a, ev, elems, parts, C, ring, tab, cb, role);
    }
};
template <int NR, int NC> struct TpbFilterF {
    static __device__ __forceinline__ void run(const MtgSolveArgs &a, int64_t ev, const double *states, double *parts, int C,
                                               const MtgMathTables *tab)
    {
        tpb_filter_eval<NR, NC>(a, ev, states, parts, C, tab);
    }
};

__device__ __forceinline__ int tpb_nr(const MtgSolveArgs &a, int64_t ev) { return a.tp_nr0 + 2 * (a.sig ? a.sig[ev] : 0); }

}  // namespace

// one wave per evaluation: the chunks' partial sums in a fixed order, plus the head (sample 0)
void mtg_launch_tpb_finish(const MtgSolveArgs &a, const double *parts, const double *head, int C, int64_t nevals,
                           hipStream_t stream);
// mtg_tp_big_compose4q.hip / mtg_tp_big_filter.hip: the two kernels that hold all six structures
void mtg_launch_tpb_compose4q(const MtgSolveArgs &a, double *elems, double *parts, int C, int64_t nevals, hipStream_t stream);
void mtg_launch_tpb_filter(const MtgSolveArgs &a, const double *states, double *parts, int C, int64_t nevals, hipStream_t stream);
