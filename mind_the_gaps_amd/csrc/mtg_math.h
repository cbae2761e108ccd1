// mtg_math.h -- FP64 elementary functions shaped for the celerite recurrence on
// gfx950.
//
// The solve kernel is FP64-VALU-issue bound (4 cycles per wave64 FP64
// instruction, one SGPR constant per instruction), and every sample of every
// evaluation needs one exp per term and one sincos per complex term.  OCML's
// exp/sincos cost ~30/~65 instructions and two dozen 64-bit constants each,
// which overflows the 102-SGPR budget and makes the compiler shuffle constants
// through VGPRs.  These versions trade polynomial degree for two small LDS
// tables (filled once per workgroup):
//
//   exp(y)    = 2^k * T[j] * (1 + p(r)),  y = (N k + j) ln2/N + r, |r| <= ln2/2N,
//               N = 2048: p of degree 3                 -> 12 VALU + 1 ds_read_b64
//   sincos(x) : x = m 2pi/N + r, |r| <= pi/N, (cos, sin)(2 pi m/N) from the table,
//               N = 2048: sin r / cos r of degree 3 / 4, angle addition
//                                                      -> 16 VALU + 1 ds_read_b128
//
// Both are accurate to ~1 ulp (tests/test_device_math_gpu.py); arguments beyond
// the exactness range of the Cody-Waite reductions fall back to OCML through a
// wave-uniform branch.
#pragma once
#include <hip/hip_runtime.h>

// Table sizes (log2 entries).  Bigger tables shorten the polynomials: every VALU
// instruction, FP64 or not, costs the wave one 4-cycle issue slot, and the sweep is
// issue bound.  2^11 + 2^11 entries = 16 KiB + 32 KiB of LDS per workgroup.
#ifndef MTG_EXP_BITS
#define MTG_EXP_BITS 11
#endif
#ifndef MTG_TRIG_BITS
#define MTG_TRIG_BITS 11
#endif
#define MTG_EXP_N (1 << MTG_EXP_BITS)
#define MTG_TRIG_N (1 << MTG_TRIG_BITS)

// TRIG = false (kernels without complex terms) leaves the trig table out of LDS, so
// that the occupancy of those light kernels is not capped by it.
template <bool TRIG>
struct MtgMathTablesT {
    double exp2_frac[MTG_EXP_N];          // 2^(j / N)
    double2 cis[TRIG ? MTG_TRIG_N : 1];   // (cos, sin)(2 pi j / N)
};
typedef MtgMathTablesT<true> MtgMathTables;

// Cooperative fill by the whole workgroup; follow with __syncthreads().
template <bool TRIG>
__device__ __forceinline__ void mtg_fill_tables(MtgMathTablesT<TRIG> *tab, int tid, int nthreads)
{
    for (int j = tid; j < MTG_EXP_N; j += nthreads)
        tab->exp2_frac[j] = exp2((double)j * (1.0 / MTG_EXP_N));
    if (TRIG)
        for (int j = tid; j < MTG_TRIG_N; j += nthreads) {
            double s, c;
            sincospi((double)j * (2.0 / MTG_TRIG_N), &s, &c);
            tab->cis[j] = make_double2(c, s);
        }
}

// The same tables copied from a resident global copy (made once per context by mtg_tables_kernel with mtg_fill_tables
// itself, so the entries are the same bits): a workgroup of ONE wave spends ~5 us computing 2048 exp2 and 2048 sincospi
// entries -- as long as the rest of a short light curve's time-parallel solve -- and ~1 us copying them.  `src` NULL:
// computed as before.  TRIG_USED = false: the (cos, sin) table is left alone (a model without complex terms never reads it).
template <bool TRIG_USED>
__device__ __forceinline__ void mtg_load_tables(MtgMathTablesT<true> *tab, const MtgMathTablesT<true> *src, int tid, int nthreads)
{
    if (!src) {
        for (int j = tid; j < MTG_EXP_N; j += nthreads) tab->exp2_frac[j] = exp2((double)j * (1.0 / MTG_EXP_N));
        if (TRIG_USED)
            for (int j = tid; j < MTG_TRIG_N; j += nthreads) {
                double s, c;
                sincospi((double)j * (2.0 / MTG_TRIG_N), &s, &c);
                tab->cis[j] = make_double2(c, s);
            }
        return;
    }
    // batches of loads issued together (a loop of load -> LDS store waits for memory once per trip)
    constexpr int BATCH = 16;
    const double2 *s = reinterpret_cast<const double2 *>(src);
    double2 *d = reinterpret_cast<double2 *>(tab);
    const int n = TRIG_USED ? (int)(sizeof(MtgMathTablesT<true>) / sizeof(double2)) : MTG_EXP_N / 2;
    for (int base = tid; base < n; base += BATCH * nthreads) {
        double2 r[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int j = base + u * nthreads;
            r[u] = j < n ? s[j] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int j = base + u * nthreads;
            if (j < n) d[j] = r[u];
        }
    }
}

// exp(r) - 1 on |r| <= ln2 / 2^(BITS+1), truncation error below 4e-17 absolute.
__device__ __forceinline__ double mtg_expm1_small(double r)
{
#if MTG_EXP_BITS >= 11
    const double p = __builtin_fma(r, 0x1.5555555555555p-3, 0.5);                  // 1/2 + r/6
#elif MTG_EXP_BITS >= 8
    double p = 0x1.5555555555555p-5;                                                // 1/24
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
#else
    double p = 0x1.6c16c16c16c17p-10;                                               // 1/720
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
#endif
    return __builtin_fma(p * r, r, r);
}

// (sin r, cos r) on |r| <= pi / 2^BITS, truncation error below 1e-16 absolute.
__device__ __forceinline__ void mtg_sincos_small(double r, double *s, double *c)
{
    const double z = r * r;
#if MTG_TRIG_BITS >= 11
    *s = __builtin_fma(r * z, -0x1.5555555555555p-3, r);                            // r - r^3/6
    *c = __builtin_fma(__builtin_fma(z, 0x1.5555555555555p-5, -0.5), z, 1.0);       // 1 - z/2 + z^2/24
#elif MTG_TRIG_BITS >= 10
    // |r| <= pi / 1024: the sine needs its r^5 term (r^5/120 = 2e-15 without), the cosine is as above (r^6/720 = 1e-18)
    *s = __builtin_fma(r * z, __builtin_fma(z, 0x1.1111111111111p-7, -0x1.5555555555555p-3), r);
    *c = __builtin_fma(__builtin_fma(z, 0x1.5555555555555p-5, -0.5), z, 1.0);
#elif MTG_TRIG_BITS >= 8
    const double ps = __builtin_fma(z, 0x1.1111111111111p-7, -0x1.5555555555555p-3);
    *s = __builtin_fma(r * z, ps, r);
    double pc = -0x1.6c16c16c16c17p-10;
    pc = __builtin_fma(pc, z, 0x1.5555555555555p-5);
    pc = __builtin_fma(pc, z, -0.5);
    *c = __builtin_fma(pc, z, 1.0);
#else
    double ps = -0x1.a01a01a01a01ap-13;
    ps = __builtin_fma(ps, z, 0x1.1111111111111p-7);
    ps = __builtin_fma(ps, z, -0x1.5555555555555p-3);
    *s = __builtin_fma(r * z, ps, r);
    double pc = 0x1.a01a01a01a01ap-16;
    pc = __builtin_fma(pc, z, -0x1.6c16c16c16c17p-10);
    pc = __builtin_fma(pc, z, 0x1.5555555555555p-5);
    pc = __builtin_fma(pc, z, -0.5);
    *c = __builtin_fma(pc, z, 1.0);
#endif
}

// Largest phase increment x = d * dx handed to mtg_phase_step.  Its reduction is exact for any x whose multiple
// count k = rint(x N / 2 pi) fits the mantissa trick below (k < 2^51: x < 6.9e12 for N = 2048): the product k C is
// formed inside an fma, the remainder is rounded once, and the only error is that of the constant C -- the frequency
// d moved by less than its own rounding, the same at every sample.  What is left is the rounding of x itself,
// ulp(x) / 2 per step (7e-12 rad at x = 1e5, 6e-5 at 1e12; a random walk over the steps of a sweep) -- against
// ulp(d t_n) / 2 at EVERY sample for a phase evaluated at the elapsed time, as the libm variant of the sweep does
// (and celerite, at the absolute time): n times larger at sample n.  Accuracy therefore never argues for the libm
// variants; they are kept for what the mantissa trick cannot hold.  (Until round 3 the limit was 1e5, a left-over of
// a two-constant reduction.  A sampler's walkers at the top of the prior box -- omega_0 ~ e^10 per day, gaps of
// days -- crossed it, and ONE such lane sends its whole wave through libm: 25 % of the configs[3] refits' time.)
#ifndef MTG_TRIG_FAST_MAX
#define MTG_TRIG_FAST_MAX 1.0e12
#endif

// ---------------------------------------------------------------------------
// Every VALU instruction costs the wave the same 4-cycle issue slot, FP64 or not,
// so the reductions also strip integer glue: rounding is done by adding
// 1.5 * 2^(52+s), which rounds to a multiple of 2^s, so the integer comes out
// pre-multiplied by the table's entry size (s = 3 for 8-byte entries, 4 for
// 16-byte ones): one cvt + one and give the LDS byte offset, no shift.
// ---------------------------------------------------------------------------

// Phase accumulation for a complex term: the running phase d (t_n - t_0) is kept as
// (m16, r): phase = (m16 / 16) 2 pi / N + r, |r| <= pi / N, m16 = 16 * (m mod N) being the
// byte offset of the table entry.  One step adds d * dx, re-reduces and returns
// (cos, sin) of the NEW phase directly -- no separate rotation of the previous pair.
// Requires d * dx <= MTG_TRIG_FAST_MAX.
template <class Tab>
__device__ __forceinline__ void mtg_phase_step(double d, double dx, double &r, int &m16, double *sn,
                                               double *cs, const Tab *tab)
{
    const double magic = 0x1.8p+56;                                                 // 1.5 * 2^(52+4)
    const double x = __builtin_fma(d, dx, r);
    const double w = __builtin_fma(x, 0x1.45f306dc9c883p+1 * MTG_TRIG_N, magic);    // x 16 N / 2 pi
    const double md16 = w - magic;                                                  // 16 rint(x N / 2 pi)
    // One-constant reduction: the fma forms md16 * C exactly, so the only error is that of C itself
    // (<= 2^-53 relative), i.e. the accumulated phase is d (1 + eps) (t_n - t_0) with ONE eps for the
    // whole sweep -- a perturbation of the frequency below its own rounding, not a drift.
    const double rr = __builtin_fma(md16, -(0x1.921fb54442d18p-2 / MTG_TRIG_N), x);  // 2 pi / 16 N
    // the low mantissa dword of w is rint(x N / 2 pi) (two's complement): shift-add, no cvt; m16 wraps
    // freely and is masked where it is used
    // (v_lshl_add_u32 spelled out: left to itself the compiler narrows the masked sum to packed
    // 16-bit arithmetic, which takes more instructions)
    asm("v_lshl_add_u32 %0, %1, 4, %0" : "+v"(m16) : "v"(__double2loint(w)));
    r = rr;
    const double2 cj = *(const double2 *)((const char *)tab->cis + (m16 & ((MTG_TRIG_N - 1) * 16)));
    double s, c;
    mtg_sincos_small(rr, &s, &c);
    *sn = __builtin_fma(cj.x, s, cj.y * c);
    *cs = __builtin_fma(-cj.y, s, cj.x * c);
}

// exp(-c dx): negc = -c and cs8 = -c * 8 N / ln2 are per-lane constants hoisted out of
// the sweep.  One-constant reduction r = y - q ln2 / N: its error is |y| 2^-53 relative,
// i.e. at most 4e-17 ABSOLUTE in the result (x e^-x <= 0.37), which is what matters for
// a propagator that multiplies bounded state.  Far below the underflow point q
// saturates (cvt) and ldexp returns 0.
#define MTG_EXP_CSCALE (0x1.71547652b82fep+3 * MTG_EXP_N)                           /* 8 N / ln2 */
#define MTG_EXP_C1 (0x1.62e42fefa39efp-4 / MTG_EXP_N)                               /* ln2 / 8 N */
template <class Tab>
__device__ __forceinline__ double mtg_exp_cdx(double negc, double cs8, double dx, const Tab *tab)
{
    const double magic = 0x1.8p+55;                                                 // 1.5 * 2^(52+3)
    const double w = __builtin_fma(dx, cs8, magic);
    const double q8 = w - magic;                                                    // 8 rint(y N / ln2)
    const int i8 = (int)q8;                                                         // saturates: huge c dx -> 0
    const double t = *(const double *)((const char *)tab->exp2_frac + (i8 & ((MTG_EXP_N - 1) * 8)));
#if MTG_EXP_BITS >= 11
    // remainder in table units, exact in the fma: y 8N/ln2 - q8 = f, r = f ln2/8N with |r| <= ln2/2N;
    // exp(r) - 1 = f (C + f (C^2/2 + f C^3/6)) with the unit folded into the coefficients
    (void)negc;
    const double f = __builtin_fma(dx, cs8, -q8);
    double p = __builtin_fma(f, MTG_EXP_C1 * MTG_EXP_C1 * MTG_EXP_C1 / 6.0, MTG_EXP_C1 * MTG_EXP_C1 / 2.0);
    p = __builtin_fma(p, f, MTG_EXP_C1) * f;
#else
    const double r = __builtin_fma(q8, -MTG_EXP_C1, negc * dx);
    const double p = mtg_expm1_small(r);
#endif
    return __builtin_ldexp(__builtin_fma(t, p, t), i8 >> (3 + MTG_EXP_BITS));
}

// 1 / d for a normal positive d: hardware seed (v_rcp_f64) plus one Newton step, good to
// 2e-15 relative -- far inside what the recurrence needs; a second step
// changes the worst lnL error against the golden vectors from 2.4e-13 to 1.9e-13 and
// costs 1.5 % of the sweep.
#ifndef MTG_RCP_NEWTON
#define MTG_RCP_NEWTON 1
#endif
__device__ __forceinline__ double mtg_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < MTG_RCP_NEWTON; ++i) {
        const double e = __builtin_fma(-d, x, 1.0);
        x = __builtin_fma(x, e, x);
    }
    return x;
}
