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
//   exp(y)    = 2^k * T[j] * (1 + p(r)),  y = (64 k + j) ln2/64 + r, |r| <= ln2/128,
//               p of degree 5                         -> 13 FP64 ops + 1 ds_read_b64
//   sincos(x) : x = m pi/32 + r, |r| <= pi/64, (cos, sin)(m pi/32) from a 64-entry
//               table, sin r / cos r of degree 7 / 8, angle addition
//                                                      -> 20 FP64 ops + 1 ds_read_b128
//
// Both are accurate to ~1 ulp (tests/test_device_math_gpu.py); arguments beyond
// the exactness range of the Cody-Waite reductions fall back to OCML through a
// wave-uniform branch.
#pragma once
#include <hip/hip_runtime.h>

struct MtgMathTables {
    double exp2_frac[64];  // 2^(j/64)
    double2 cis[64];       // (cos, sin)(2 pi j / 64)
};

// One entry per lane; call with all 64 lanes of the first wave active, then barrier.
__device__ __forceinline__ void mtg_fill_tables(MtgMathTables *tab, int lane)
{
    if (lane < 64) {
        tab->exp2_frac[lane] = exp2((double)lane * (1.0 / 64.0));
        double s, c;
        sincospi((double)lane * (1.0 / 32.0), &s, &c);
        tab->cis[lane] = make_double2(c, s);
    }
}

// exp(y) for y <= 0.  Any y <= 0 is safe: y is clamped at -1e4, where the result
// has long underflowed to 0 through ldexp.
__device__ __forceinline__ double mtg_exp(double y, const MtgMathTables *tab)
{
    y = __builtin_fmax(y, -1.0e4);
    const double kd = __builtin_rint(y * 0x1.71547652b82fep+6);           // y * 64 / ln2
    double r = __builtin_fma(kd, -0x1.62e42fee00000p-7, y);                // ln2/64, 32 high bits
    r = __builtin_fma(kd, -0x1.a39ef35793c76p-39, r);                      // ln2/64, low part
    const int ki = (int)kd;
    const double t = tab->exp2_frac[ki & 63];
    // exp(r) - 1 = r + r^2 (1/2 + r/6 + r^2/24 + r^3/120 + r^4/720), |r| <= 0.0055
    double p = 0x1.6c16c16c16c17p-10;
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p * r, r, r);
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 6);
}

// Largest argument for which m = rint(x 32/pi) < 2^22 keeps m * P1, m * P2 exact.
#define MTG_TRIG_FAST_MAX 4.0e5

__device__ __forceinline__ void mtg_sincos_fast(double x, double *sn, double *cs,
                                                const MtgMathTables *tab)
{
    const double md = __builtin_rint(x * 0x1.45f306dc9c883p+3);            // x * 32 / pi
    double r = __builtin_fma(md, -0x1.921fb54000000p-4, x);                // pi/32 in 30 + 30 + 53 bits
    r = __builtin_fma(md, -0x1.10b4611800000p-34, r);
    r = __builtin_fma(md, -0x1.313198a2e0370p-65, r);
    const double2 cj = tab->cis[(int)md & 63];
    const double z = r * r;
    // sin r = r + r^3 (-1/6 + z/120 - z^2/5040), cos r = 1 + z (-1/2 + z/24 - z^2/720 + z^3/40320)
    double ps = -0x1.a01a01a01a01ap-13;
    ps = __builtin_fma(ps, z, 0x1.1111111111111p-7);
    ps = __builtin_fma(ps, z, -0x1.5555555555555p-3);
    double pc = 0x1.a01a01a01a01ap-16;
    pc = __builtin_fma(pc, z, -0x1.6c16c16c16c17p-10);
    pc = __builtin_fma(pc, z, 0x1.5555555555555p-5);
    pc = __builtin_fma(pc, z, -0.5);
    const double s = __builtin_fma(r * z, ps, r);
    const double c = __builtin_fma(pc, z, 1.0);
    // angle addition with the table entry (cj.x, cj.y) = (cos, sin)(m pi/32)
    *sn = __builtin_fma(cj.x, s, cj.y * c);
    *cs = __builtin_fma(-cj.y, s, cj.x * c);
}

// sin/cos(x) for x >= 0 of any size: table path when EVERY lane of the wave is in
// range (the branch is wave-uniform), OCML otherwise.
__device__ __forceinline__ void mtg_sincos(double x, double *sn, double *cs, const MtgMathTables *tab)
{
    if (__builtin_expect(__any(!(x <= MTG_TRIG_FAST_MAX)), 0)) {
        sincos(x, sn, cs);
    } else {
        mtg_sincos_fast(x, sn, cs, tab);
    }
}

// 1 / d for a normal positive d (pivots live in (1e-24, 1e22)): hardware seed
// plus Newton steps.
#ifndef MTG_RCP_NEWTON
#define MTG_RCP_NEWTON 2
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
