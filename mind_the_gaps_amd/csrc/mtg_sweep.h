// mtg_sweep.h -- the serial sweep of one evaluation (celerite CholeskySolver.compute + log_determinant +
// dot_solve fused into one pass over the N samples), shared by the one-structure kernel of mtg_kernels.hip and
// the several-structures-in-one-launch kernel of mtg_kernels_multi.hip.
#ifndef MTG_SWEEP_H
#define MTG_SWEEP_H

#include "mtg_device.h"
#include "mtg_math.h"

#include <math.h>

#define MTG_LN_2PI 1.8378770664093454835606594728112
#define MTG_BLOCK 256

// ---------------------------------------------------------------------------
// fused factorisation + forward solve, one lane per evaluation
// ---------------------------------------------------------------------------
// Waves per SIMD the register allocator must leave room for (512 VGPRs / waves):
// the state is J(J+1)/2 + 4J + ... doubles per lane, so the target drops with J.
#ifndef MTG_WAVES_BIAS
#define MTG_WAVES_BIAS 0
#endif
__host__ __device__ constexpr int mtg_waves_for(int J)
{
    return (J <= 2 ? 5 : J <= 3 ? 3 : J <= 8 ? 2 : 1) + MTG_WAVES_BIAS;
}

// Per-lane state of one evaluation, all statically indexed -> VGPRs.
template <int NR, int NC>
struct MtgLane {
    static constexpr int J = NR + 2 * NC;
    double ar[NR > 0 ? NR : 1], cr[NR > 0 ? NR : 1];
    double ac[NC > 0 ? NC : 1], bc[NC > 0 ? NC : 1], cc[NC > 0 ? NC : 1], dc[NC > 0 ? NC : 1];
    double jit, slope, icpt;
    double S[J * (J + 1) / 2];
    double Wt[J];  // V_n - S U_n  (W_n = Wt / D_n)
    double f[J];
    double pr[NC > 0 ? NC : 1];                        // phase d_k (t_n - t_0) = pm pi/32 + pr
    int pm[NC > 0 ? NC : 1];                           //   pm = 16 * (m mod N_trig) (table path)
    double ncr[NR > 0 ? NR : 1], cr64[NR > 0 ? NR : 1];  // -c and -c 8 N_exp/ln2 of the real terms
    double ncc[NC > 0 ? NC : 1], cc64[NC > 0 ? NC : 1];  // same for the complex terms
    double invD, z, dot, dprod;
    int dmin_hi;  // smallest high dword of a pivot: <= 0 means some D_n <= 0 (K not positive definite)
    int dexp;
};

// The sweep over the N samples: ONE basic block per step (no branch besides the
// back edge), so the scheduler can hoist the five table look-ups and the next
// sample's loads above the polynomial/recurrence arithmetic.
//   FAST: every lane's d_k * max(dx) is inside the exact range of the table sincos.
//   MEAN: a mean function has to be subtracted or a jitter term added (false: the mean is
//         identically zero, the frozen per-light-curve constant having been folded into y at
//         upload, and the model has no JitterTerm).
//   NB0:  the last NB0 complex terms have b = 0 by construction (Lorentzian, three-parameter ComplexTerm,
//         Cosinus): their U is a (cos, sin) -- a multiplication instead of a multiplication and a multiply-add.
template <int NR, int NC, bool FAST, bool MEAN, int NB0 = 0, class Tab = MtgMathTablesT<(NC > 0)>>
__device__ __forceinline__ void mtg_sweep(MtgLane<NR, NC> &L, const MtgSolveArgs &a, const double2 *yv_base,
                                          uint32_t yv_records, uint32_t yoff, const double2 *dxt_base,
                                          uint32_t dxt_records, uint32_t toff, const Tab *tab)
{
    // Every fused multiply-add of the sweep is written out: the same source is instantiated in more than one kernel
    // (one structure per launch, several in one) and a row must come out the same to the last bit in all of them,
    // which the compiler's own choice of what to contract does not promise.
#pragma clang fp contract(off)
    constexpr int J = NR + 2 * NC;
    constexpr int NT = NR + NC;  // distinct exp(-c dx) factors
    // Samples come through buffer loads: resource in SGPRs, the light curve's byte
    // offset per lane (voffset) and the running sample offset on the scalar unit
    // (soffset += 16 per step) -- no VALU instruction is spent on addressing, and the
    // hardware range check makes the one-past-the-end prefetch of the last step a
    // harmless zero.  (y, sigma^2) and (dx, t) are interleaved: one 16-byte load each.
    const __amdgpu_buffer_rsrc_t ryv =
        __builtin_amdgcn_make_buffer_rsrc((void *)yv_base, 0, (int)yv_records, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdt =
        __builtin_amdgcn_make_buffer_rsrc((void *)dxt_base, 0, (int)dxt_records, 0x00020000);
    auto ld = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };

    const double t0 = ld(rdt, toff, 0).y;  // phases are measured from the first sample
    // One step of the recurrence for the sample (dxc, tc, yc, vc): a single basic block.
    auto step = [&](const double2 dtc, const double2 yvc) __attribute__((always_inline)) {
        const double dxc = dtc.x, tc = dtc.y, yc = yvc.x, vc = yvc.y;
        // -- per-term propagators and generators (celerite phi, U, V) ---------
        double ph[NT > 0 ? NT : 1];
        double U[J], V[J];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            ph[j] = mtg_exp_cdx(L.ncr[j], L.cr64[j], dxc, tab);
            U[j] = L.ar[j];
            V[j] = 1.0;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            ph[NR + k] = mtg_exp_cdx(L.ncc[k], L.cc64[k], dxc, tab);
            // (cos, sin) of d_k (t_n - t_0): the kernel depends on time differences
            // only, so the phase origin is free
            double cn, sn;
            if (FAST) {
                mtg_phase_step(L.dc[k], dxc, L.pr[k], L.pm[k], &sn, &cn, tab);
            } else {
                // huge d_k dx somewhere in this wave: evaluate at the elapsed time the way celerite
                // does at the absolute one.  (Rotating the previous pair by sincos(d dx) instead
                // lets the pair drift by ~1e-16 per step, which an ill-conditioned covariance --
                // amplitude >> noise -- amplifies far beyond celerite's own error.)
                sincos(L.dc[k] * (tc - t0), &sn, &cn);
            }
            if (k >= NC - NB0) {
                U[NR + 2 * k] = L.ac[k] * cn;
                U[NR + 2 * k + 1] = L.ac[k] * sn;
            } else {
                U[NR + 2 * k] = fma(L.ac[k], cn, L.bc[k] * sn);
                U[NR + 2 * k + 1] = fma(L.ac[k], sn, -(L.bc[k] * cn));
            }
            V[NR + 2 * k] = cn;
            V[NR + 2 * k + 1] = sn;
        }
        // -- S <- (phi phi^T) o (S + D W W^T) ;  f <- phi o (f + W z) ----------
        const double zs = L.z * L.invD;
        L.dot = fma(L.z, zs, L.dot);  // z_{n-1}^2 / D_{n-1}: the previous sample's term of r^T K^-1 r
        double wd[J];
#pragma unroll
        for (int i = 0; i < J; ++i) wd[i] = L.Wt[i] * L.invD;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            const int ti = i < NR ? i : NR + (i - NR) / 2;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const int tj = j < NR ? j : NR + (j - NR) / 2;
                const double pp = ph[ti] * ph[tj];
                L.S[i * (i + 1) / 2 + j] = pp * fma(L.Wt[i], wd[j], L.S[i * (i + 1) / 2 + j]);
            }
            L.f[i] = ph[ti] * fma(L.Wt[i], zs, L.f[i]);
        }
        // -- D_n = A_n - U^T S U ; Wt = V - S U ; z_n = r_n - U^T f -------------
        // U^T V = sum of the a_j (the kernel at lag 0), so A_n - U^T S U = sigma_n^2 + jitter + U^T Wt:
        // the subtraction V - S U rides on the multiply-add chain and D needs no second pass over q
        double D = MEAN ? vc + L.jit : vc;
        double zn = MEAN ? yc - fma(L.slope, tc, L.icpt) : yc;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double w = V[i];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                w = fma(-L.S[hi * (hi + 1) / 2 + lo], U[j], w);
            }
            L.Wt[i] = w;
            D = fma(U[i], w, D);
            zn = fma(-U[i], L.f[i], zn);
        }
        L.dmin_hi = min(L.dmin_hi, __double2hiint(D));  // sign / zero test on the high dword
        L.invD = mtg_rcp(D);
        L.z = zn;
        L.dprod *= D;  // ln det K = ln prod D_n, exponent peeled off by the caller
    };
    // ln det: the pivot product is renormalised every two steps (D in (1e-70, 1e70))
    auto renorm = [&]() __attribute__((always_inline)) {
        const double pr = L.dprod;
        L.dprod = __builtin_amdgcn_frexp_mant(pr);
        L.dexp += __builtin_amdgcn_frexp_exp(pr);
    };

    // Two steps per trip with ping-pong sample registers: the sample of step n + 1 is
    // loaded under the arithmetic of step n and nothing is copied between registers.
    const uint32_t N = (uint32_t)a.N;
    double2 dtA = ld(rdt, toff, 0), yvA = ld(ryv, yoff, 0);
    uint32_t soff = 0;
    for (uint32_t n = 0; n + 1 < N; n += 2) {
        const double2 dtB = ld(rdt, toff, soff + 16), yvB = ld(ryv, yoff, soff + 16);
        step(dtA, yvA);
        soff += 32;
        dtA = ld(rdt, toff, soff); yvA = ld(ryv, yoff, soff);
        step(dtB, yvB);
        renorm();
    }
    if (N & 1u) {
        step(dtA, yvA);
        renorm();
    }
    L.dot = fma(L.z * L.z, L.invD, L.dot);  // the last sample's term
}


// One evaluation, row `e` of the batch, on this lane: coefficients -> sweep -> lnL and status.  `tab` is the
// workgroup's table set (any MtgMathTablesT that has what the structure needs).
template <int NR, int NC, int NB0, class Tab>
__device__ __forceinline__ void mtg_solve_row(const MtgSolveArgs &a, int64_t e, const Tab *tabp)
{
#pragma clang fp contract(off)
    constexpr int J = NR + 2 * NC;  // celerite rank
    const Tab &tab = *tabp;
    // ---- coefficients of this evaluation -----------------------------------
    MtgLane<NR, NC> L;
    const double *cf = a.coef + e;
    const int64_t cs = a.cstride;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        L.ar[j] = cf[a.lay.ar(j) * cs];
        L.cr[j] = cf[a.lay.cr(j) * cs];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        L.ac[k] = cf[a.lay.ac(k) * cs];
        L.bc[k] = cf[a.lay.bc(k) * cs];
        L.cc[k] = cf[a.lay.cc(k) * cs];
        L.dc[k] = cf[a.lay.dc(k) * cs];
        dmax = fmax(dmax, fabs(L.dc[k]));
    }
    L.jit = cf[a.lay.jit() * cs];
    L.slope = cf[a.lay.mean(0) * cs];
    L.icpt = cf[a.lay.mean(1) * cs];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) L.S[i] = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) { L.Wt[i] = 0.0; L.f[i] = 0.0; }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        L.pr[k] = 0.0; L.pm[k] = 0;
        L.ncc[k] = -L.cc[k]; L.cc64[k] = L.cc[k] * -MTG_EXP_CSCALE;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) { L.ncr[j] = -L.cr[j]; L.cr64[j] = L.cr[j] * -MTG_EXP_CSCALE; }
    L.invD = 0.0; L.z = 0.0; L.dot = 0.0; L.dprod = 1.0; L.dexp = 0; L.dmin_hi = 0x7fffffff;

    const uint32_t lc = a.lc_index ? (uint32_t)a.lc_index[e] : 0u;
    const uint64_t lc_bytes = (uint64_t)a.N * 16u;
    // The sweep reads samples through buffer descriptors with 32-bit byte offsets while the resident
    // set may be far larger than 4 GiB (288 GB of HBM): every wave places its descriptors at the
    // first light curve its evaluations need and reaches the others by their distance from it.
    // Batches are grouped by light curve as a rule (a wave of a (walker x light curve) sweep touches
    // one or two); an evaluation further than a.window_bytes from its wave's first light curve goes
    // to the left-over list, which a second launch sweeps one evaluation per wave.
    uint32_t lo = 0xffffffffu;
    for (unsigned long long m = __ballot(1); m; m &= m - 1ull)  // scalar loop over the active lanes
        lo = min(lo, (uint32_t)__builtin_amdgcn_readlane((int)lc, __ffsll((long long)m) - 1));
    const uint64_t base_bytes = (uint64_t)lo * lc_bytes, rel = (uint64_t)(lc - lo) * lc_bytes;
    if (((uint64_t)lc + 1u) * lc_bytes > a.yv_bytes) {
        // a device-side lc_index outside the resident set (the host cannot see it): no likelihood
        a.out[e] = -INFINITY;
        a.status[e] = MTG_ST_NONFINITE;
        return;
    }
    if (rel + lc_bytes > a.window_bytes) {
        if (a.left_list) a.left_list[atomicAdd(a.left_count, 1)] = (int)e;
        return;
    }
    const uint64_t yv_left = a.yv_bytes > base_bytes ? a.yv_bytes - base_bytes : 0;
    const double2 *yv_base = (const double2 *)((const char *)a.yv + base_bytes);
    const uint32_t yv_rec = yv_left > 0xffffffffull ? 0xffffffffu : (uint32_t)yv_left;
    const uint32_t yoff = (uint32_t)rel;
    // shared sampling: one (dx, t) row for everybody; per-light-curve sampling: the same window
    const double2 *dxt_base = a.t_stride ? (const double2 *)((const char *)a.dxt + base_bytes) : a.dxt;
    const uint32_t dxt_rec = a.t_stride ? yv_rec : (uint32_t)lc_bytes;
    const uint32_t toff = a.t_stride ? yoff : 0u;

    // the table sincos of mtg_phase_step serves every lane of the wave while d_k * dx <= MTG_TRIG_FAST_MAX
    const bool fast = !__any(!(dmax * *a.dxmax <= MTG_TRIG_FAST_MAX));
    if (fast) {
        if (a.has_mean) mtg_sweep<NR, NC, true, true, NB0>(L, a, yv_base, yv_rec, yoff, dxt_base, dxt_rec, toff, &tab);
        else mtg_sweep<NR, NC, true, false, NB0>(L, a, yv_base, yv_rec, yoff, dxt_base, dxt_rec, toff, &tab);
    } else {
        mtg_sweep<NR, NC, false, true, NB0>(L, a, yv_base, yv_rec, yoff, dxt_base, dxt_rec, toff, &tab);
    }

    const double logdet = fma((double)L.dexp, 0.69314718055994530942, log(L.dprod));
    double ll = -0.5 * fma((double)a.N, MTG_LN_2PI, L.dot + logdet);
    int st = MTG_ST_OK;
    if (L.dmin_hi <= 0) { st = MTG_ST_NOTPD; ll = -INFINITY; }
    else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
    a.out[e] = ll;
    a.status[e] = st;
}

#endif  // MTG_SWEEP_H
