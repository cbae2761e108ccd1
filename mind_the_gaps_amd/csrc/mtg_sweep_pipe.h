// mtg_sweep_pipe.h -- the serial sweep of mtg_sweep.h cut in two along the only line that has no recurrence across
// it, each half on a wave of its own: a PRODUCER wave evaluates what depends on the sample alone -- the propagators
// exp(-c dx) and the phases (cos, sin) of the complex terms -- for its 64 evaluations, a CONSUMER wave runs the
// recurrences (U, S, f, W~, D, z, ln det) on what the producer left in LDS a few samples earlier.
//
// Why: between ~10^4 and ~3 10^4 rows a launch of the one-lane-per-evaluation sweep puts ONE wave on (at most) half
// of the SIMDs and every wave walks its ~165 instructions per sample alone -- 3.5 ms for 10^4 samples whatever the
// number of rows (profiles/r03_order_sweep.txt).  That is one GPU's share of the Protassov refits at 8 GPUs
// (250 light curves x 128 walkers per half-step).  The exp / sincos are ~70 of those instructions and depend on the
// sample only, not on the state: a second wave on an idle SIMD takes them over.  (Splitting one evaluation over a
// lane PAIR instead -- rows of S on two lanes, DPP exchange -- does not pay: the symmetric S update is 15 entries x 2
// instructions on one lane and still 15 x 2 on the lane that owns three full rows, a 64-bit DPP exchange costs two
// moves per double against the one multiply-add it saves, and only the generators split evenly; DESIGN.md.)
//
// What the measurements shaped (scripts/pipe_ab.py, scripts/pipe_pmc.sh):
//  * the hand-over is the expensive part, and its price is the STORES: one wave's ds_write_b128 holds its issue port
//    for ~21 cycles (MI355X_MICROARCH.md, LDS: 13 cycles per store with both halves of the store path busy, twice that
//    from one half), five multiply-adds' worth -- so only what cannot be recomputed cheaply crosses: NT propagators
//    and NC (cos, sin) pairs; U = a (cos, sin) + b (sin, -cos) is four instructions on the consumer against two
//    stores on the producer;
//  * four samples per hand-over, three chunks in the ring: a barrier drains both pipelines (~150 cycles);
//  * the barrier of a chunk comes after the arithmetic of the NEXT chunk's first sample and before that sample's
//    stores, so the chunk's own stores land under arithmetic, not under an s_waitcnt;
//  * the consumer runs two chunks behind and fetches the generators of sample n + 1 while it works on sample n.
//
// The arithmetic is that of mtg_sweep.h, expression by expression, under `fp contract(off)`: a row comes out the same
// to the last bit as from mtg_solve_kernel / mtg_solve_kernel_multi (tests/test_pipe_gpu.py).
#ifndef MTG_SWEEP_PIPE_H
#define MTG_SWEEP_PIPE_H

#include "mtg_sweep.h"

#define MTG_PIPE_BLOCK 256   // four waves: producers of pair 0 and 1, consumers of pair 0 and 1
#define MTG_PIPE_ROWS 128    // evaluations per workgroup
// samples per hand-over (one barrier each), by the 16-byte slots a sample takes in the kernel's widest structure:
// 3 chunks x CH samples x N2 KiB x 2 pairs next to 48 KiB of tables in 160 KiB of LDS; even: the pivot product is
// renormalised in pairs.  Rank 3 hands over two slots per sample and has room for eight samples per chunk; measured
// (scripts/pipe_ab.py, 32 000 rows, N = 1e4): 1.47 ms with four, 1.58 with six, 1.62 with eight -- the unrolled trip
// of 24 samples outgrows what the instruction cache keeps of the two roles.
#ifndef MTG_PIPE_CHUNK_SMALL
#define MTG_PIPE_CHUNK_SMALL 4
#endif
__host__ __device__ constexpr int mtg_pipe_chunk(int n2) { return n2 <= 2 ? MTG_PIPE_CHUNK_SMALL : 4; }
#define MTG_PIPE_RING 3      // chunks in the ring: one being written, one being read, one in between

// What the producer hands over per sample and lane: the NT = NR + NC propagators, then (cos, sin) of every complex
// term; packed as 16-byte slots [slot][lane] so that a wave's ds_write_b128 / ds_read_b128 touch consecutive
// addresses.  3 chunks x 4 samples x N2 KiB x 2 pairs next to 48 KiB of tables in 160 KiB of LDS: N2 <= 4.
template <int NR, int NC>
struct MtgPipeShape {
    static constexpr int NT = NR + NC;
    static constexpr int NV = NT + 2 * NC;
    static constexpr int N2 = (NV + 1) / 2;
};

// the producer's hand-over: the LDS traffic of this wave has landed, then all four waves meet.  (Spelled out because
// __syncthreads() may also wait for the global loads in flight -- the coming samples, prefetched on purpose.)
// `later`: samples whose generators must not be started on above the barrier -- they depend on nothing but the
// samples themselves, and the scheduler would hoist a whole trip's exp() above the trip's first hand-over, which
// makes that interval several times the others and the pair wait for each other in every one of them.
template <bool FAST>
__device__ __forceinline__ void mtg_pipe_barrier_before(double2 &d0, double2 &d1)
{
    if constexpr (FAST) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(d0.x), "+v"(d1.x) : : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(d0.x), "+v"(d0.y), "+v"(d1.x), "+v"(d1.y) : : "memory");
}
__device__ __forceinline__ void mtg_pipe_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// the consumer's: its arithmetic must not sink below the barrier (nothing but registers ties it to the LDS reads that
// the memory clobber pins) -- a consumer that reaches every barrier early and computes afterwards serialises the pair
__device__ __forceinline__ void mtg_pipe_barrier_after(double &z, double &invD)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(z), "+v"(invD) : : "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// producer: propagators and phases of every sample into ring[chunk mod R][sample][slot][lane]
// ---------------------------------------------------------------------------------------------------------------
template <int NR, int NC, bool FAST, int CH, class Tab>
__device__ __forceinline__ void mtg_pipe_produce(const MtgSolveArgs &a, int64_t e, uint32_t toff, double2 *ring,
                                                 const Tab *tab)
{
#pragma clang fp contract(off)
    constexpr int R = MTG_PIPE_RING, TRIP = R * CH;
    constexpr int NT = NR + NC;
    constexpr int N2 = MtgPipeShape<NR, NC>::N2;
    constexpr int NV = MtgPipeShape<NR, NC>::NV;
    const double *cf = a.coef + e;
    const int64_t cs = a.cstride;
    double ncr[NR > 0 ? NR : 1], cr64[NR > 0 ? NR : 1];
    double dc[NC > 0 ? NC : 1], ncc[NC > 0 ? NC : 1], cc64[NC > 0 ? NC : 1];
    double pr[NC > 0 ? NC : 1];
    int pm[NC > 0 ? NC : 1];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double c = cf[a.lay.cr(j) * cs];
        ncr[j] = -c; cr64[j] = c * -MTG_EXP_CSCALE;
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const double c = cf[a.lay.cc(k) * cs];
        dc[k] = cf[a.lay.dc(k) * cs];
        ncc[k] = -c; cc64[k] = c * -MTG_EXP_CSCALE;
        pr[k] = 0.0; pm[k] = 0;
    }
    const uint64_t dxt_left = a.t_stride ? a.dxt_bytes : (uint64_t)a.N * 16u;
    const __amdgpu_buffer_rsrc_t rdt = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.dxt, 0, (int)(dxt_left > 0xffffffffull ? 0xffffffffu : (uint32_t)dxt_left), 0x00020000);
    auto ld = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    const double t0 = ld(rdt, toff, 0).y;

    struct Gen { double v[NV + 1]; };
    auto compute = [&](const double2 dtc, Gen &g) __attribute__((always_inline)) {
        double *v = g.v;
#ifdef MTG_PIPE_DBG_NOPRODUCE
        for (int q = 0; q <= NV; ++q) v[q] = dtc.x;
        return;
#endif
        const double dxc = dtc.x, tc = dtc.y;
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = mtg_exp_cdx(ncr[j], cr64[j], dxc, tab);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            v[NR + k] = mtg_exp_cdx(ncc[k], cc64[k], dxc, tab);
            double cn, sn;
            if (FAST) mtg_phase_step(dc[k], dxc, pr[k], pm[k], &sn, &cn, tab);
            else sincos(dc[k] * (tc - t0), &sn, &cn);
            v[NT + 2 * k] = cn;
            v[NT + 2 * k + 1] = sn;
        }
        v[NV] = 0.0;
    };
    auto store = [&](const Gen &g, double2 *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < N2; ++q) dst[q * 64] = make_double2(g.v[2 * q], g.v[2 * q + 1]);
    };
    // Hand-over of the chunk written last.  It comes AFTER the arithmetic of the next chunk's first sample (every value
    // of `next` is pinned before the barrier) and BEFORE that sample's stores: the time the chunk's last ds_writes
    // need to land passes under arithmetic instead of under an s_waitcnt, and the wave arrives with nothing in flight.
    auto hand_over = [&](Gen &next, double2 &later0, double2 &later1) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NV; ++q) asm volatile("" : "+v"(next.v[q]));
        mtg_pipe_barrier_before<FAST>(later0, later1);
    };

    // A trip of the main loop is R chunks = TRIP samples with static ring slots; a sample's register is reloaded in
    // place, with the sample TRIP further on, as soon as it has been used.  (Loads past the light curve's end return
    // zeros or the neighbour's samples: produced, never consumed.)
    const uint32_t N = (uint32_t)a.N, nch = (N + CH - 1) / CH;
    double2 d[TRIP];
#pragma unroll
    for (int s = 0; s < TRIP; ++s) d[s] = ld(rdt, toff, 16u * s);
    uint32_t soff = 16u * TRIP;   // byte offset of the next trip's first sample
    uint32_t c = 0;
    Gen g[2];
    compute(d[0], g[0]);
    for (; c + R <= nch; c += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int i = u * CH + s;                 // this sample of the trip
                store(g[s & 1], ring + i * N2 * 64);
                d[i] = ld(rdt, toff, soff + 16u * i);
                compute(d[(i + 1) % TRIP], g[(s + 1) & 1]);
            }
            hand_over(g[0], d[(u * CH + CH + 1) % TRIP], d[(u * CH + CH + 2) % TRIP]);
        }
        soff += 16u * TRIP;
    }
    // the last nch mod R chunks (g[0] holds the first sample of chunk c)
    double2 *slot = ring;
#pragma unroll 1
    for (; c < nch; ++c) {
        double2 nx[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) nx[s] = ld(rdt, toff, 16u * (CH * c + s + 1));
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            store(g[s & 1], slot + s * N2 * 64);
            compute(nx[s], g[(s + 1) & 1]);
        }
        hand_over(g[0], nx[0], nx[1]);
        slot += CH * N2 * 64;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// consumer: the recurrences of mtg_sweep.h's step on the generators in LDS; writes lnL and status
// ---------------------------------------------------------------------------------------------------------------
template <int NR, int NC, int NB0, bool MEAN, int CH>
__device__ __forceinline__ void mtg_pipe_consume(const MtgSolveArgs &a, int64_t e, bool active, uint32_t yoff,
                                                 uint32_t toff, const double2 *ring)
{
#pragma clang fp contract(off)
    constexpr int R = MTG_PIPE_RING, TRIP = R * CH;
    constexpr int J = NR + 2 * NC;
    constexpr int NT = NR + NC;
    constexpr int N2 = MtgPipeShape<NR, NC>::N2;
    constexpr int NV = MtgPipeShape<NR, NC>::NV;
    const double *cf = a.coef + e;
    const int64_t cs = a.cstride;
    double ar[NR > 0 ? NR : 1], ac[NC > 0 ? NC : 1], bc[NC > 0 ? NC : 1];
#pragma unroll
    for (int j = 0; j < NR; ++j) ar[j] = cf[a.lay.ar(j) * cs];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        ac[k] = cf[a.lay.ac(k) * cs];
        bc[k] = cf[a.lay.bc(k) * cs];
    }
    const double jit = cf[a.lay.jit() * cs], slope = cf[a.lay.mean(0) * cs], icpt = cf[a.lay.mean(1) * cs];
    double S[J * (J + 1) / 2], Wt[J], f[J];
#pragma unroll
    for (int i = 0; i < J * (J + 1) / 2; ++i) S[i] = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) { Wt[i] = 0.0; f[i] = 0.0; }
    double invD = 0.0, z = 0.0, dot = 0.0, dprod = 1.0;
    int dexp = 0, dmin_hi = 0x7fffffff;

    const __amdgpu_buffer_rsrc_t ryv = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.yv, 0, (int)(a.yv_bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)a.yv_bytes), 0x00020000);
    const uint64_t dxt_left = a.t_stride ? a.dxt_bytes : (uint64_t)a.N * 16u;
    const __amdgpu_buffer_rsrc_t rdt = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.dxt, 0, (int)(dxt_left > 0xffffffffull ? 0xffffffffu : (uint32_t)dxt_left), 0x00020000);
    auto ld = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto ldt = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {   // t_n alone: the second half of (dx, t)
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff + 8u, 0));
    };

    struct Gen { double v[NV + 1]; };
    auto fetch = [&](Gen &g, const double2 *src) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < N2; ++q) {
            const double2 p = src[q * 64];
            g.v[2 * q] = p.x;
            if (2 * q + 1 <= NV) g.v[2 * q + 1] = p.y;
        }
    };
    // one step: mtg_sweep.h's, with phi / cos / sin fetched instead of computed
    auto step = [&](const double2 yvc, const double tc, const Gen &g) __attribute__((always_inline)) {
#ifdef MTG_PIPE_DBG_NOCONSUME
        z += yvc.x + g.v[0]; return;
#endif
        const double yc = yvc.x, vc = yvc.y;
        const double *ph = g.v;
        double U[J], V[J];
#pragma unroll
        for (int j = 0; j < NR; ++j) { U[j] = ar[j]; V[j] = 1.0; }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const double cn = g.v[NT + 2 * k], sn = g.v[NT + 2 * k + 1];
            if (k >= NC - NB0) {
                U[NR + 2 * k] = ac[k] * cn;
                U[NR + 2 * k + 1] = ac[k] * sn;
            } else {
                U[NR + 2 * k] = fma(ac[k], cn, bc[k] * sn);
                U[NR + 2 * k + 1] = fma(ac[k], sn, -(bc[k] * cn));
            }
            V[NR + 2 * k] = cn;
            V[NR + 2 * k + 1] = sn;
        }
        const double zs = z * invD;
        dot = fma(z, zs, dot);
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
        double D = MEAN ? vc + jit : vc;
        double zn = MEAN ? yc - fma(slope, tc, icpt) : yc;
#pragma unroll
        for (int i = 0; i < J; ++i) {
            double w = V[i];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int hi = i > j ? i : j, lo = i > j ? j : i;
                w = fma(-S[hi * (hi + 1) / 2 + lo], U[j], w);
            }
            Wt[i] = w;
            D = fma(U[i], w, D);
            zn = fma(-U[i], f[i], zn);
        }
        dmin_hi = min(dmin_hi, __double2hiint(D));
        invD = mtg_rcp(D);
        z = zn;
        dprod *= D;
    };
    auto renorm = [&]() __attribute__((always_inline)) {
        const double p = dprod;
        dprod = __builtin_amdgcn_frexp_mant(p);
        dexp += __builtin_amdgcn_frexp_exp(p);
    };

    const uint32_t N = (uint32_t)a.N, nch = (N + CH - 1) / CH;
    double2 y[TRIP];
    double tt[MEAN ? TRIP : 1];
#pragma unroll
    for (int s = 0; s < TRIP; ++s) {
        y[s] = ld(ryv, yoff, 16u * s);
        if constexpr (MEAN) tt[s] = ldt(rdt, toff, 16u * s);
    }
    uint32_t soff = 16u * TRIP;
    mtg_pipe_barrier();   // chunk 0 is there
    mtg_pipe_barrier();   // chunk 1 is there (the launcher asks for N >= 64)
    Gen g[2];
    fetch(g[0], ring);
    uint32_t c = 0;
    // whole trips of R chunks whose barriers are all due (chunk c + u + 2 exists for every u < R)
    for (; c + R + 2 <= nch; c += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int i = u * CH + s;
                // the generators of the next sample -- of this chunk or the first of the next one, which is complete
                fetch(g[(s + 1) & 1], ring + ((i + 1) % TRIP) * N2 * 64);
                double tc = 0.0;
                if constexpr (MEAN) tc = tt[i];
                step(y[i], tc, g[s & 1]);
                y[i] = ld(ryv, yoff, soff + 16u * i);
                if constexpr (MEAN) tt[i] = ldt(rdt, toff, soff + 16u * i);
                if (s & 1) renorm();
            }
            mtg_pipe_barrier_after(z, invD);                          // chunk c + u + 2 is there
        }
        soff += 16u * TRIP;
    }
    // the remaining chunks (at most R + 1, the last maybe partial), sample by sample
    const double2 *slot = ring;   // c is a multiple of R here
    int u = 0;
#pragma unroll 1
    for (; c < nch; ++c) {
        const uint32_t n0 = CH * c, left = N - n0 < (uint32_t)CH ? N - n0 : (uint32_t)CH;
#pragma unroll 1
        for (uint32_t s = 0; s < left; ++s) {
            fetch(g[0], slot + s * N2 * 64);
            const double2 ys = ld(ryv, yoff, 16u * (n0 + s));
            double tc = 0.0;
            if constexpr (MEAN) tc = ldt(rdt, toff, 16u * (n0 + s));
            step(ys, tc, g[0]);
            if ((s & 1u) || s + 1 == left) renorm();
        }
        if (c + 2 < nch) mtg_pipe_barrier_after(z, invD);
        u = u + 1 == R ? 0 : u + 1;
        slot = ring + u * (CH * N2 * 64);
    }
    dot = fma(z * z, invD, dot);

    const double logdet = fma((double)dexp, 0.69314718055994530942, log(dprod));
    double ll = -0.5 * fma((double)a.N, MTG_LN_2PI, dot + logdet);
    int st = MTG_ST_OK;
    if (dmin_hi <= 0) { st = MTG_ST_NOTPD; ll = -INFINITY; }
    else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
    if (active) {
        a.out[e] = ll;
        a.status[e] = st;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One QUARTET (producers of pair 0 and 1, consumers of pair 0 and 1; 128 rows) of a launch over all structures of a
// model -- the body of mtg_pipe_kernel (mtg_kernels_pipe.hip) and of either half of mtg_pipe_pair_kernel
// (mtg_kernels_pipe_pair.hip: two models' quartets in one workgroup).
// ---------------------------------------------------------------------------------------------------------------
template <int NR, int NC, int LASTB0>
struct MtgPipeB0 { static constexpr int value = (LASTB0 && NC > 0 && NR < 5 && NC < 4 && NR + 2 * NC <= 6) ? 1 : 0; };

template <int NR, int NC, int NB0, int CH>
__device__ __forceinline__ void mtg_pipe_rows(const MtgSolveArgs &a, int64_t e, bool active, int wave, double2 *ring,
                                              const MtgMathTables *tab)
{
    // the row's light curve (inside the one descriptor window: the launcher checks yv_bytes <= window_bytes)
    uint32_t lc = a.lc_index ? (uint32_t)a.lc_index[e] : 0u;
    const uint64_t lc_bytes = (uint64_t)a.N * 16u;
    bool lost = false;
    if (((uint64_t)lc + 1u) * lc_bytes > a.yv_bytes) { lost = active; lc = 0; active = false; }
    const uint32_t yoff = (uint32_t)((uint64_t)lc * lc_bytes);
    const uint32_t toff = a.t_stride ? yoff : 0u;
    if (wave < 2) {
        // table or libm sincos: decided per 64 rows exactly as mtg_solve_row decides it per wave
        double dmax = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) dmax = fmax(dmax, fabs(a.coef[e + a.lay.dc(k) * a.cstride]));
        const bool fast = !__any(active && !(dmax * *a.dxmax <= MTG_TRIG_FAST_MAX));
        if (fast) mtg_pipe_produce<NR, NC, true, CH>(a, e, toff, ring, tab);
        else mtg_pipe_produce<NR, NC, false, CH>(a, e, toff, ring, tab);
    } else {
        if (lost) {  // a device-side lc_index outside the resident set: no likelihood (as mtg_solve_row)
            a.out[e] = -INFINITY;
            a.status[e] = MTG_ST_NONFINITE;
        }
        if (a.has_mean) mtg_pipe_consume<NR, NC, NB0, true, CH>(a, e, active, yoff, toff, ring);
        else mtg_pipe_consume<NR, NC, NB0, false, CH>(a, e, active, yoff, toff, ring);
    }
}

template <int NR0, int NC0, int NSIG, int LASTB0, int CH, int S = 0>
__device__ __forceinline__ void mtg_pipe_dispatch(int k, const MtgSolveArgs &a, int64_t e, bool active, int wave,
                                                  double2 *ring, const MtgMathTables *tab)
{
    if (k == S) mtg_pipe_rows<NR0 + 2 * S, NC0 - S, MtgPipeB0<NR0 + 2 * S, NC0 - S, LASTB0>::value, CH>(a, e, active, wave, ring, tab);
    else if constexpr (S + 1 < NSIG) mtg_pipe_dispatch<NR0, NC0, NSIG, LASTB0, CH, S + 1>(k, a, e, active, wave, ring, tab);
}

// workgroup `block` -> (structure k, index of the block inside the structure's segment, rows before the segment, rows
// of the segment); false: the grid is sized for the worst padding and this block has no rows
template <int NSIG>
__device__ __forceinline__ bool mtg_pipe_locate(const MtgSolveArgs &a, int64_t &block, int &k, int64_t &first, int64_t &count)
{
    k = 0; first = 0; count = 0;
    if (NSIG > 1) {
        for (; k < NSIG; ++k) {
            count = a.seg_counts[k];
            const int64_t blocks = (count + MTG_PIPE_ROWS - 1) / MTG_PIPE_ROWS;
            if (block < blocks) break;
            block -= blocks;
            first += count;
        }
        return k < NSIG;
    }
    count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    return block * MTG_PIPE_ROWS < count;
}
template <int NSIG>
__device__ __forceinline__ bool mtg_pipe_has_block(const MtgSolveArgs &a, int64_t block)
{
    int k;
    int64_t first, count;
    return mtg_pipe_locate<NSIG>(a, block, k, first, count);
}

// wave 0..3 of the quartet (0, 1 producers; 2, 3 consumers), ring = [2 pairs][MTG_PIPE_RING * CH * N2 * 64] slots
template <int NR0, int NC0, int NSIG, int LASTB0, int CH>
__device__ __forceinline__ void mtg_pipe_quartet(const MtgSolveArgs &a, int64_t block, int wave, int lane, double2 *ring,
                                                 const MtgMathTables *tab)
{
    constexpr int N2 = MtgPipeShape<NR0, NC0>::N2;
    int k;
    int64_t first, count;
    if (!mtg_pipe_locate<NSIG>(a, block, k, first, count)) return;
    const int pair = wave & 1;
    const int64_t gid = block * MTG_PIPE_ROWS + pair * 64 + lane;
    bool active = gid < count;
    // idle lanes walk along on row 0 of the batch (any row with readable coefficients) and store nothing
    int64_t e = 0;
    if (active) e = a.list ? (int64_t)a.list[first + gid] : gid;
    if (active && a.status[e] != MTG_ST_OK) active = false;  // prior said -inf, or another rank's row
    mtg_pipe_dispatch<NR0, NC0, NSIG, LASTB0, CH>(k, a, e, active, wave, ring + pair * (MTG_PIPE_RING * CH * N2 * 64) + lane, tab);
}

#endif  // MTG_SWEEP_PIPE_H
