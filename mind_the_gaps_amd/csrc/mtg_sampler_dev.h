// mtg_sampler_dev.h -- device functions of the ensemble sampler (stretch move, accept step, speculative iteration):
// shared by the per-iteration kernels of mtg_sampler.hip and the persistent kernel of mtg_persist.hip, so that both
// run the same arithmetic on the same Philox counters -- the same chain to the last bit.
#pragma once
#include "mtg_device.h"
#include "mtg_prepare.h"

#include <math.h>

namespace {

struct Philox {
    uint32_t c[4];
};

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox{{c0, c1, c2, c3}};
}

// 53-bit uniform in [0, 1) from two 32-bit words
__host__ __device__ inline double u01(uint32_t hi, uint32_t lo)
{
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}

enum { PURPOSE_SPLIT = 1, PURPOSE_PROPOSE = 2, PURPOSE_ACCEPT = 3 };

}  // namespace

// Red/blue split + stretch proposal + theta -> coefficients for the `half`-th half of every
// ensemble; one workgroup per ensemble.
//   split (half 0 only; half 1 re-reads it): a uniformly random permutation of 0..W-1, obtained
//     by ranking one 64-bit Philox key per walker (ties -- probability ~W^2 2^-65 -- broken by
//     walker index); perm[e][0..W/2) is the first half.  Keys in LDS, W broadcast reads per walker.
//   proposal: z = ((a - 1) u + 1)^2 / a,  q = c_partner - (c_partner - s) z,  factor = (P - 1) ln z.
//   expansion: mtg_prepare_one on the proposal (prior verdict, coefficient columns, structure lists).
// s_key: W 64-bit keys, then W ranks (int), in LDS.
// The red/blue split of one iteration: a uniformly random permutation of 0..W-1 from ranked 64-bit Philox keys (see
// mtg_propose_part); perm[e][0..W/2) is the first half.  Ends with a barrier: the permutation is readable.
// The three steps of the split, callable apart (the speculative kernel runs the first two on its idle threads while the
// first 256 take the accept step: the permutation of the coming iteration depends on nothing but its number).
//   keys: one 64-bit Philox key per walker, ranks cleared            (all threads; barrier needed before the ranking)
__device__ __forceinline__ void mtg_split_keys(const MtgEnsembleArgs &g, uint32_t iteration, uint64_t *s_key)
{
    const int W = g.W;
    const int e = blockIdx.x;
    int *s_rank = (int *)(s_key + W);
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        const Philox r = philox4x32_10(iteration, PURPOSE_SPLIT, (uint32_t)e + g.e_base, (uint32_t)w, g.seed_lo, g.seed_hi);
        s_key[w] = ((uint64_t)r.c[0] << 32) | r.c[1];
        s_rank[w] = 0;
    }
}
//   ranking: W^2 comparisons spread over threads [t0, t0 + nt) of the workgroup -- `parts` threads per walker, each
//   counting over its share of the keys (one thread per walker and 512 serial comparisons were 12 us of the 25 us this
//   kernel took at W = 256)                                          (barrier needed before the ranks are read)
__device__ __forceinline__ void mtg_split_rank(const MtgEnsembleArgs &g, uint64_t *s_key, int t0, int nt)
{
    const int W = g.W;
    int *s_rank = (int *)(s_key + W);
    const int tid = (int)threadIdx.x - t0;
    if (tid < 0 || tid >= nt) return;
    const int parts = nt >= W ? nt / W : 1;
    const int span = (W + parts - 1) / parts;
    for (int i = tid; i < W * parts; i += nt) {
        const int w = i % W, part = i / W;
        const uint64_t mine = s_key[w];
        const int j0 = part * span, j1 = j0 + span < W ? j0 + span : W;
        int rank = 0;
        for (int j = j0; j < j1; ++j) {
            const uint64_t other = s_key[j];
            rank += (other < mine) || (other == mine && j < w);
        }
        if (parts > 1) atomicAdd(&s_rank[w], rank);
        else s_rank[w] = rank;
    }
}
//   the permutation itself                                            (barrier needed before it is read back)
__device__ __forceinline__ void mtg_split_write(const MtgEnsembleArgs &g, uint64_t *s_key)
{
    const int W = g.W;
    int32_t *p = g.perm + (int64_t)blockIdx.x * W;
    const int *s_rank = (const int *)(s_key + W);
    for (int w = threadIdx.x; w < W; w += blockDim.x) p[s_rank[w]] = w;
}

__device__ __forceinline__ void mtg_split_part(const MtgEnsembleArgs &g, uint32_t iteration, uint64_t *s_key)
{
    mtg_split_keys(g, iteration, s_key);
    __syncthreads();
    mtg_split_rank(g, s_key, 0, (int)blockDim.x);
    __syncthreads();
    mtg_split_write(g, s_key);
    __syncthreads();  // the permutation is read back below (same workgroup: visible after the barrier)
}

__device__ __forceinline__ void mtg_propose_part(const MtgEnsembleArgs &g, int half, uint32_t iteration, const MtgPrepArgs &pa,
                                                 uint64_t *s_key)
{
    const int W = g.W, P = g.P, H = W / 2;
    const int e = blockIdx.x;
    int32_t *p = g.perm + (int64_t)e * W;
    if (half == 0) mtg_split_part(g, iteration, s_key);
    double *q = const_cast<double *>(pa.theta);  // the proposals ARE the batch the expansion reads
    for (int k0 = 0; k0 < H; k0 += blockDim.x) {  // uniform trip count: mtg_prepare_one votes per wave
        const int k = k0 + (int)threadIdx.x;
        const bool live = k < H;
        const int64_t i = (int64_t)e * H + (live ? k : 0);
        if (live) {
            const Philox r = philox4x32_10(iteration, PURPOSE_PROPOSE + 16 * half, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
            const double u = u01(r.c[0], r.c[1]);
            const double zr = (g.a - 1.0) * u + 1.0;
            const double z = zr * zr / g.a;
            const int w = p[half * H + k];
            const int partner = p[(1 - half) * H + (int)(u01(r.c[2], r.c[3]) * (double)H)];
            const double *s = g.coords + ((int64_t)e * W + w) * P;
            const double *c = g.coords + ((int64_t)e * W + partner) * P;
            double *qo = q + i * P;
            for (int d = 0; d < P; ++d) qo[d] = c[d] - (c[d] - s[d]) * z;
            g.factor[i] = (double)(P - 1) * log(z);
        }
        mtg_prepare_one(pa, i, live);
    }
}

// Accept / reject of the half-step whose proposals are q[], with log-probabilities new_lnp[] / status[]: state
// update, per-ensemble running best; clears `clear_counts` (structure counters the solver of this half-step is done
// with) and, after the second half, appends the ensemble's state to the chain.  One workgroup per ensemble; its first
// 256 threads do the work, every thread takes part in the barriers.
__device__ __forceinline__ void mtg_accept_part(const MtgEnsembleArgs &g, int half, uint32_t iteration, const double *q,
                                                const double *new_lnp, const int32_t *status, int *clear_counts,
                                                double *chain_row, double *lnp_chain_row, double *s_best, int *s_idx)
{
    const int W = g.W, P = g.P, H = W / 2;
    const int e = blockIdx.x;
    const bool worker = threadIdx.x < 256;
    if (e == 0 && threadIdx.x < 64 && clear_counts) clear_counts[threadIdx.x] = 0;
    double my_best = -INFINITY;
    int my_idx = -1;
    if (worker)
        for (int k = threadIdx.x; k < H; k += 256) {
            const int64_t i = (int64_t)e * H + k;
            const int w = g.perm[(int64_t)e * W + half * H + k];
            const Philox r = philox4x32_10(iteration, PURPOSE_ACCEPT + 16 * half, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
            const double lu = log(u01(r.c[0], r.c[1]));
            const double cand = new_lnp[i];
            if (status[i] == MTG_ST_NOTPD) atomicAdd(g.n_notpd, 1);
            const int64_t wi = (int64_t)e * W + w;
            const double diff = g.factor[i] + cand - g.lnp[wi];
            if (diff > lu) {  // false for NaN and for cand = -inf
                for (int d = 0; d < P; ++d) g.coords[wi * P + d] = q[i * P + d];
                g.lnp[wi] = cand;
                g.naccept[wi] += 1;
                if (cand > my_best) { my_best = cand; my_idx = (int)i; }
            }
        }
    // best accepted proposal of the ensemble: inside each of the four worker waves by shuffles, then one barrier instead
    // of the nine of a tree over 256 LDS slots (measured: no difference -- this kernel's ~14 us are a chain of about ten
    // dependent global-memory round trips: split -> partner -> coordinates -> proposal -> expansion -> list append)
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(my_best, off);
        const int oi = __shfl_down(my_idx, off);
        if (ob > my_best) { my_best = ob; my_idx = oi; }
    }
    if (worker && (threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = my_best; s_idx[threadIdx.x >> 6] = my_idx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int wv = 1; wv < 4; ++wv)
            if (s_best[wv] > s_best[0]) { s_best[0] = s_best[wv]; s_idx[0] = s_idx[wv]; }
    if (threadIdx.x == 0 && s_idx[0] >= 0 && s_best[0] > g.best_lnp[e]) {
        g.best_lnp[e] = s_best[0];
        for (int d = 0; d < P; ++d) g.best_coords[(int64_t)e * P + d] = q[(int64_t)s_idx[0] * P + d];
    }
    // emcee stores the ensemble after both halves moved (every update of this ensemble's walkers
    // was made by this workgroup, before the barriers above)
    if (chain_row && worker)
        for (int j = threadIdx.x; j < W * P; j += 256)
            chain_row[(int64_t)e * W * P + j] = g.coords[(int64_t)e * W * P + j];
    if (lnp_chain_row && worker)
        for (int w = threadIdx.x; w < W; w += 256) lnp_chain_row[(int64_t)e * W + w] = g.lnp[(int64_t)e * W + w];
}

// ---------------------------------------------------------------------------------------------------------------
// Speculative iteration: BOTH half-steps of an iteration in one batch of 3 E H rows.
//
// A small ensemble leaves most of the GPU idle: the time-parallel solve of 64 rows takes as long as that of 192.
// The second half-step's proposals depend on the first half-step's outcome only through the partner's coordinates
// -- the partner either accepted its proposal or kept its place -- so both candidates are evaluated beside the first
// half-step's proposals, and the accept step picks the one that applies:
//     rows [0, EH)       first half-step's proposals (as mtg_propose_part, half 0)
//     rows [EH, 2 EH)    second half-step's proposals with the partner where it IS
//     rows [2 EH, 3 EH)  ... with the partner where its own proposal would put it
// One solve and one launch of this kernel per iteration instead of two and two.  Same Philox counters as the
// sequential form, hence the same chain to the last bit where the solver's arithmetic for a row does not depend on
// the batch (tests/test_device_sampler_gpu.py compares the two).
// phase stamps of the speculative kernel (measurements only: -DMTG_SAMPLER_STAMPS prints, for iteration 100, the time
// between the phase boundaries as thread 0 of workgroup 0 sees them, in 10 ns ticks of the constant-rate clock)
#ifdef MTG_SAMPLER_STAMPS
#define MTG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) mtg_stamps[k] = wall_clock64(); } while (0)
__device__ unsigned long long mtg_stamps[16];
#else
#define MTG_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void mtg_propose_both(const MtgEnsembleArgs &g, uint32_t iteration, const MtgPrepArgs &pa, uint64_t *s_key,
                                                 double *s_q)
{
    const int W = g.W, P = g.P, H = W / 2;
    const int e = blockIdx.x;
    const int64_t EH = (int64_t)g.E * H;
    // the split of this iteration: made beforehand for the whole run where that is small (mtg_split_all_kernel: it depends
    // on nothing but the iteration's number, and ranking W keys against each other is 2.3 (W = 128) to 5.0 us (W = 256) of
    // ONE compute unit's time -- whatever the number of threads -- in a kernel that is nothing but latency), or made here
    const int32_t *p = (g.perm_next ? g.perm_next : g.perm) + (int64_t)e * W;
    MTG_STAMP(6);
    if (!g.perm_next) mtg_split_part(g, iteration, s_key);
    MTG_STAMP(7);
    double *q = const_cast<double *>(pa.theta);
    // the first half-step's proposals: one thread each
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const int64_t i = (int64_t)e * H + k;
        const Philox r = philox4x32_10(iteration, PURPOSE_PROPOSE, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
        const double u = u01(r.c[0], r.c[1]);
        const double zr = (g.a - 1.0) * u + 1.0;
        const double z = zr * zr / g.a;
        const int w = p[k];
        const int partner = p[H + (int)(u01(r.c[2], r.c[3]) * (double)H)];
        const double *s = g.coords + ((int64_t)e * W + w) * P;
        const double *c = g.coords + ((int64_t)e * W + partner) * P;
        double *qo = q + i * P, *ql = s_q + (int64_t)k * P;
        for (int d = 0; d < P; ++d) { const double v = c[d] - (c[d] - s[d]) * z; qo[d] = v; ql[d] = v; }
        g.factor[i] = (double)(P - 1) * log(z);
    }
    __syncthreads();  // ... which other threads of this workgroup read as the partner's would-be place
    MTG_STAMP(8);
    // the second half-step's two candidates: one thread per candidate
    for (int t = threadIdx.x; t < 2 * H; t += blockDim.x) {
        const int k = t % H, which = t / H;
        const int64_t i = (int64_t)e * H + k;
        const Philox r = philox4x32_10(iteration, PURPOSE_PROPOSE + 16, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
        const double u = u01(r.c[0], r.c[1]);
        const double zr = (g.a - 1.0) * u + 1.0;
        const double z = zr * zr / g.a;
        const int w = p[H + k];
        const int j = (int)(u01(r.c[2], r.c[3]) * (double)H);              // the partner's slot in the first half
        const double *s = g.coords + ((int64_t)e * W + w) * P;
        const double *c = which ? s_q + (int64_t)j * P                      // where its proposal would put it (the LDS copy)
                                : g.coords + ((int64_t)e * W + p[j]) * P;   // where it is
        double *qo = q + ((which ? 2 * EH : EH) + i) * P, *ql = s_q + ((int64_t)(1 + which) * H + k) * P;
        for (int d = 0; d < P; ++d) { const double v = c[d] - (c[d] - s[d]) * z; qo[d] = v; ql[d] = v; }
        if (!which) g.factor[EH + i] = (double)(P - 1) * log(z);
    }
    __syncthreads();  // every row is expanded by the thread with its number, not by the one that wrote it
    MTG_STAMP(9);
}

// theta -> prior + coefficients of the 3 H rows just proposed, read from their copy in LDS (the one place of the kernel
// that holds the expansion: two inlined copies of its list-appending atomics trip the compiler's back end)
__device__ __forceinline__ void mtg_expand_proposals(const MtgEnsembleArgs &g, const MtgPrepArgs &pa, const double *s_q)
{
    const int H = g.W / 2, P = g.P;
    const int e = blockIdx.x;
    const int64_t EH = (int64_t)g.E * H;
    for (int t0 = 0; t0 < 3 * H; t0 += blockDim.x) {  // uniform trip count: mtg_prepare_one votes per wave
        const int t = t0 + (int)threadIdx.x;
        const bool live = t < 3 * H;
        const int block = live ? t / H : 0, k = live ? t % H : 0;
        mtg_prepare_one(pa, (int64_t)block * EH + (int64_t)e * H + k, live, s_q + ((int64_t)block * H + k) * P);
    }
}

// Accept / reject of both half-steps of a speculative iteration (rows as above).  s_acc: H ints of LDS.
__device__ __forceinline__ void mtg_accept_both(const MtgEnsembleArgs &g, uint32_t iteration, const double *q, const double *new_lnp,
                                                const int32_t *status, int *clear_counts, double *chain_row,
                                                double *lnp_chain_row, double *s_best, int *s_idx, int *s_acc)
{
    const int W = g.W, P = g.P, H = W / 2;
    const int e = blockIdx.x;
    const int64_t EH = (int64_t)g.E * H;
    const bool worker = threadIdx.x < 256;
    if (e == 0 && threadIdx.x < 64 && clear_counts) clear_counts[threadIdx.x] = 0;
    double my_best = -INFINITY;
    int64_t my_idx = -1;
    for (int half = 0; half < 2; ++half) {
        if (worker)
            for (int k = threadIdx.x; k < H; k += 256) {
                const int64_t i = (int64_t)e * H + k;
                int64_t row = i;  // the row that holds this walker's proposal
                if (half == 1) {
                    const Philox rp = philox4x32_10(iteration, PURPOSE_PROPOSE + 16, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
                    const int j = (int)(u01(rp.c[2], rp.c[3]) * (double)H);
                    row = (s_acc[j] ? 2 * EH : EH) + i;
                }
                const int w = g.perm[(int64_t)e * W + half * H + k];
                const Philox r = philox4x32_10(iteration, PURPOSE_ACCEPT + 16 * half, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
                const double lu = log(u01(r.c[0], r.c[1]));
                const double cand = new_lnp[row];
                if (status[row] == MTG_ST_NOTPD) atomicAdd(g.n_notpd, 1);
                const int64_t wi = (int64_t)e * W + w;
                const double diff = g.factor[half ? EH + i : i] + cand - g.lnp[wi];
                const bool accept = diff > lu;  // false for NaN and for cand = -inf
                if (half == 0) s_acc[k] = accept ? 1 : 0;
                if (accept) {
                    for (int d = 0; d < P; ++d) g.coords[wi * P + d] = q[row * P + d];
                    g.lnp[wi] = cand;
                    g.naccept[wi] += 1;
                    if (cand > my_best) { my_best = cand; my_idx = row; }
                }
            }
        __syncthreads();  // half 0: the accept flags; half 1: this workgroup's updates of the ensemble
        MTG_STAMP(1 + half);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(my_best, off);
        const int64_t oi = __shfl_down(my_idx, off);
        if (ob > my_best) { my_best = ob; my_idx = oi; }
    }
    if (worker && (threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = my_best; s_idx[threadIdx.x >> 6] = (int)my_idx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int wv = 1; wv < 4; ++wv)
            if (s_best[wv] > s_best[0]) { s_best[0] = s_best[wv]; s_idx[0] = s_idx[wv]; }
    if (threadIdx.x == 0 && s_idx[0] >= 0 && s_best[0] > g.best_lnp[e]) {
        g.best_lnp[e] = s_best[0];
        for (int d = 0; d < P; ++d) g.best_coords[(int64_t)e * P + d] = q[(int64_t)s_idx[0] * P + d];
    }
    if (chain_row && worker)
        for (int j = threadIdx.x; j < W * P; j += 256)
            chain_row[(int64_t)e * W * P + j] = g.coords[(int64_t)e * W * P + j];
    if (lnp_chain_row && worker)
        for (int w = threadIdx.x; w < W; w += 256) lnp_chain_row[(int64_t)e * W + w] = g.lnp[(int64_t)e * W + w];
}

// The accept step of one speculative iteration and the proposals of the next with the ensemble's state in LDS (small
// ensembles: W P <= MTG_SPEC_LDS_DOUBLES, H <= 256, the splits made beforehand).  mtg_accept_both + mtg_propose_both are a
// chain of about ten dependent global-memory round trips of ONE workgroup (profiles/r05_sampler_stamps.txt: seven phases
// of 1-2.6 us around 3-5 us of expansion); here everything the iteration reads -- coordinates, log-probabilities, both
// splits, the evaluated proposals with their results -- is fetched at once, the phases then run out of LDS and registers
// and only write to memory.  Same counters, same arithmetic, same stores: the chain of the other form to the last bit.
#define MTG_SPEC_LDS_DOUBLES 2048
__device__ __forceinline__ void mtg_spec_both_lds(const MtgEnsembleArgs &g, uint32_t iteration, uint32_t next_iteration,
                                                  const MtgPrepArgs &pa, const double *new_lnp, const int32_t *status,
                                                  int *clear_counts, double *chain_row, double *lnp_chain_row, double *s_best,
                                                  int *s_idx, int *s_acc, double *s_q, double *s_c, double *s_l, int *s_p, double *s_r)
{
    const int W = g.W, P = g.P, H = W / 2, nt = (int)blockDim.x, tid = (int)threadIdx.x;
    const int e = blockIdx.x;
    const int64_t EH = (int64_t)g.E * H, i = (int64_t)e * H + tid;
    const bool mine = tid < H;  // thread k holds walker slot k of either half
    double *q = const_cast<double *>(pa.theta);
    if (e == 0 && tid < 64 && clear_counts) clear_counts[tid] = 0;
    // ---- everything this iteration reads, in one go: every load is issued before the first of them is waited for (fixed trip
    // counts over registers: a loop that loads and stores to LDS waits for memory once per trip) ---------------------------
    constexpr int CPT = MTG_SPEC_LDS_DOUBLES / 256;  // coordinates per thread at most (blocks of 256 threads or more)
    double rc[CPT], rq[3][CPT / 2], rl[2];
    int rp_next[2];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int j = tid + u * nt;
        rc[u] = j < W * P ? g.coords[(int64_t)e * W * P + j] : 0.0;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)  // the proposals being decided (block b of the batch), where the new ones will go afterwards
#pragma unroll
        for (int u = 0; u < CPT / 2; ++u) {
            const int j = tid + u * nt;
            rq[b][u] = j < H * P ? q[((int64_t)b * EH + (int64_t)e * H) * P + j] : 0.0;
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int w = tid + u * nt;
        rl[u] = w < W ? g.lnp[(int64_t)e * W + w] : 0.0;
        rp_next[u] = w < W ? g.perm_next[(int64_t)e * W + w] : 0;
    }
    int w0 = 0, w1 = 0, j1 = 0, st0 = 0, st1a = 0, st1b = 0;
    double lu0 = 0.0, lu1 = 0.0, cand0 = 0.0, cand1a = 0.0, cand1b = 0.0, f0 = 0.0, f1 = 0.0;
    if (mine) {
        w0 = g.perm[(int64_t)e * W + tid];
        w1 = g.perm[(int64_t)e * W + H + tid];
        cand0 = new_lnp[i]; st0 = status[i]; f0 = g.factor[i];
        cand1a = new_lnp[EH + i]; st1a = status[EH + i];
        cand1b = new_lnp[2 * EH + i]; st1b = status[2 * EH + i];
        f1 = g.factor[EH + i];
    }
    // The expansion that follows this function walks through the model's description -- ~1.4 KB of the argument segment, read
    // with scalar loads whose addresses depend on what the previous one returned; the segment was written by the host
    // moments ago, so every 64-byte line of it is a trip to memory the first time (the expansion: 4.0 / 4.8 / 6.4 us at the
    // three sizes of profiles/r05_sampler_stamps.txt, 2.6 / 3.6 / 4.7 with the lines touched beforehand).  Touch every line
    // now, behind the vector loads above: the trips run under this function's phases.
    {
        int touched = 0;
        const int *words = (const int *)&pa.model;
#pragma unroll
        for (int off = 0; off < (int)(sizeof(MtgModel) / sizeof(int)); off += 16) touched ^= words[off];
        if (touched == 0x5eed5eed && clear_counts) clear_counts[63] = 0;  // (keeps the loads; what it writes is what is there)
    }
    // The iteration's random numbers depend on nothing but counters: every quarter of the workgroup makes one kind of them
    // for all H slots (a lone wave walks through a Philox block and a logarithm in ~0.5 us -- five blocks and four
    // logarithms one after the other were most of this kernel's time outside the expansion).
    //   s_r[0..H)  ln u of the first half's accept        s_r[H..2H)   ln u of the second half's accept
    //   s_r[2H..)  z, (P - 1) ln z of the first half's proposal, then of the second half's;   s_ri: partner slots
    double *s_lu0 = s_r, *s_lu1 = s_r + H, *s_z1 = s_r + 2 * H, *s_f1 = s_r + 3 * H, *s_z2 = s_r + 4 * H, *s_f2 = s_r + 5 * H;
    int *s_j1 = (int *)(s_r + 6 * H), *s_pj = s_j1 + H, *s_j2 = s_pj + H;
    {
        const int quarter = nt / 4, role = tid / quarter;
        for (int k = tid - role * quarter; k < H; k += quarter) {
            if (role == 0) {
                const Philox r = philox4x32_10(iteration, PURPOSE_ACCEPT, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
                s_lu0[k] = log(u01(r.c[0], r.c[1]));
            } else if (role == 1) {
                const Philox r = philox4x32_10(iteration, PURPOSE_ACCEPT + 16, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
                s_lu1[k] = log(u01(r.c[0], r.c[1]));
                const Philox rp = philox4x32_10(iteration, PURPOSE_PROPOSE + 16, (uint32_t)e + g.e_base, (uint32_t)k, g.seed_lo, g.seed_hi);
                s_j1[k] = (int)(u01(rp.c[2], rp.c[3]) * (double)H);
            } else {
                const Philox r = philox4x32_10(next_iteration, role == 2 ? PURPOSE_PROPOSE : PURPOSE_PROPOSE + 16, (uint32_t)e + g.e_base,
                                               (uint32_t)k, g.seed_lo, g.seed_hi);
                const double u = u01(r.c[0], r.c[1]);
                const double zr = (g.a - 1.0) * u + 1.0;
                const double z = zr * zr / g.a;
                (role == 2 ? s_z1 : s_z2)[k] = z;
                (role == 2 ? s_f1 : s_f2)[k] = (double)(P - 1) * log(z);
                (role == 2 ? s_pj : s_j2)[k] = (int)(u01(r.c[2], r.c[3]) * (double)H);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int j = tid + u * nt;
        if (j < W * P) s_c[j] = rc[u];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int u = 0; u < CPT / 2; ++u) {
            const int j = tid + u * nt;
            if (j < H * P) s_q[(int64_t)b * H * P + j] = rq[b][u];
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int w = tid + u * nt;
        if (w < W) { s_l[w] = rl[u]; s_p[w] = rp_next[u]; }
    }
    __syncthreads();
    MTG_STAMP(1);
    if (mine) { lu0 = s_lu0[tid]; lu1 = s_lu1[tid]; j1 = s_j1[tid]; }
    // ---- accept / reject, first half then second -----------------------------------------------------------------------
    double my_best = -INFINITY;
    int64_t my_idx = -1;
    if (mine) {
        if (st0 == MTG_ST_NOTPD) atomicAdd(g.n_notpd, 1);
        const int64_t wi = (int64_t)e * W + w0;
        const bool accept = f0 + cand0 - s_l[w0] > lu0;  // false for NaN and for cand = -inf
        s_acc[tid] = accept ? 1 : 0;
        if (accept) {
            for (int d = 0; d < P; ++d) { const double v = s_q[(int64_t)tid * P + d]; g.coords[wi * P + d] = v; s_c[w0 * P + d] = v; }
            g.lnp[wi] = cand0; s_l[w0] = cand0;
            g.naccept[wi] += 1;
            my_best = cand0; my_idx = i;
        }
    }
    __syncthreads();  // the accept flags
    MTG_STAMP(2);
    if (mine) {
        const int which = s_acc[j1] ? 2 : 1;
        const double cand = which == 2 ? cand1b : cand1a;
        if ((which == 2 ? st1b : st1a) == MTG_ST_NOTPD) atomicAdd(g.n_notpd, 1);
        const int64_t wi = (int64_t)e * W + w1;
        if (f1 + cand - s_l[w1] > lu1) {
            for (int d = 0; d < P; ++d) { const double v = s_q[((int64_t)which * H + tid) * P + d]; g.coords[wi * P + d] = v; s_c[w1 * P + d] = v; }
            g.lnp[wi] = cand; s_l[w1] = cand;
            g.naccept[wi] += 1;
            if (cand > my_best) { my_best = cand; my_idx = (int64_t)which * EH + i; }
        }
    }
    // best accepted proposal of the ensemble (as mtg_accept_both: the four worker waves by shuffles, then thread 0)
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(my_best, off);
        const int64_t oi = __shfl_down(my_idx, off);
        if (ob > my_best) { my_best = ob; my_idx = oi; }
    }
    if (tid < 256 && (tid & 63) == 0) { s_best[tid >> 6] = my_best; s_idx[tid >> 6] = (int)my_idx; }
    __syncthreads();  // the state after both halves; the waves' bests
    MTG_STAMP(5);
    if (tid == 0) {
        for (int wv = 1; wv < 4; ++wv)
            if (s_best[wv] > s_best[0]) { s_best[0] = s_best[wv]; s_idx[0] = s_idx[wv]; }
        if (s_idx[0] >= 0 && s_best[0] > g.best_lnp[e]) {
            g.best_lnp[e] = s_best[0];
            // row b EH + e H + k of the batch is slot (b H + k) of this ensemble's copy
            const int64_t row = s_idx[0], b = row / EH, k = row - b * EH - (int64_t)e * H;
            for (int d = 0; d < P; ++d) g.best_coords[(int64_t)e * P + d] = s_q[(b * H + k) * P + d];
        }
    }
    if (chain_row)
        for (int j = tid; j < W * P; j += nt) chain_row[(int64_t)e * W * P + j] = s_c[j];
    if (lnp_chain_row)
        for (int w = tid; w < W; w += nt) lnp_chain_row[(int64_t)e * W + w] = s_l[w];
    __syncthreads();  // thread 0 is done with the decided proposals: their place takes the new ones
    MTG_STAMP(7);
    // ---- the coming iteration's proposals (as mtg_propose_both, the state read from LDS) --------------------------------
    if (mine) {
        const double z = s_z1[tid];
        const double *sw = s_c + (int64_t)s_p[tid] * P;
        const double *c = s_c + (int64_t)s_p[H + s_pj[tid]] * P;
        double *qo = q + i * P, *ql = s_q + (int64_t)tid * P;
        for (int d = 0; d < P; ++d) { const double v = c[d] - (c[d] - sw[d]) * z; qo[d] = v; ql[d] = v; }
        g.factor[i] = s_f1[tid];
    }
    __syncthreads();  // ... which other threads read as the partner's would-be place
    MTG_STAMP(8);
    for (int t = tid; t < 2 * H; t += nt) {
        const int k = t % H, which = t / H;
        const int64_t ik = (int64_t)e * H + k;
        const double z = s_z2[k];
        const int j = s_j2[k];
        const double *sw = s_c + (int64_t)s_p[H + k] * P;
        const double *c = which ? s_q + (int64_t)j * P : s_c + (int64_t)s_p[j] * P;
        double *qo = q + ((which ? 2 * EH : EH) + ik) * P, *ql = s_q + ((int64_t)(1 + which) * H + k) * P;
        for (int d = 0; d < P; ++d) { const double v = c[d] - (c[d] - sw[d]) * z; qo[d] = v; ql[d] = v; }
        if (!which) g.factor[EH + ik] = s_f2[k];
    }
    __syncthreads();
    MTG_STAMP(9);
}

