// mtg_sampler.hip -- device-resident lock-step ensemble sampler.
//
// emcee's stretch move as the reference drives it (gpmodelling.py:245-248; emcee
// 3.1.4 semantics restated in SURVEY.md Appendix B), for E independent ensembles of
// W walkers at once, with the walkers, their log-probabilities, the random
// numbers and the accept/reject step all resident on the GPU: one iteration is
//     mtg_split_kernel      random red/blue split of every ensemble
//     2 x { mtg_propose_kernel -> prepare + solve (the likelihood) -> mtg_accept_kernel }
// enqueued on one stream with no host synchronisation in between.
//
// Random numbers: Philox4x32-10 (Salmon et al. 2011), counter-based, so every draw is a
// pure function of (seed, iteration, purpose, ensemble, walker) -- reproducible and
// independent of launch geometry; tests/test_device_sampler_gpu.py replays the same
// stream on the host.
#include "mtg_device.h"

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

// Random red/blue split: a Fisher-Yates shuffle of 0..W-1 per ensemble (one thread per
// ensemble; W <= a few hundred, E in the thousands).  perm[e][0..W/2) is the first half.
__global__ void __launch_bounds__(64)
mtg_split_kernel(int E, int W, uint32_t iteration, uint32_t seed_lo, uint32_t seed_hi, int32_t *perm)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int32_t *p = perm + (int64_t)e * W;
    for (int i = 0; i < W; ++i) p[i] = i;
    for (int i = W - 1; i > 0; i -= 2) {
        // one Philox call feeds two swaps
        const Philox r = philox4x32_10(iteration, PURPOSE_SPLIT, (uint32_t)e, (uint32_t)i, seed_lo, seed_hi);
        int j = (int)(u01(r.c[0], r.c[1]) * (double)(i + 1));
        int32_t t = p[i]; p[i] = p[j]; p[j] = t;
        if (i - 1 > 0) {
            j = (int)(u01(r.c[2], r.c[3]) * (double)i);
            t = p[i - 1]; p[i - 1] = p[j]; p[j] = t;
        }
    }
}

// Stretch proposal for the `half`-th half of every ensemble:
//   z = ((a - 1) u + 1)^2 / a,  q = c_partner - (c_partner - s) z,  factor = (P - 1) ln z.
__global__ void __launch_bounds__(256)
mtg_propose_kernel(int E, int W, int P, int half, uint32_t iteration, uint32_t seed_lo, uint32_t seed_hi,
                   double a, const int32_t *__restrict__ perm, const double *__restrict__ coords,
                   double *__restrict__ q, double *__restrict__ factor)
{
    const int H = W / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)E * H) return;
    const int e = (int)(i / H), k = (int)(i % H);
    const int32_t *p = perm + (int64_t)e * W;
    const Philox r = philox4x32_10(iteration, PURPOSE_PROPOSE + 16 * half, (uint32_t)e, (uint32_t)k, seed_lo, seed_hi);
    const double u = u01(r.c[0], r.c[1]);
    const double zr = (a - 1.0) * u + 1.0;
    const double z = zr * zr / a;
    const int w = p[half * H + k];
    const int partner = p[(1 - half) * H + (int)(u01(r.c[2], r.c[3]) * (double)H)];
    const double *s = coords + ((int64_t)e * W + w) * P;
    const double *c = coords + ((int64_t)e * W + partner) * P;
    double *qo = q + i * P;
    for (int d = 0; d < P; ++d) qo[d] = c[d] - (c[d] - s[d]) * z;
    factor[i] = (double)(P - 1) * log(z);
}

// Accept / reject, state update, per-ensemble running best.  One workgroup per ensemble.
__global__ void __launch_bounds__(256)
mtg_accept_kernel(int E, int W, int P, int half, uint32_t iteration, uint32_t seed_lo, uint32_t seed_hi,
                  const int32_t *__restrict__ perm, const double *__restrict__ q,
                  const double *__restrict__ factor, const double *__restrict__ new_lnp,
                  const int32_t *__restrict__ status, double *__restrict__ coords, double *__restrict__ lnp,
                  int32_t *__restrict__ naccept, double *__restrict__ best_lnp,
                  double *__restrict__ best_coords, int32_t *__restrict__ n_notpd)
{
    const int H = W / 2;
    const int e = blockIdx.x;
    __shared__ double s_best[256];
    __shared__ int s_idx[256];
    double my_best = -INFINITY;
    int my_idx = -1;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const int64_t i = (int64_t)e * H + k;
        const int w = perm[(int64_t)e * W + half * H + k];
        const Philox r = philox4x32_10(iteration, PURPOSE_ACCEPT + 16 * half, (uint32_t)e, (uint32_t)k, seed_lo, seed_hi);
        const double lu = log(u01(r.c[0], r.c[1]));
        const double cand = new_lnp[i];
        if (status[i] == MTG_ST_NOTPD) atomicAdd(n_notpd, 1);
        const int64_t wi = (int64_t)e * W + w;
        const double diff = factor[i] + cand - lnp[wi];
        if (diff > lu) {  // false for NaN and for cand = -inf
            for (int d = 0; d < P; ++d) coords[wi * P + d] = q[i * P + d];
            lnp[wi] = cand;
            naccept[wi] += 1;
            if (cand > my_best) { my_best = cand; my_idx = (int)i; }
        }
    }
    s_best[threadIdx.x] = my_best;
    s_idx[threadIdx.x] = my_idx;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && s_best[threadIdx.x + s] > s_best[threadIdx.x]) {
            s_best[threadIdx.x] = s_best[threadIdx.x + s];
            s_idx[threadIdx.x] = s_idx[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_idx[0] >= 0 && s_best[0] > best_lnp[e]) {
        best_lnp[e] = s_best[0];
        for (int d = 0; d < P; ++d) best_coords[(int64_t)e * P + d] = q[(int64_t)s_idx[0] * P + d];
    }
}

// Running best of the INITIAL state (before any move).
__global__ void __launch_bounds__(256)
mtg_initial_best_kernel(int E, int W, int P, const double *__restrict__ coords, const double *__restrict__ lnp,
                        double *__restrict__ best_lnp, double *__restrict__ best_coords)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double b = -INFINITY;
    int bi = 0;
    for (int w = 0; w < W; ++w) {
        const double v = lnp[(int64_t)e * W + w];
        if (v > b) { b = v; bi = w; }
    }
    best_lnp[e] = b;
    for (int d = 0; d < P; ++d) best_coords[(int64_t)e * P + d] = coords[((int64_t)e * W + bi) * P + d];
}

void mtg_launch_split(int E, int W, uint32_t iteration, uint64_t seed, int32_t *perm, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_split_kernel, dim3((E + 63) / 64), dim3(64), 0, s, E, W, iteration,
                       (uint32_t)seed, (uint32_t)(seed >> 32), perm);
}

void mtg_launch_propose(int E, int W, int P, int half, uint32_t iteration, uint64_t seed, double a,
                        const int32_t *perm, const double *coords, double *q, double *factor, hipStream_t s)
{
    const int64_t n = (int64_t)E * (W / 2);
    hipLaunchKernelGGL(mtg_propose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, E, W, P, half,
                       iteration, (uint32_t)seed, (uint32_t)(seed >> 32), a, perm, coords, q, factor);
}

void mtg_launch_accept(int E, int W, int P, int half, uint32_t iteration, uint64_t seed, const int32_t *perm,
                       const double *q, const double *factor, const double *new_lnp, const int32_t *status,
                       double *coords, double *lnp, int32_t *naccept, double *best_lnp, double *best_coords,
                       int32_t *n_notpd, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_accept_kernel, dim3(E), dim3(256), 0, s, E, W, P, half, iteration, (uint32_t)seed,
                       (uint32_t)(seed >> 32), perm, q, factor, new_lnp, status, coords, lnp, naccept, best_lnp,
                       best_coords, n_notpd);
}

void mtg_launch_initial_best(int E, int W, int P, const double *coords, const double *lnp, double *best_lnp,
                             double *best_coords, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_initial_best_kernel, dim3((E + 255) / 256), dim3(256), 0, s, E, W, P, coords, lnp,
                       best_lnp, best_coords);
}
