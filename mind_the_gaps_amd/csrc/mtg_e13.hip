// mtg_e13.hip -- the flux-PDF adjustment of Emmanoulopoulos, McHardy & Papadakis (2013) on the device, as the reference
// runs it on every simulated segment when the simulator is made with pdf = "Lognormal" or "Uniform"
// (/root/reference/mind_the_gaps/simulator.py:65-140, E13Simulator.adjust_lightcurve_pdf; the distributions of
// stats.py:116-146): a white series drawn from the wanted flux PDF (mean = the simulator's, standard deviation = the
// segment's) repeatedly takes the Fourier AMPLITUDES of the TK95 segment and gives its values back by RANK, until it
// stops changing (np.allclose, rtol 1e-4) or max_iter is used up:
//     amplitudes = |rfft(segment)|                                   once
//     x          = pdf.rvs(n)            values = sort(x), descending once
//     loop:  spectrum = amplitudes exp(i angle(rfft(x)));  adjusted = irfft(spectrum)
//            new[argsort(-adjusted)] = values;  allclose(new, x) -> done;  x = new
// (only the ranks of `adjusted` are used, so the reference's amplitude normalisation 1 / (n // 2 + 1) and hipFFT's
// unnormalised inverse drop out).  Everything is batched over the `S` segments of one simulation chunk: the transforms
// are two batched hipFFT plans (plumbing, made by the caller), the rank matching two device-wide radix sorts per iteration
// (rocPRIM, as in mtg_sort.hip): all S n (adjusted, global index) pairs by value, descending, then -- stably -- by segment
// number, which leaves every segment's indices in the order of its ranks.  (rocPRIM's SEGMENTED sort gives a long segment
// to ONE workgroup: 16 segments of 870 000 samples -- the chunk of BASELINE configs[3] -- kept 16 compute units busy, 2.3 ms
// per segment and iteration; the two device-wide sorts use the whole GPU.)  The rest are the kernels below.  A segment that has
// converged is frozen (its slots of the batched transforms keep running and are not looked at), so every segment's result
// is the one a loop of its own would give.  Random numbers: Philox4x32-10 keyed by (seed, global series index, element).
#include "mtg_device.h"

#include <math.h>

#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct Philox {
    uint32_t c[4];
};

__device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

__device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
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

__device__ inline double u01(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53; }

enum { PURPOSE_E13 = 11 };   // (8, 9, 10: spectrum, shift, noise of mtg_simulate.hip)

// mean and (population) standard deviation of every segment, np.std's two passes; one workgroup per segment
__global__ void __launch_bounds__(1024) mtg_e13_std_kernel(int64_t n, const double *seg, double *stdv)
{
    const int64_t s = blockIdx.x;
    const double *x = seg + s * n;
    __shared__ double part[1024];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) acc += x[j];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    const double mu = part[0] / (double)n;
    __syncthreads();
    acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) { const double d = x[j] - mu; acc += d * d; }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) stdv[s] = sqrt(part[0] / (double)n);
}

// x[s][j] ~ the wanted PDF with the simulator's mean and segment s's standard deviation:
//   kind 1 lognormal (stats.py:116-129): s_ln = sqrt(ln(var / mean^2 + 1)), scale = mean^2 / sqrt(var + mean^2), x = scale exp(s_ln z)
//   kind 2 uniform   (stats.py:132-146): mean -+ sqrt(3) std

__global__ void __launch_bounds__(256) mtg_e13_draw_kernel(int64_t S, int64_t s0, int64_t sbase, int64_t n, int kind, double mean,
                                                          const double *stdv, uint32_t seed_lo, uint32_t seed_hi, double *x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per PAIR of elements (one Philox block)
    const int64_t half = (n + 1) / 2;
    if (i >= S * half) return;
    const int64_t s = i / half, p = i % half, sg = s0 + s + sbase;
    const Philox r = philox4x32_10((uint32_t)p, PURPOSE_E13, (uint32_t)sg, (uint32_t)(p >> 32), seed_lo, seed_hi);
    const double sd = stdv[s], var = sd * sd;
    double v0, v1;
    if (kind == 1) {
        const double u1 = 1.0 - u01(r.c[0], r.c[1]), u2 = u01(r.c[2], r.c[3]);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        const double sl = sqrt(log(var / (mean * mean) + 1.0)), scale = mean * mean / sqrt(var + mean * mean);
        v0 = scale * exp(sl * rad * cs);
        v1 = scale * exp(sl * rad * sn);
    } else {
        const double hw = sqrt(3.0) * sd;
        v0 = mean - hw + 2.0 * hw * u01(r.c[0], r.c[1]);
        v1 = mean - hw + 2.0 * hw * u01(r.c[2], r.c[3]);
    }
    x[s * n + 2 * p] = v0;
    if (2 * p + 1 < n) x[s * n + 2 * p + 1] = v1;
}

__global__ void __launch_bounds__(256) mtg_e13_iota_kernel(int64_t S, int64_t n, int32_t *idx, int32_t *done)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S * n) idx[i] = (int32_t)i;          // GLOBAL index s n + j: the payload of the rank sort
    if (done && i < S) done[i] = 0;
}

// segment number of every entry of a list of global indices
__global__ void __launch_bounds__(256) mtg_e13_segment_kernel(int64_t total, int64_t n, const int32_t *idx, uint32_t *segment)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) segment[i] = (uint32_t)(idx[i] / n);
}

// values[r] = x[order[r]]
__global__ void __launch_bounds__(256) mtg_e13_gather_kernel(int64_t total, const int32_t *order, const double *x, double *values)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) values[i] = x[order[i]];
}

// amp[s][k] = |spec[s][k]|
__global__ void __launch_bounds__(256) mtg_e13_abs_kernel(int64_t total, const double2 *spec, double *amp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    amp[i] = hypot(spec[i].x, spec[i].y);
}

// spec <- amp exp(i angle(spec))   (np.angle(0) = 0: a vanishing coefficient takes the amplitude on the real axis)
__global__ void __launch_bounds__(256) mtg_e13_phase_kernel(int64_t total, const double *amp, double2 *spec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double2 z = spec[i];
    const double mag = hypot(z.x, z.y), a = amp[i];
    spec[i] = mag > 0.0 ? make_double2(a * (z.x / mag), a * (z.y / mag)) : make_double2(a, 0.0);
}

// fresh[s][order[s][r]] = values[s][r] for segments still running;  notconv[s] = 1 where |fresh - x| > 1e-8 + 1e-4 |x| anywhere
// (np.allclose(new, current, rtol=1e-4): elementwise against the CURRENT series)
__global__ void __launch_bounds__(256) mtg_e13_scatter_kernel(int64_t S, int64_t n, const int32_t *order, const double *values,
                                                             const double *x, const int32_t *done, double *fresh, int32_t *notconv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * n) return;
    const int64_t s = i / n;      // (the list is grouped by segment, n entries each: position r of segment s is i = s n + r)
    if (done[s]) return;
    const int64_t at = order[i];
    const double v = values[i], c = x[at];
    fresh[at] = v;
    if (!(fabs(v - c) <= 1.0e-8 + 1.0e-4 * fabs(c))) notconv[s] = 1;
}

// x <- fresh for segments still running
__global__ void __launch_bounds__(256) mtg_e13_take_kernel(int64_t S, int64_t n, const double *fresh, const int32_t *done, double *x)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * n) return;
    if (done[i / n]) return;
    x[i] = fresh[i];
}

// after the copy: a segment whose step changed nothing beyond the tolerance is done; count the ones still running
__global__ void __launch_bounds__(256) mtg_e13_verdict_kernel(int64_t S, int32_t *done, int32_t *notconv, int32_t *running)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    if (!done[s]) {
        if (!notconv[s]) done[s] = 1;
        else atomicAdd(running, 1);
    }
    notconv[s] = 0;
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// bytes of the rocPRIM workspace for S segments of n elements (the larger of the two sorts)
size_t mtg_e13_sort_temp_bytes(int64_t S, int64_t n)
{
    size_t a = 0, b = 0;
    double *k = nullptr;
    int32_t *v = nullptr;
    uint32_t *g = nullptr;
    (void)rocprim::radix_sort_pairs_desc(nullptr, a, k, k, v, v, (size_t)(S * n), 0u, 64u, (hipStream_t) nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, b, g, g, v, v, (size_t)(S * n), 0u, 32u, (hipStream_t) nullptr);
    return a > b ? a : b;
}

void mtg_launch_e13_std(int64_t S, int64_t n, const double *seg, double *stdv, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_std_kernel, dim3((unsigned)S), dim3(1024), 0, st, n, seg, stdv);
}

void mtg_launch_e13_draw(int64_t S, int64_t s0, int64_t sbase, int64_t n, int kind, double mean, const double *stdv, uint64_t seed,
                         double *x, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_draw_kernel, dim3(blocks(S * ((n + 1) / 2))), dim3(256), 0, st, S, s0, sbase, n, kind, mean, stdv,
                       (uint32_t)seed, (uint32_t)(seed >> 32), x);
}

void mtg_launch_e13_iota(int64_t S, int64_t n, int32_t *idx, int32_t *done, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_iota_kernel, dim3(blocks(S * n)), dim3(256), 0, st, S, n, idx, done);
}

void mtg_launch_e13_abs(int64_t total, const double2 *spec, double *amp, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_abs_kernel, dim3(blocks(total)), dim3(256), 0, st, total, spec, amp);
}

void mtg_launch_e13_phase(int64_t total, const double *amp, double2 *spec, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_phase_kernel, dim3(blocks(total)), dim3(256), 0, st, total, amp, spec);
}

// order[s n + r] = GLOBAL index of the r-th largest entry of keys[s][:]: all pairs by value, descending, then stably by
// segment (bits of S).  keys_out, order_tmp, segment, segment_out: S n entries of scratch each.
hipError_t mtg_launch_e13_rank(int64_t S, int64_t n, const double *keys, double *keys_out, const int32_t *idx, int32_t *order_tmp,
                               uint32_t *segment, uint32_t *segment_out, int32_t *order, void *temp, size_t temp_bytes, hipStream_t st)
{
    const int64_t total = S * n;
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys, keys_out, idx, order_tmp, (size_t)total, 0u, 64u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mtg_e13_segment_kernel, dim3(blocks(total)), dim3(256), 0, st, total, n, order_tmp, segment);
    unsigned bits = 1;
    while (bits < 32 && ((uint64_t)(S - 1) >> bits) != 0) ++bits;
    return rocprim::radix_sort_pairs(temp, temp_bytes, segment, segment_out, order_tmp, order, (size_t)total, 0u, bits, st);
}

// values[s][:] = x[s][:] sorted, descending (the same two sorts, then a gather)
hipError_t mtg_launch_e13_sort_values(int64_t S, int64_t n, const double *x, double *keys_out, const int32_t *idx, int32_t *order_tmp,
                                      uint32_t *segment, uint32_t *segment_out, int32_t *order, double *values, void *temp,
                                      size_t temp_bytes, hipStream_t st)
{
    hipError_t e = mtg_launch_e13_rank(S, n, x, keys_out, idx, order_tmp, segment, segment_out, order, temp, temp_bytes, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mtg_e13_gather_kernel, dim3(blocks(S * n)), dim3(256), 0, st, S * n, order, x, values);
    return hipGetLastError();
}

// one step's bookkeeping after the rank sort: scatter + convergence test, copy, verdicts; *running (device) = segments not done
void mtg_launch_e13_step(int64_t S, int64_t n, const int32_t *order, const double *values, double *x, double *fresh, int32_t *done,
                         int32_t *notconv, int32_t *running, hipStream_t st)
{
    hipLaunchKernelGGL(mtg_e13_scatter_kernel, dim3(blocks(S * n)), dim3(256), 0, st, S, n, order, values, x, done, fresh, notconv);
    hipLaunchKernelGGL(mtg_e13_take_kernel, dim3(blocks(S * n)), dim3(256), 0, st, S, n, fresh, done, x);
    hipLaunchKernelGGL(mtg_e13_verdict_kernel, dim3(blocks(S)), dim3(256), 0, st, S, done, notconv, running);
}
