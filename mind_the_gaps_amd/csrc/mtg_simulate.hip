// mtg_simulate.hip -- posterior-predictive light-curve simulation on the device
// (SURVEY.md 8(f) row f2): the step BEFORE the hot path in the Protassov loop.
//
// Timmer & Koenig (1995) as the reference runs it
// (/root/reference/mind_the_gaps/simulator.py:369-420,468-501 and
// gpmodelling.py:478-539), for S posterior samples at once:
//   X_k = (g1 + i g2) sqrt(PSD(w_k) / 2),  k = 1..nfft/2   (Nyquist term real)     get_fft
//   counts = irfft(X) * sqrt(nfft dt sqrt(2 pi));  rate = counts / dt - <rate> + mean
//   random segment of the observed duration, bin-average onto the observing pattern  downsample
//   Gaussian or Poisson noise + error bars                                          noise_models.py
// The PSD is the celerite one of the context's model (Term.get_psd), evaluated from
// the coefficient columns mtg_prepare_kernel produced; the inverse FFTs are one batched
// hipFFT Z2D plan; random numbers are Philox4x32-10 (see mtg_sampler.hip).  The DC term
// is set to 0 instead of the reference's arbitrary 1e6: the series mean is removed
// anyway and 0 makes it vanish exactly.
#include "mtg_device.h"

#include <math.h>

namespace {

struct Philox {
    uint32_t c[4];
};

__device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

__device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                       uint32_t k1)
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

__device__ inline double u01(uint32_t hi, uint32_t lo)
{
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53;
}

// two standard normals from one Philox block (Box-Muller; u1 in (0, 1])
__device__ inline void normal2(const Philox &r, double *g1, double *g2)
{
    const double u1 = 1.0 - u01(r.c[0], r.c[1]), u2 = u01(r.c[2], r.c[3]);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    *g1 = rad * c;
    *g2 = rad * s;
}

enum { PURPOSE_SPECTRUM = 8, PURPOSE_SHIFT = 9, PURPOSE_NOISE = 10 };

}  // namespace

// celerite Term.get_psd at angular frequency w for simulation s (coefficient columns)
__device__ inline double mtg_psd(const double *cf, int64_t cs, const MtgCoefLayout &lay, int nr, int nc, double w)
{
    const double w2 = w * w;
    double p = 0.0;
    for (int j = 0; j < nr; ++j) {
        const double a = cf[lay.ar(j) * cs], c = cf[lay.cr(j) * cs];
        p += a * c / (c * c + w2);
    }
    for (int k = 0; k < nc; ++k) {
        const double a = cf[lay.ac(k) * cs], b = cf[lay.bc(k) * cs], c = cf[lay.cc(k) * cs], d = cf[lay.dc(k) * cs];
        const double w02 = c * c + d * d;
        p += ((a * c + b * d) * w02 + (a * c - b * d) * w2) / (w2 * w2 + 2.0 * (c * c - d * d) * w2 + w02 * w02);
    }
    return 0.79788456080286535588 * p;  // sqrt(2 / pi)
}

// X[s][k], k = 0..nfft/2 (hipFFT Z2D input layout).  sbase: index of the call's first series in the caller's global
// numbering (mtg_set_stream_base) -- it enters the random counters only, never an address.
__global__ void __launch_bounds__(256)
mtg_tk95_spectrum_kernel(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, double dt, const double *coef, int64_t cstride,
                         MtgCoefLayout lay, int nr0, int nc0, const int32_t *sig, const double *psd_table,
                         int64_t psd_rows, uint32_t seed_lo, uint32_t seed_hi, const double *given, double2 *X)
{
    const int64_t nk = nfft / 2 + 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * nk) return;
    const int64_t s = i / nk, k = i % nk;
    double re = 0.0, im = 0.0;
    if (k > 0) {
        const int64_t sg = s0 + s;
        double power;
        if (psd_table) {  // any callable PSD, evaluated by the host at the angular frequencies 2 pi k / (nfft dt)
            power = psd_table[(psd_rows > 1 ? sg : 0) * nk + k];
        } else {
            const int nr = nr0 + 2 * sig[sg], nc = nc0 - sig[sg];
            const double w = 6.28318530717958647692 * (double)k / ((double)nfft * dt);
            power = mtg_psd(coef + sg, cstride, lay, nr, nc, w);
        }
        const double amp = sqrt(0.5 * power);
        if (given) {  // the caller's own standard normals (mtg_set_simulate_draws): [series][re | im][k]
            re = given[(2 * sg) * nk + k];
            im = given[(2 * sg + 1) * nk + k];
        } else {
            const Philox r = philox4x32_10((uint32_t)k, PURPOSE_SPECTRUM, (uint32_t)(sg + sbase), (uint32_t)(k >> 32), seed_lo, seed_hi);
            normal2(r, &re, &im);
        }
        re *= amp; im *= amp;
        if (2 * k == nfft) im = 0.0;  // Nyquist term of an even-length series is real
    }
    X[i] = make_double2(re, im);
}

// cut_random_segment (simulator.py:536-539): start ~ U(time[0], time[-1] - duration), first fine sample at
// or after it; fixed_start >= 0 (tests) overrides the draw
__device__ inline int64_t tk95_segment_start(int64_t sg, int64_t nfft, int64_t seg_len, int64_t fixed_start,
                                             uint32_t seed_lo, uint32_t seed_hi)
{
    if (fixed_start >= 0) return fixed_start;
    const Philox rs = philox4x32_10(0u, PURPOSE_SHIFT, (uint32_t)sg, 0u, seed_lo, seed_hi);
    const double span = (double)(nfft - 1) - (double)seg_len;  // in units of dt
    int64_t j0 = span > 0.0 ? (int64_t)ceil(u01(rs.c[0], rs.c[1]) * span) : 0;
    if (j0 > nfft - seg_len) j0 = nfft - seg_len;
    if (j0 < 0) j0 = 0;
    return j0;
}

// The cut segment itself, as rates on the fine grid (the light curve the reference hands to its E13
// amplitude adjustment before down-sampling): out[sg - out_first][j] = series[s][j0 + j] scale / dt + mean  (out_first = 0: a buffer
// for the whole set; = s0: one for this chunk alone).
__global__ void __launch_bounds__(256)
mtg_tk95_segment_kernel(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, int64_t seg_len, double dt, double scale, double mean_rate,
                        const double *series, uint32_t seed_lo, uint32_t seed_hi, const int64_t *given_start, double *out, int64_t out_first)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * seg_len) return;
    const int64_t s = i / seg_len, j = i % seg_len, sg = s0 + s;
    const int64_t j0 = tk95_segment_start(sg + sbase, nfft, seg_len, given_start ? given_start[sg] : -1, seed_lo, seed_hi);
    out[(sg - out_first) * seg_len + j] = series[s * nfft + j0 + j] * scale / dt + mean_rate;
}

// numpy.random.poisson's two regimes on a Philox stream keyed by (epoch, series): Knuth's multiplication below 10, the
// transformed rejection of Hoermann (1993, PTRS) above; NaN for a negative mean (numpy raises: the epoch is flagged instead)
__device__ inline double tk95_poisson(double lam, uint32_t n, uint32_t sg, uint32_t seed_lo, uint32_t seed_hi)
{
    double counts;
    if (!(lam >= 0.0)) {
        counts = NAN;
    } else if (lam < 10.0) {
        const double limit = exp(-lam);
        double prod = 1.0;
        int kcount = 0;
        for (uint32_t ctr = 1; ctr < 64; ++ctr) {
            const Philox r = philox4x32_10(n, PURPOSE_NOISE, sg, ctr, seed_lo, seed_hi);
            prod *= u01(r.c[0], r.c[1]);
            if (prod <= limit) break;
            ++kcount;
            prod *= u01(r.c[2], r.c[3]);
            if (prod <= limit) break;
            ++kcount;
        }
        counts = (double)kcount;
    } else {
        const double slam = sqrt(lam), loglam = log(lam);
        const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
        const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
        counts = floor(lam + 0.5);
        for (uint32_t ctr = 1; ctr < 256; ++ctr) {
            const Philox r = philox4x32_10(n, PURPOSE_NOISE, sg, ctr, seed_lo, seed_hi);
            const double U = u01(r.c[0], r.c[1]) - 0.5, V = u01(r.c[2], r.c[3]);
            const double us = 0.5 - fabs(U);
            const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
            if (us >= 0.07 && V <= vr) { counts = kf; break; }
            if (kf < 0.0 || (us < 0.013 && V > us)) continue;
            if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) {
                counts = kf;
                break;
            }
        }
    }
    return counts;
}

// Segment cut + bin average onto the observing pattern + noise.  One thread per (simulation, epoch).
__global__ void __launch_bounds__(256)
mtg_tk95_observe_kernel(int64_t S, int64_t s0, int64_t sbase, int64_t N, int64_t nfft, int64_t seg_len, double dt, double scale,
                        double mean_rate, const double *series, const int32_t *win_lo, const int32_t *win_hi,
                        int noise_kind, double sigma_noise, const double *exposures, int64_t fixed_start,
                        uint32_t seed_lo, uint32_t seed_hi, const int64_t *given_start, double *clean, double *rates, double *dy,
                        MtgKraftTables kraft)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * N) return;
    const int64_t s = i / N, n = i % N, sg = s0 + s;
    const int64_t j0 = tk95_segment_start(sg + sbase, nfft, seg_len, given_start ? given_start[sg] : fixed_start, seed_lo, seed_hi);
    const double *x = series + s * nfft + j0;
    const int lo = win_lo[n], hi = win_hi[n];
    double acc = 0.0;
    for (int j = lo; j < hi; ++j) acc += x[j];
    // rate = counts / dt with counts = irfft * scale (hipFFT's C2R is unnormalised: 1 / nfft is in `scale`)
    const double rate = hi > lo ? acc / (double)(hi - lo) * scale / dt + mean_rate : NAN;
    const int64_t o = sg * N + n;
    if (clean) clean[o] = rate;
    double yv = rate, ev = 0.0;
    if (noise_kind == 1) {  // GaussianNoise (noise_models.py:152-184)
        const Philox r = philox4x32_10((uint32_t)n, PURPOSE_NOISE, (uint32_t)(sg + sbase), 0u, seed_lo, seed_hi);
        double g1, g2;
        normal2(r, &g1, &g2);
        yv = rate + sigma_noise * g1;
        ev = sigma_noise;
    } else if (noise_kind == 2) {  // PoissonNoise without background (noise_models.py:29-78)
        const double expo = exposures[n];
        const double counts = tk95_poisson(rate * expo, (uint32_t)n, (uint32_t)(sg + sbase), seed_lo, seed_hi);
        yv = counts / expo;
        ev = sqrt(counts) / expo;
    } else if (noise_kind == 3) {  // KraftNoise (noise_models.py:81-150): Poisson with background; the faint epochs' Bayesian estimates
        const double expo = exposures[n], bkg = kraft.bkg_counts[n], err = kraft.bkg_rate_err[n];
        const double total = tk95_poisson(rate * expo + bkg, (uint32_t)n, (uint32_t)(sg + sbase), seed_lo, seed_hi);
        yv = (total - bkg) / expo;
        ev = sqrt((sqrt(total) / expo) * (sqrt(total) / expo) + err * err);
        if (total < kraft.threshold) {   // posterior median and half-width of the 68 % interval, tabulated per (epoch, counts)
            const int64_t at = n * kraft.K + (int64_t)total;
            yv = kraft.median[at] / expo;
            ev = kraft.half[at] / expo;
        }
    }
    rates[o] = yv;
    dy[o] = ev;
}

// Make the simulated light curves the context's resident set: per-light-curve mean (the frozen
// ConstantModel(lightcurve.mean) of gpmodelling.py:83-87) and the interleaved (y - mean, (dy + 1e-12)^2)
// pairs the solve kernel reads (yerr = dy + 1e-12, gpmodelling.py:54).  One workgroup per light curve.
__global__ void __launch_bounds__(256)
mtg_tk95_resident_kernel(int64_t N, const double *rates, const double *dy, double2 *yv, double *means)
{
    const int64_t l = blockIdx.x;
    __shared__ double part[256];
    double acc = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) acc += rates[l * N + n];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const double mu = part[0] / (double)N;
    if (threadIdx.x == 0) means[l] = mu;
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
        const double e = dy[l * N + n] + 1e-12;
        yv[l * N + n] = make_double2(rates[l * N + n] - mu, e * e);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Inverse real transform of a length with large prime factors, by hand (chirp-z / Bluestein on power-of-two transforms).
//
// The simulator's grid is what the reference's arithmetic makes it (simulator.py:259-262): 1 087 853 = 13^2 x 41 x 157
// points for BASELINE configs[3].  hipFFT takes such a length through its own Bluestein path, and BUILDING that plan costs
// 0.9 s every time (2.0 s the first time in a process; scripts/plan_time_probe.py) against 15 ms for a power of two --
// on the critical path of every Protassov test, a fifth of one GPU's share at 8 GPUs.  So:
//   x_j = sum_k Y_k e^{+2 pi i jk/n},  jk = (j^2 + k^2 - (j - k)^2) / 2,  w_l = e^{i pi l^2 / n}:
//   x_j = w_j sum_k (Y_k w_k) conj(w_{j-k})   -- a convolution, done with two complex transforms of m = 2^p >= 2n - 1.
// Two series share one complex transform: both are real, so IDFT(Y1 + i Y2) = x1 + i x2.  The chirp's phase is reduced
// in integers (l^2 mod 2n), so w is accurate to an ulp however long the series.  Y is the Hermitian extension of the
// half spectrum hipFFT's Z2D would have read (imaginary parts of the k = 0 and Nyquist entries ignored likewise).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mtg_czt_chirp_kernel(int64_t n, double2 *chirp)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const uint64_t q = ((uint64_t)l * (uint64_t)l) % (uint64_t)(2 * n);   // l < 2^30: l^2 < 2^60
    double sn, cs;
    sincospi((double)q / (double)n, &sn, &cs);
    chirp[l] = make_double2(cs, sn);
}

// b_l = conj(w_l) / m for |l| < n, wrapped into [0, m) (the 1 / m of the unnormalised inverse transform rides along)
__global__ void __launch_bounds__(256) mtg_czt_b_kernel(int64_t n, int64_t m, const double2 *chirp, double2 *b)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= m) return;
    const int64_t d = l < n ? l : (m - l < n ? m - l : -1);
    const double inv = 1.0 / (double)m;
    b[l] = d < 0 ? make_double2(0.0, 0.0) : make_double2(chirp[d].x * inv, -chirp[d].y * inv);
}

// a[p][k] = (Y1_k + i Y2_k) w_k for k < n, 0 up to m; series per p and per p + 1 of the call's chunk (per = 2; the second may
// not exist) -- or, per = 1, one series per transform (Y2 = 0: a series' values then do not depend on its neighbour)
__global__ void __launch_bounds__(256) mtg_czt_pack_kernel(int64_t S, int per, int64_t n, int64_t m, const double2 *X, const double2 *chirp, double2 *a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pairs = (S + per - 1) / per;
    if (i >= pairs * m) return;
    const int64_t p = i / m, k = i % m, nk = n / 2 + 1;
    double2 out = make_double2(0.0, 0.0);
    if (k < n) {
        const int64_t kk = k <= n / 2 ? k : n - k;
        const bool edge = k == 0 || 2 * k == n;          // entries Z2D takes as real
        double2 y1 = X[(per * p) * nk + kk];
        double2 y2 = per == 2 && 2 * p + 1 < S ? X[(2 * p + 1) * nk + kk] : make_double2(0.0, 0.0);
        if (kk != k) { y1.y = -y1.y; y2.y = -y2.y; }
        if (edge) { y1.y = 0.0; y2.y = 0.0; }
        const double zr = y1.x - y2.y, zi = y1.y + y2.x;  // Y1 + i Y2
        const double2 w = chirp[k];
        out = make_double2(zr * w.x - zi * w.y, zr * w.y + zi * w.x);
    }
    a[p * m + k] = out;
}

__global__ void __launch_bounds__(256) mtg_czt_mul_kernel(int64_t pairs, int64_t m, const double2 *bhat, double2 *a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs * m) return;
    const double2 u = a[i], v = bhat[i % m];
    a[i] = make_double2(u.x * v.x - u.y * v.y, u.x * v.y + u.y * v.x);
}

// series[per p][j] + i series[per p + 1][j] = w_j c[p][j]
__global__ void __launch_bounds__(256) mtg_czt_unpack_kernel(int64_t S, int per, int64_t n, int64_t m, const double2 *c, const double2 *chirp, double *series)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pairs = (S + per - 1) / per;
    if (i >= pairs * n) return;
    const int64_t p = i / n, j = i % n;
    const double2 u = c[p * m + j], w = chirp[j];
    series[(per * p) * n + j] = u.x * w.x - u.y * w.y;
    if (per == 2 && 2 * p + 1 < S) series[(2 * p + 1) * n + j] = u.x * w.y + u.y * w.x;
}

void mtg_launch_czt_tables(int64_t n, int64_t m, double2 *chirp, double2 *b, hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_czt_chirp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, chirp);
    hipLaunchKernelGGL(mtg_czt_b_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, n, m, chirp, b);
}
void mtg_launch_czt_pack(int64_t S, int per, int64_t n, int64_t m, const double2 *X, const double2 *chirp, double2 *a, hipStream_t stream)
{
    const int64_t total = ((S + per - 1) / per) * m;
    hipLaunchKernelGGL(mtg_czt_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, S, per, n, m, X, chirp, a);
}
void mtg_launch_czt_mul(int64_t pairs, int64_t m, const double2 *bhat, double2 *a, hipStream_t stream)
{
    const int64_t total = pairs * m;
    hipLaunchKernelGGL(mtg_czt_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pairs, m, bhat, a);
}
void mtg_launch_czt_unpack(int64_t S, int per, int64_t n, int64_t m, const double2 *c, const double2 *chirp, double *series, hipStream_t stream)
{
    const int64_t total = ((S + per - 1) / per) * n;
    hipLaunchKernelGGL(mtg_czt_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, S, per, n, m, c, chirp, series);
}

void mtg_launch_tk95_resident(int64_t L, int64_t N, const double *rates, const double *dy, double2 *yv, double *means,
                              hipStream_t st)
{
    hipLaunchKernelGGL(mtg_tk95_resident_kernel, dim3((unsigned)L), dim3(256), 0, st, N, rates, dy, yv, means);
}

void mtg_launch_tk95_spectrum(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, double dt, const double *coef, int64_t cstride,
                              MtgCoefLayout lay, int nr0, int nc0, const int32_t *sig, const double *psd_table,
                              int64_t psd_rows, uint64_t seed, const double *given, double2 *X, hipStream_t st)
{
    const int64_t n = S * (nfft / 2 + 1);
    hipLaunchKernelGGL(mtg_tk95_spectrum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, s0, sbase, nfft, dt,
                       coef, cstride, lay, nr0, nc0, sig, psd_table, psd_rows, (uint32_t)seed, (uint32_t)(seed >> 32), given, X);
}

void mtg_launch_tk95_segment(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, int64_t seg_len, double dt, double scale,
                             double mean_rate, const double *series, uint64_t seed, const int64_t *given_start, double *out,
                             hipStream_t st, int64_t out_first)
{
    const int64_t n = S * seg_len;
    hipLaunchKernelGGL(mtg_tk95_segment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, s0, sbase, nfft, seg_len,
                       dt, scale, mean_rate, series, (uint32_t)seed, (uint32_t)(seed >> 32), given_start, out, out_first);
}

void mtg_launch_tk95_observe(int64_t S, int64_t s0, int64_t sbase, int64_t N, int64_t nfft, int64_t seg_len, double dt, double scale,
                             double mean_rate, const double *series, const int32_t *win_lo, const int32_t *win_hi,
                             int noise_kind, double sigma_noise, const double *exposures, int64_t fixed_start,
                             uint64_t seed, const int64_t *given_start, double *clean, double *rates, double *dy, hipStream_t st,
                             const MtgKraftTables &kraft)
{
    const int64_t n = S * N;
    hipLaunchKernelGGL(mtg_tk95_observe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, s0, sbase, N, nfft,
                       seg_len, dt, scale, mean_rate, series, win_lo, win_hi, noise_kind, sigma_noise, exposures,
                       fixed_start, (uint32_t)seed, (uint32_t)(seed >> 32), given_start, clean, rates, dy, kraft);
}


// ---------------------------------------------------------------------------
// Walker-averaged autocorrelation function of a chain (the convergence check of derive_posteriors,
// gpmodelling.py:260-272 -> emcee.autocorr.integrated_time -> function_1d per walker and dimension)
// ---------------------------------------------------------------------------
// chain[n_t][S] with S = E * W * P series (ensemble, walker, dimension fastest).  emcee computes, per series, the
// autocorrelation by zero-padded FFT, normalises it by its lag-0 value and averages over the walkers.  The
// inverse transform is linear, so the power spectra are normalised (lag 0 of a series = the sum of its squares)
// and averaged BEFORE it: S forward transforms, E * P inverse ones.  The chain is centred in its own
// [time][series] layout (coalesced over the series), transposed once to [series][time], and the transforms run
// over contiguous series: hipFFT's stock kernels.  (Strided plans over the chain's own layout worked too, but
// every new length cost a run-time compilation of ~1.4 s inside rocFFT.)

// Sums over time in two deterministic steps (a chain can be 10^5 steps of only a few dozen series, so the time
// axis has to be spread over workgroups): partial[tile][s] = sum over the tile's steps of v or (v - mean)^2, then
// the tiles added in order.  MTG_ACF_TILE steps per tile.
#define MTG_ACF_TILE 256
// SQUARES = false: partial sums of the chain;  true: x[t][s] = chain - mean (zero for t >= n_t) written and the
// partial sums of its squares
template <bool SQUARES>
__global__ void __launch_bounds__(256) mtg_acf_tile_kernel(int64_t n_t, int64_t n2, int64_t S, const double *chain, const double *mean,
                                                         double *x, double *partial)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int64_t t0 = (int64_t)blockIdx.y * MTG_ACF_TILE, t1 = t0 + MTG_ACF_TILE < n2 ? t0 + MTG_ACF_TILE : n2;
    const double m = SQUARES ? mean[s] : 0.0;
    double acc = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        if (SQUARES) {
            const double v = t < n_t ? chain[t * S + s] - m : 0.0;
            x[t * S + s] = v;
            acc = fma(v, v, acc);
        } else if (t < n_t) {
            acc += chain[t * S + s];
        }
    }
    partial[(int64_t)blockIdx.y * S + s] = acc;
}

// out[s] = scale * sum over the tiles (in order) of partial[tile][s]
__global__ void __launch_bounds__(256) mtg_acf_fold_kernel(int64_t tiles, int64_t S, double scale, const double *partial, double *out)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double acc = 0.0;
    for (int64_t k = 0; k < tiles; ++k) acc += partial[k * S + s];
    out[s] = acc * scale;
}

// [rows][cols] -> [cols][rows] through a 32 x 33 LDS tile (both sides coalesced)
__global__ void __launch_bounds__(256) mtg_acf_transpose_kernel(int64_t rows, int64_t cols, const double *in, double *out)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads, four rows each
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// g[e][p][k] = mean over the walkers of ensemble e of |f[e][w][p][k]|^2 / sumsq[e][w][p]   (a real spectrum);
// f: one row of nk coefficients per series, the series in (ensemble, walker, dimension) order
__global__ void __launch_bounds__(256) mtg_acf_power_kernel(int64_t nk, int64_t E, int W, int P, const double2 *f, const double *sumsq,
                                                          double2 *g)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk * E * P) return;
    const int64_t ep = i / nk, k = i % nk, e = ep / P;
    const int p = (int)(ep % P);
    const int64_t first = e * (int64_t)W * P + p;
    double acc = 0.0;
    for (int w = 0; w < W; ++w) {
        const int64_t sidx = first + (int64_t)w * P;
        const double2 v = f[sidx * nk + k];
        acc += (v.x * v.x + v.y * v.y) / sumsq[sidx];   // 0 / 0 = NaN for a walker that never moved, as emcee has it
    }
    g[i] = make_double2(acc / (double)W, 0.0);
}

// rho[t][ep] = scale * r[ep][t]   (t < n_t; r has n2 values per row)
__global__ void __launch_bounds__(256) mtg_acf_out_kernel(int64_t n_t, int64_t n2, int64_t EP, double scale, const double *r, double *rho)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_t * EP) return;
    const int64_t t = i / EP, ep = i % EP;
    rho[i] = scale * r[ep * n2 + t];
}

void mtg_launch_acf_transpose(int64_t rows, int64_t cols, const double *in, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_acf_transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, s, rows,
                       cols, in, out);
}

void mtg_launch_acf_power(int64_t nk, int64_t E, int W, int P, const double2 *f, const double *sumsq, double2 *g, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_acf_power_kernel, dim3((unsigned)((nk * E * P + 255) / 256)), dim3(256), 0, s, nk, E, W, P, f, sumsq, g);
}

void mtg_launch_acf_out(int64_t n_t, int64_t n2, int64_t EP, double scale, const double *r, double *rho, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_acf_out_kernel, dim3((unsigned)((n_t * EP + 255) / 256)), dim3(256), 0, s, n_t, n2, EP, scale, r, rho);
}

// x[t][s] (t < n2) = centred, zero-padded chain; sumsq[s]; `scratch`: (n2 / MTG_ACF_TILE + 1) * S + S doubles
void mtg_launch_acf_center(int64_t n_t, int64_t n2, int64_t S, const double *chain, double *x, double *sumsq, double *scratch,
                           hipStream_t s)
{
    const int64_t tiles = (n2 + MTG_ACF_TILE - 1) / MTG_ACF_TILE, tiles_in = (n_t + MTG_ACF_TILE - 1) / MTG_ACF_TILE;
    double *partial = scratch, *mean = scratch + tiles * S;
    const dim3 block(64), cols((unsigned)((S + 63) / 64));
    hipLaunchKernelGGL((mtg_acf_tile_kernel<false>), dim3(cols.x, (unsigned)tiles_in), block, 0, s, n_t, n2, S, chain,
                       (const double *)nullptr, (double *)nullptr, partial);
    hipLaunchKernelGGL(mtg_acf_fold_kernel, cols, block, 0, s, tiles_in, S, 1.0 / (double)n_t, partial, mean);
    hipLaunchKernelGGL((mtg_acf_tile_kernel<true>), dim3(cols.x, (unsigned)tiles), block, 0, s, n_t, n2, S, chain, mean, x, partial);
    hipLaunchKernelGGL(mtg_acf_fold_kernel, cols, block, 0, s, tiles, S, 1.0, partial, sumsq);
}
