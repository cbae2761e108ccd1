// mtg_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels of the celerite
// log-likelihood hot path.
//
//   mtg_lc_setup   : celerite.GP.compute(t, yerr) (called with yerr = dy + 1e-12
//                    at reference gpmodelling.py:54): interleaved (y, sigma^2 = yerr^2)
//                    and (dx_n, t_n) pairs, one 16-byte load each per sample.
//   mtg_prepare    : set_parameter_vector + log_prior + Term.coefficients
//                    (gpmodelling.py:147-151, celerite_models.py:7-90,
//                    celerite built-in terms): theta -> prior verdict and the
//                    (a, c | a, b, c, d) coefficient columns, grouped by
//                    structure signature (SHOTerm: 1 complex or 2 real terms).
//   mtg_solve<NR,NC>: celerite CholeskySolver.compute + log_determinant +
//                    dot_solve fused into ONE sweep over the N samples with the
//                    whole semiseparable state (S: J(J+1)/2, W, f) in VGPRs;
//                    one lane = one (theta, light curve) evaluation, one 8-byte
//                    store per evaluation.  Nothing per-sample is written.
//
// Roofline: nominally HBM (24 N bytes of t, y, sigma^2 per evaluation), in
// practice FP64 VALU issue; there is no dense contraction here, so no MFMA.
#include "mtg_device.h"
#include "mtg_prepare.h"
#include "mtg_sweep.h"

// ---------------------------------------------------------------------------
// light-curve set-up
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mtg_lc_setup_kernel(int64_t N, int64_t L, int64_t t_rows, const double *__restrict__ t,
                    const double *__restrict__ y, const double *__restrict__ yerr,
                    const double *__restrict__ y_offset, double2 *__restrict__ dxt,
                    double2 *__restrict__ yv, unsigned long long *__restrict__ dxmax_bits)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < L * N; i += stride) {
        const double s = yerr[i];  // celerite squares the yerr handed to compute()
        // y_offset: the frozen ConstantModel(lightcurve.mean) of gpmodelling.py:83-87,
        // one value per light curve, subtracted once instead of once per evaluation
        const double mu = y_offset ? y_offset[i / N] : 0.0;
        yv[i] = make_double2(y[i] - mu, s * s);
    }
    double mx = 0.0;
    bool sorted = true;
    for (int64_t i = i0; i < t_rows * N; i += stride) {
        const int64_t n = i % N;
        const double d = n == 0 ? 0.0 : t[i] - t[i - 1];
        dxt[i] = make_double2(d, t[i]);
        sorted = sorted && d >= 0.0;  // NaN times count as unsorted
        mx = fmax(mx, d);
    }
    // max over the grid: non-negative doubles order like their bit patterns
    atomicMax(dxmax_bits, (unsigned long long)__double_as_longlong(mx));
    if (!sorted) dxmax_bits[1] = 1ull;  // celerite.GP.compute raises ValueError for unsorted times
}

void mtg_launch_lc_setup(int64_t N, int64_t L, int64_t t_rows, const double *t, const double *y,
                         const double *yerr, const double *y_offset, double2 *dxt, double2 *yv,
                         double *dxmax, hipStream_t stream)
{
    int64_t blocks = (L * N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mtg_lc_setup_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, N, L,
                       t_rows, t, y, yerr, y_offset, dxt, yv, (unsigned long long *)dxmax);
}

// ---------------------------------------------------------------------------
// theta -> prior + coefficients
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mtg_prepare_kernel(MtgPrepArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    mtg_prepare_one(a, e, e < a.B);
}

void mtg_launch_prepare(const MtgPrepArgs &a, hipStream_t stream)
{
    const int64_t blocks = (a.B + 255) / 256;
    hipLaunchKernelGGL(mtg_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
}


template <int NR, int NC, int NB0 = 0>
__global__ void __launch_bounds__(MTG_BLOCK, mtg_waves_for(NR + 2 * NC)) mtg_solve_kernel(MtgSolveArgs a)
{
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    // solo launch (left-overs of a resident set beyond one window): one evaluation per wave, on lane 0
    const int per_block = a.solo ? MTG_BLOCK / 64 : MTG_BLOCK;
    if ((int64_t)blockIdx.x * per_block >= count) return;  // whole workgroup idle (e.g. empty signature list)
    __shared__ MtgMathTablesT<(NC > 0)> tab;
    mtg_fill_tables(&tab, threadIdx.x, MTG_BLOCK);
    __syncthreads();
    if (a.solo && (threadIdx.x & 63) != 0) return;
    const int64_t gid = (int64_t)blockIdx.x * per_block + (a.solo ? threadIdx.x >> 6 : threadIdx.x);
    if (gid >= count) return;
    int64_t first = 0;  // sorted order: this structure's segment of the sorted batch
    if (a.seg_counts)
        for (int i = 0; i < a.seg_k; ++i) first += a.seg_counts[i];
    const int64_t e = a.list ? (int64_t)a.list[first + gid] : gid;
    // prior said -inf (the structure lists hold accepted rows only; a sorted single-structure batch keeps its rejected
    // rows at the end of the order)
    if (a.status[e] != MTG_ST_OK) return;

    mtg_solve_row<NR, NC, NB0>(a, e, &tab);
}

// accuracy probe of the device math (tests only): e = exp(-x), (s, c) = sincos(x)
__global__ void __launch_bounds__(256)
mtg_math_probe_kernel(int64_t n, const double *x, double *e, double *s, double *c, double *rcp)
{
    __shared__ MtgMathTables tab;
    mtg_fill_tables(&tab, threadIdx.x, 256);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const double xv = live ? x[i] : 1.0;
    // the sweep's own functions: exp(-c dx) with c = 1, one phase step from phase 0
    const double ev = mtg_exp_cdx(-1.0, -MTG_EXP_CSCALE, xv, &tab);
    double sv, cv;
    if (__any(!(xv <= MTG_TRIG_FAST_MAX))) {
        sincos(xv, &sv, &cv);
    } else {
        double r = 0.0;
        int m16 = 0;
        mtg_phase_step(1.0, xv, r, m16, &sv, &cv, &tab);
    }
    if (live) {
        e[i] = ev;
        s[i] = sv;
        c[i] = cv;
        rcp[i] = mtg_rcp(xv);
    }
}

void mtg_launch_math_probe(int64_t n, const double *x, double *e, double *s, double *c, double *rcp,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       n, x, e, s, c, rcp);
}

template <int NR, int NC, int NB0 = 0>
static void mtg_launch_solve(const MtgSolveArgs &a, int64_t nlanes, hipStream_t stream)
{
    const int64_t blocks = (nlanes + MTG_BLOCK - 1) / MTG_BLOCK;
    if (blocks <= 0) return;
    hipLaunchKernelGGL((mtg_solve_kernel<NR, NC, NB0>), dim3((unsigned)blocks), dim3(MTG_BLOCK), 0, stream, a);
}

// Compiled structures: NR real + NC complex terms, J = NR + 2 NC <= MTG_MAX_J.
#define MTG_MAX_NR 10
#define MTG_MAX_NC 5
template <int NR, int NC, bool OK = (NR + NC > 0 && NR + 2 * NC <= MTG_MAX_J)>
struct MtgSel { static constexpr mtg_solve_launcher fn = mtg_launch_solve<NR, NC>; };
template <int NR, int NC>
struct MtgSel<NR, NC, false> { static constexpr mtg_solve_launcher fn = nullptr; };
#define MTG_ROW(nr)                                                                      \
    { MtgSel<(nr), 0>::fn, MtgSel<(nr), 1>::fn, MtgSel<(nr), 2>::fn, MtgSel<(nr), 3>::fn, \
      MtgSel<(nr), 4>::fn, MtgSel<(nr), 5>::fn }

// A white kernel -- JitterTerm alone, which celerite accepts and reference docs/notebooks/celerite_variance.ipynb cell 26
// fits: K = diag(sigma_n^2 + jitter), so the sweep has no state: D_n = sigma_n^2 + jitter, z_n = r_n.  The rows, lists and
// verdicts of mtg_solve_kernel; plain loads (nothing here is worth a buffer descriptor).
__global__ void __launch_bounds__(MTG_BLOCK) mtg_white_kernel(MtgSolveArgs a)
{
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    const int per_block = a.solo ? MTG_BLOCK / 64 : MTG_BLOCK;
    if (a.solo && (threadIdx.x & 63) != 0) return;
    const int64_t gid = (int64_t)blockIdx.x * per_block + (a.solo ? threadIdx.x >> 6 : threadIdx.x);
    if (gid >= count) return;
    int64_t first = 0;
    if (a.seg_counts)
        for (int i = 0; i < a.seg_k; ++i) first += a.seg_counts[i];
    const int64_t e = a.list ? (int64_t)a.list[first + gid] : gid;
    if (a.status[e] != MTG_ST_OK) return;
    const double *cf = a.coef + e;
    const double jit = cf[a.lay.jit() * a.cstride], slope = cf[a.lay.mean(0) * a.cstride], icpt = cf[a.lay.mean(1) * a.cstride];
    const uint64_t lc = a.lc_index ? (uint64_t)(uint32_t)a.lc_index[e] : 0u;
    if ((lc + 1u) * (uint64_t)a.N * 16u > a.yv_bytes) {  // a device-side lc_index outside the resident set
        a.out[e] = -INFINITY;
        a.status[e] = MTG_ST_NONFINITE;
        return;
    }
    const double2 *yv = a.yv + lc * (uint64_t)a.N;
    const double2 *dxt = a.dxt + (a.t_stride ? lc * (uint64_t)a.N : 0u);
    double dot = 0.0, dprod = 1.0;
    int dexp = 0, dmin_hi = 0x7fffffff;
    for (int64_t n = 0; n < a.N; ++n) {
        const double2 s = yv[n];
        const double D = s.y + jit, z = s.x - fma(slope, dxt[n].y, icpt);
        dmin_hi = min(dmin_hi, __double2hiint(D));
        dot = fma(z, z / D, dot);
        dprod *= D;
        dexp += __builtin_amdgcn_frexp_exp(dprod);
        dprod = __builtin_amdgcn_frexp_mant(dprod);
    }
    const double logdet = fma((double)dexp, 0.69314718055994530942, log(dprod));
    double ll = -0.5 * fma((double)a.N, MTG_LN_2PI, dot + logdet);
    int st = MTG_ST_OK;
    if (dmin_hi <= 0) { st = MTG_ST_NOTPD; ll = -INFINITY; }
    else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
    a.out[e] = ll;
    a.status[e] = st;
}

static void mtg_launch_white(const MtgSolveArgs &a, int64_t nlanes, hipStream_t stream)
{
    const int64_t blocks = (nlanes + MTG_BLOCK - 1) / MTG_BLOCK;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(mtg_white_kernel, dim3((unsigned)blocks), dim3(MTG_BLOCK), 0, stream, a);
}

static const mtg_solve_launcher mtg_solver_table[MTG_MAX_NR + 1][MTG_MAX_NC + 1] = {
    MTG_ROW(0), MTG_ROW(1), MTG_ROW(2), MTG_ROW(3), MTG_ROW(4), MTG_ROW(5),
    MTG_ROW(6), MTG_ROW(7), MTG_ROW(8), MTG_ROW(9), MTG_ROW(10)};

// the same structures with the LAST complex term known to have b = 0, for the small ranks (J <= 6)
template <int NR, int NC, bool OK = (NC > 0 && NR + 2 * NC <= 6)>
struct MtgSelB0 { static constexpr mtg_solve_launcher fn = mtg_launch_solve<NR, NC, 1>; };
template <int NR, int NC>
struct MtgSelB0<NR, NC, false> { static constexpr mtg_solve_launcher fn = nullptr; };
#define MTG_ROW_B0(nr) { nullptr, MtgSelB0<(nr), 1>::fn, MtgSelB0<(nr), 2>::fn, MtgSelB0<(nr), 3>::fn }
static const mtg_solve_launcher mtg_solver_table_b0[5][4] = {MTG_ROW_B0(0), MTG_ROW_B0(1), MTG_ROW_B0(2), MTG_ROW_B0(3),
                                                            MTG_ROW_B0(4)};

// last_b0: the model's last complex term has b = 0 whatever its parameters (mtg_set_model works it out)
int mtg_solver_uses_b0(int nr, int nc, int last_b0)
{
    if (nr < 0 || nc < 0 || nr > MTG_MAX_NR || nc > MTG_MAX_NC) return 0;
    return last_b0 && nr < 5 && nc < 4 && mtg_solver_table_b0[nr][nc] ? 1 : 0;
}

mtg_solve_launcher mtg_find_solver(int nr, int nc, int last_b0)
{
    if (nr < 0 || nc < 0 || nr > MTG_MAX_NR || nc > MTG_MAX_NC) return nullptr;
    if (last_b0 && nr < 5 && nc < 4 && mtg_solver_table_b0[nr][nc]) return mtg_solver_table_b0[nr][nc];
    if (nr + nc == 0) return mtg_launch_white;
    return mtg_solver_table[nr][nc];
}
