// mtg_sampler.hip -- device-resident lock-step ensemble sampler.
//
// emcee's stretch move as the reference drives it (gpmodelling.py:245-248; emcee
// 3.1.4 semantics restated in SURVEY.md Appendix B), for E independent ensembles of
// W walkers at once, with the walkers, their log-probabilities, the random
// numbers and the accept/reject step all resident on the GPU: one iteration is
//     2 x { solve (the likelihood) -> mtg_sampler_step_kernel }
// enqueued on one stream with no host synchronisation in between.  The step kernel takes the accept step of
// the half-step just evaluated (state update, running best, chain row after the second half, clearing the
// structure lists the solver used) and then makes the proposals of the next half-step: the red/blue split
// (first half-step of an iteration), the stretch move, and their expansion into celerite coefficients
// (mtg_prepare_one) -- two launches per half-step where the straightforward
// split | propose | memset | prepare | solve | accept | copy sequence needs six.
//
// Random numbers: Philox4x32-10 (Salmon et al. 2011), counter-based, so every draw is a
// pure function of (seed, iteration, purpose, ensemble, walker) -- reproducible and
// independent of launch geometry; tests/test_device_sampler_gpu.py replays the same
// stream on the host.
#include "mtg_sampler_dev.h"

__global__ void __launch_bounds__(1024)
mtg_sampler_spec_kernel(MtgEnsembleArgs g, int do_accept, uint32_t iteration, const double *new_lnp, const int32_t *status,
                        int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose, uint32_t next_iteration,
                        MtgPrepArgs pa, int state_in_lds)
{
    extern __shared__ uint64_t s_key[];   // W keys, W ranks (int), H accept flags (int)
    __shared__ double s_best[256];
    __shared__ int s_idx[256];
    int *s_acc = (int *)(s_key + g.W) + g.W;
    // the 3 H proposals of the coming iteration, kept beside their global copy (mtg_prepare_one reads them from here)
    double *s_q = (double *)(s_key + g.W + (3 * g.W / 2 + 1) / 2 + 1);
    MTG_STAMP(0);
    const bool in_lds = state_in_lds && do_accept && do_propose;  // a small ensemble in the middle of a run
    if (in_lds) {
        double *s_c = s_q + (int64_t)3 * (g.W / 2) * g.P, *s_l = s_c + (int64_t)g.W * g.P;
        mtg_spec_both_lds(g, iteration, next_iteration, pa, new_lnp, status, clear_counts, chain_row, lnp_chain_row, s_best, s_idx,
                          s_acc, s_q, s_c, s_l, (int *)(s_l + g.W), s_l + g.W + (g.W + 1) / 2);
    } else {
        if (do_accept) {
            mtg_accept_both(g, iteration, pa.theta, new_lnp, status, clear_counts, chain_row, lnp_chain_row, s_best, s_idx, s_acc);
            __syncthreads();
        }
        MTG_STAMP(5);
        if (do_propose) mtg_propose_both(g, next_iteration, pa, s_key, s_q);
    }
    if (do_propose) mtg_expand_proposals(g, pa, s_q);
#ifdef MTG_SAMPLER_STAMPS
    __syncthreads();
    MTG_STAMP(10);
    if (threadIdx.x == 0 && blockIdx.x == 0 && iteration == 100) {
        if (in_lds)
            printf("sampler stamps, state in LDS (10 ns ticks): fetch %llu accept0 %llu accept1+best %llu chain %llu | propose1 %llu propose2 %llu expand %llu | total %llu\n",
                   mtg_stamps[1] - mtg_stamps[0], mtg_stamps[2] - mtg_stamps[1], mtg_stamps[5] - mtg_stamps[2], mtg_stamps[7] - mtg_stamps[5],
                   mtg_stamps[8] - mtg_stamps[7], mtg_stamps[9] - mtg_stamps[8], mtg_stamps[10] - mtg_stamps[9], mtg_stamps[10] - mtg_stamps[0]);
        else
            printf("sampler stamps (10 ns ticks): accept0 %llu accept1 %llu best+chain %llu | split %llu propose1 %llu propose2 %llu expand %llu | total %llu\n",
                   mtg_stamps[1] - mtg_stamps[0], mtg_stamps[2] - mtg_stamps[1], mtg_stamps[5] - mtg_stamps[2], mtg_stamps[7] - mtg_stamps[6],
                   mtg_stamps[8] - mtg_stamps[7], mtg_stamps[9] - mtg_stamps[8], mtg_stamps[10] - mtg_stamps[9], mtg_stamps[10] - mtg_stamps[0]);
    }
#endif
}

// One kernel between two solves: the accept step of the half-step just evaluated, then -- do_propose -- the
// proposals of the next one (expanded into the OTHER bank of structure lists: workgroup 0 clears the bank the
// solver has just used while the others may already be appending to the next one).  do_accept = 0: the very first
// proposals of a run.
__global__ void __launch_bounds__(1024)
mtg_sampler_step_kernel(MtgEnsembleArgs g, int do_accept, int half, uint32_t iteration, const double *new_lnp,
                        const int32_t *status, int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose,
                        int next_half, uint32_t next_iteration, MtgPrepArgs pa)
{
    extern __shared__ uint64_t s_key[];
    __shared__ double s_best[256];
    __shared__ int s_idx[256];
    if (do_accept) {
        mtg_accept_part(g, half, iteration, pa.theta, new_lnp, status, clear_counts, chain_row, lnp_chain_row, s_best, s_idx);
        __syncthreads();  // this workgroup's updates of the ensemble are visible to all its threads
    }
    if (do_propose) mtg_propose_part(g, next_half, next_iteration, pa, s_key);
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

void mtg_launch_sampler_step(const MtgEnsembleArgs &g, int do_accept, int half, uint32_t iteration, const double *new_lnp,
                             const int32_t *status, int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose,
                             int next_half, uint32_t next_iteration, const MtgPrepArgs &pa, hipStream_t s)
{
    // the proposals need W / 2 threads, the accept step up to 256; the split of a first half-step ranks W keys against
    // each other and takes as many threads as a workgroup of a few ensembles can have (many ensembles: the GPU is
    // full anyway)
    int threads = 256;
    if (do_propose && next_half == 0 && g.E <= 64) threads = 1024;
    hipLaunchKernelGGL(mtg_sampler_step_kernel, dim3((unsigned)g.E), dim3(threads), (size_t)g.W * (sizeof(uint64_t) + sizeof(int)), s,
                       g, do_accept, half, iteration, new_lnp, status, clear_counts, chain_row, lnp_chain_row, do_propose, next_half,
                       next_iteration, pa);
}

// The red/blue splits of `steps` consecutive iterations, one workgroup per (ensemble, iteration): perm_all[s][e][W].
__global__ void __launch_bounds__(1024)
mtg_split_all_kernel(MtgEnsembleArgs g, uint32_t iteration0, int32_t *perm_all)
{
    extern __shared__ uint64_t s_key[];   // W keys, W ranks (int)
    const uint32_t iteration = iteration0 + blockIdx.y;
    mtg_split_keys(g, iteration, s_key);
    __syncthreads();
    mtg_split_rank(g, s_key, 0, (int)blockDim.x);
    __syncthreads();
    const int *s_rank = (const int *)(s_key + g.W);
    int32_t *p = perm_all + ((int64_t)blockIdx.y * g.E + blockIdx.x) * g.W;
    for (int w = threadIdx.x; w < g.W; w += blockDim.x) p[s_rank[w]] = w;
}

void mtg_launch_split_all(const MtgEnsembleArgs &g, uint32_t iteration0, int steps, int32_t *perm_all, hipStream_t s)
{
    const int threads = g.W > 64 ? 1024 : 256;
    hipLaunchKernelGGL(mtg_split_all_kernel, dim3((unsigned)g.E, (unsigned)steps), dim3(threads),
                       (size_t)g.W * (sizeof(uint64_t) + sizeof(int)), s, g, iteration0, perm_all);
}

void mtg_launch_sampler_spec(const MtgEnsembleArgs &g, int do_accept, uint32_t iteration, const double *new_lnp,
                             const int32_t *status, int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose,
                             uint32_t next_iteration, const MtgPrepArgs &pa, hipStream_t s)
{
    // (the split ranks W keys against each other: W^2 comparisons over the threads; a small ensemble is quicker through
    // the barriers of four waves than of sixteen)
    // (measured, iterations/s with 256 / 1024 threads: W = 32 44.1e3 / 43.0e3, W = 128 18.2e3 / 19.0e3, W = 256 6.8e3 / 7.7e3)
    const int threads = do_propose && g.E <= 64 && g.W > 64 ? 1024 : 256;
    // the ensemble's state in LDS for the length of the kernel (mtg_spec_both_lds): small ensembles whose splits were made
    // beforehand, between two solves of a run (MTG_SAMPLER_LDS=0 in an MTG_MEASURE build: never)
    static const bool lds_wanted = !(mtg_measure_env("MTG_SAMPLER_LDS") && atoi(mtg_measure_env("MTG_SAMPLER_LDS")) == 0);
    const int state_in_lds = lds_wanted && do_accept && do_propose && g.perm_next && g.W / 2 <= 256 && g.W / 2 <= threads &&
                             (int64_t)g.W * g.P <= MTG_SPEC_LDS_DOUBLES ? 1 : 0;
    // keys (8 W), ranks (4 W), accept flags (4 W/2), padding to 8 bytes, then 3 W/2 proposals of P doubles; with the state
    // in LDS also W P coordinates, W log-probabilities, W ints of the next split
    size_t lds = (size_t)(g.W + (3 * g.W / 2 + 1) / 2 + 1) * sizeof(uint64_t) + (size_t)(3 * (g.W / 2)) * g.P * sizeof(double);
    // ... W ints of the next split (padded to doubles), and the iteration's random numbers: 6 H doubles + 3 H ints
    if (state_in_lds) lds += ((size_t)g.W * g.P + g.W + (g.W + 1) / 2 + 6 * (g.W / 2) + (3 * (g.W / 2) + 1) / 2) * sizeof(double);
    hipLaunchKernelGGL(mtg_sampler_spec_kernel, dim3((unsigned)g.E), dim3(threads), lds, s, g, do_accept, iteration,
                       new_lnp, status, clear_counts, chain_row, lnp_chain_row, do_propose, next_iteration, pa, state_in_lds);
}

void mtg_launch_initial_best(int E, int W, int P, const double *coords, const double *lnp, double *best_lnp,
                             double *best_coords, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_initial_best_kernel, dim3((E + 255) / 256), dim3(256), 0, s, E, W, P, coords, lnp,
                       best_lnp, best_coords);
}

// ---------------------------------------------------------------------------
// Conditional mean / variance at the training times
// ---------------------------------------------------------------------------
// celerite.GP.predict(y, return_var=True) as GPModelling.standarized_residuals uses it
// (gpmodelling.py:353-370).  celerite forms the dense N x N cross-covariance for the
// variance; here both come from the semiseparable factorisation in O(N J^2):
//   alpha = K^-1 r              forward + backward solve
//   mu_n  = mean_n + r_n - d_n alpha_n,                d_n = sigma_n^2 + jitter
//   var_n = d_n - d_n^2 (K^-1)_nn
//   (K^-1)_nn = 1/D_n + W_n^T Phi_{n+1} G_{n+1} Phi_{n+1} W_n,
//   G_m = U_m U_m^T / D_m + (I - U_m W_m^T) Phi_{m+1} G_{m+1} Phi_{m+1} (I - W_m U_m^T)
// This is a diagnostic evaluated for one or a few parameter vectors, so it is a plain
// generic-J kernel (one thread per evaluation, per-step generators kept in a global
// workspace), not a tuned template family.
#define MTG_PJ MTG_MAX_J


__global__ void __launch_bounds__(64) mtg_predict_kernel(MtgPredictArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.B) return;
    if (a.status_in[e] != MTG_ST_OK) { a.status[e] = a.status_in[e]; return; }
    const int NR = a.nr0 + 2 * a.sig[e], NC = a.nc0 - a.sig[e], J = NR + 2 * NC;
    const double *cf = a.coef + e;
    const int64_t cs = a.cstride;
    double ar[MTG_PJ], cr[MTG_PJ], ac[MTG_PJ / 2], bc[MTG_PJ / 2], cc[MTG_PJ / 2], dc[MTG_PJ / 2];
    for (int j = 0; j < NR; ++j) { ar[j] = cf[a.lay.ar(j) * cs]; cr[j] = cf[a.lay.cr(j) * cs]; }
    for (int k = 0; k < NC; ++k) {
        ac[k] = cf[a.lay.ac(k) * cs]; bc[k] = cf[a.lay.bc(k) * cs];
        cc[k] = cf[a.lay.cc(k) * cs]; dc[k] = cf[a.lay.dc(k) * cs];
    }
    const double asum = cf[a.lay.asum() * cs], slope = cf[a.lay.mean(0) * cs], icpt = cf[a.lay.mean(1) * cs];
    const double jitter = cf[a.lay.jit() * cs];
    const int64_t lc = a.lc_index ? a.lc_index[e] : 0;
    const int64_t N = a.N;
    const double2 *yv = a.yv + lc * N, *dxt = a.dxt + lc * a.t_stride;
    const int stride = 3 * J + 2;
    double *wk = a.work + e * N * stride;

    // ---- forward sweep: factorisation + z = L^-1 r, generators stored -------------
    double S[MTG_PJ][MTG_PJ], f[MTG_PJ], Wp[MTG_PJ], U[MTG_PJ], V[MTG_PJ], ph[MTG_PJ];
    const double t_first = dxt[0].y;
    for (int i = 0; i < J; ++i) { f[i] = 0.0; Wp[i] = 0.0; for (int j = 0; j < J; ++j) S[i][j] = 0.0; }
    double Dp = 1.0, zp = 0.0;
    bool bad = false;
    for (int64_t n = 0; n < N; ++n) {
        const double dx = dxt[n].x, t = dxt[n].y;
        for (int j = 0; j < NR; ++j) { ph[j] = exp(-cr[j] * dx); U[j] = ar[j]; V[j] = 1.0; }
        for (int k = 0; k < NC; ++k) {
            const double p = exp(-cc[k] * dx);
            // at the elapsed time, as celerite does at the absolute one: a (cos, sin) pair rotated
            // step by step drifts, and an ill-conditioned covariance amplifies the drift
            double sn, cn;
            sincos(dc[k] * (t - t_first), &sn, &cn);
            ph[NR + 2 * k] = ph[NR + 2 * k + 1] = p;
            U[NR + 2 * k] = ac[k] * cn + bc[k] * sn; U[NR + 2 * k + 1] = ac[k] * sn - bc[k] * cn;
            V[NR + 2 * k] = cn; V[NR + 2 * k + 1] = sn;
        }
        for (int i = 0; i < J; ++i) {
            for (int j = 0; j < J; ++j) S[i][j] = ph[i] * ph[j] * (S[i][j] + Dp * Wp[i] * Wp[j]);
            f[i] = ph[i] * (f[i] + Wp[i] * zp);
        }
        double D = yv[n].y + asum, z = yv[n].x - (slope * t + icpt);
        double Wn[MTG_PJ];
        for (int i = 0; i < J; ++i) {
            double q = 0.0;
            for (int j = 0; j < J; ++j) q += S[i][j] * U[j];
            Wn[i] = V[i] - q;
            D -= U[i] * q;
            z -= U[i] * f[i];
        }
        bad = bad || !(D > 0.0);
        double *w = wk + n * stride;
        for (int i = 0; i < J; ++i) { Wn[i] /= D; w[i] = U[i]; w[J + i] = Wn[i]; w[2 * J + i] = ph[i]; Wp[i] = Wn[i]; }
        w[3 * J] = D; w[3 * J + 1] = z;
        Dp = D; zp = z;
    }
    if (bad) { a.status[e] = MTG_ST_NOTPD; return; }

    // ---- backward sweep: alpha = L^-T D^-1 z and diag(K^-1) -------------------------
    double g[MTG_PJ], G[MTG_PJ][MTG_PJ];
    for (int i = 0; i < J; ++i) { g[i] = 0.0; for (int j = 0; j < J; ++j) G[i][j] = 0.0; }
    double Un[MTG_PJ], phn[MTG_PJ];  // generators of sample n + 1
    double xn = 0.0;
    for (int i = 0; i < J; ++i) { Un[i] = 0.0; phn[i] = 0.0; }
    for (int64_t n = N - 1; n >= 0; --n) {
        const double *w = wk + n * stride;
        const double D = w[3 * J], z = w[3 * J + 1];
        // g_n = Phi_{n+1} (g_{n+1} + U_{n+1} x_{n+1});  x_n = z_n / D_n - W_n^T g_n
        double x = z / D;
        for (int i = 0; i < J; ++i) { g[i] = phn[i] * (g[i] + Un[i] * xn); x -= w[J + i] * g[i]; }
        // (K^-1)_nn = 1/D_n + (Phi_{n+1} W_n)^T G_{n+1} (Phi_{n+1} W_n)
        double pw[MTG_PJ], kinv = 1.0 / D;
        for (int i = 0; i < J; ++i) pw[i] = phn[i] * w[J + i];
        for (int i = 0; i < J; ++i) {
            double s = 0.0;
            for (int j = 0; j < J; ++j) s += G[i][j] * pw[j];
            kinv += pw[i] * s;
        }
        const double d = yv[n].y + jitter;
        const double r = yv[n].x - (slope * dxt[n].y + icpt);
        a.mu[e * N + n] = (slope * dxt[n].y + icpt) + r - d * x;
        a.var[e * N + n] = d - d * d * kinv;
        // G_n = U_n U_n^T / D_n + (I - U_n W_n^T) X (I - W_n U_n^T),  X = Phi_{n+1} G_{n+1} Phi_{n+1}
        double X[MTG_PJ][MTG_PJ], XW[MTG_PJ], wxw = 0.0;
        for (int i = 0; i < J; ++i)
            for (int j = 0; j < J; ++j) X[i][j] = phn[i] * phn[j] * G[i][j];
        for (int i = 0; i < J; ++i) {
            double s = 0.0;
            for (int j = 0; j < J; ++j) s += X[i][j] * w[J + j];
            XW[i] = s;
        }
        for (int i = 0; i < J; ++i) wxw += w[J + i] * XW[i];
        for (int i = 0; i < J; ++i)
            for (int j = 0; j < J; ++j)
                G[i][j] = X[i][j] - w[i] * XW[j] - XW[i] * w[j] + w[i] * w[j] * (wxw + 1.0 / D);
        for (int i = 0; i < J; ++i) { Un[i] = w[i]; phn[i] = w[2 * J + i]; }
        xn = x;
    }
    a.status[e] = MTG_ST_OK;
}

// K^-1 applied to M right-hand sides with the factors mtg_predict_kernel left in its workspace
// (celerite.GP.apply_inverse / solver.solve; what GP.predict at NEW times needs: K^-1 r for the
// mean, K^-1 K_*^T for the variance).  One lane per right-hand side; x is [N][M] (row n holds
// sample n of every right-hand side, so the lanes' accesses coalesce) and is overwritten;
// every lane reads the same generator row (broadcast).
//   forward   f_n = Phi_n (f_{n-1} + W_{n-1} z_{n-1}),  z_n = b_n - U_n^T f_n
//   backward  g_n = Phi_{n+1} (g_{n+1} + U_{n+1} x_{n+1}),  x_n = z_n / D_n - W_n^T g_n
__global__ void __launch_bounds__(64)
mtg_apply_inverse_kernel(const double *__restrict__ work, int64_t N, int J, int64_t M, double *__restrict__ x)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int stride = 3 * J + 2;
    double f[MTG_PJ], g[MTG_PJ];
    for (int i = 0; i < J; ++i) { f[i] = 0.0; g[i] = 0.0; }
    double zp = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        const double *w = work + n * stride;
        double z = x[n * M + m];
        for (int i = 0; i < J; ++i) {
            if (n > 0) f[i] = w[2 * J + i] * (f[i] + w[J + i - stride] * zp);
            z -= w[i] * f[i];
        }
        x[n * M + m] = z;
        zp = z;
    }
    double xn = 0.0;
    for (int64_t n = N - 1; n >= 0; --n) {
        const double *w = work + n * stride;
        double v = x[n * M + m] / w[3 * J];
        for (int i = 0; i < J; ++i) {
            if (n < N - 1) g[i] = w[stride + 2 * J + i] * (g[i] + w[stride + i] * xn);
            v -= w[J + i] * g[i];
        }
        x[n * M + m] = v;
        xn = v;
    }
}

void mtg_launch_apply_inverse(const double *work, int64_t N, int J, int64_t M, double *x, hipStream_t s)
{
    hipLaunchKernelGGL(mtg_apply_inverse_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, s, work, N, J, M, x);
}

void mtg_launch_predict(const void *args_void, hipStream_t s)
{
    const MtgPredictArgs &a = *static_cast<const MtgPredictArgs *>(args_void);
    hipLaunchKernelGGL(mtg_predict_kernel, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, s, a);
}
