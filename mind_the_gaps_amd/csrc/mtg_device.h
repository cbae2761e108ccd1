// mtg_device.h -- data structures shared by the host side of the C-ABI and the
// gfx950 kernels.  Written for MI355X (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtg.h"

// Measurement knobs (kernel A/B experiments: scripts/build_variant.sh compiles with -DMTG_MEASURE).  The shipped library
// reads NO environment variable: every switch a user may touch is an mtg_set_* entry of include/mtg.h, and nothing that
// can change the bits of a result (the rank-10 chunk count, ...) hangs on the environment of the process.
#ifdef MTG_MEASURE
#include <stdlib.h>
static inline const char *mtg_measure_env(const char *name) { return getenv(name); }
#else
static inline const char *mtg_measure_env(const char *) { return nullptr; }
#endif

// What celerite.GP(kernel, mean, fit_mean) holds (reference gpmodelling.py:51),
// flattened so that it travels as a kernel argument (scalar loads only).
struct MtgModel {
    int nterms;
    int mean_kind;
    int PF;      // full parameter vector length (kernel + mean)
    int P;       // free parameters = length of one theta row
    int nk;      // kernel parameters (mean parameters start here)
    int nsho;    // SHOTerm count: each may expand to 1 complex or 2 real terms
    int nr0, nc0;        // structure with every SHO under-damped (Q >= 1/2)
    int nr_max, nc_max;  // widest real / complex expansion (workspace layout)
    int last_b0;         // the last complex slot always holds a term with b = 0 (Lorentzian, ComplexTerm3, Cosinus)
    int kinds[MTG_MAX_TERMS];
    int poff[MTG_MAX_TERMS];
    int src[MTG_MAX_PARAMS];  // theta column feeding full[k], or -1 = frozen
    double defaults[MTG_MAX_PARAMS];
    double lo[MTG_MAX_PARAMS];
    double hi[MTG_MAX_PARAMS];
    double extra[MTG_MAX_TERMS];
};

// Coefficient workspace: structure-of-arrays, one column per evaluation so that
// lane e reads/writes coef[slot * stride + e] (coalesced).  Slots:
//   a_real[nr_max] c_real[nr_max] a_comp[nc_max] b_comp[nc_max] c_comp[nc_max]
//   d_comp[nc_max] asum(= sum a + jitter) mean_slope mean_intercept jitter
struct MtgCoefLayout {
    int nr_max, nc_max;
    __host__ __device__ int ar(int j) const { return j; }
    __host__ __device__ int cr(int j) const { return nr_max + j; }
    __host__ __device__ int ac(int k) const { return 2 * nr_max + k; }
    __host__ __device__ int bc(int k) const { return 2 * nr_max + nc_max + k; }
    __host__ __device__ int cc(int k) const { return 2 * nr_max + 2 * nc_max + k; }
    __host__ __device__ int dc(int k) const { return 2 * nr_max + 3 * nc_max + k; }
    __host__ __device__ int asum() const { return 2 * nr_max + 4 * nc_max; }
    __host__ __device__ int mean(int i) const { return 2 * nr_max + 4 * nc_max + 1 + i; }
    __host__ __device__ int jit() const { return 2 * nr_max + 4 * nc_max + 3; }
    __host__ __device__ int nslots() const { return 2 * nr_max + 4 * nc_max + 4; }
};

struct MtgPrepArgs {
    MtgModel model;
    const double *theta;  // [B][P]
    int64_t B;
    int add_prior;
    double *coef;
    int64_t cstride;
    int nsig;       // number of structure signatures (nsho + 1); 1 = identity mapping
    int *lists;     // [nsig][cstride] evaluation indices grouped by signature
    int *counts;    // [nsig]
    double *out;    // [B]
    int32_t *status;  // [B]
    int32_t *sig;   // [B] number of over-damped SHO terms per evaluation, or NULL
    // a walker-sharded ensemble expands and solves only rows [row_lo, row_hi); the others get MTG_ST_REMOTE
    // (skipped by every solver like a prior rejection) until the all-gather brings the owner's result
    int64_t row_lo = 0, row_hi = INT64_MAX;
};

// status of a row another rank evaluates (internal: never visible after the exchange)
#define MTG_ST_REMOTE 4

struct MtgSolveArgs {
    const double *coef;
    int64_t cstride;
    MtgCoefLayout lay;
    const int *list;       // NULL = identity mapping
    const int *count_ptr;  // NULL = B
    int64_t B;
    const int32_t *lc_index;  // NULL = light curve 0
    int32_t *status;
    double *out;
    const double2 *dxt;  // [N] or [L][N] pairs (dx_n, t_n), dx_0 = 0
    const double2 *yv;   // [L][N] pairs (y_n, sigma_n^2 = yerr_n^2)
    int64_t N;
    int64_t t_stride;    // 0 (shared sampling) or N
    uint64_t yv_bytes;   // L * N * 16: may exceed 4 GiB (the sweep windows its 32-bit offsets per wave)
    uint64_t dxt_bytes;
    uint64_t window_bytes;  // reach of one buffer descriptor, <= 2^32 - 1 (smaller only in tests)
    // resident sets beyond the window: evaluations a wave cannot reach from its first light curve are
    // appended to left_list (count in *left_count) and swept by a second launch with solo = 1, one
    // evaluation per wave; NULL when the whole set lies inside one window
    int *left_list;
    int *left_count;
    int solo;
    const double *dxmax;  // [1] max_n dx_n (device): decides table vs OCML sincos per wave
    int mean_kind;
    int has_mean;  // 0: the mean is identically zero for every evaluation of this launch and the model has no jitter term
    // time-parallel path of the rank-10 structures only (mtg_tp_big.h): workspace laid out by
    // mtg_tp_big_plan(J, B, tp_chunks) and the number of chunks per evaluation; NULL / 0 otherwise
    double *tp_ws;
    int tp_chunks;
    int tp_gsize;   // elements per scan group (mtg_tp_big_gsize)
    int tp_direct;  // likelihood without the filter pass (mtg_tp_big.h)
    int tp_nr0, tp_nc0;    // structure with no over-damped SHO term; evaluation ev has (tp_nr0 + 2 sig[ev], tp_nc0 - sig[ev])
    const int32_t *sig;    // [B] over-damped SHO terms per evaluation (mtg_prepare_one), or NULL = 0
    // serial sweep in sorted order (mtg_sort.hip): `list` is the whole batch sorted by (structure, light curve) and
    // this launch takes the seg_k-th segment of it, which starts after seg_counts[0 .. seg_k); NULL: list starts at 0
    const int *seg_counts = nullptr;
    int seg_k = 0;
    // the context's resident copy of the exp2 / (cos, sin) tables (struct MtgMathTablesT<true> of mtg_math.h, made once by
    // mtg_launch_tables): the time-parallel kernels copy it into LDS instead of computing it per workgroup; NULL: computed
    const void *tables = nullptr;
};

// fills `tables` (sizeof(MtgMathTablesT<true>) bytes of device memory) -- mtg_timeparallel.hip
void mtg_launch_tables(void *tables, hipStream_t stream);
size_t mtg_tables_bytes();

// doubles per filtering element (A | b | eta | C | Jm | five likelihood scalars) of the time-parallel kernel,
// rounded up to an odd number: lane l's element starts l * MTG_TP_ELEM doubles into LDS, and an even stride
// puts many lanes on the same bank (rank 3 would be 32 doubles: all 64 lanes on ONE bank, every access
// serialised 64-fold -- measured as 5.6 us per scan round instead of ~1)
#define MTG_TP_ELEM(J) (((J) * (J) + 2 * (J) + (J) * ((J) + 1) + 5) | 1)

struct MtgPredictArgs {
    const double *coef;     // SoA coefficient workspace (mtg_prepare_kernel)
    int64_t cstride;
    MtgCoefLayout lay;
    int nr0, nc0;           // structure with no over-damped SHO term
    const int32_t *sig;     // [B] over-damped SHO terms of each evaluation
    int64_t B;
    const int32_t *lc_index;
    const int32_t *status_in;   // prior verdict from prepare
    const double2 *dxt, *yv;
    int64_t N, t_stride;
    double *work;           // [B][N][3 J + 2]: U, W, phi (per slot), D, z
    double *mu, *var;       // [B][N]
    int32_t *status;        // [B]
};

typedef void (*mtg_solve_launcher)(const MtgSolveArgs &, int64_t nlanes, hipStream_t);
// Table lookup of the compiled <NR, NC> instantiations (mtg_kernels.hip).
mtg_solve_launcher mtg_find_solver(int nr, int nc, int last_b0 = 0);
// Time-parallel (one wave per evaluation) instantiations, J <= 6 (mtg_timeparallel.hip); the
// launcher's second argument is the number of evaluations.
mtg_solve_launcher mtg_find_tp_solver(int nr, int nc);
// Same with 256 chunks (four waves) per evaluation, J <= 5: for batches of at most a few hundred.
mtg_solve_launcher mtg_find_tp_wide_solver(int nr, int nc);
// All nsig = (SHO terms + 1) structures (nr0 + 2k, nc0 - k) of a model in one launch: the launcher
// takes list = base of the per-structure lists, count_ptr = base of the counts; lanes 64 or 256.
mtg_solve_launcher mtg_find_tp_fused_solver(int nr0, int nc0, int nsig, int lanes);
void mtg_launch_prepare(const MtgPrepArgs &, hipStream_t);
// mtg_sort.hip: evaluation indices sorted (stable) by key = structure * L + light curve, rejected rows last
int mtg_sort_key_bits(int64_t L, int nsig);
size_t mtg_sort_temp_bytes(int64_t B, int bits);
hipError_t mtg_launch_sort_by_lightcurve(int64_t B, const int32_t *status, const int32_t *sig, const int32_t *lc, int64_t L,
                                         int nsig, uint32_t *keys_in, uint32_t *keys_out, int *order, void *temp,
                                         size_t temp_bytes, hipStream_t stream);
// every structure of a sorted batch in one launch (mtg_kernels_multi.hip): a.list = the sorted order, a.seg_counts =
// the rows per structure; nullptr when the combination is not compiled
mtg_solve_launcher mtg_find_multi_solver(int nr0, int nc0, int nsig, int last_b0);
// the serial sweep as a producer / consumer pipeline of two waves per 64 rows (mtg_kernels_pipe.hip), for batches that
// leave the one-lane-per-evaluation launch a single wave on half of the SIMDs; nsig = 1: list / count_ptr as for
// mtg_find_solver's kernels, nsig > 1: the sorted order and seg_counts as for mtg_find_multi_solver's
mtg_solve_launcher mtg_find_pipe_solver(int nr0, int nc0, int nsig, int last_b0);
#define MTG_PIPE_ROWS_PER_CU 128
// two models' pipelined sweeps in ONE launch (mtg_kernels_pipe_pair.hip): a workgroup of eight waves runs a quartet of
// each, two waves per SIMD sharing one table set; shapes as for mtg_find_pipe_solver; NULL = this pair is not compiled
struct MtgPipeShapeId { int nr0, nc0, nsig, last_b0; };
typedef void (*mtg_pipe_pair_launcher)(const MtgSolveArgs &a, int64_t nlanes_a, const MtgSolveArgs &b, int64_t nlanes_b, hipStream_t);
mtg_pipe_pair_launcher mtg_find_pipe_pair_solver(const MtgPipeShapeId &a, const MtgPipeShapeId &b);
// does mtg_find_solver(nr, nc, last_b0) return the b = 0 specialisation?
int mtg_solver_uses_b0(int nr, int nc, int last_b0);
void mtg_launch_lc_setup(int64_t N, int64_t L, int64_t t_rows, const double *t, const double *y,
                         const double *yerr, const double *y_offset, double2 *dxt, double2 *yv,
                         double *dxmax, hipStream_t);
// device-resident ensemble sampler (mtg_sampler.hip)
struct MtgEnsembleArgs {
    int E, W, P;
    uint32_t e_base;      // index of ensemble 0 in the caller's global numbering: random counters only (mtg_set_stream_base)
    uint32_t seed_lo, seed_hi;
    double a;             // stretch scale
    int32_t *perm;        // [E][W] red/blue split of the current iteration
    const int32_t *perm_next = nullptr;   // [E][W] split of the iteration about to be proposed, made beforehand (or NULL: made in place)
    double *coords;       // [E][W][P]
    double *lnp;          // [E][W]
    double *factor;       // [E][W/2] (P - 1) ln z of the current proposals ([2][E][W/2] for a speculative iteration)
    int32_t *naccept;     // [E][W]
    double *best_lnp;     // [E]
    double *best_coords;  // [E][P]
    int32_t *n_notpd;
};
// One launch between two solves: accept of half-step (iteration, half) -- proposals in pa.theta, results in
// new_lnp / status -- then the proposals of (next_iteration, next_half) expanded through pa (either part optional).
void mtg_launch_sampler_step(const MtgEnsembleArgs &g, int do_accept, int half, uint32_t iteration, const double *new_lnp,
                             const int32_t *status, int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose,
                             int next_half, uint32_t next_iteration, const MtgPrepArgs &pa, hipStream_t);
// A whole iteration speculatively (mtg_sampler.hip): accept of both half-steps of `iteration` from the 3 E W/2 rows
// in new_lnp / status, then the 3 E W/2 proposals of next_iteration (g.factor holds 2 E W/2 entries).
void mtg_launch_sampler_spec(const MtgEnsembleArgs &g, int do_accept, uint32_t iteration, const double *new_lnp,
                             const int32_t *status, int *clear_counts, double *chain_row, double *lnp_chain_row, int do_propose,
                             uint32_t next_iteration, const MtgPrepArgs &pa, hipStream_t);
// the splits of `steps` iterations from iteration0 on, perm_all[steps][E][W] (mtg_sampler.hip)
void mtg_launch_split_all(const MtgEnsembleArgs &g, uint32_t iteration0, int steps, int32_t *perm_all, hipStream_t);
void mtg_launch_initial_best(int E, int W, int P, const double *coords, const double *lnp, double *best_lnp,
                             double *best_coords, hipStream_t);
// TK95 light-curve simulation (mtg_simulate.hip)
void mtg_launch_tk95_spectrum(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, double dt, const double *coef, int64_t cstride,
                              MtgCoefLayout lay, int nr0, int nc0, const int32_t *sig, const double *psd_table,
                              int64_t psd_rows, uint64_t seed, const double *given, double2 *X, hipStream_t);
void mtg_launch_tk95_segment(int64_t S, int64_t s0, int64_t sbase, int64_t nfft, int64_t seg_len, double dt, double scale,
                             double mean_rate, const double *series, uint64_t seed, const int64_t *given_start, double *out,
                             hipStream_t, int64_t out_first = 0);
// KraftNoise (noise_models.py:81-150) for noise_kind 3: background counts and rate errors per epoch, and the posterior
// median / half-width of the 68 % interval of the source counts for total counts 0 .. K - 1 (< threshold), [N][K]
struct MtgKraftTables {
    const double *bkg_counts = nullptr, *bkg_rate_err = nullptr, *median = nullptr, *half = nullptr;
    int K = 0;
    double threshold = 0.0;
};
// the E13 flux-PDF adjustment of the cut segments (mtg_e13.hip)
size_t mtg_e13_sort_temp_bytes(int64_t S, int64_t n);
void mtg_launch_e13_std(int64_t S, int64_t n, const double *seg, double *stdv, hipStream_t);
void mtg_launch_e13_draw(int64_t S, int64_t s0, int64_t sbase, int64_t n, int kind, double mean, const double *stdv, uint64_t seed,
                         double *x, hipStream_t);
void mtg_launch_e13_iota(int64_t S, int64_t n, int32_t *idx, int32_t *done, hipStream_t);
void mtg_launch_e13_abs(int64_t total, const double2 *spec, double *amp, hipStream_t);
void mtg_launch_e13_phase(int64_t total, const double *amp, double2 *spec, hipStream_t);
hipError_t mtg_launch_e13_sort_values(int64_t S, int64_t n, const double *x, double *keys_out, const int32_t *idx, int32_t *order_tmp,
                                      uint32_t *segment, uint32_t *segment_out, int32_t *order, double *values, void *temp,
                                      size_t temp_bytes, hipStream_t);
hipError_t mtg_launch_e13_rank(int64_t S, int64_t n, const double *keys, double *keys_out, const int32_t *idx, int32_t *order_tmp,
                               uint32_t *segment, uint32_t *segment_out, int32_t *order, void *temp, size_t temp_bytes, hipStream_t);
void mtg_launch_e13_step(int64_t S, int64_t n, const int32_t *order, const double *values, double *x, double *fresh, int32_t *done,
                         int32_t *notconv, int32_t *running, hipStream_t);
void mtg_launch_tk95_observe(int64_t S, int64_t s0, int64_t sbase, int64_t N, int64_t nfft, int64_t seg_len, double dt, double scale,
                             double mean_rate, const double *series, const int32_t *win_lo, const int32_t *win_hi,
                             int noise_kind, double sigma_noise, const double *exposures, int64_t fixed_start,
                             uint64_t seed, const int64_t *given_start, double *clean, double *rates, double *dy, hipStream_t,
                             const MtgKraftTables &kraft = MtgKraftTables());
void mtg_launch_tk95_resident(int64_t L, int64_t N, const double *rates, const double *dy, double2 *yv, double *means,
                              hipStream_t);
// inverse real transform of a length with large prime factors on power-of-two transforms (chirp-z; mtg_simulate.hip)
void mtg_launch_czt_tables(int64_t n, int64_t m, double2 *chirp, double2 *b, hipStream_t);
void mtg_launch_czt_pack(int64_t S, int per, int64_t n, int64_t m, const double2 *X, const double2 *chirp, double2 *a, hipStream_t);
void mtg_launch_czt_mul(int64_t pairs, int64_t m, const double2 *bhat, double2 *a, hipStream_t);
void mtg_launch_czt_unpack(int64_t S, int per, int64_t n, int64_t m, const double2 *c, const double2 *chirp, double *series, hipStream_t);
void mtg_launch_predict(const void *predict_args, hipStream_t);
void mtg_launch_apply_inverse(const double *work, int64_t N, int J, int64_t M, double *x, hipStream_t);
void mtg_launch_math_probe(int64_t n, const double *x, double *e, double *s, double *c, double *rcp,
                           hipStream_t);

// walker-averaged autocorrelation function of a chain (mtg_simulate.hip)
void mtg_launch_acf_center(int64_t n_t, int64_t n2, int64_t S, const double *chain, double *x, double *sumsq, double *scratch,
                           hipStream_t s);
void mtg_launch_acf_power(int64_t nk, int64_t E, int W, int P, const double2 *f, const double *sumsq, double2 *g, hipStream_t s);
void mtg_launch_acf_transpose(int64_t rows, int64_t cols, const double *in, double *out, hipStream_t s);
void mtg_launch_acf_out(int64_t n_t, int64_t n2, int64_t EP, double scale, const double *r, double *rho, hipStream_t s);
