/*
 * include/mtg.h -- C-ABI of libmtg_hip.so, the MI355X (gfx950) engine behind
 * mind_the_gaps' GP log-likelihood hot path.
 *
 * Every entry point replaces a piece of the reference's Python -> celerite
 * interface (paths relative to /root/reference); the reference has no FFI of
 * its own (celerite is a pybind11 dependency), so these are the functions a
 * ctypes binding inside mind_the_gaps/gpmodelling.py would bind
 * (INTEGRATION.md shows that stub).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative MTG_E_* code and never throws; the caller owns every
 * pointer it passes, the library borrows it for the duration of the call;
 * device buffers created by the library belong to the context.  One in-flight
 * call per context; any number of contexts (one per process per GPU is the
 * intended use).  All arithmetic is IEEE float64.
 */
#ifndef MTG_H
#define MTG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTG_API __attribute__((visibility("default")))

/* ---- error codes -------------------------------------------------------- */
#define MTG_OK 0
#define MTG_E_ARG (-1)       /* bad argument (message in mtg_last_error)        */
#define MTG_E_NODEVICE (-2)  /* no usable gfx950 device / HIP runtime failure   */
#define MTG_E_HIP (-3)       /* a HIP call failed                               */
#define MTG_E_STATE (-4)     /* light curves or model not set yet               */
#define MTG_E_UNSUPPORTED (-5) /* kernel structure outside the compiled table  */

/* ---- per-evaluation status (status[b]) ----------------------------------- */
#define MTG_ST_OK 0        /* lnP finite                                        */
#define MTG_ST_PRIOR 1     /* log_prior = -inf: lnP = -inf, likelihood skipped
                              (gpmodelling.py:149-151)                           */
#define MTG_ST_NOTPD 2     /* non-positive pivot D_n: celerite raises
                              LinAlgError here; lnP is written as -inf           */
#define MTG_ST_NONFINITE 3 /* lnL or ln det not finite: celerite returns -inf   */

/* ---- term kinds: celerite coefficient builders, evaluated on the device --- */
#define MTG_TERM_REAL 0       /* celerite.terms.RealTerm(log_a, log_c)                  */
#define MTG_TERM_COMPLEX3 1   /* celerite.terms.ComplexTerm(log_a, log_c, log_d), b = 0 */
#define MTG_TERM_COMPLEX4 2   /* celerite.terms.ComplexTerm(log_a, log_b, log_c, log_d) */
#define MTG_TERM_SHO 3        /* celerite.terms.SHOTerm(log_S0, log_Q, log_omega0)      */
#define MTG_TERM_MATERN32 4   /* celerite.terms.Matern32Term(log_sigma, log_rho; eps)   */
#define MTG_TERM_JITTER 5     /* celerite.terms.JitterTerm(log_sigma)                   */
#define MTG_TERM_DRW 6        /* mind_the_gaps/models/celerite_models.py:55-68          */
#define MTG_TERM_LORENTZIAN 7 /* mind_the_gaps/models/celerite_models.py:7-34           */
#define MTG_TERM_COSINUS 8    /* mind_the_gaps/models/celerite_models.py:36-52          */
#define MTG_TERM_BPL 9        /* mind_the_gaps/models/celerite_models.py:71-90          */
#define MTG_N_TERM_KINDS 10

#define MTG_MEAN_CONSTANT 0   /* celerite.modeling.ConstantModel(value)                 */
#define MTG_MEAN_LINEAR 1     /* mind_the_gaps/models/mean_models.py:24-31 (slope, intercept) */

#define MTG_MAX_TERMS 12
#define MTG_MAX_PARAMS 40     /* full parameter vector: kernel + mean parameters        */
#define MTG_MAX_J 10          /* celerite rank J = n_real + 2 n_complex                 */
#define MTG_MAX_WALKERS 4096  /* walkers per ensemble of the device sampler             */

typedef struct mtg_ctx mtg_ctx;

/* Library / device discovery.  mtg_device_count() < 0 means no HIP runtime. */
MTG_API int mtg_device_count(void);
MTG_API const char *mtg_version(void);
MTG_API int mtg_term_nparams(int kind);

/*
 * Context = one GPU + its light curves + its model + workspaces.
 * Replaces the per-process state the reference keeps in `self.gp`
 * (gpmodelling.py:51-55) and copies into every multiprocessing.Pool worker
 * (gpmodelling.py:245).  Returns NULL when `device` is not a usable GPU.
 */
MTG_API mtg_ctx *mtg_create(int device);
/*
 * A context whose kernels run on one of `parts` (1 to 8) equal, disjoint slices of the device's compute units (CU i is
 * in slice i mod parts; the context's stream carries the CU mask).  Two contexts on the two halves of a GPU run their
 * launches side by side whatever order they arrive in: the null and the alternative model's refits of the Protassov
 * loop (docs/notebooks/tutorial_ppp.ipynb:334-340, one after the other in the reference), each a half-step of
 * ~32 000 rows at 8 GPUs -- one wave per SIMD on its half -- take max(t_null, t_alt) together instead of
 * t_null + t_alt (ppp.protassov_test(concurrent_refits=True)).  Calls on a caller's stream
 * (mtg_loglike_batch_device with a stream of the caller's) run wherever that stream runs.
 */
MTG_API mtg_ctx *mtg_create_on_slice(int device, int part, int parts);
MTG_API void mtg_destroy(mtg_ctx *ctx);
/* Message for the last non-zero return on this context (ctx may be NULL for
 * mtg_create failures). */
MTG_API const char *mtg_last_error(const mtg_ctx *ctx);
/*
 * Testing aid.  The sweep reads samples through buffer descriptors with 32-bit byte
 * offsets; for resident sets beyond 4 GiB every wave places its descriptor at the
 * first light curve its 64 evaluations need and reaches the others within a window
 * of `bytes` (default and maximum 2^32 - 1).  Evaluations further away (never, when
 * a batch is grouped by light curve) are collected and swept one per wave by a
 * second launch.  A small window lets a test exercise that logic on a small set.
 */
MTG_API int mtg_set_window_bytes(mtg_ctx *ctx, uint64_t bytes);

/*
 * celerite.GP.compute(t, yerr) for L light curves at once; the reference calls
 * it with yerr = dy + 1e-12 (gpmodelling.py:54) and the host layer above this
 * ABI does the same.  t: [N] when t_per_lc == 0 (shared sampling, the PPP case
 * of gpmodelling.py:538) or [L][N]; y, yerr: [L][N], host pointers.  The
 * library checks that every t row is sorted (celerite raises ValueError
 * otherwise -> MTG_E_ARG), forms sigma^2 = yerr^2 (celerite squares yerr) and
 * dx_n = t_n - t_{n-1} on the device and keeps everything resident until the
 * next call / mtg_destroy.
 * y_offset: [L] or NULL.  The reference freezes the mean of every light curve at
 * ITS OWN average, ConstantModel(lightcurve.mean) with fit_mean=False
 * (gpmodelling.py:83-87): pass those L values here (they are subtracted from y
 * once, at upload) and give the model a frozen mean of 0.  The set may be as
 * large as the HBM allows (L * N * 32 bytes resident; the upload is staged in
 * blocks); one light curve must stay below 2^28 samples.
 */
MTG_API int mtg_set_lightcurves(mtg_ctx *ctx, int64_t N, int64_t L, const double *t, int t_per_lc,
                                const double *y, const double *yerr, const double *y_offset);
/* Same with DEVICE pointers (light curves born on the GPU); no sortedness
 * check, the data are copied device-to-device on the context's stream. */
MTG_API int mtg_set_lightcurves_device(mtg_ctx *ctx, int64_t N, int64_t L, const double *d_t,
                                       int t_per_lc, const double *d_y, const double *d_yerr,
                                       const double *d_y_offset);

/*
 * The model: what `celerite.GP(kernel, mean=..., fit_mean=...)`
 * (gpmodelling.py:51) holds.  The FULL parameter vector is the kernel
 * parameters of every term in `+` order followed by the mean parameters
 * (1 for MTG_MEAN_CONSTANT, 2 for MTG_MEAN_LINEAR), PF entries in all.
 *   kinds[nterms]      MTG_TERM_* tags
 *   term_extra[nterms] per-term constant (Matern32Term eps), may be NULL
 *   full_values[PF]    current value of every parameter (used for frozen ones)
 *   free_index[P]      position in the full vector of each entry of theta,
 *                      i.e. celerite's unfrozen-parameter order
 *                      (get_parameter_vector, gpmodelling.py:55)
 *   bounds[PF][2]      (lo, hi) of every parameter, +-inf for None
 *                      (get_parameter_bounds(include_frozen=True))
 */
MTG_API int mtg_set_model(mtg_ctx *ctx, int nterms, const int32_t *kinds, const double *term_extra,
                          int mean_kind, int PF, const double *full_values, int P,
                          const int32_t *free_index, const double *bounds);

/*
 * GPModelling._log_probability (gpmodelling.py:127-152; add_prior = 1) and
 * -GPModelling._neg_log_like (gpmodelling.py:155-169; add_prior = 0) for B
 * parameter vectors in one launch.  theta: [B][P] host; lc_index: [B] light
 * curve of each row (NULL = all 0); out: [B] lnP; status: [B] MTG_ST_*.
 * Replaces emcee's `pool.map(self._log_probability, coords)` (gpmodelling.py:247-248)
 * and scipy's serial finite-difference calls (gpmodelling.py:192).
 */
MTG_API int mtg_loglike_batch(mtg_ctx *ctx, int64_t B, const double *theta, const int32_t *lc_index,
                              int add_prior, double *out, int32_t *status);
/*
 * Same with DEVICE pointers, enqueued on `stream` (a hipStream_t passed as void*) without
 * synchronising: inputs stay resident in HBM, results are valid once the stream reaches this point.
 * NULL is what it is everywhere in HIP: the (legacy) default stream -- PyTorch's
 * torch.cuda.current_stream().cuda_stream is 0 for its default stream, and work the caller has
 * queued there (the producer of d_theta, the consumer of d_out) is ordered with the launch.
 * MTG_STREAM_CONTEXT selects the context's own non-blocking stream, which is NOT ordered against the
 * default stream as far as the CALLER's work goes.  The library's own calls are always ordered one after
 * the other whatever streams they run on (they share the context's workspaces): a call that follows one on
 * a different stream waits for it through an event, and mtg_synchronize waits for the last call on a
 * caller's stream as well as for the context's stream.
 *
 * Order of evaluation: the throughput kernel gives every evaluation a lane and each lane reads its own
 * light curve, so it wants the lanes of a wave on one or two light curves.  Large batches are therefore
 * swept in the order of a stable sort by (structure, light curve) of the library's own making, whatever
 * order the caller's rows have (results are per row and do not depend on it); mtg_set_sort: 0 = keep the
 * caller's order, 1 = always sort, 2 (default) = sort unless the host-pointer entry point sees that the
 * rows are grouped by light curve already.  A model whose rows fall into more than one structure (an SHO
 * term on either side of Q = 1/2) is always swept in sorted order under 1 and 2: the per-structure lists
 * are filled with atomics, and only the sort makes the assignment of rows to waves -- on which the last
 * bits of a row can depend, see MTG_TRIG_FAST_MAX -- the same in every run, as a seeded chain needs it.
 */
#define MTG_STREAM_CONTEXT ((void *)(intptr_t)-1)
MTG_API int mtg_loglike_batch_device(mtg_ctx *ctx, int64_t B, const double *d_theta,
                                     const int32_t *d_lc_index, int add_prior, double *d_out,
                                     int32_t *d_status, void *stream);

/*
 * celerite solver entry `compute(jitter, a_real, c_real, a_comp, b_comp, c_comp,
 * d_comp, ...)` + log_likelihood for user-defined Python terms whose
 * coefficients are evaluated on the host (celerite_models.py:9,17 override
 * points).  All evaluations share one structure (jr real, jc complex terms).
 * a_real..d_comp: [B][jr] / [B][jc] host, jitter: [B] (NULL = 0),
 * mean_params: [B][1 or 2] (NULL = 0).
 */
MTG_API int mtg_loglike_coeffs(mtg_ctx *ctx, int64_t B, int jr, int jc, const double *a_real,
                               const double *c_real, const double *a_comp, const double *b_comp,
                               const double *c_comp, const double *d_comp, const double *jitter,
                               int mean_kind, const double *mean_params, const int32_t *lc_index,
                               double *out, int32_t *status);

/*
 * Kernel choice for mtg_loglike_batch[_device] / mtg_ensemble_*: 0 = always one lane per
 * evaluation (throughput kernel), 1 = one wave per evaluation, parallel in time, whenever the
 * structure has that kernel (J <= 6, and the J = 10 structures of five SHO terms), 2 = automatic
 * (default): time-parallel for light curves of at least 256 samples and batches of at most 1024
 * evaluations (J <= 6: 4-10x faster there); J = 10: at least 1024 samples and at most 8192 evaluations
 * (the light curve is cut into as many chunks as fill the GPU: 40-100x faster than the serial sweep for
 * the 32-256 evaluations of an ensemble half-step).  3 = as 1, restricted to the one-wave-per-evaluation kernel
 * (J <= 6): under 1 and 2 the number of rows in a batch picks between kernels whose sums differ in the last bits;
 * under 0 (one-lane sweep and its pipelined form, bit-identical to each other) and under 3 a row's result does not
 * depend on what else is in the batch -- what a job that splits its rows over several GPUs needs to reproduce the
 * one-GPU result exactly (ppp.protassov_test(reproducible=True)).
 */
MTG_API int mtg_set_time_parallel(mtg_ctx *ctx, int mode);
/*
 * Index of this context's first resident ensemble (mtg_ensemble_*) and first simulated series
 * (mtg_simulate_tk95) in the caller's global numbering, default 0.  The counter-based random streams are keyed by
 * (seed, ..., first_index + local index): a job that splits E ensembles or S series over several contexts (GPUs)
 * with the same seed draws, for every one of them, exactly the numbers ONE context holding them all would draw --
 * the result of the reference's loop over simulated light curves (docs/notebooks/tutorial_ppp.ipynb:326-343,
 * gpmodelling.py:505-513) no longer depends on how many GPUs share it.  Read at mtg_ensemble_init / at every
 * mtg_simulate_tk95; enters random counters only, never an address.
 */
MTG_API int mtg_set_stream_base(mtg_ctx *ctx, int64_t first_index);
/*
 * mtg_simulate_tk95 transforms a grid length with large prime factors (the reference's grid arithmetic, simulator.py:259-262,
 * gives BASELINE configs[3] 1 087 853 = 13^2 x 41 x 157 points) on power-of-two transforms by hand (chirp-z), two real
 * series per complex transform: hipFFT's own plan for such a length takes 0.9 s to build.  on = 0: one series per
 * transform, so that a series' values do not depend on which other series the call holds -- what a set simulated in
 * blocks over several GPUs needs to be the set of one call, bit for bit (with mtg_set_stream_base); default 1.
 */
MTG_API int mtg_set_simulate_pairs(mtg_ctx *ctx, int on);
/*
 * The flux PDF of the light curves mtg_simulate_tk95 makes (Simulator(..., pdf=...), simulator.py:149-150): kind 0 (default)
 * Gaussian = the TK95 series as it is; 1 lognormal, 2 uniform = every cut segment goes through the amplitude / rank
 * adjustment of Emmanoulopoulos et al. 2013 as the reference runs it (simulator.py:65-140 E13Simulator; the distributions
 * of stats.py:116-146 with the simulator's mean and the segment's standard deviation), at most max_iter + 1 iterations
 * (the reference's `max_iter`), ON THE DEVICE (csrc/mtg_e13.hip: batched hipFFT transforms, one segmented radix sort per
 * iteration) before the segment is averaged into the epochs; everything after it -- noise, make_resident -- as for the
 * Gaussian case, so a posterior-predictive run with a non-Gaussian PDF never leaves the GPU.  The white series is drawn
 * from the context's Philox stream (keyed by seed and global series index), or taken from the caller for the NEXT call:
 * mtg_set_simulate_pdf_draws(ctx, S, n, draws[S][n]) with n = seg_len (tests hold the device against the host with it).
 * mtg_simulate_pdf_report: segments of the last call that used all their iterations without converging (the reference
 * warns), and the largest iteration count.
 */
MTG_API int mtg_set_simulate_pdf(mtg_ctx *ctx, int kind, int max_iter);
/*
 * KraftNoise (noise_models.py:81-150) for mtg_simulate_tk95(noise_kind = 3), on the device: total counts ~ Poisson(rate x
 * exposure + bkg_counts[n]); net rate = (total - bkg) / exposure, dy = sqrt((sqrt(total) / exposure)^2 + bkg_rate_err[n]^2);
 * epochs with total < threshold (the reference's kraft_counts, 15) take the median of the Kraft, Burrows & Nousek (1991)
 * posterior of the source counts and half the width of its 68 % interval instead -- functions of (total, background)
 * alone, tabulated by the caller for total = 0 .. K - 1 and every epoch: median[N][K], half[N][K] (counts; K >= threshold).
 * N must be the number of epochs of the resident sampling; N = 0 forgets the tables.
 */
MTG_API int mtg_set_simulate_kraft(mtg_ctx *ctx, int64_t N, int K, double threshold, const double *bkg_counts, const double *bkg_rate_err,
                                   const double *median, const double *half);
MTG_API int mtg_set_simulate_pdf_draws(mtg_ctx *ctx, int64_t S, int64_t n, const double *draws);
MTG_API int mtg_simulate_pdf_report(const mtg_ctx *ctx, int64_t *not_converged, int *iterations);
/* Which transform mtg_simulate_tk95 takes: 0 (default) = by grid length (hipFFT's plan for lengths of radices 2-13, the
 * hand-written chirp-z otherwise), 1 = always hipFFT's plan, 2 = always chirp-z (tests compare the two). */
MTG_API int mtg_set_simulate_transform(mtg_ctx *ctx, int mode);
/*
 * The reference's simulator draws from numpy's GLOBAL generator -- get_fft (simulator.py:468-501): real, im =
 * np.random.normal(0, size=(2, N // 2 + 1)); cut_random_segment (simulator.py:536-539): np.random.uniform -- so a user who
 * calls np.random.seed() gets the same light curve every time.  For that user the host draws those numbers in the
 * reference's order and hands them to the NEXT mtg_simulate_tk95 of this context in place of its own Philox streams:
 * normals [S][2][nk] (per series the row of real parts, then the row of imaginary parts, nk = nfft / 2 + 1; entry 0 of
 * either is unused as in the reference) and starts [S], the index of the first fine-grid sample of each series' cut.
 * The spectrum scaling, the inverse transform, the cut, the window averages stay on the device.  Consumed by one call
 * (which must have the same S and nfft; every start + seg_len <= nfft); S = 0 clears.  `Simulator(..., stream="numpy")`.
 */
MTG_API int mtg_set_simulate_draws(mtg_ctx *ctx, int64_t S, int64_t nk, const double *normals, const int64_t *starts);
MTG_API int mtg_set_sort(mtg_ctx *ctx, int mode);
/*
 * The serial sweep as a two-wave pipeline (csrc/mtg_kernels_pipe.hip): 0 = never, 1 = whenever the model has the
 * kernel (ranks 3-6 with a complex term; light curves of at least 64 samples; measurements and tests), 2 (default) =
 * for batches beyond the time-parallel kernels' range that still fit one workgroup of 128 rows per compute unit
 * (32 768 rows on an MI355X), where the one-lane-per-evaluation launch would leave a lone wave on half of the SIMDs:
 * e.g. one GPU's share of the Protassov refits at 8 GPUs -- 250 light curves x 128 walkers per half-step of the
 * reference's loop (docs/notebooks/tutorial_ppp.ipynb:326-343, gpmodelling.py:247-248).  A row's result is the
 * same to the last bit as from the one-lane kernel.
 */
MTG_API int mtg_set_pipeline(mtg_ctx *ctx, int mode);
/*
 * Two contexts on one device whose pipelined half-steps go out in ONE launch (csrc/mtg_kernels_pipe_pair.hip): the null
 * and the alternative kernel of the posterior-predictive test, which the reference fits to every simulated light curve
 * one after the other (docs/notebooks/tutorial_ppp.ipynb:326-343) and which here advance side by side, each context
 * driven by a host thread of its own.  A pipelined sweep takes a whole compute unit (its tables and rings fill the LDS),
 * so two contexts' launches alternate on the compute units and leave every SIMD one wave that issues ~60 % of the time;
 * paired, a workgroup of eight waves runs 128 rows of each model on one table set, two waves per SIMD.  Each call that
 * would dispatch mtg_pipe_kernel meets its partner's (on the host, bounded wait: mtg_set_pair_patience, default 250 ms);
 * a partner that does not come in time means "alone this time" (the next wait is half as long); four misses in a row,
 * another sampling or a pair of model shapes without a compiled kernel break the pair for good and both go on alone.
 * Results are those of the unpaired kernels to the last bit.  mtg_pair_stats: launches that were shared / made alone
 * since pairing, and whether the pair is broken.  mtg_destroy unpairs; unpairing (or destroying) one context while the
 * partner's thread is inside a call is safe: the rendezvous is reference-counted and the partner launches alone.
 */
MTG_API int mtg_pair_contexts(mtg_ctx *a, mtg_ctx *b);
MTG_API int mtg_unpair_contexts(mtg_ctx *ctx);
/* Longest host-side wait (ms >= 1) of a paired context for its partner's half-step; the context must be paired. */
MTG_API int mtg_set_pair_patience(mtg_ctx *ctx, int milliseconds);
MTG_API int mtg_pair_stats(const mtg_ctx *ctx, int64_t *paired_launches, int64_t *solo_launches, int *broken);
/*
 * Speculative iterations of mtg_ensemble_run (default 1 = where they pay, 0 = never).  The time-parallel solve of a
 * small batch takes as long for three times the rows -- most of the GPU is idle -- and the second half-step of a
 * stretch-move iteration depends on the first only through each partner's coordinates: where it is, or where its own
 * proposal would put it.  Both candidates are then evaluated beside the first half-step's proposals, one solve and one
 * sampler launch per iteration instead of two and two (1.9 x the iterations per second for BASELINE configs[0] and
 * [1]).  Taken when all 3 E W/2 rows get a workgroup of their own in one occupancy round (light curves of >= 4096
 * samples: <= 512 rows up to rank 5, <= 256 at rank 6; <= 1024 rows below), J <= 6, not walker-sharded.  The random numbers, hence the chain, are those of
 * the sequential form -- to the last bit where both forms run the same kernel (the batch size picks it: e.g. rank 5,
 * 256 walkers: four waves per evaluation for the 128 rows of a half-step, two for the 384 of a speculative iteration).
 * Mode 2 is mode 1 with every iteration's split ranked inside its sampler launch instead of all of a run's splits in
 * one launch up front (what mode 1 itself does once steps * E * W * 4 bytes pass 64 MiB): same chain, bit for bit.
 */
MTG_API int mtg_set_speculation(mtg_ctx *ctx, int mode);
/* Name of the kernel the last batch was dispatched to, e.g. "mtg_solve_kernel<1,2,1>" (first structure of the
 * model when several were launched); "" before the first call.  For measurements (bench.py roofline.kernel). */
MTG_API const char *mtg_last_solver(const mtg_ctx *ctx);
/*
 * J = 10 time-parallel path: enabled != 0 (default) takes the likelihood from the chunk-composition pass
 * and the scan alone and sends only evaluations whose terms cancel badly (or meet a non-positive pivot)
 * through the filter pass; 0 runs the filter pass for every evaluation.  Same results to ~1e-13; a
 * switch for tests and measurements.
 */
MTG_API int mtg_set_tp_direct(mtg_ctx *ctx, int enabled);
/* Block until everything enqueued on the context's stream has finished, and the last
 * mtg_loglike_batch_device call made on a caller's stream. */
MTG_API int mtg_synchronize(mtg_ctx *ctx);
/* Device time (ms, HIP events on the launch stream) of the last
 * mtg_loglike_batch / mtg_loglike_coeffs call: kernels only, no copies. */
MTG_API double mtg_last_kernel_ms(const mtg_ctx *ctx);
/*
 * Per-call kernel timing for the next `capacity` mtg_loglike_batch[_device]
 * calls: HIP events are recorded on the launch stream before the prepare
 * kernel, between prepare and the solve kernel(s), and after them.
 * mtg_profile_read waits for the recorded calls, writes their durations (ms)
 * and returns how many were recorded (<= capacity), ending the session.
 */
MTG_API int mtg_profile_begin(mtg_ctx *ctx, int capacity);
MTG_API int mtg_profile_read(mtg_ctx *ctx, int capacity, double *prepare_ms, double *solve_ms);
/*
 * Device-resident lock-step ensembles: emcee's stretch move (a = 2, random red/blue
 * split) as `derive_posteriors` drives it (gpmodelling.py:245-248), for E independent
 * ensembles of W walkers with state, random numbers (Philox4x32-10 keyed by `seed`)
 * and accept/reject on the GPU; one iteration = 2 x W/2 evaluations per ensemble, no
 * host synchronisation inside mtg_ensemble_run.
 *   coords          [E][W][P] initial walkers (host), e.g. from spread_walkers; W even,
 *                   2 P <= W <= MTG_MAX_WALKERS
 *   lc_of_ensemble  [E] light curve of every ensemble; NULL = ensemble e -> light
 *                   curve e (E == L), or all -> 0 when one light curve is resident
 *   chain/lnp_chain [steps][E][W][P] / [steps][E][W] host buffers or NULL (emcee's
 *                   get_chain / get_log_prob layout per ensemble)
 * mtg_ensemble_get copies out the current state, the best sample seen per ensemble
 * (max_loglikelihood / max_parameters, gpmodelling.py:431-443), acceptance counts,
 * the iteration count and the number of proposals whose covariance was not positive
 * definite (celerite would have raised LinAlgError; they are rejected here).
 */
MTG_API int mtg_ensemble_init(mtg_ctx *ctx, int64_t E, int W, uint64_t seed, const double *coords,
                              const int32_t *lc_of_ensemble);
MTG_API int mtg_ensemble_run(mtg_ctx *ctx, int steps, double *chain, double *lnp_chain);
/* Resume: after mtg_ensemble_init with the coordinates and the seed of a saved state (mtg_ensemble_get), put the
 * iteration counter -- every random number is a function of (seed, iteration, ...) --, the saved
 * log-probabilities lnp [E][W] and, optionally, the acceptance counts and the running best back;
 * mtg_ensemble_run then continues the chain bit for bit.  lnp = NULL keeps the values mtg_ensemble_init has
 * just computed for the saved coordinates: the same numbers to rounding, but from ONE batch of E W rows where
 * the run evaluated half-ensembles -- another row count may mean another kernel and summation order, and a
 * last-bit difference can flip an accept decision; pass the saved values for a bit-for-bit continuation. */
MTG_API int mtg_ensemble_restore(mtg_ctx *ctx, int64_t iteration, const double *lnp, const int32_t *naccept,
                                 const double *best_lnp, const double *best_coords);
MTG_API int mtg_ensemble_get(mtg_ctx *ctx, double *coords, double *lnp, double *best_lnp, double *best_coords,
                             int32_t *naccept, int64_t *iteration, int32_t *n_notpd);

/*
 * Walker-averaged, normalised autocorrelation function of a chain -- the inner loop of the convergence
 * check of mind_the_gaps/gpmodelling.py:260-272 (emcee.autocorr.integrated_time: function_1d of every
 * walker and dimension by zero-padded FFT, each normalised by its lag-0 value, averaged over the walkers).
 * chain: host [n_t][E][W][P] for E independent ensembles (E = 1: one chain); rho: host out [n_t][E][P].  On
 * the device: E * W * P batched forward transforms, power spectra normalised and averaged over each ensemble's
 * walkers, E * P inverse transforms (hipFFT, the only library call).  A walker that never moved yields NaN in
 * its dimension, as emcee does.
 */
MTG_API int mtg_chain_autocorr(mtg_ctx *ctx, int64_t n_t, int64_t E, int W, int P, const double *chain, double *rho);
/* Pairs of hipFFT plans mtg_chain_autocorr has built on this context so far (it keeps the four most recently used
 * shapes): a call on a cached shape leaves the count alone. */
MTG_API int64_t mtg_chain_autocorr_plans_built(const mtg_ctx *ctx);
/* hipFFT's one-time start-up (~1.4 s: the first plan of a process) paid now, on the context's device; safe to call
 * from a helper thread (HIP's current device is per thread: the call selects ctx's; NULL = the thread's current). */
MTG_API int mtg_fft_warmup(mtg_ctx *ctx);
/* The inverse-transform plan mtg_simulate_tk95 needs for series of nfft points, made now and kept by the context
 * (one plan per length, whatever the number of simulations).  A Bluestein plan -- 1 087 853 points in BASELINE
 * configs[3] -- takes 0.9 s to build: from a helper thread, while the context samples the observed light curve, it
 * costs nothing.  Safe beside any other call on the context; mtg_simulate_tk95 waits for it. */
MTG_API int mtg_simulate_plan(mtg_ctx *ctx, int64_t nfft);

/*
 * Walker sharding of the resident ensembles across the GPUs of a job (one process per GPU; the
 * reference's counterpart is the multiprocessing.Pool.map over the half-ensemble of
 * mind_the_gaps/gpmodelling.py:245-248).  Every process calls mtg_ensemble_init with the SAME
 * coordinates and seed, then one of the two calls below with its rank: from then on
 * mtg_ensemble_run evaluates only rows [rank * chunk, (rank + 1) * chunk), chunk = ceil(E * W/2 /
 * world), of every half-step's proposals and gets the others' log-probabilities (8 bytes per
 * walker and a status word) before the accept step, which every rank then takes identically:
 * all ranks hold the same chain.  A row is evaluated by the same kernel whatever the rank, but the
 * kernel is chosen by the number of rows a rank evaluates: chains agree bit for bit with the
 * one-process chain when that choice is the same (always with mtg_set_time_parallel(ctx, 0)).
 *
 *  mtg_ensemble_shard_rccl   one ncclAllGather of doubles and one of int32 per half-step, grouped,
 *                            on the context's stream: no host synchronisation in the run.  id128:
 *                            the 128-byte ncclUniqueId made by ONE process with
 *                            mtg_rccl_unique_id and handed to the others by the caller (any
 *                            channel: torch.distributed.broadcast, MPI, a file).  Collective:
 *                            returns when every rank of `world` has called it.  librccl.so.1 is
 *                            looked up at run time, the copy already loaded in the process first
 *                            (mtg_rccl_load(path) names one explicitly before anything else loads it).
 *  mtg_ensemble_shard_host   the exchange is a callback of the caller (another transport: gloo,
 *                            MPI ...; or processes that share one GPU, which RCCL refuses): after
 *                            each half-step's solve the library hands it host arrays lnp[count],
 *                            status[count] with rows [lo, hi) filled in; it must fill in all others
 *                            (the chunk layout above) and return 0.  Synchronises the stream twice per
 *                            half-step.
 *  mtg_ensemble_unshard      back to every row on this process.  mtg_ensemble_init also resets it.
 */
typedef int (*mtg_exchange_fn)(void *user, double *lnp, int32_t *status, int64_t count, int64_t lo, int64_t hi);
MTG_API int mtg_rccl_load(const char *path);
MTG_API int mtg_rccl_unique_id(void *id128);
MTG_API int mtg_ensemble_shard_rccl(mtg_ctx *ctx, const void *id128, int rank, int world);
MTG_API int mtg_ensemble_shard_host(mtg_ctx *ctx, int rank, int world, mtg_exchange_fn fn, void *user);
MTG_API int mtg_ensemble_unshard(mtg_ctx *ctx);
/* How the resident ensembles are sharded: kind 0 none / 1 RCCL / 2 host callback, this process's rank and the
 * world it was given, and -- RCCL -- the communicator's own idea of its size (ncclCommCount).  Any pointer may
 * be NULL. */
MTG_API int mtg_ensemble_shard_info(const mtg_ctx *ctx, int *kind, int *rank, int *world, int *comm_ranks);
/* Time the RCCL exchange: HIP events on the launch stream around the first `capacity` all-gather pairs of the
 * following mtg_ensemble_run calls; mtg_ensemble_shard_profile_read waits for them, writes their durations (ms)
 * and returns how many were recorded, ending the session.  (What an exchange costs on the stream -- including
 * the wait for the slowest rank's solve -- not the wire time alone.) */
MTG_API int mtg_ensemble_shard_profile(mtg_ctx *ctx, int capacity);
MTG_API int mtg_ensemble_shard_profile_read(mtg_ctx *ctx, int capacity, double *exchange_ms);

/*
 * Posterior-predictive light-curve simulation, the step before the hot path in the
 * Protassov loop: GPModelling.generate_from_posteriors (gpmodelling.py:478-539) ->
 * Simulator.generate_lightcurve + add_noise (simulator.py:300-420, get_fft :468-501),
 * Gaussian flux PDF (Timmer & Koenig 1995), for S posterior samples theta[S][P] of the
 * context's model on the context's (shared) sampling of N epochs.
 *   nfft, sim_dt   length and step of the fine regular grid (len(sim_timestamps), sim_dt)
 *   seg_len        fine samples in the randomly cut segment (sim_duration / sim_dt)
 *   win_lo/win_hi  [N] fine-sample ranges [lo, hi) of the segment averaged into every
 *                  epoch (the `strategy` windows of simulator.py:266-267, 340-367)
 *   noise_kind     0 none, 1 Gaussian(sigma_noise), 2 Poisson over exposures[N], 3 Kraft (Poisson with background and the
 *                  Bayesian estimates of the faint epochs: mtg_set_simulate_kraft first)
 *   clean          [S][N] noise-free rates or NULL; rates, dy [S][N]: noisy rates and
 *                  their 1-sigma errors (host buffers); lc_means [S] or NULL
 *   make_resident  != 0: the simulated set becomes the context's light curves (frozen mean
 *                  = each curve's average, yerr = dy + 1e-12), ready for
 *                  mtg_loglike_batch / mtg_ensemble_* with no host round trip
 *   psd_table      NULL: the PSD is celerite's Term.get_psd of the context's model at theta[S][P].  Otherwise
 *                  (any callable PSD, simulator.py:149,272-280) the spectrum tabulated by the caller at the
 *                  angular frequencies 2 pi k / (nfft sim_dt), k = 0 .. nfft/2: [psd_rows][nfft/2 + 1] with
 *                  psd_rows = 1 (shared) or S; theta and the model are then not used
 *   segments       NULL, or [S][seg_len]: the cut segments themselves as rates on the fine grid (what the
 *                  reference hands to its E13 flux-PDF adjustment before down-sampling)
 * The inverse FFTs are one batched hipFFT plan; random numbers are Philox4x32-10 keyed by `seed`.
 */
MTG_API int mtg_simulate_tk95(mtg_ctx *ctx, int64_t S, const double *theta, const double *psd_table, int64_t psd_rows,
                              uint64_t seed, int64_t nfft, double sim_dt, double mean_rate, int64_t seg_len,
                              const int32_t *win_lo, const int32_t *win_hi, int noise_kind, double sigma_noise,
                              const double *exposures, double *clean, double *rates, double *dy, double *lc_means,
                              double *segments, int make_resident);
/*
 * Test entry: the down-sampling step of mtg_simulate_tk95 alone on caller-provided fine-grid series
 * [S][nfft] with the segment starting at fine sample `start`: rates[S][N] = plain average of
 * series[s][start + lo .. start + hi) per epoch (simulator.py:340-367).
 */
MTG_API int mtg_tk95_observe_series(mtg_ctx *ctx, int64_t S, int64_t nfft, int64_t seg_len, int64_t start,
                                    const double *series, const int32_t *win_lo, const int32_t *win_hi, double *rates);

/*
 * celerite.GP.predict(y, return_var=True) at the training times, as
 * GPModelling.standarized_residuals calls it (gpmodelling.py:366): conditional mean
 * mu[b][n] (WITHOUT the per-light-curve y_offset, which the caller adds back) and
 * variance var[b][n] (without jitter; the caller adds kernel.jitter when
 * include_noise) for B parameter vectors, in O(N J^2) per vector through the
 * factorisation instead of celerite's dense N x N cross-covariance.  Rows outside
 * the prior or with a non positive-definite covariance get their status and no
 * output.
 */
MTG_API int mtg_predict(mtg_ctx *ctx, int64_t B, const double *theta, const int32_t *lc_index, double *mu,
                        double *var, int32_t *status);

/*
 * celerite.GP.apply_inverse(y) / CholeskySolver.solve: x <- K^-1 x for M right-hand sides,
 * x[N][M] (row n = sample n of every right-hand side), K the covariance of light curve
 * `lc_index` at parameter vector `theta`, in O(N J^2 + N J M) from the same factorisation
 * mtg_predict uses.  It is what celerite's predict at NEW times is made of (mean
 * K_* K^-1 r, covariance K_** - K_* K^-1 K_*^T); the host side assembles those.
 * *status: MTG_ST_* of the parameter vector (nothing is written to x unless MTG_ST_OK).
 */
MTG_API int mtg_apply_inverse(mtg_ctx *ctx, const double *theta, int32_t lc_index, int64_t M, double *x,
                              int32_t *status);

/*
 * Accuracy probe of the device elementary functions the recurrence uses
 * (tests only): exp_neg[i] = exp(-x[i]), sin/cos(x[i]), rcp_x[i] = 1 / x[i] for
 * n host values x >= 0.
 */
MTG_API int mtg_math_probe(mtg_ctx *ctx, int64_t n, const double *x, double *exp_neg, double *sin_x,
                           double *cos_x, double *rcp_x);
/* 1 if (jr, jc) has a compiled kernel. */
MTG_API int mtg_structure_supported(int jr, int jc);

#ifdef __cplusplus
}
#endif
#endif /* MTG_H */
