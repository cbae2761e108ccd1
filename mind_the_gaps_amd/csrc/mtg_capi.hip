// mtg_capi.hip -- host side of the C-ABI declared in include/mtg.h.
// One context = one MI355X + resident light curves + model + workspaces.
#include "mtg_device.h"
#include "mtg_tp_scan.h"
#include "mtg_trace.h"

#include <dlfcn.h>
#include <hipfft/hipfft.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

// hipFFT plans are made from helper threads too (mtg_fft_warmup, mtg_simulate_plan) while another thread may be making
// or destroying one for a convergence check: plan creation and destruction go through one process-wide lock
// (executions do not).
std::mutex g_fft_plan_mu;

struct DevBuf {  // owning device allocation; locals free themselves on every return path
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4;  // grow with slack: batches vary between calls
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

struct mtg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::string err;

    // light curves (resident)
    int64_t N = 0, L = 0;
    int t_per_lc = 0;
    uint64_t window_bytes = 0xffffffffull;  // reach of one buffer descriptor of the sweep (mtg_set_window_bytes)
    DevBuf dxt, yv, dxmax;  // interleaved (dx, t) and (y, sigma^2) pairs
    DevBuf t_tmp, y_tmp, dy_tmp, off_tmp;  // upload staging

    // model
    bool has_model = false;
    MtgModel model;

    // workspaces
    DevBuf coef, lists, counts, tp_ws, sig;
    DevBuf tables;            // resident exp2 / (cos, sin) tables for the time-parallel kernels (mtg_launch_tables)
    bool tables_ready = false;
    int64_t cstride = 0;
    int nsig_ws = 1;  // signature lists the workspace was laid out for
    int bank = 0;     // which of the two banks of structure lists / counters the next expansion and solve use (the device
                      // sampler alternates: the proposals of a half-step are expanded while the other bank is cleared)
    // staging for the host-pointer entry points
    DevBuf theta, lc, out, status;

    // small batches: one wave per evaluation, parallel in time (0 never, 1 whenever compiled, 2 auto)
    int tp_mode = 2;   // (3: as 1, but the one-wave-per-evaluation kernel only -- results independent of the batch size)
    int tp_direct = 1;  // rank-10 time-parallel path: likelihood without the filter pass (mtg_set_tp_direct)
    int pipe_mode = 2;  // two-wave pipeline of the serial sweep (mtg_set_pipeline): 0 never, 1 whenever compiled, 2 auto
    int cus = 0;        // compute units of the device

    // device-resident ensembles (mtg_ensemble_*)
    int64_t ens_E = 0;
    int ens_W = 0, ens_P = 0;
    uint64_t ens_seed = 0;
    int64_t stream_base = 0;      // mtg_set_stream_base: global index of the context's first ensemble / simulated series
    int64_t ens_base = 0;         // ... as it was when the resident ensembles were made
    uint32_t ens_iteration = 0;
    int64_t ens_L = 0, ens_N = 0;  // shape of the resident set the ensembles index into
    DevBuf ens_coords, ens_lnp, ens_perm, ens_q, ens_factor, ens_new, ens_st, ens_lc_full, ens_lc_half,
        ens_naccept, ens_best_lnp, ens_best_coords, ens_notpd, ens_chain, ens_lnp_chain;
    DevBuf ens_lc_spec;                 // light-curve index of the 3 E W/2 rows of a speculative iteration
    DevBuf ens_perm_all;                // the splits of a whole speculative run, made before it ([steps][E][W])
    // walker sharding (mtg_ensemble_shard_*): this rank evaluates rows [shard_lo, shard_hi) of every
    // half-step's proposals; the exchange brings everybody's log-probabilities before the accept step
    int shard_kind = 0;  // 0 none, 1 RCCL all-gather on the stream, 2 host callback
    std::atomic<int> shard_generation{0};   // bumped by every (un)sharding: a communicator that comes up late is dropped
    int shard_rank = 0, shard_world = 1;
    int64_t shard_chunk = 0, shard_lo = 0, shard_hi = 0;
    void *shard_comm = nullptr;         // ncclComm_t
    mtg_exchange_fn shard_fn = nullptr;
    void *shard_user = nullptr;
    double *shard_h_lnp = nullptr;      // pinned staging of the host exchange
    int32_t *shard_h_st = nullptr;
    int64_t shard_h_rows = 0;
    int64_t live_rows = 0;              // rows the next solve really evaluates (0: all) -- kernel choice only
    bool no_prior_batch = false;        // the batch being solved was expanded WITHOUT the prior (run_model_batch): kernel choice only

    // mtg_chain_autocorr: the convergence check is repeated on a growing chain, so plans and buffers stay.  Four plan
    // pairs, the least recently used one making room: the tutorial's loop checks the null and the alternative model's
    // chains in turn (two shapes), and a single slot was rebuilt at every check -- 9 ms each, a third of that loop
    // (scripts/tutorial_loop_probe.py)
    struct AcfPlans { hipfftHandle fwd = 0, inv = 0; bool have = false; int64_t n2 = 0, S = 0, P = 0; uint64_t used = 0; } acf_slots[4];
    uint64_t acf_clock = 0;
    DevBuf acf_chain, acf_x, acf_f, acf_g, acf_r, acf_ss, acf_tmp;
    int64_t acf_plans_built = 0;   // plan pairs made so far (mtg_chain_autocorr_plans_built: a cached shape must not add to it)

    // mtg_simulate_tk95: the inverse transform's plan (made once per length: a Bluestein plan for the 1 087 853 points
    // of BASELINE configs[3] takes 0.9 s to build, as long as the 2000 simulations it then runs) and its buffers.
    // mtg_simulate_plan may build it from a helper thread while the context is busy elsewhere: sim_mu.
    std::mutex sim_mu;
    // C2R plans of the simulator: [0] the bulk plan (sim_batch_for's batch for the length), [1] a short call's (fewer
    // series than that batch: a single light curve runs ONE transform); each remade when its (length, batch) changes
    struct SimPlan { hipfftHandle h = 0; bool have = false; int64_t nfft = 0; int batch = 0; } sim_plans[2];
    DevBuf sim_spec, sim_series;
    // ... and the hand-made chirp-z transform for lengths hipFFT would take through a Bluestein plan (0.9 s to build
    // against 15 ms for the power-of-two plans this needs; mtg_simulate.hip): chirp w [nfft], transform of the wrapped
    // conjugate chirp [m], work area [pairs][m], Z2Z plans of length m ([0] the bulk batch, [1] a short call's)
    struct SimCzt {
        bool tables = false;
        bool pairs_on = true;   // two series per complex transform (mtg_set_simulate_pairs)
        int64_t nfft = 0, m = 0;
        DevBuf chirp, bhat, work;
        struct { hipfftHandle h = 0; bool have = false; int64_t m = 0; int pairs = 0; } plans[2];
    } czt;
    // draws handed in by the caller for the NEXT mtg_simulate_tk95 (mtg_set_simulate_draws): standard normals
    // [S][2][nfft / 2 + 1] and segment starts [S]
    struct { DevBuf normals, starts; std::vector<int64_t> starts_host; int64_t S = 0, nk = 0; } given;
    // KraftNoise for noise_kind 3 (mtg_set_simulate_kraft): per-epoch background and the faint epochs' tables
    struct { DevBuf bkg, err, med, half; int K = 0; double threshold = 0.0; int64_t N = 0; } kraft;
    // the flux PDF of the simulated light curves (mtg_set_simulate_pdf): 0 Gaussian = TK95 as it is, 1 lognormal, 2 uniform
    // = the E13 adjustment of every cut segment on the device (mtg_e13.hip); its buffers, plans and last run's report
    struct E13 {
        int kind = 0, max_iter = 400;
        DevBuf seg, x, fresh, values, adj, keys, amp, spec, idx, order, order_tmp, segment, segment_out, flags, stdv, temp;
        hipfftHandle fwd = 0, inv = 0;
        bool have = false;
        int64_t n = 0, batch = 0;
        int64_t not_converged = 0;
        int iterations = 0;
        DevBuf given;            // caller's draws for the next simulation (mtg_set_simulate_pdf_draws): [S][n]
        int64_t given_S = 0, given_n = 0;
    } e13;

    // side streams: the structures (signatures) of a small batch run next to each other
    hipStream_t side[MTG_MAX_J / 2] = {};
    hipEvent_t side_done[MTG_MAX_J / 2] = {};
    hipEvent_t fork = nullptr;

    // per-call kernel timing (mtg_profile_*): event triples start / solve / end
    std::vector<hipEvent_t> prof_ev;
    int prof_cap = 0, prof_n = 0;

    // order of the serial sweep (mtg_sort.hip): 0 the caller's order, 1 always sorted by (structure, light curve),
    // 2 sorted unless the caller's order is known to be grouped already (host entry points look at lc_index)
    int sort_mode = 2;
    int spec_mode = 1;                  // speculative iterations of small ensembles: 0 never, 1 where they pay, 2 = 1 without the splits up front (mtg_ensemble_run)
    int lc_grouped_hint = 0;   // set by the host-pointer entry points for the call in flight
    DevBuf sort_keys, sort_keys_out, sort_order, sort_tmp;
    char last_solver[96] = "";   // what the last solve dispatched (mtg_last_solver)

    // Calls may come on the caller's streams (mtg_loglike_batch_device) and on the context's own; they share the
    // workspaces, so consecutive calls on different streams are chained with events: every call on a foreign
    // stream ends by recording `foreign_done` on it, every call begins by waiting for whatever ran last elsewhere.
    hipEvent_t foreign_done = nullptr, own_done = nullptr;
    bool foreign_pending = false;   // work recorded in foreign_done that the context's stream has not waited for
    bool own_dirty = false;         // the context's stream has had work since the last foreign call waited for it
    hipStream_t last_foreign = nullptr;

    // timing of the walker-sharded exchange (mtg_ensemble_shard_profile): event pairs around the first exchanges of a run
    std::vector<hipEvent_t> shard_ev;
    int shard_ev_cap = 0, shard_ev_n = 0;

    // mtg_pair_contexts: the partner whose pipelined half-steps share a launch with this context's (MtgPair below)
    // Shared ownership: a thread inside pair_launch holds a reference of its own, so that mtg_unpair_contexts /
    // mtg_destroy on the partner's thread cannot free the rendezvous under it.  Read and written with
    // std::atomic_load / std::atomic_store only.
    std::shared_ptr<struct MtgPair> pair;

    // mtg_set_simulate_transform: 0 = by grid length (default), 1 = hipFFT's own plan, 2 = chirp-z
    int sim_transform = 0;
};

// Two contexts whose pipelined sweeps go out in ONE launch (mtg_kernels_pipe_pair.hip): the two models of the Protassov
// test, each driven by a host thread of its own (mtg_ensemble_run: a loop of asynchronous launches).  Whoever reaches
// a pipelined half-step first leaves its arguments here, records `ready` on its stream and waits -- on the HOST, for as
// long as the partner takes to get to its own half-step, microseconds in steady state --; the second one makes its
// stream wait for `ready`, launches both models' rows in one grid, records `done`, and the first one's stream waits
// for that.  Nothing waits without a bound: a partner that does not come within `patience_ms` (stalled between two C
// calls, its run over, its batch on another kernel) means "alone this time"; MTG_PAIR_MAX_MISSES consecutive misses
// (each waited for half as long as the one before) break the pair for good and everybody launches alone from then on.
enum { MTG_PAIR_MAX_MISSES = 4 };
struct MtgPair {
    ~MtgPair()
    {
        for (hipEvent_t e : {ready[0], ready[1], done})
            if (e) (void)hipEventDestroy(e);
    }
    std::mutex mu;
    std::condition_variable cv;
    mtg_ctx *members[2] = {nullptr, nullptr};
    hipEvent_t ready[2] = {nullptr, nullptr}, done = nullptr;
    bool waiting = false;     // the slot holds a half-step
    int who = 0;              // ... of this member
    MtgSolveArgs sa;
    int64_t rows = 0;
    MtgPipeShapeId shape{};
    uint64_t launched = 0;    // pair launches so far (a waiter leaves when it moves)
    bool broken = false;
    int patience_ms = 250;
    int misses = 0;           // consecutive half-steps whose partner did not come
    int64_t n_pair = 0, n_solo = 0;
};

namespace {

// the current bank of structure lists ([nsig] signature lists, then [nsig] left-over lists) and counters (64)
inline int *bank_lists(const mtg_ctx *ctx) { return ctx->lists.as<int>() + (int64_t)ctx->bank * 2 * ctx->nsig_ws * ctx->cstride; }
inline int *bank_counts(const mtg_ctx *ctx) { return ctx->counts.as<int>() + ctx->bank * 64; }

void shard_release(mtg_ctx *ctx);
int shard_exchange(mtg_ctx *ctx, int64_t EH, hipStream_t s);

int fail(mtg_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail((ctx), MTG_E_HIP, "%s failed: %s", #call, hipGetErrorString(e__));   \
    } while (0)

int nparams(int kind)
{
    switch (kind) {
    case MTG_TERM_REAL: return 2;
    case MTG_TERM_COMPLEX3: return 3;
    case MTG_TERM_COMPLEX4: return 4;
    case MTG_TERM_SHO: return 3;
    case MTG_TERM_MATERN32: return 2;
    case MTG_TERM_JITTER: return 1;
    case MTG_TERM_DRW: return 2;
    case MTG_TERM_LORENTZIAN: return 3;
    case MTG_TERM_COSINUS: return 2;
    case MTG_TERM_BPL: return 3;
    default: return -1;
    }
}

int use_device(mtg_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return MTG_OK;
}

// Begin a call that launches on stream `s`: wait for what the previous calls left running on other streams.
int enter_stream(mtg_ctx *ctx, hipStream_t s)
{
    if (s == ctx->stream) {
        if (ctx->foreign_pending) {
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->foreign_done, 0));
            ctx->foreign_pending = false;
        }
        ctx->own_dirty = true;
        return MTG_OK;
    }
    if (ctx->own_dirty) {
        HIP_TRY(ctx, hipEventRecord(ctx->own_done, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->own_done, 0));
        ctx->own_dirty = false;
    }
    if (ctx->foreign_pending && ctx->last_foreign != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->foreign_done, 0));
    return MTG_OK;
}

// End of a call on a foreign stream: later calls (and mtg_synchronize) wait for this point.
int leave_stream(mtg_ctx *ctx, hipStream_t s)
{
    if (s == ctx->stream) return MTG_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->foreign_done, s));
    ctx->foreign_pending = true;
    ctx->last_foreign = s;
    return MTG_OK;
}

// the context's own stream, ordered after whatever ran last on a caller's stream
#define CTX_STREAM(ctx, s)                         \
    hipStream_t s = (ctx)->stream;                 \
    do {                                           \
        int rc__ = enter_stream((ctx), s);         \
        if (rc__) return rc__;                     \
    } while (0)

int reserve_workspace(mtg_ctx *ctx, int64_t B, int nslots, int nsig)
{
    // stride padded to a multiple of 64 so that every column starts 512-B aligned
    int64_t stride = (B + 63) / 64 * 64;
    if (stride < 64) stride = 64;
    HIP_TRY(ctx, ctx->coef.reserve((size_t)stride * nslots * sizeof(double)));
    // [nsig] signature lists, then [nsig] left-over lists of the windowed sweep (sweep_launch)
    HIP_TRY(ctx, ctx->lists.reserve((size_t)stride * (nsig > 1 ? nsig : 1) * 2 * 2 * sizeof(int)));   // two banks
    ctx->nsig_ws = nsig > 1 ? nsig : 1;
    HIP_TRY(ctx, ctx->counts.reserve(2 * 64 * sizeof(int)));
    HIP_TRY(ctx, ctx->sig.reserve((size_t)stride * sizeof(int32_t)));
    ctx->cstride = stride;
    return MTG_OK;
}

// every structure of the model (nr0 + 2k, nc0 - k) must have a compiled kernel; reserves the
// coefficient workspace for B evaluations
int check_model_workspace(mtg_ctx *ctx, int64_t B)
{
    const MtgModel &m = ctx->model;
    const int nsig = m.nsho + 1;
    for (int k = 0; k < nsig; ++k) {
        const int nr = m.nr0 + 2 * k, nc = m.nc0 - k;
        if (nr + nc == 0) continue;
        if (!mtg_find_solver(nr, nc))
            return fail(ctx, MTG_E_UNSUPPORTED,
                        "no compiled kernel for %d real + %d complex terms (J=%d)", nr, nc,
                        nr + 2 * nc);
    }
    MtgCoefLayout lay{m.nr_max, m.nc_max};
    return reserve_workspace(ctx, B, lay.nslots(), nsig);
}

// arguments of the theta -> coefficients expansion into the context's workspace
MtgPrepArgs make_prep_args(mtg_ctx *ctx, int64_t B, const double *d_theta, int add_prior, double *d_out,
                           int32_t *d_status)
{
    MtgPrepArgs pa;
    pa.model = ctx->model;
    pa.theta = d_theta;
    pa.B = B;
    pa.add_prior = add_prior;
    pa.coef = ctx->coef.as<double>();
    pa.cstride = ctx->cstride;
    pa.nsig = ctx->model.nsho + 1;
    pa.lists = bank_lists(ctx);
    pa.counts = bank_counts(ctx);
    pa.out = d_out;
    pa.status = d_status;
    pa.sig = ctx->sig.as<int32_t>();  // structure of every evaluation: the rank-10 time-parallel path dispatches on it
    pa.row_lo = 0;
    pa.row_hi = INT64_MAX;
    return pa;
}

// One launch of the serial sweep for structure k of the model.  When the resident set is larger than
// the reach of a buffer descriptor the kernel leaves the evaluations a wave cannot reach from its
// first light curve on a list; a second launch sweeps them one per wave (its workgroups find the
// list empty and leave at once when batches are grouped by light curve, the usual case).
int sweep_launch(mtg_ctx *ctx, mtg_solve_launcher fn, MtgSolveArgs sa, int64_t B, int k, hipStream_t s)
{
    sa.solo = 0;
    sa.left_list = nullptr;
    sa.left_count = nullptr;
    if (sa.yv_bytes <= sa.window_bytes) {
        fn(sa, B, s);
        return MTG_OK;
    }
    int *left_list = bank_lists(ctx) + ((int64_t)ctx->nsig_ws + k) * ctx->cstride;
    int *left_count = bank_counts(ctx) + 32 + k;
    HIP_TRY(ctx, hipMemsetAsync(left_count, 0, sizeof(int), s));
    sa.left_list = left_list;
    sa.left_count = left_count;
    fn(sa, B, s);
    sa.solo = 1;
    sa.list = left_list;
    sa.count_ptr = left_count;
    sa.seg_counts = nullptr;  // (the left-over list starts at 0, whatever segment of a sorted order it came from)
    sa.seg_k = 0;
    sa.left_list = nullptr;
    sa.left_count = nullptr;
    fn(sa, B * 64, s);  // one wave per left-over evaluation
    return MTG_OK;
}

int solve_prepared(mtg_ctx *ctx, int64_t B, const int32_t *d_lc, double *d_out, int32_t *d_status, hipStream_t s,
                   bool may_sort = false);

// theta -> coefficients -> solver(s) for B evaluations; timing events around the launches
int run_model_batch(mtg_ctx *ctx, int64_t B, const double *d_theta, const int32_t *d_lc,
                    int add_prior, double *d_out, int32_t *d_status, hipStream_t s)
{
    int rc = check_model_workspace(ctx, B);
    if (rc) return rc;
    const int nsig = ctx->model.nsho + 1;
    const bool prof = ctx->prof_n < ctx->prof_cap;
    hipEvent_t *pe = prof ? &ctx->prof_ev[3 * (size_t)ctx->prof_n] : nullptr;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, s));
    if (prof) HIP_TRY(ctx, hipEventRecord(pe[0], s));
    {
        mtg_trace::Range range("mtg:prepare (theta -> prior, coefficients)");
        if (nsig > 1) HIP_TRY(ctx, hipMemsetAsync(bank_counts(ctx), 0, 64 * sizeof(int), s));
        mtg_launch_prepare(make_prep_args(ctx, B, d_theta, add_prior, d_out, d_status), s);
    }
    if (prof) HIP_TRY(ctx, hipEventRecord(pe[1], s));
    {
        mtg_trace::Range range("mtg:solve (factorisation + forward solve)");
        ctx->no_prior_batch = !add_prior;
        rc = solve_prepared(ctx, B, d_lc, d_out, d_status, s, true);
        ctx->no_prior_batch = false;
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, s));
    if (prof) {
        HIP_TRY(ctx, hipEventRecord(pe[2], s));
        ctx->prof_n += 1;
    }
    ctx->timed = true;
    return MTG_OK;
}

// MTG_SWEEP_MULTI=0 (MTG_MEASURE builds only): one launch per structure even where the one-launch kernel exists
bool sweep_multi_enabled()
{
    static const bool on = !(mtg_measure_env("MTG_SWEEP_MULTI") && atoi(mtg_measure_env("MTG_SWEEP_MULTI")) == 0);
    return on;
}

// A pipelined half-step of a paired context (MtgPair): *paired = 1 when its rows went out -- or will go out, ordered
// before anything that follows on `s` -- in a launch shared with the partner's; 0: the caller launches alone.
int pair_launch(mtg_ctx *ctx, const MtgSolveArgs &sa, int64_t B, const MtgPipeShapeId &shape, hipStream_t s, int *paired)
{
    *paired = 0;
    const std::shared_ptr<MtgPair> p = std::atomic_load(&ctx->pair);   // ours for the whole call, whatever the partner does
    if (!p) return MTG_OK;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->broken) { p->n_solo += 1; return MTG_OK; }
    const int me = p->members[0] == ctx ? 0 : 1;
    if (!p->waiting) {
        HIP_TRY(ctx, hipEventRecord(p->ready[me], s));
        p->sa = sa; p->rows = B; p->shape = shape; p->who = me; p->waiting = true;
        const uint64_t seen = p->launched;
        const int patience = std::max(1, p->patience_ms >> std::min(p->misses, 8));
        p->cv.wait_for(lk, std::chrono::milliseconds(patience), [&] { return p->launched != seen || p->broken; });
        if (p->launched != seen) {   // the partner launched both
            p->misses = 0;
            HIP_TRY(ctx, hipStreamWaitEvent(s, p->done, 0));
            *paired = 1;
            return MTG_OK;
        }
        p->waiting = false;          // nobody came: alone this time; for good after a few misses in a row
        p->misses += 1;
        if (p->misses >= MTG_PAIR_MAX_MISSES) p->broken = true;
        p->n_solo += 1;
        return MTG_OK;
    }
    // the partner's half-step is waiting: both in one launch, on this stream
    const MtgSolveArgs &other = p->sa;
    mtg_pipe_pair_launcher fn = nullptr;
    bool mine_first = false;
    if (other.N == sa.N && p->who != me) {
        // (member 0's model first, as the pairs are listed: null, alternative; then the other way round)
        const bool other_is_0 = p->who == 0;
        fn = other_is_0 ? mtg_find_pipe_pair_solver(p->shape, shape) : mtg_find_pipe_pair_solver(shape, p->shape);
        mine_first = !other_is_0;
        if (!fn) {
            fn = other_is_0 ? mtg_find_pipe_pair_solver(shape, p->shape) : mtg_find_pipe_pair_solver(p->shape, shape);
            mine_first = other_is_0;
        }
    }
    if (!fn) {   // different samplings, or a pair of shapes that is not compiled: both go alone from now on
        p->broken = true;
        p->n_solo += 1;
        lk.unlock();
        p->cv.notify_all();
        return MTG_OK;
    }
    if (hipStreamWaitEvent(s, p->ready[p->who], 0) != hipSuccess) {   // nothing launched: the waiter goes alone
        p->broken = true;
        p->n_solo += 1;
        lk.unlock();
        p->cv.notify_all();
        return fail(ctx, MTG_E_HIP, "pair_launch: hipStreamWaitEvent failed");
    }
    if (mine_first) fn(sa, B, other, p->rows, s);
    else fn(other, p->rows, sa, B, s);
    // The partner's rows ARE in flight from here on: whatever happens next, its wait must end with "launched" (a waiter
    // that timed out would launch them a second time), and `done` is what its stream orders itself behind.
    const hipError_t recorded = hipEventRecord(p->done, s);
    p->waiting = false;
    p->launched += 1;
    p->n_pair += 1;
    p->misses = 0;
    *paired = 1;
    if (recorded != hipSuccess) p->broken = true;
    lk.unlock();
    p->cv.notify_all();
    if (recorded != hipSuccess) return fail(ctx, MTG_E_HIP, "pair_launch: hipEventRecord(done) failed: %s", hipGetErrorString(recorded));
    return MTG_OK;
}

// Launch the solver(s) for B prepared evaluations living in ctx->coef (lists / counts filled).
// may_sort: the caller's order is arbitrary (mtg_loglike_batch[_device]); the device sampler's batches are grouped
// by ensemble, hence by light curve, by construction.
// What every solver launch of B prepared evaluations takes: the resident light curves, the coefficient columns, the
// context's tables (made at the first call).
int solve_args_base(mtg_ctx *ctx, int64_t B, const int32_t *d_lc, double *d_out, int32_t *d_status, hipStream_t s, MtgSolveArgs &sa)
{
    const MtgModel &m = ctx->model;
    MtgCoefLayout lay{m.nr_max, m.nc_max};
    sa.coef = ctx->coef.as<double>();
    sa.cstride = ctx->cstride;
    sa.lay = lay;
    sa.B = B;
    sa.lc_index = d_lc;
    sa.status = d_status;
    sa.out = d_out;
    sa.dxt = ctx->dxt.as<double2>();
    sa.yv = ctx->yv.as<double2>();
    sa.N = ctx->N;
    sa.t_stride = ctx->t_per_lc ? ctx->N : 0;
    sa.dxmax = ctx->dxmax.as<double>();
    sa.yv_bytes = (uint64_t)ctx->L * (uint64_t)ctx->N * 16u;
    sa.dxt_bytes = (uint64_t)(ctx->t_per_lc ? ctx->L : 1) * (uint64_t)ctx->N * 16u;
    sa.window_bytes = ctx->window_bytes;
    sa.mean_kind = m.mean_kind;
    // the mean vanishes identically when it is a frozen constant equal to 0 (the
    // per-light-curve frozen mean lives in y_offset)
    sa.has_mean = !(m.mean_kind == MTG_MEAN_CONSTANT && m.src[m.nk] < 0 && m.defaults[m.nk] == 0.0);
    for (int i = 0; i < m.nterms; ++i)
        if (m.kinds[i] == MTG_TERM_JITTER) sa.has_mean = 1;  // the plain sweep variant also skips the jitter add
    if (!ctx->tables_ready) {   // once per context, on the stream of its first batch (every later one is ordered behind it)
        HIP_TRY(ctx, ctx->tables.reserve(mtg_tables_bytes()));
        mtg_launch_tables(ctx->tables.p, s);
        HIP_TRY(ctx, hipGetLastError());
        ctx->tables_ready = true;
    }
    sa.tables = ctx->tables.p;
    if (const char *env = mtg_measure_env("MTG_GLOBAL_TABLES"))   // MTG_MEASURE builds only: 0 = every workgroup computes its own
        if (atoi(env) == 0) sa.tables = nullptr;
    return MTG_OK;
}

int solve_prepared(mtg_ctx *ctx, int64_t B, const int32_t *d_lc, double *d_out, int32_t *d_status, hipStream_t s, bool may_sort)
{
    const MtgModel &m = ctx->model;
    const int nsig = m.nsho + 1;
    MtgSolveArgs sa;
    {
        const int rc = solve_args_base(ctx, B, d_lc, d_out, d_status, s, sa);
        if (rc) return rc;
    }
    // A small batch of long light curves leaves a one-lane-per-evaluation launch idle for N serial
    // steps: give every evaluation a whole wave (or four) instead (mtg_timeparallel.hip); the rank-10
    // structures get as many chunks per evaluation as fill the GPU (mtg_tp_big.h).
    // Measured crossovers: J <= 6 (one workgroup per evaluation) pays up to several thousand evaluations -- the serial
    // sweep runs one wave per 64 evaluations, latency bound, on a fraction of the SIMDs until ~10^5 of them; the J = 10
    // path costs ~3 x the serial sweep's work per sample, spread over every SIMD instead of B / 64 of
    // them, against ~1.05 us x N for the serial sweep whatever B <= 65 536 is.
    const int Jmodel = m.nr0 + 2 * m.nc0;
    // rows that do work: a walker-sharded half-step skips the rows of the other ranks (MTG_ST_REMOTE)
    const int64_t Bw = ctx->live_rows > 0 ? ctx->live_rows : B;
    bool pays;
    // (scripts/crossover_probe.py, serial sweep / time-parallel in ms: N = 1e4, J = 5: 3.4 / 0.35 at 1024 evaluations,
    // 3.4 / 1.0 at 4096, 3.4 / 1.8 at 8192, equal at 16 384; N = 1e3, J = 5: 0.36 / 0.29 at 4096, 0.36 / 0.51 at 8192)
    // (round 4, against the pipelined sweep that now takes over beyond: scripts/pipe_probe.py, N = 1e4, time-parallel / pipeline
    // in ms: J = 3: 0.82 / 1.37 at 8192 rows, 1.20 / 1.39 at 12 288, 1.51 / 1.40 at 16 000; J = 5: 1.85 / 2.40 at 8192, 2.73 / 2.46 at 12 288)
    if (Jmodel <= 6) pays = ctx->N >= 256 && Bw <= (ctx->N >= 4096 ? (Jmodel <= 3 ? 12288 : 8192) : 4096);
    else pays = ctx->N >= 1024 && Bw <= 8192;
    // A term with a free b (ComplexTerm with four parameters, BendingPowerlaw) has a power spectrum that goes negative
    // where b d > a c -- which is exactly what those terms' own log_prior forbids, so a batch expanded WITH the prior never
    // solves such a row.  Without it (the optimiser's -lnL, gpmodelling.py:155-169) it may, and there the state-space form
    // the time-parallel kernels work in has an indefinite stationary covariance: their filter pass was found 1e-7 off on
    // such a row (tests/test_fuzz_gpu.py at MTG_FUZZ_OFFSET=112000, case 70; scripts/fuzz_case.py) where celerite's own
    // recursion -- the sweep -- is exact to rounding.  Those batches keep the sweep.
    bool free_b = false;
    for (int i = 0; i < m.nterms; ++i) free_b = free_b || m.kinds[i] == MTG_TERM_COMPLEX4 || m.kinds[i] == MTG_TERM_BPL;
    const bool tp_allowed = !(free_b && ctx->no_prior_batch);
    const bool small = tp_allowed && (ctx->tp_mode == 1 || ctx->tp_mode == 3 || (ctx->tp_mode == 2 && pays));
    sa.tp_ws = nullptr;
    sa.tp_chunks = 0;
    sa.tp_gsize = 0;
    sa.tp_direct = ctx->tp_direct;
    sa.tp_nr0 = m.nr0; sa.tp_nc0 = m.nc0;
    sa.sig = ctx->sig.as<int32_t>();
    bool small_ok = small;
    // mode 3 promises bits that do not depend on the batch; the rank-10 path sizes its chunks and scan groups by the
    // batch (mtg_tp_big_chunks), so under mode 3 such a model keeps the serial sweep
    if (ctx->tp_mode == 3 && Jmodel > 6) small_ok = false;
    if (Jmodel == 0) small_ok = false;  // a white kernel: nothing to parallelise over time (mtg_white_kernel)
    if (small_ok && Jmodel > 6) {
        for (int k = 0; k < nsig; ++k)
            if (!mtg_find_tp_solver(m.nr0 + 2 * k, m.nc0 - k)) small_ok = false;
        const int C = mtg_tp_big_chunks(ctx->N, Bw);
        int g = mtg_tp_big_gsize(Bw, C);
        if (const char *env = mtg_measure_env("MTG_TP_GSIZE")) {  // MTG_MEASURE builds only
            const int v = atoi(env);
            if (v == 4 || v == 8 || v == 16) g = v;
        }
        const size_t need = (size_t)mtg_tp_big_plan(Jmodel, B, C, g).total * sizeof(double);
        if (B > 65535 || need > ((size_t)16 << 30)) small_ok = false;  // grid / workspace limits: the serial sweep
        if (small_ok) {
            HIP_TRY(ctx, ctx->tp_ws.reserve(need));
            sa.tp_ws = ctx->tp_ws.as<double>();
            sa.tp_chunks = C;
            sa.tp_gsize = g;
        }
    }
    // four waves per evaluation: while every evaluation's workgroup is resident at once (rank <= 3: two per CU, their
    // elements take 68 KB of LDS; above: one)
    // (scripts/spec_probe.py, J = 3, N = 1e4: 384 rows 70.9 us against 96.0 us with one wave each, 512 rows 77.4 / 97.2)
    // (mode 3: the one-wave kernel whatever the batch, so that a row's bits do not depend on how many rows travel with it)
    const bool wide = ctx->tp_mode != 3 && Bw <= (Jmodel <= 3 ? 512 : 256) && ctx->N >= 4096;
    // two waves per evaluation between 257 and 512 rows of rank 4 or 5: half a CU's LDS each, all resident at once
    // (scripts/spec_probe.py, J = 5, N = 1e4, 384 rows: see DESIGN.md)
    const bool mid = ctx->tp_mode != 3 && !wide && Bw <= 512 && ctx->N >= 4096 && (Jmodel == 4 || Jmodel == 5);
    // Between the time-parallel kernels' range and ~one wave per SIMD the serial sweep is one lone wave per 64 rows on
    // a fraction of the SIMDs, N dependent steps of ~166 instructions: the pipelined form puts the generators of those
    // rows on a second wave (mtg_kernels_pipe.hip) -- one workgroup of 128 rows per CU, all resident at once.
    mtg_solve_launcher pipe = nullptr;
    if (!small_ok && ctx->pipe_mode != 0 && ctx->N >= 64 && sa.yv_bytes <= sa.window_bytes &&
        (ctx->pipe_mode == 1 || (ctx->N >= 256 && B <= (int64_t)MTG_PIPE_ROWS_PER_CU * ctx->cus)))
        pipe = mtg_find_pipe_solver(m.nr0, m.nc0, nsig, m.last_b0);
    mtg_solve_launcher fused = nullptr;
    int fused_lanes = 64;
    if (small_ok && nsig > 1) {
        if (wide && (fused = mtg_find_tp_fused_solver(m.nr0, m.nc0, nsig, 256))) fused_lanes = 256;
        if (!fused && mid && (fused = mtg_find_tp_fused_solver(m.nr0, m.nc0, nsig, 128))) fused_lanes = 128;
        if (!fused) fused = mtg_find_tp_fused_solver(m.nr0, m.nc0, nsig, 64);
    }
    sa.solo = 0; sa.left_list = nullptr; sa.left_count = nullptr;
    // The serial sweep reads each lane's own light curve: sort the evaluations by (structure, light curve) unless the
    // caller's order is known to be grouped (mtg_sort.hip).  One light curve, or no index at all: nothing to sort.
    // More than one structure: the per-structure lists are appended to with one atomic per wave, so their order -- which
    // rows share a wave -- changes from run to run, and a row's last bits may depend on its wave (a lane with a huge
    // d dx sends the whole wave through the libm sincos).  A seeded chain has to be reproducible: the stable sort gives
    // the lanes of every structure the caller's order, whatever the arrival order of the waves was.
    const int *sorted = nullptr;
    const bool for_order = may_sort && d_lc && ctx->L > 1 && (ctx->sort_mode == 1 || (ctx->sort_mode == 2 && !ctx->lc_grouped_hint));
    const bool for_determinism = nsig > 1 && ctx->sort_mode != 0;
    if ((for_order || for_determinism) && !small_ok && B > 64 && (uint64_t)ctx->L * (uint64_t)nsig < 0x7fffffffull) {
        mtg_trace::Range range("mtg:sort (evaluations by structure, light curve)");
        const size_t tmp = mtg_sort_temp_bytes(B, mtg_sort_key_bits(ctx->L, nsig));
        HIP_TRY(ctx, ctx->sort_keys.reserve((size_t)B * 4));
        HIP_TRY(ctx, ctx->sort_keys_out.reserve((size_t)B * 4));
        HIP_TRY(ctx, ctx->sort_order.reserve((size_t)B * 4));
        HIP_TRY(ctx, ctx->sort_tmp.reserve(tmp > 0 ? tmp : 16));
        HIP_TRY(ctx, mtg_launch_sort_by_lightcurve(B, d_status, nsig > 1 ? ctx->sig.as<int32_t>() : nullptr, d_lc, ctx->L, nsig,
                                                   ctx->sort_keys.as<uint32_t>(), ctx->sort_keys_out.as<uint32_t>(),
                                                   ctx->sort_order.as<int>(), ctx->sort_tmp.p, tmp, s));
        sorted = ctx->sort_order.as<int>();
    }
    if (small_ok && Jmodel > 6) {  // rank 10: every structure in one sequence of launches (mtg_tp_big.h)
        sa.list = nullptr;
        sa.count_ptr = nullptr;
        snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_tpb_compose4q_kernel (+ mtg_tpb_reduce_kernel<10>, C = %d)",
                 sa.tp_chunks);
        mtg_launch_tp_big(sa, B, s);
    } else if (fused) {  // every signature in one launch
        sa.list = bank_lists(ctx);
        sa.count_ptr = bank_counts(ctx);
        snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_tp_fused_kernel<%d,%d,%d,%d>", m.nr0, m.nc0, nsig, fused_lanes);
        fused(sa, B, s);
    } else if (pipe && (nsig == 1 || sorted)) {
        sa.list = sorted;            // nsig == 1: the sorted order, or NULL = the caller's
        sa.count_ptr = nullptr;
        sa.seg_counts = nsig > 1 ? bank_counts(ctx) : nullptr;
        sa.seg_k = 0;
        const MtgPipeShapeId shape{m.nr0, m.nc0, nsig, m.last_b0 ? 1 : 0};
        int paired = 0;
        if (std::atomic_load(&ctx->pair)) {
            const int rc = pair_launch(ctx, sa, B, shape, s, &paired);
            if (rc) return rc;
        }
        if (paired) {
            snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_pipe_pair_kernel (this model: <%d,%d,%d,%d>)", m.nr0, m.nc0, nsig, m.last_b0 ? 1 : 0);
        } else {
            snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_pipe_kernel<%d,%d,%d,%d>", m.nr0, m.nc0, nsig, m.last_b0 ? 1 : 0);
            pipe(sa, B, s);
        }
    } else if (mtg_solve_launcher multi = sorted && nsig > 1 && sa.yv_bytes <= sa.window_bytes && sweep_multi_enabled()
                                              ? mtg_find_multi_solver(m.nr0, m.nc0, nsig, m.last_b0) : nullptr) {
        // every structure of the sorted order in one launch of identical workgroups (mtg_kernels_multi.hip)
        sa.list = sorted;
        sa.count_ptr = nullptr;
        sa.seg_counts = bank_counts(ctx);
        sa.seg_k = 0;
        snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_solve_kernel_multi<%d,%d,%d,%d>", m.nr0, m.nc0, nsig, m.last_b0 ? 1 : 0);
        multi(sa, B, s);
    } else {
        // A time-parallel launch is latency bound: a structure holding three evaluations takes as long
        // as one holding 250 (J = 10: ~10 ms each), and one after the other on the same stream they
        // add up.  The structures work on disjoint evaluations, so each gets its own stream: forked
        // after the expansion, joined before whatever follows on `s`.
        // The serial sweep is latency bound in the same way -- N dependent steps, ~0.4 us each, whatever the number of
        // rows -- and a sampler's half-step of 256 000 walkers with a handful of them over-damped paid 14.9 ms for
        // the first structure and 3.2-4.3 ms more for those few (profiles/r03_c3_halfstep_trace.txt).  On their own
        // stream they take wave slots as the big launch frees them and finish under it.
        static const bool sweep_fan_out = !(mtg_measure_env("MTG_SWEEP_FANOUT") && atoi(mtg_measure_env("MTG_SWEEP_FANOUT")) == 0);
        const bool fan_out = (small_ok || sweep_fan_out) && nsig > 1 && nsig - 1 <= MTG_MAX_J / 2;
        if (fan_out) {
            if (!ctx->fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming));
            int prio_low = 0, prio_high = 0;
            HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
            for (int k = 0; k + 1 < nsig; ++k) {
                // (above the caller's stream: the few rows of a rare structure should not queue behind the common one)
                if (!ctx->side[k]) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->side[k], hipStreamNonBlocking, prio_high));
                if (!ctx->side_done[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_done[k], hipEventDisableTiming));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->fork, s));
        }
        // the side streams first: their (usually few) waves are resident before the common structure's launch fills
        // every slot its registers allow (J = 6: two waves of 204 VGPRs leave no room for a third of 166)
        for (int kk = 0; kk < nsig; ++kk) {
            const int k = fan_out ? nsig - 1 - kk : kk;
            const int nr = m.nr0 + 2 * k, nc = m.nc0 - k;
            mtg_solve_launcher fn = mtg_find_solver(nr, nc, m.last_b0);
            if (!fn) continue;
            mtg_solve_launcher tp = small_ok ? mtg_find_tp_solver(nr, nc) : nullptr;
            if (tp && wide && mtg_find_tp_wide_solver(nr, nc)) tp = mtg_find_tp_wide_solver(nr, nc);
            sa.list = nsig > 1 ? bank_lists(ctx) + (int64_t)k * ctx->cstride : nullptr;
            sa.count_ptr = nsig > 1 ? bank_counts(ctx) + k : nullptr;
            sa.seg_counts = nullptr; sa.seg_k = 0;
            if (sorted && !tp) {  // the k-th segment of the sorted order
                sa.list = sorted;
                sa.seg_counts = nsig > 1 ? bank_counts(ctx) : nullptr;
                sa.seg_k = k;
            }
            if (k == 0) {
                if (tp) snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_tp_kernel<%d,%d,%d>", nr, nc, tp == mtg_find_tp_solver(nr, nc) ? 64 : 256);
                else if (nr + nc == 0) snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_white_kernel");
                else snprintf(ctx->last_solver, sizeof ctx->last_solver, "mtg_solve_kernel<%d,%d,%d>", nr, nc, mtg_solver_uses_b0(nr, nc, m.last_b0));
            }
            hipStream_t sk = fan_out && k > 0 ? ctx->side[k - 1] : s;
            if (sk != s) HIP_TRY(ctx, hipStreamWaitEvent(sk, ctx->fork, 0));
            if (tp) {
                sa.solo = 0; sa.left_list = nullptr; sa.left_count = nullptr;
                tp(sa, B, sk);
            } else {
                const int rc = sweep_launch(ctx, fn, sa, B, k, sk);
                if (rc) return rc;
            }
            if (sk != s) HIP_TRY(ctx, hipEventRecord(ctx->side_done[k - 1], sk));
        }
        if (fan_out)
            for (int k = 1; k < nsig; ++k)
                if (mtg_find_solver(m.nr0 + 2 * k, m.nc0 - k, m.last_b0)) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->side_done[k - 1], 0));
    }
    HIP_TRY(ctx, hipGetLastError());
    return MTG_OK;
}

int check_ready(mtg_ctx *ctx, bool need_model)
{
    if (!ctx) return MTG_E_ARG;
    if (ctx->N <= 0) return fail(ctx, MTG_E_STATE, "mtg_set_lightcurves has not been called");
    if (need_model && !ctx->has_model) return fail(ctx, MTG_E_STATE, "mtg_set_model has not been called");
    return MTG_OK;
}

}  // namespace

extern "C" {

MTG_API int mtg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

MTG_API const char *mtg_version(void) { return "mtg-hip 0.1 (gfx950)"; }

MTG_API int mtg_term_nparams(int kind) { return nparams(kind); }

MTG_API int mtg_structure_supported(int jr, int jc) { return mtg_find_solver(jr, jc) ? 1 : 0; }

static mtg_ctx *create_context(int device, int part, int parts)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(nullptr, MTG_E_NODEVICE, "no HIP device available (%s)",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        fail(nullptr, MTG_E_ARG, "device %d out of range [0, %d)", device, n);
        return nullptr;
    }
    if (parts < 1 || parts > 8 || part < 0 || part >= parts) {
        fail(nullptr, MTG_E_ARG, "slice %d of %d compute-unit slices: 1 to 8 slices", part, parts);
        return nullptr;
    }
    mtg_ctx *ctx = new (std::nothrow) mtg_ctx();
    if (!ctx) return nullptr;
    ctx->device = device;
    bool ok = hipSetDevice(device) == hipSuccess;
    if (ok && (hipDeviceGetAttribute(&ctx->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->cus <= 0))
        ctx->cus = 256;
    if (ok && parts > 1) {
        // a contiguous run of mask bits per slice (MTG_CU_SLICE_INTERLEAVED=1: bit i to slice i mod parts): the contexts
        // of different slices run their kernels side by side on disjoint compute units (a queue's CU mask), whatever the
        // order their launches arrive in.  (The driver deals the bits of a mask round over the XCDs: a contiguous run
        // leaves every XCD with its share of enabled compute units, which a dispatch split over all XCDs needs.)
        uint32_t mask[16] = {};
        int mine = 0;
        const bool interleaved = mtg_measure_env("MTG_CU_SLICE_INTERLEAVED") && atoi(mtg_measure_env("MTG_CU_SLICE_INTERLEAVED")) == 1;
        const int per = ctx->cus / parts;
        for (int i = 0; i < ctx->cus && i < 512; ++i)
            if (interleaved ? i % parts == part : (i / per == part || (part == parts - 1 && i / per >= parts))) {
                mask[i / 32] |= 1u << (i % 32);
                ++mine;
            }
        ok = hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)((ctx->cus + 31) / 32), mask) == hipSuccess;
        ctx->cus = mine;
    } else if (ok) {
        ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    }
    if (!ok || hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->foreign_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->own_done, hipEventDisableTiming) != hipSuccess) {
        fail(nullptr, MTG_E_HIP, "could not create stream/events on device %d", device);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

MTG_API mtg_ctx *mtg_create(int device) { return create_context(device, 0, 1); }

MTG_API mtg_ctx *mtg_create_on_slice(int device, int part, int parts) { return create_context(device, part, parts); }

MTG_API int mtg_unpair_contexts(mtg_ctx *ctx);

MTG_API void mtg_destroy(mtg_ctx *ctx)
{
    if (!ctx) return;
    (void)mtg_unpair_contexts(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->foreign_pending) (void)hipEventSynchronize(ctx->foreign_done);
    DevBuf *bufs[] = {&ctx->sort_keys, &ctx->sort_keys_out, &ctx->sort_order, &ctx->sort_tmp, &ctx->dxt, &ctx->yv, &ctx->t_tmp, &ctx->y_tmp, &ctx->dy_tmp, &ctx->off_tmp, &ctx->dxmax, &ctx->coef, &ctx->lists,
                      &ctx->counts, &ctx->tp_ws, &ctx->sig, &ctx->tables, &ctx->theta, &ctx->lc, &ctx->out, &ctx->status,
                      &ctx->ens_coords, &ctx->ens_lnp, &ctx->ens_perm, &ctx->ens_q, &ctx->ens_factor,
                      &ctx->ens_new, &ctx->ens_st, &ctx->ens_lc_full, &ctx->ens_lc_half, &ctx->ens_lc_spec, &ctx->ens_perm_all, &ctx->ens_naccept,
                      &ctx->ens_best_lnp, &ctx->ens_best_coords, &ctx->ens_notpd, &ctx->ens_chain,
                      &ctx->ens_lnp_chain};
    for (DevBuf *b : bufs) b->release();
    shard_release(ctx);
    {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        for (auto &sl : ctx->acf_slots)
            if (sl.have) { (void)hipfftDestroy(sl.fwd); (void)hipfftDestroy(sl.inv); }
        for (auto &sp : ctx->sim_plans)
            if (sp.have) (void)hipfftDestroy(sp.h);
        for (auto &sp : ctx->czt.plans)
            if (sp.have) (void)hipfftDestroy(sp.h);
        if (ctx->e13.have) { (void)hipfftDestroy(ctx->e13.fwd); (void)hipfftDestroy(ctx->e13.inv); ctx->e13.have = false; }
    }
    for (DevBuf *b : {&ctx->kraft.bkg, &ctx->kraft.err, &ctx->kraft.med, &ctx->kraft.half}) b->release();
    for (DevBuf *b : {&ctx->e13.seg, &ctx->e13.x, &ctx->e13.fresh, &ctx->e13.values, &ctx->e13.adj, &ctx->e13.keys, &ctx->e13.amp,
                      &ctx->e13.spec, &ctx->e13.idx, &ctx->e13.order, &ctx->e13.order_tmp, &ctx->e13.segment, &ctx->e13.segment_out, &ctx->e13.flags, &ctx->e13.stdv, &ctx->e13.temp,
                      &ctx->e13.given})
        b->release();
    ctx->sim_spec.release();
    ctx->sim_series.release();
    for (DevBuf *b : {&ctx->acf_chain, &ctx->acf_x, &ctx->acf_f, &ctx->acf_g, &ctx->acf_r, &ctx->acf_ss, &ctx->acf_tmp}) b->release();
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->shard_ev) (void)hipEventDestroy(e);
    if (ctx->foreign_done) (void)hipEventDestroy(ctx->foreign_done);
    if (ctx->own_done) (void)hipEventDestroy(ctx->own_done);
    for (hipStream_t st : ctx->side) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->side_done) if (ev) (void)hipEventDestroy(ev);
    if (ctx->fork) (void)hipEventDestroy(ctx->fork);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

MTG_API int mtg_set_window_bytes(mtg_ctx *ctx, uint64_t bytes)
{
    if (!ctx) return MTG_E_ARG;
    if (bytes < 16 || bytes > 0xffffffffull) return fail(ctx, MTG_E_ARG, "mtg_set_window_bytes: 16 <= bytes < 2^32");
    if (ctx->N > 0 && (uint64_t)ctx->N * 16u > bytes)
        return fail(ctx, MTG_E_ARG, "mtg_set_window_bytes: a resident light curve (%lld samples) does not fit", (long long)ctx->N);
    ctx->window_bytes = bytes;
    return MTG_OK;
}

MTG_API const char *mtg_last_error(const mtg_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

static int set_lightcurves_common(mtg_ctx *ctx, int64_t N, int64_t L, const double *t, int t_per_lc,
                                  const double *y, const double *yerr, const double *y_offset,
                                  hipMemcpyKind kind)
{
    if (!ctx) return MTG_E_ARG;
    if (N <= 0 || L <= 0 || !t || !y || !yerr)
        return fail(ctx, MTG_E_ARG, "mtg_set_lightcurves: need N > 0, L > 0 and non-NULL t, y, yerr");
    // one light curve must fit the reach of a buffer descriptor; the set itself may fill the HBM
    if ((uint64_t)N * 16u > ctx->window_bytes)
        return fail(ctx, MTG_E_ARG, "light curve too long: N * 16 bytes must stay below %llu",
                    (unsigned long long)ctx->window_bytes);
    if (L > 0x7fffffff) return fail(ctx, MTG_E_ARG, "too many light curves (32-bit indices)");
    int rc = use_device(ctx);
    if (rc) return rc;
    mtg_trace::Range range("mtg:set_lightcurves (upload, sigma^2, dx)");
    const int64_t t_rows = t_per_lc ? L : 1;
    rc = enter_stream(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, ctx->dxt.reserve((size_t)t_rows * N * 16));
    HIP_TRY(ctx, ctx->yv.reserve((size_t)L * N * 16));
    // staged in blocks of light curves (<= ~256 MiB of staging per array), so that a resident set
    // of tens of GB does not need its own size again in scratch
    int64_t block = ((int64_t)256 << 20) / (N * 8);
    if (block < 1) block = 1;
    if (block > L) block = L;
    HIP_TRY(ctx, ctx->t_tmp.reserve((size_t)(t_per_lc ? block : 1) * N * 8));
    HIP_TRY(ctx, ctx->y_tmp.reserve((size_t)block * N * 8));
    HIP_TRY(ctx, ctx->dy_tmp.reserve((size_t)block * N * 8));
    HIP_TRY(ctx, ctx->dxmax.reserve(64));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dxmax.p, 0, 16, ctx->stream));  // [0] max dx, [1] "unsorted" flag
    if (y_offset) HIP_TRY(ctx, ctx->off_tmp.reserve((size_t)block * 8));
    for (int64_t l0 = 0; l0 < L; l0 += block) {
        const int64_t lb = l0 + block <= L ? block : L - l0;
        const int64_t tr = t_per_lc ? lb : (l0 == 0 ? 1 : 0);  // a shared sampling is set up once
        if (tr)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->t_tmp.p, t + (t_per_lc ? l0 * N : 0), (size_t)tr * N * 8, kind, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->y_tmp.p, y + l0 * N, (size_t)lb * N * 8, kind, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dy_tmp.p, yerr + l0 * N, (size_t)lb * N * 8, kind, ctx->stream));
        const double *d_off = nullptr;
        if (y_offset) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->off_tmp.p, y_offset + l0, (size_t)lb * 8, kind, ctx->stream));
            d_off = ctx->off_tmp.as<double>();
        }
        mtg_launch_lc_setup(N, lb, tr, ctx->t_tmp.as<double>(), ctx->y_tmp.as<double>(), ctx->dy_tmp.as<double>(),
                            d_off, ctx->dxt.as<double2>() + (t_per_lc ? l0 * N : 0), ctx->yv.as<double2>() + l0 * N,
                            ctx->dxmax.as<double>(), ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        // the staging buffers are reused by the next block; pageable host copies have returned by
        // now, device-to-device ones are ordered on the stream
    }
    uint64_t flags[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->dxmax.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flags[1]) {  // device-resident times cannot be checked on the host
        ctx->N = 0; ctx->L = 0;
        return fail(ctx, MTG_E_ARG, "the input coordinates must be sorted");
    }
    ctx->N = N; ctx->L = L; ctx->t_per_lc = t_per_lc ? 1 : 0;
    return MTG_OK;
}

MTG_API int mtg_set_lightcurves(mtg_ctx *ctx, int64_t N, int64_t L, const double *t, int t_per_lc,
                                const double *y, const double *yerr, const double *y_offset)
{
    if (!ctx) return MTG_E_ARG;
    if (N <= 0 || L <= 0 || !t || !y || !yerr)
        return fail(ctx, MTG_E_ARG, "mtg_set_lightcurves: need N > 0, L > 0 and non-NULL t, y, yerr");
    // celerite.GP.compute raises ValueError for unsorted times
    const int64_t t_rows = t_per_lc ? L : 1;
    for (int64_t r = 0; r < t_rows; ++r)
        for (int64_t n = 1; n < N; ++n)
            if (!(t[r * N + n] >= t[r * N + n - 1]))
                return fail(ctx, MTG_E_ARG, "the input coordinates must be sorted");
    return set_lightcurves_common(ctx, N, L, t, t_per_lc, y, yerr, y_offset, hipMemcpyHostToDevice);
}

MTG_API int mtg_set_lightcurves_device(mtg_ctx *ctx, int64_t N, int64_t L, const double *d_t,
                                       int t_per_lc, const double *d_y, const double *d_yerr,
                                       const double *d_y_offset)
{
    return set_lightcurves_common(ctx, N, L, d_t, t_per_lc, d_y, d_yerr, d_y_offset, hipMemcpyDeviceToDevice);
}

MTG_API int mtg_set_model(mtg_ctx *ctx, int nterms, const int32_t *kinds, const double *term_extra,
                          int mean_kind, int PF, const double *full_values, int P,
                          const int32_t *free_index, const double *bounds)
{
    if (!ctx) return MTG_E_ARG;
    if (nterms <= 0 || nterms > MTG_MAX_TERMS || !kinds)
        return fail(ctx, MTG_E_ARG, "mtg_set_model: nterms must be in [1, %d]", MTG_MAX_TERMS);
    if (mean_kind != MTG_MEAN_CONSTANT && mean_kind != MTG_MEAN_LINEAR)
        return fail(ctx, MTG_E_ARG, "mtg_set_model: unknown mean kind %d", mean_kind);
    MtgModel m;
    memset(&m, 0, sizeof m);
    m.nterms = nterms;
    m.mean_kind = mean_kind;
    int off = 0;
    for (int i = 0; i < nterms; ++i) {
        const int np = nparams(kinds[i]);
        if (np < 0) return fail(ctx, MTG_E_ARG, "mtg_set_model: unknown term kind %d", kinds[i]);
        m.kinds[i] = kinds[i];
        m.poff[i] = off;
        m.extra[i] = term_extra ? term_extra[i] : 0.01;
        off += np;
        switch (kinds[i]) {
        case MTG_TERM_REAL: case MTG_TERM_DRW: m.nr0 += 1; break;
        case MTG_TERM_JITTER: break;
        case MTG_TERM_SHO: m.nc0 += 1; m.nsho += 1; break;
        default: m.nc0 += 1; break;
        }
    }
    m.nk = off;
    // the last complex slot: slots are handed out in term order, an over-damped SHOTerm takes none -- so it holds
    // the last term that is complex whatever its parameters, if that term comes after every SHOTerm
    for (int i = nterms - 1; i >= 0; --i) {
        const int kd = kinds[i];
        if (kd == MTG_TERM_REAL || kd == MTG_TERM_DRW || kd == MTG_TERM_JITTER) continue;
        m.last_b0 = kd == MTG_TERM_LORENTZIAN || kd == MTG_TERM_COMPLEX3 || kd == MTG_TERM_COSINUS;
        break;
    }
    const int nmean = mean_kind == MTG_MEAN_LINEAR ? 2 : 1;
    if (PF != off + nmean)
        return fail(ctx, MTG_E_ARG, "mtg_set_model: PF = %d but the terms + mean hold %d parameters",
                    PF, off + nmean);
    if (PF > MTG_MAX_PARAMS) return fail(ctx, MTG_E_ARG, "mtg_set_model: more than %d parameters", MTG_MAX_PARAMS);
    if (P < 0 || P > PF || (P > 0 && !free_index) || !full_values)
        return fail(ctx, MTG_E_ARG, "mtg_set_model: bad P / free_index / full_values");
    m.PF = PF;
    m.P = P;
    for (int k = 0; k < PF; ++k) {
        m.src[k] = -1;
        m.defaults[k] = full_values[k];
        m.lo[k] = bounds ? bounds[2 * k] : -INFINITY;
        m.hi[k] = bounds ? bounds[2 * k + 1] : INFINITY;
    }
    for (int i = 0; i < P; ++i) {
        if (free_index[i] < 0 || free_index[i] >= PF || m.src[free_index[i]] != -1)
            return fail(ctx, MTG_E_ARG, "mtg_set_model: free_index[%d] = %d invalid or repeated", i,
                        free_index[i]);
        m.src[free_index[i]] = i;
    }
    m.nr_max = m.nr0 + 2 * m.nsho;
    m.nc_max = m.nc0;
    for (int k = 0; k <= m.nsho; ++k) {
        const int nr = m.nr0 + 2 * k, nc = m.nc0 - k;
        if (!mtg_find_solver(nr, nc))
            return fail(ctx, MTG_E_UNSUPPORTED,
                        "mtg_set_model: no compiled kernel for %d real + %d complex terms (J = %d > %d?)",
                        nr, nc, nr + 2 * nc, MTG_MAX_J);
    }
    ctx->model = m;
    ctx->has_model = true;
    return MTG_OK;
}

MTG_API int mtg_loglike_batch_device(mtg_ctx *ctx, int64_t B, const double *d_theta,
                                     const int32_t *d_lc_index, int add_prior, double *d_out,
                                     int32_t *d_status, void *stream)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!d_out || !d_status || (!d_theta && ctx->model.P > 0))))
        return fail(ctx, MTG_E_ARG, "mtg_loglike_batch_device: bad arguments");
    if (B == 0) return MTG_OK;
    if (B > INT32_MAX) return fail(ctx, MTG_E_ARG, "batch too large");
    rc = use_device(ctx);
    if (rc) return rc;
    hipStream_t s = stream == MTG_STREAM_CONTEXT ? ctx->stream : (hipStream_t)stream;  // NULL: HIP's default stream
    rc = enter_stream(ctx, s);
    if (rc) return rc;
    ctx->lc_grouped_hint = 0;   // device-resident indices: the host cannot see their order
    rc = run_model_batch(ctx, B, d_theta, d_lc_index, add_prior, d_out, d_status, s);
    // whatever happened: kernels may already be queued on the caller's stream, and the next call on another stream
    // must wait for them before it touches the shared workspaces (the first error is the one reported)
    const std::string first_error = ctx->err;
    const int rc_leave = leave_stream(ctx, s);
    if (rc) { ctx->err = first_error; return rc; }
    return rc_leave;
}

MTG_API int mtg_loglike_batch(mtg_ctx *ctx, int64_t B, const double *theta, const int32_t *lc_index,
                              int add_prior, double *out, int32_t *status)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (B < 0 || (B > 0 && (!out || !status || (!theta && ctx->model.P > 0))))
        return fail(ctx, MTG_E_ARG, "mtg_loglike_batch: bad arguments");
    if (B == 0) return MTG_OK;
    if (B > INT32_MAX) return fail(ctx, MTG_E_ARG, "batch too large");
    // (the same pass tells whether the caller's order is already grouped by light curve: then the sweep keeps it)
    int64_t runs = 1;
    bool ascending = true;
    if (lc_index)
        for (int64_t b = 0; b < B; ++b) {
            if (lc_index[b] < 0 || lc_index[b] >= ctx->L)
                return fail(ctx, MTG_E_ARG, "lc_index[%lld] = %d outside [0, %lld)", (long long)b,
                            lc_index[b], (long long)ctx->L);
            if (b > 0 && lc_index[b] != lc_index[b - 1]) ++runs;
            if (b > 0 && lc_index[b] < lc_index[b - 1]) ascending = false;
        }
    rc = use_device(ctx);
    if (rc) return rc;
    // grouped: already in ascending order (sorting changes nothing), or in runs of equal indices long enough that a
    // wave of 64 lanes straddles two or three light curves at most
    ctx->lc_grouped_hint = !lc_index || ascending || B / runs >= 32;
    const int P = ctx->model.P;
    CTX_STREAM(ctx, s);
    HIP_TRY(ctx, ctx->theta.reserve((size_t)B * (P > 0 ? P : 1) * 8));
    HIP_TRY(ctx, ctx->out.reserve((size_t)B * 8));
    HIP_TRY(ctx, ctx->status.reserve((size_t)B * 4));
    if (P > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->theta.p, theta, (size_t)B * P * 8, hipMemcpyHostToDevice, s));
    const int32_t *d_lc = nullptr;
    if (lc_index) {
        HIP_TRY(ctx, ctx->lc.reserve((size_t)B * 4));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->lc.p, lc_index, (size_t)B * 4, hipMemcpyHostToDevice, s));
        d_lc = ctx->lc.as<int32_t>();
    }
    rc = run_model_batch(ctx, B, ctx->theta.as<double>(), d_lc, add_prior, ctx->out.as<double>(),
                         ctx->status.as<int32_t>(), s);
    if (rc) return rc;
    mtg_trace::Range range("mtg:gather (lnP, status -> host)");
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(status, ctx->status.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTG_OK;
}

MTG_API int mtg_loglike_coeffs(mtg_ctx *ctx, int64_t B, int jr, int jc, const double *a_real,
                               const double *c_real, const double *a_comp, const double *b_comp,
                               const double *c_comp, const double *d_comp, const double *jitter,
                               int mean_kind, const double *mean_params, const int32_t *lc_index,
                               double *out, int32_t *status)
{
    int rc = check_ready(ctx, false);
    if (rc) return rc;
    if (B < 0 || jr < 0 || jc < 0 || (B > 0 && (!out || !status)))
        return fail(ctx, MTG_E_ARG, "mtg_loglike_coeffs: bad arguments");
    if ((jr > 0 && (!a_real || !c_real)) || (jc > 0 && (!a_comp || !b_comp || !c_comp || !d_comp)))
        return fail(ctx, MTG_E_ARG, "mtg_loglike_coeffs: NULL coefficient array");
    if (mean_kind != MTG_MEAN_CONSTANT && mean_kind != MTG_MEAN_LINEAR)
        return fail(ctx, MTG_E_ARG, "mtg_loglike_coeffs: unknown mean kind %d", mean_kind);
    if (B == 0) return MTG_OK;
    if (B > INT32_MAX) return fail(ctx, MTG_E_ARG, "batch too large");
    mtg_solve_launcher fn = mtg_find_solver(jr, jc);
    if (!fn)
        return fail(ctx, MTG_E_UNSUPPORTED, "no compiled kernel for %d real + %d complex terms", jr, jc);
    if (lc_index)
        for (int64_t b = 0; b < B; ++b)
            if (lc_index[b] < 0 || lc_index[b] >= ctx->L)
                return fail(ctx, MTG_E_ARG, "lc_index[%lld] = %d outside [0, %lld)", (long long)b,
                            lc_index[b], (long long)ctx->L);
    rc = use_device(ctx);
    if (rc) return rc;
    MtgCoefLayout lay{jr, jc};
    rc = reserve_workspace(ctx, B, lay.nslots(), 1);
    if (rc) return rc;
    const int64_t cs = ctx->cstride;
    CTX_STREAM(ctx, s);
    // host-side transpose [B][j] -> SoA columns, then one upload
    const int nmean = mean_kind == MTG_MEAN_LINEAR ? 2 : 1;
    double *h = (double *)malloc((size_t)cs * lay.nslots() * 8);
    if (!h) return fail(ctx, MTG_E_ARG, "out of host memory");
    memset(h, 0, (size_t)cs * lay.nslots() * 8);
    for (int64_t b = 0; b < B; ++b) {
        double asum = jitter ? jitter[b] : 0.0;
        for (int j = 0; j < jr; ++j) {
            h[lay.ar(j) * cs + b] = a_real[b * jr + j];
            h[lay.cr(j) * cs + b] = c_real[b * jr + j];
            asum += a_real[b * jr + j];
        }
        for (int k = 0; k < jc; ++k) {
            h[lay.ac(k) * cs + b] = a_comp[b * jc + k];
            h[lay.bc(k) * cs + b] = b_comp[b * jc + k];
            h[lay.cc(k) * cs + b] = c_comp[b * jc + k];
            h[lay.dc(k) * cs + b] = d_comp[b * jc + k];
            asum += a_comp[b * jc + k];
        }
        h[lay.asum() * cs + b] = asum;
        h[lay.jit() * cs + b] = jitter ? jitter[b] : 0.0;
        // slots are (slope, intercept); a constant mean is slope 0
        const double m0 = mean_params ? mean_params[b * nmean] : 0.0;
        h[lay.mean(0) * cs + b] = nmean == 2 ? m0 : 0.0;
        h[lay.mean(1) * cs + b] = nmean == 2 ? mean_params[b * nmean + 1] : m0;
    }
    hipError_t e = hipMemcpyAsync(ctx->coef.p, h, (size_t)cs * lay.nslots() * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    free(h);
    if (e != hipSuccess) return fail(ctx, MTG_E_HIP, "coefficient upload failed: %s", hipGetErrorString(e));
    HIP_TRY(ctx, ctx->out.reserve((size_t)B * 8));
    HIP_TRY(ctx, ctx->status.reserve((size_t)B * 4));
    HIP_TRY(ctx, hipMemsetAsync(ctx->status.p, 0, (size_t)B * 4, s));
    const int32_t *d_lc = nullptr;
    if (lc_index) {
        HIP_TRY(ctx, ctx->lc.reserve((size_t)B * 4));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->lc.p, lc_index, (size_t)B * 4, hipMemcpyHostToDevice, s));
        d_lc = ctx->lc.as<int32_t>();
    }
    MtgSolveArgs sa;
    sa.coef = ctx->coef.as<double>();
    sa.cstride = cs;
    sa.lay = lay;
    sa.list = nullptr;
    sa.count_ptr = nullptr;
    sa.B = B;
    sa.lc_index = d_lc;
    sa.status = ctx->status.as<int32_t>();
    sa.out = ctx->out.as<double>();
    sa.dxt = ctx->dxt.as<double2>();
    sa.yv = ctx->yv.as<double2>();
    sa.N = ctx->N;
    sa.t_stride = ctx->t_per_lc ? ctx->N : 0;
    sa.dxmax = ctx->dxmax.as<double>();
    sa.yv_bytes = (uint64_t)ctx->L * (uint64_t)ctx->N * 16u;
    sa.dxt_bytes = (uint64_t)(ctx->t_per_lc ? ctx->L : 1) * (uint64_t)ctx->N * 16u;
    sa.window_bytes = ctx->window_bytes;
    sa.mean_kind = mean_kind;
    sa.has_mean = mean_params != nullptr || jitter != nullptr;
    sa.tp_ws = nullptr;
    sa.tp_chunks = 0;
    sa.tp_gsize = 0;
    sa.tp_direct = 0;
    sa.tp_nr0 = jr; sa.tp_nc0 = jc;
    sa.sig = nullptr;
    ctx->timed = true;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, s));
    rc = sweep_launch(ctx, fn, sa, B, 0, s);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, s));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(status, ctx->status.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTG_OK;
}

MTG_API int mtg_ensemble_init(mtg_ctx *ctx, int64_t E, int W, uint64_t seed, const double *coords,
                              const int32_t *lc_of_ensemble)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    const int P = ctx->model.P;
    if (E <= 0 || W < 2 || (W & 1) || !coords || P <= 0)
        return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: need E > 0, an even W >= 2, P > 0 and coords");
    if (W < 2 * P)
        return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: fewer walkers (%d) than twice the dimension (%d)", W, 2 * P);
    if (E * (int64_t)W > INT32_MAX) return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: too many walkers");
    if (W > MTG_MAX_WALKERS)
        return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: at most %d walkers per ensemble", MTG_MAX_WALKERS);
    if (!lc_of_ensemble && E != ctx->L && ctx->L != 1)
        return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: %lld ensembles but %lld light curves and no map",
                    (long long)E, (long long)ctx->L);
    rc = use_device(ctx);
    if (rc) return rc;
    const int64_t EW = E * W, EH = E * (W / 2);
    std::vector<int32_t> lc_full((size_t)EW), lc_half((size_t)EH), lc_spec((size_t)(3 * EH));
    for (int64_t e = 0; e < E; ++e) {
        const int32_t l = lc_of_ensemble ? lc_of_ensemble[e] : (ctx->L == 1 ? 0 : (int32_t)e);
        if (l < 0 || l >= ctx->L) return fail(ctx, MTG_E_ARG, "mtg_ensemble_init: light curve %d out of range", l);
        for (int w = 0; w < W; ++w) lc_full[(size_t)(e * W + w)] = l;
        for (int k = 0; k < W / 2; ++k)
            lc_half[(size_t)(e * (W / 2) + k)] = lc_spec[(size_t)(e * (W / 2) + k)] = lc_spec[(size_t)(EH + e * (W / 2) + k)] =
                lc_spec[(size_t)(2 * EH + e * (W / 2) + k)] = l;
    }
    CTX_STREAM(ctx, s);
    HIP_TRY(ctx, ctx->ens_coords.reserve((size_t)EW * P * 8));
    HIP_TRY(ctx, ctx->ens_lnp.reserve((size_t)EW * 8));
    HIP_TRY(ctx, ctx->ens_perm.reserve((size_t)EW * 4));
    // (proposals, their factors, log-probabilities and statuses: room for the 3 E W/2 rows of a speculative iteration)
    HIP_TRY(ctx, ctx->ens_q.reserve((size_t)3 * EH * P * 8));
    HIP_TRY(ctx, ctx->ens_factor.reserve((size_t)2 * EH * 8));
    HIP_TRY(ctx, ctx->ens_new.reserve((size_t)3 * EH * 8));
    HIP_TRY(ctx, ctx->ens_st.reserve((size_t)3 * EH * 4));
    HIP_TRY(ctx, ctx->ens_lc_spec.reserve((size_t)3 * EH * 4));
    HIP_TRY(ctx, ctx->ens_lc_full.reserve((size_t)EW * 4));
    HIP_TRY(ctx, ctx->ens_lc_half.reserve((size_t)EH * 4));
    HIP_TRY(ctx, ctx->ens_naccept.reserve((size_t)EW * 4));
    HIP_TRY(ctx, ctx->ens_best_lnp.reserve((size_t)E * 8));
    HIP_TRY(ctx, ctx->ens_best_coords.reserve((size_t)E * P * 8));
    HIP_TRY(ctx, ctx->ens_notpd.reserve(64));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_coords.p, coords, (size_t)EW * P * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_lc_full.p, lc_full.data(), (size_t)EW * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_lc_half.p, lc_half.data(), (size_t)EH * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_lc_spec.p, lc_spec.data(), (size_t)3 * EH * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->ens_naccept.p, 0, (size_t)EW * 4, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->ens_notpd.p, 0, 4, s));
    // log-probability of the initial state (emcee evaluates p0 once); the rows are grouped by ensemble, hence by light curve
    ctx->lc_grouped_hint = 1;
    rc = run_model_batch(ctx, EW, ctx->ens_coords.as<double>(), ctx->ens_lc_full.as<int32_t>(), 1,
                         ctx->ens_lnp.as<double>(), ctx->ens_st.as<int32_t>(), s);
    if (rc) return rc;
    mtg_launch_initial_best((int)E, W, P, ctx->ens_coords.as<double>(), ctx->ens_lnp.as<double>(),
                            ctx->ens_best_lnp.as<double>(), ctx->ens_best_coords.as<double>(), s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));  // lc_full / lc_half live on this stack frame
    ctx->ens_E = E; ctx->ens_W = W; ctx->ens_P = P; ctx->ens_seed = seed; ctx->ens_iteration = 0; ctx->ens_base = ctx->stream_base;
    ctx->ens_L = ctx->L; ctx->ens_N = ctx->N;
    shard_release(ctx);  // a new set of ensembles starts unsharded (mtg_ensemble_shard_* after this call)
    return MTG_OK;
}

// ---------------------------------------------------------------------------
// Walker sharding of the device-resident ensembles (replaces the reference's
// multiprocessing.Pool.map per half-step, gpmodelling.py:245-248)
// ---------------------------------------------------------------------------
// Every rank runs the whole sampler -- same Philox key, same proposals, same accept step -- but evaluates
// only rows [rank * chunk, (rank + 1) * chunk) of each half-step's proposals; one all-gather of the
// log-probabilities (8 bytes per walker, plus the status word) on the launch stream brings the rest
// before the accept kernel.  RCCL is looked up at run time: a process that has PyTorch loaded must use
// PyTorch's copy of librccl.so.1 (two copies in one process is asking for trouble), and a library
// linked against /opt/rocm's would bring that one in first.
namespace {

struct Id128 { char b[128]; };  // ncclUniqueId
struct Rccl {
    void *lib = nullptr;
    // (ncclComm_t, ncclUniqueId, ncclDataType_t of rccl.h, restated as plain types)
    int (*GetUniqueId)(void *id128) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, Id128 id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommCount)(void *comm, int *count) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
} g_rccl;
enum { RCCL_INT32 = 2, RCCL_FLOAT64 = 8 };  // ncclInt32, ncclFloat64

bool rccl_load(const char *path)
{
    if (g_rccl.lib) return true;
    void *h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process (PyTorch's)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        g_rccl.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
        return false;
    }
    auto sym = [&](const char *n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GroupStart ||
        !g_rccl.GroupEnd || !g_rccl.GetErrorString) {
        g_rccl.why = "librccl.so.1 lacks one of the nccl* entry points";
        return false;
    }
    g_rccl.lib = h;
    return true;
}

#define RCCL_TRY(ctx, call)                                                                          \
    do {                                                                                             \
        int r__ = (call);                                                                            \
        if (r__ != 0) return fail((ctx), MTG_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); \
    } while (0)

void shard_release(mtg_ctx *ctx)
{
    if (ctx->shard_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->shard_comm);
    if (ctx->shard_h_lnp) (void)hipHostFree(ctx->shard_h_lnp);
    if (ctx->shard_h_st) (void)hipHostFree(ctx->shard_h_st);
    ctx->shard_comm = nullptr; ctx->shard_h_lnp = nullptr; ctx->shard_h_st = nullptr; ctx->shard_h_rows = 0;
    ctx->shard_kind = 0; ctx->shard_rank = 0; ctx->shard_world = 1;
    ctx->shard_generation.fetch_add(1);
    ctx->shard_chunk = ctx->shard_lo = ctx->shard_hi = 0;
    ctx->shard_fn = nullptr; ctx->shard_user = nullptr;
}

// rows of the half-step batch this rank evaluates, and buffers large enough for the padded all-gather
int shard_layout(mtg_ctx *ctx, int rank, int world)
{
    if (ctx->ens_E <= 0) return fail(ctx, MTG_E_STATE, "mtg_ensemble_init has not been called");
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, MTG_E_ARG, "mtg_ensemble_shard: rank %d of %d", rank, world);
    const int64_t EH = ctx->ens_E * (ctx->ens_W / 2);
    const int64_t chunk = (EH + world - 1) / world;
    int rc = use_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    shard_release(ctx);
    // (DevBuf::reserve keeps the contents only when it does not grow: the buffers hold nothing between runs)
    HIP_TRY(ctx, ctx->ens_new.reserve((size_t)std::max<int64_t>(ctx->ens_E * ctx->ens_W, chunk * world) * 8));
    HIP_TRY(ctx, ctx->ens_st.reserve((size_t)std::max<int64_t>(ctx->ens_E * ctx->ens_W, chunk * world) * 4));
    ctx->shard_rank = rank; ctx->shard_world = world; ctx->shard_chunk = chunk;
    ctx->shard_lo = std::min<int64_t>((int64_t)rank * chunk, EH);
    ctx->shard_hi = std::min<int64_t>(ctx->shard_lo + chunk, EH);
    return MTG_OK;
}

// after the solve of a half-step: everybody's log-probabilities and status words into ens_new / ens_st
int shard_exchange(mtg_ctx *ctx, int64_t EH, hipStream_t s)
{
    double *lnp = ctx->ens_new.as<double>();
    int32_t *st = ctx->ens_st.as<int32_t>();
    const int64_t chunk = ctx->shard_chunk, lo = ctx->shard_lo, hi = ctx->shard_hi;
    if (ctx->shard_kind == 1) {
        mtg_trace::Range range("mtg:all-gather of the half-step's log-probabilities (RCCL)");
        // in place: this rank's block already sits at rank * chunk of the receive buffer
        const int64_t at = (int64_t)ctx->shard_rank * chunk;
        const bool timed = ctx->shard_ev_n < ctx->shard_ev_cap;
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->shard_ev[2 * (size_t)ctx->shard_ev_n], s));
        struct Stamp {  // the closing event, whatever way the block is left
            mtg_ctx *c; hipStream_t st; bool on;
            ~Stamp() { if (on) { (void)hipEventRecord(c->shard_ev[2 * (size_t)c->shard_ev_n + 1], st); c->shard_ev_n += 1; } }
        } stamp{ctx, s, timed};
        RCCL_TRY(ctx, g_rccl.GroupStart());
        int r1 = g_rccl.AllGather(lnp + at, lnp, (size_t)chunk, RCCL_FLOAT64, ctx->shard_comm, s);
        int r2 = r1 ? r1 : g_rccl.AllGather(st + at, st, (size_t)chunk, RCCL_INT32, ctx->shard_comm, s);
        const int r3 = g_rccl.GroupEnd();   // (always closed, whatever the calls inside it said)
        if (r2 || r3)
            return fail(ctx, MTG_E_HIP, "ncclAllGather of the half-step's log-probabilities failed: %s",
                        g_rccl.GetErrorString(r2 ? r2 : r3));
        return MTG_OK;
    }
    // host callback: stage this rank's rows, let the caller fill in the others, upload everything
    mtg_trace::Range range("mtg:exchange of the half-step's log-probabilities (host callback)");
    if (ctx->shard_h_rows < EH) {
        if (ctx->shard_h_lnp) (void)hipHostFree(ctx->shard_h_lnp);
        if (ctx->shard_h_st) (void)hipHostFree(ctx->shard_h_st);
        ctx->shard_h_lnp = nullptr; ctx->shard_h_st = nullptr; ctx->shard_h_rows = 0;
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->shard_h_lnp, (size_t)EH * 8));
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->shard_h_st, (size_t)EH * 4));
        ctx->shard_h_rows = EH;
    }
    if (hi > lo) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->shard_h_lnp + lo, lnp + lo, (size_t)(hi - lo) * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->shard_h_st + lo, st + lo, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    const int rc = ctx->shard_fn(ctx->shard_user, ctx->shard_h_lnp, ctx->shard_h_st, EH, lo, hi);
    if (rc) return fail(ctx, MTG_E_STATE, "the exchange callback of the walker-sharded ensemble returned %d", rc);
    HIP_TRY(ctx, hipMemcpyAsync(lnp, ctx->shard_h_lnp, (size_t)EH * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->shard_h_st, (size_t)EH * 4, hipMemcpyHostToDevice, s));
    return MTG_OK;
}

}  // namespace

MTG_API int mtg_rccl_load(const char *path)
{
    return rccl_load(path) ? MTG_OK : MTG_E_UNSUPPORTED;
}

MTG_API int mtg_rccl_unique_id(void *id128)
{
    if (!id128) return MTG_E_ARG;
    if (!rccl_load(nullptr)) return MTG_E_UNSUPPORTED;
    return g_rccl.GetUniqueId(id128) == 0 ? MTG_OK : MTG_E_HIP;
}

MTG_API int mtg_ensemble_shard_rccl(mtg_ctx *ctx, const void *id128, int rank, int world)
{
    if (!ctx || !id128) return MTG_E_ARG;
    if (!rccl_load(nullptr)) return fail(ctx, MTG_E_UNSUPPORTED, "%s", g_rccl.why.c_str());
    int rc = shard_layout(ctx, rank, world);
    if (rc) return rc;
    Id128 id;
    memcpy(id.b, id128, sizeof id.b);
    void *comm = nullptr;
    const int generation = ctx->shard_generation.load();
    RCCL_TRY(ctx, g_rccl.CommInitRank(&comm, world, id, rank));
    if (ctx->shard_generation.load() != generation) {
        // ncclCommInitRank took so long that the caller gave up and (un)sharded the context another way meanwhile
        // (distributed.shard_device_ensemble's fall-back to the host-staged exchange): this communicator is nobody's
        (void)g_rccl.CommDestroy(comm);
        return fail(ctx, MTG_E_STATE, "mtg_ensemble_shard_rccl: the context was re-sharded while ncclCommInitRank was running");
    }
    ctx->shard_comm = comm;
    ctx->shard_kind = 1;
    return MTG_OK;
}

MTG_API int mtg_ensemble_shard_host(mtg_ctx *ctx, int rank, int world, mtg_exchange_fn fn, void *user)
{
    if (!ctx || !fn) return MTG_E_ARG;
    int rc = shard_layout(ctx, rank, world);
    if (rc) return rc;
    ctx->shard_fn = fn;
    ctx->shard_user = user;
    ctx->shard_kind = 2;
    return MTG_OK;
}

MTG_API int mtg_ensemble_shard_info(const mtg_ctx *ctx, int *kind, int *rank, int *world, int *comm_ranks)
{
    if (!ctx) return MTG_E_ARG;
    if (kind) *kind = ctx->shard_kind;
    if (rank) *rank = ctx->shard_rank;
    if (world) *world = ctx->shard_world;
    if (comm_ranks) {
        *comm_ranks = 0;
        if (ctx->shard_kind == 1 && ctx->shard_comm && g_rccl.CommCount) (void)g_rccl.CommCount(ctx->shard_comm, comm_ranks);
    }
    return MTG_OK;
}

MTG_API int mtg_ensemble_shard_profile(mtg_ctx *ctx, int capacity)
{
    if (!ctx || capacity < 0) return MTG_E_ARG;
    int rc = use_device(ctx);
    if (rc) return rc;
    while ((int)ctx->shard_ev.size() < 2 * capacity) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->shard_ev.push_back(e);
    }
    ctx->shard_ev_cap = capacity;
    ctx->shard_ev_n = 0;
    return MTG_OK;
}

MTG_API int mtg_ensemble_shard_profile_read(mtg_ctx *ctx, int capacity, double *exchange_ms)
{
    if (!ctx || capacity < 0) return MTG_E_ARG;
    const int n = ctx->shard_ev_n < capacity ? ctx->shard_ev_n : capacity;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(ctx->shard_ev[2 * (size_t)i + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->shard_ev[2 * (size_t)i], ctx->shard_ev[2 * (size_t)i + 1]));
        if (exchange_ms) exchange_ms[i] = ms;
    }
    ctx->shard_ev_cap = 0;
    return n;
}

MTG_API int mtg_ensemble_unshard(mtg_ctx *ctx)
{
    if (!ctx) return MTG_E_ARG;
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    shard_release(ctx);
    return MTG_OK;
}

MTG_API int mtg_ensemble_run(mtg_ctx *ctx, int steps, double *chain, double *lnp_chain)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (ctx->ens_E <= 0) return fail(ctx, MTG_E_STATE, "mtg_ensemble_init has not been called");
    if (steps < 0) return fail(ctx, MTG_E_ARG, "mtg_ensemble_run: negative step count");
    if (ctx->ens_P != ctx->model.P) return fail(ctx, MTG_E_STATE, "the model changed since mtg_ensemble_init");
    if (ctx->ens_L != ctx->L || ctx->ens_N != ctx->N)
        return fail(ctx, MTG_E_STATE, "the resident light curves changed shape since mtg_ensemble_init");
    rc = use_device(ctx);
    if (rc) return rc;
    const int E = (int)ctx->ens_E, W = ctx->ens_W, P = ctx->ens_P, H = W / 2;
    const int64_t EW = (int64_t)E * W, EH = (int64_t)E * H;
    CTX_STREAM(ctx, s);
    mtg_trace::Range range("mtg:ensemble_run (stretch moves, device resident)");
    if (chain) HIP_TRY(ctx, ctx->ens_chain.reserve((size_t)steps * EW * P * 8));
    if (lnp_chain) HIP_TRY(ctx, ctx->ens_lnp_chain.reserve((size_t)steps * EW * 8));
    rc = check_model_workspace(ctx, EH);
    if (rc) return rc;
    // Between two solves ONE launch does the accept step of the half-step just evaluated and the proposals (with
    // their expansion) of the next: mtg_sampler_step_kernel.  The structure lists have two banks: the proposals of
    // half-step h + 1 are appended to one while workgroup 0 clears the counters of the other, which the solver of
    // half-step h has just used.
    HIP_TRY(ctx, hipMemsetAsync(ctx->counts.p, 0, 2 * 64 * sizeof(int), s));
    struct BankGuard {  // everybody else uses bank 0
        mtg_ctx *c;
        ~BankGuard() { c->bank = 0; }
    } bank_guard{ctx};
    const bool sharded = ctx->shard_kind != 0;
    struct LiveRows {  // the solver's kernel choice looks at the rows this rank evaluates
        mtg_ctx *c;
        LiveRows(mtg_ctx *ctx_, int64_t n) : c(ctx_) { c->live_rows = n; }
        ~LiveRows() { c->live_rows = 0; }
    } live_rows(ctx, sharded ? (ctx->shard_hi > ctx->shard_lo ? ctx->shard_hi - ctx->shard_lo : 1) : 0);
    auto prep_args = [&](int bank) {
        ctx->bank = bank;
        MtgPrepArgs pa = make_prep_args(ctx, EH, ctx->ens_q.as<double>(), 1, ctx->ens_new.as<double>(), ctx->ens_st.as<int32_t>());
        if (sharded) {
            pa.row_lo = ctx->shard_lo;
            pa.row_hi = ctx->shard_hi;
        }
        return pa;
    };
    MtgEnsembleArgs g;
    g.E = E; g.W = W; g.P = P;
    g.e_base = (uint32_t)ctx->ens_base;
    g.seed_lo = (uint32_t)ctx->ens_seed; g.seed_hi = (uint32_t)(ctx->ens_seed >> 32);
    g.a = 2.0;
    g.perm = ctx->ens_perm.as<int32_t>();
    g.coords = ctx->ens_coords.as<double>();
    g.lnp = ctx->ens_lnp.as<double>();
    g.factor = ctx->ens_factor.as<double>();
    g.naccept = ctx->ens_naccept.as<int32_t>();
    g.best_lnp = ctx->ens_best_lnp.as<double>();
    g.best_coords = ctx->ens_best_coords.as<double>();
    g.n_notpd = ctx->ens_notpd.as<int32_t>();
    int bank = 0;
    // A small ensemble leaves most of the GPU idle and its solve takes as long for 3 H rows as for H: both half-steps of
    // an iteration then go into one batch (mtg_sampler.hip: speculative iteration).  Where: the time-parallel kernels
    // with every row on a workgroup of its own in one occupancy round -- 256 workgroups of four waves for long light
    // curves (one per CU: their elements fill the LDS; two per CU up to rank 3, and for ranks 4 and 5 with two waves
    // each), 1024 single-wave ones for short.  Same chain either way where both forms run the same kernel.
    const int Jmodel = ctx->model.nr0 + 2 * ctx->model.nc0;
    const int64_t rows3 = 3 * EH;
    const bool spec = steps > 0 && ctx->spec_mode != 0 && !sharded && ctx->tp_mode != 0 && Jmodel <= 6 && ctx->N >= 256 &&
                      rows3 <= (ctx->N >= 4096 ? (Jmodel <= 5 ? 512 : 256) : 1024);
    if (spec) {
        rc = check_model_workspace(ctx, rows3);
        if (rc) return rc;
        auto prep3 = [&](int bank_) {
            ctx->bank = bank_;
            return make_prep_args(ctx, rows3, ctx->ens_q.as<double>(), 1, ctx->ens_new.as<double>(), ctx->ens_st.as<int32_t>());
        };
        // the splits of the whole run in one launch over the whole GPU, where they are small (an ensemble or a few): the
        // sampler kernel of an iteration is one workgroup's chain of latencies and ranking W keys is 2-5 us of it
        int32_t *perm_all = nullptr;
        const size_t perm_bytes = (size_t)steps * EW * sizeof(int32_t);
        if (perm_bytes <= ((size_t)64 << 20) && steps <= 65535 && ctx->spec_mode != 2) {
            HIP_TRY(ctx, ctx->ens_perm_all.reserve(perm_bytes));
            perm_all = ctx->ens_perm_all.as<int32_t>();
            mtg_launch_split_all(g, ctx->ens_iteration, steps, perm_all, s);
        }
        g.perm_next = perm_all;   // (slice 0: the iteration the first launch proposes)
        mtg_launch_sampler_spec(g, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, ctx->ens_iteration, prep3(bank), s);
        for (int it = 0; it < steps; ++it) {
            const uint32_t iter = ctx->ens_iteration;
            ctx->bank = bank;
            if (perm_all) {
                g.perm = perm_all + (size_t)it * EW;                                   // this iteration's split, for the accept step
                g.perm_next = it + 1 < steps ? perm_all + (size_t)(it + 1) * EW : nullptr;  // the next one's, for its proposals
            }
            rc = solve_prepared(ctx, rows3, ctx->ens_lc_spec.as<int32_t>(), ctx->ens_new.as<double>(), ctx->ens_st.as<int32_t>(), s);
            if (rc) return rc;
            const bool more = it + 1 < steps;
            int *used_counts = bank_counts(ctx);
            mtg_launch_sampler_spec(g, 1, iter, ctx->ens_new.as<double>(), ctx->ens_st.as<int32_t>(), used_counts,
                                    chain ? ctx->ens_chain.as<double>() + (size_t)it * EW * P : nullptr,
                                    lnp_chain ? ctx->ens_lnp_chain.as<double>() + (size_t)it * EW : nullptr, more ? 1 : 0, iter + 1,
                                    prep3(bank ^ 1), s);
            bank ^= 1;
            ctx->ens_iteration += 1;
        }
    }
    if (steps > 0 && !spec)  // the first proposals of the run
        mtg_launch_sampler_step(g, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, ctx->ens_iteration, prep_args(bank), s);
    for (int it = 0; it < steps && !spec; ++it) {
        const uint32_t iter = ctx->ens_iteration;
        for (int half = 0; half < 2; ++half) {
            ctx->bank = bank;
            rc = solve_prepared(ctx, EH, ctx->ens_lc_half.as<int32_t>(), ctx->ens_new.as<double>(),
                                ctx->ens_st.as<int32_t>(), s);
            if (rc) return rc;
            if (sharded) {
                rc = shard_exchange(ctx, EH, s);
                if (rc) return rc;
            }
            const bool last = half == 1;
            const bool more = !(last && it + 1 == steps);   // another half-step follows in this call
            int *used_counts = bank_counts(ctx);
            mtg_launch_sampler_step(g, 1, half, iter, ctx->ens_new.as<double>(), ctx->ens_st.as<int32_t>(), used_counts,
                                    last && chain ? ctx->ens_chain.as<double>() + (size_t)it * EW * P : nullptr,
                                    last && lnp_chain ? ctx->ens_lnp_chain.as<double>() + (size_t)it * EW : nullptr,
                                    more ? 1 : 0, last ? 0 : 1, last ? iter + 1 : iter, prep_args(bank ^ 1), s);
            bank ^= 1;
        }
        ctx->ens_iteration += 1;
    }
    HIP_TRY(ctx, hipGetLastError());
    if (chain)
        HIP_TRY(ctx, hipMemcpyAsync(chain, ctx->ens_chain.p, (size_t)steps * EW * P * 8, hipMemcpyDeviceToHost, s));
    if (lnp_chain)
        HIP_TRY(ctx, hipMemcpyAsync(lnp_chain, ctx->ens_lnp_chain.p, (size_t)steps * EW * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTG_OK;
}

MTG_API int mtg_ensemble_restore(mtg_ctx *ctx, int64_t iteration, const double *lnp, const int32_t *naccept,
                                 const double *best_lnp, const double *best_coords)
{
    if (!ctx) return MTG_E_ARG;
    if (ctx->ens_E <= 0) return fail(ctx, MTG_E_STATE, "mtg_ensemble_init has not been called");
    if (iteration < 0 || iteration > 0xffffffffll) return fail(ctx, MTG_E_ARG, "mtg_ensemble_restore: iteration out of range");
    int rc = use_device(ctx);
    if (rc) return rc;
    const int64_t E = ctx->ens_E, EW = E * ctx->ens_W;
    CTX_STREAM(ctx, s);
    // the saved log-probabilities as they are: mtg_ensemble_init has just evaluated the saved coordinates again, but
    // in ONE batch of E W rows, where the run evaluated them in half-steps of E W/2 (or a rank's share of them) -- the
    // kernel and its summation order follow the row count, so those values may differ in the last bits and flip an
    // accept decision of the continued chain
    if (lnp) HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_lnp.p, lnp, (size_t)EW * 8, hipMemcpyHostToDevice, s));
    if (naccept) HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_naccept.p, naccept, (size_t)EW * 4, hipMemcpyHostToDevice, s));
    if (best_lnp) HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_best_lnp.p, best_lnp, (size_t)E * 8, hipMemcpyHostToDevice, s));
    if (best_coords)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->ens_best_coords.p, best_coords, (size_t)E * ctx->ens_P * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->ens_iteration = (uint32_t)iteration;
    return MTG_OK;
}

MTG_API int mtg_ensemble_get(mtg_ctx *ctx, double *coords, double *lnp, double *best_lnp, double *best_coords,
                             int32_t *naccept, int64_t *iteration, int32_t *n_notpd)
{
    if (!ctx) return MTG_E_ARG;
    if (ctx->ens_E <= 0) return fail(ctx, MTG_E_STATE, "mtg_ensemble_init has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    const int64_t E = ctx->ens_E, EW = E * ctx->ens_W;
    const int P = ctx->ens_P;
    CTX_STREAM(ctx, s);
    if (coords) HIP_TRY(ctx, hipMemcpyAsync(coords, ctx->ens_coords.p, (size_t)EW * P * 8, hipMemcpyDeviceToHost, s));
    if (lnp) HIP_TRY(ctx, hipMemcpyAsync(lnp, ctx->ens_lnp.p, (size_t)EW * 8, hipMemcpyDeviceToHost, s));
    if (best_lnp) HIP_TRY(ctx, hipMemcpyAsync(best_lnp, ctx->ens_best_lnp.p, (size_t)E * 8, hipMemcpyDeviceToHost, s));
    if (best_coords)
        HIP_TRY(ctx, hipMemcpyAsync(best_coords, ctx->ens_best_coords.p, (size_t)E * P * 8, hipMemcpyDeviceToHost, s));
    if (naccept) HIP_TRY(ctx, hipMemcpyAsync(naccept, ctx->ens_naccept.p, (size_t)EW * 4, hipMemcpyDeviceToHost, s));
    if (n_notpd) HIP_TRY(ctx, hipMemcpyAsync(n_notpd, ctx->ens_notpd.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (iteration) *iteration = ctx->ens_iteration;
    return MTG_OK;
}

MTG_API int mtg_fft_warmup(mtg_ctx *ctx)
{
    // HIP's current device is per THREAD: a helper thread starts on device 0 whatever the context's is
    if (ctx && hipSetDevice(ctx->device) != hipSuccess) return MTG_E_HIP;
    // hipFFT needs ~1.4 s the first time a plan is made in a process (rocFFT loads its kernels): callers that will
    // need mtg_chain_autocorr or mtg_simulate_tk95 later can pay that early, from another thread
    std::lock_guard<std::mutex> plans(g_fft_plan_mu);
    hipfftHandle plan = 0;
    if (hipfftPlan1d(&plan, 64, HIPFFT_D2Z, 1) != HIPFFT_SUCCESS) return MTG_E_HIP;
    (void)hipfftDestroy(plan);
    return MTG_OK;
}

MTG_API int mtg_chain_autocorr(mtg_ctx *ctx, int64_t n_t, int64_t E, int W, int P, const double *chain, double *rho)
{
    if (!ctx) return MTG_E_ARG;
    if (n_t < 2 || E < 1 || W < 1 || P < 1 || !chain || !rho) return fail(ctx, MTG_E_ARG, "mtg_chain_autocorr: bad arguments");
    int64_t n = 1;
    while (n < n_t) n *= 2;
    const int64_t n2 = 2 * n, nk = n + 1, S = E * (int64_t)W * P, EP = E * P;
    if (n2 > ((int64_t)1 << 30) || S > ((int64_t)1 << 24) || n2 * S > ((int64_t)1 << 32))
        return fail(ctx, MTG_E_ARG, "mtg_chain_autocorr: chain too large");
    int rc = use_device(ctx);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    mtg_trace::Range range("mtg:chain_autocorr (convergence check)");
    DevBuf &d_chain = ctx->acf_chain, &d_x = ctx->acf_x, &d_f = ctx->acf_f, &d_g = ctx->acf_g, &d_r = ctx->acf_r, &d_ss = ctx->acf_ss;
    HIP_TRY(ctx, d_chain.reserve((size_t)n2 * S * 8));        // the chain, then the transposed series [S][n2]
    HIP_TRY(ctx, d_x.reserve((size_t)n2 * S * 8));            // centred and padded, [n2][S]
    HIP_TRY(ctx, d_f.reserve((size_t)nk * S * 16));
    HIP_TRY(ctx, d_g.reserve((size_t)nk * EP * 16));
    HIP_TRY(ctx, d_r.reserve((size_t)(n2 + n_t) * EP * 8));   // inverse transforms [EP][n2], then rho [n_t][EP]
    HIP_TRY(ctx, d_ss.reserve((size_t)S * 8));
    HIP_TRY(ctx, ctx->acf_tmp.reserve((size_t)((n2 / 256 + 2) * S) * 8));
    HIP_TRY(ctx, hipMemcpyAsync(d_chain.p, chain, (size_t)n_t * S * 8, hipMemcpyHostToDevice, s));
    mtg_launch_acf_center(n_t, n2, S, d_chain.as<double>(), d_x.as<double>(), d_ss.as<double>(), ctx->acf_tmp.as<double>(), s);
    mtg_launch_acf_transpose(n2, S, d_x.as<double>(), d_chain.as<double>(), s);
    // contiguous batched transforms (stock kernels: no run-time compilation inside rocFFT)
    mtg_ctx::AcfPlans *slot = nullptr;
    for (auto &sl : ctx->acf_slots)
        if (sl.have && sl.n2 == n2 && sl.S == S && sl.P == EP) slot = &sl;
    if (!slot) {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        slot = &ctx->acf_slots[0];
        for (auto &sl : ctx->acf_slots)  // an empty slot, else the least recently used
            if (!sl.have || (slot->have && sl.used < slot->used)) slot = &sl;
        if (slot->have) { (void)hipfftDestroy(slot->fwd); (void)hipfftDestroy(slot->inv); slot->have = false; }
        int len = (int)n2;
        if (hipfftPlanMany(&slot->fwd, 1, &len, nullptr, 1, (int)n2, nullptr, 1, (int)nk, HIPFFT_D2Z, (int)S) != HIPFFT_SUCCESS)
            return fail(ctx, MTG_E_HIP, "mtg_chain_autocorr: hipfftPlanMany (forward) failed");
        if (hipfftPlanMany(&slot->inv, 1, &len, nullptr, 1, (int)nk, nullptr, 1, (int)n2, HIPFFT_Z2D, (int)EP) != HIPFFT_SUCCESS) {
            (void)hipfftDestroy(slot->fwd);
            return fail(ctx, MTG_E_HIP, "mtg_chain_autocorr: hipfftPlanMany (inverse) failed");
        }
        slot->have = true; slot->n2 = n2; slot->S = S; slot->P = EP;
        ctx->acf_plans_built += 1;
        if (hipfftSetStream(slot->fwd, s) != HIPFFT_SUCCESS || hipfftSetStream(slot->inv, s) != HIPFFT_SUCCESS)
            return fail(ctx, MTG_E_HIP, "mtg_chain_autocorr: hipfftSetStream failed");
    }
    slot->used = ++ctx->acf_clock;
    if (hipfftExecD2Z(slot->fwd, d_chain.as<double>(), (hipfftDoubleComplex *)d_f.p) != HIPFFT_SUCCESS)
        return fail(ctx, MTG_E_HIP, "mtg_chain_autocorr: hipfftExecD2Z failed");
    mtg_launch_acf_power(nk, E, W, P, d_f.as<double2>(), d_ss.as<double>(), d_g.as<double2>(), s);
    if (hipfftExecZ2D(slot->inv, (hipfftDoubleComplex *)d_g.p, d_r.as<double>()) != HIPFFT_SUCCESS)
        return fail(ctx, MTG_E_HIP, "mtg_chain_autocorr: hipfftExecZ2D failed");
    double *d_rho = d_r.as<double>() + n2 * EP;
    mtg_launch_acf_out(n_t, n2, EP, 1.0 / (double)n2, d_r.as<double>(), d_rho, s);   // hipFFT does not normalise
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(rho, d_rho, (size_t)n_t * EP * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTG_OK;
}

// transforms per execution of the simulator's hipFFT plan (lengths hipFFT transforms natively; the others take the
// chirp-z path below): a function of the length alone for a full call, so that one plan serves every such call (16
// transforms of 10^6 points fill the GPU; short transforms are batched by the hundred; MTG_SIM_BATCH overrides, for
// measurements) -- as long as the spectrum and series buffers of one execution, 16 nk + 8 nfft bytes per transform, stay
// within 2 GiB.  A call of fewer series than that (Simulator.generate_lightcurve() asks for ONE) gets a plan of its own
// size in the context's second slot: no transforms of empty slots, and a native plan costs milliseconds to build.
static int sim_batch_for(int64_t nfft, int64_t S = INT64_MAX)
{
    int64_t b = ((int64_t)1 << 24) / nfft;
    b = b < 16 ? 16 : b > 256 ? 256 : b;
    if (const char *env = mtg_measure_env("MTG_SIM_BATCH")) b = atoi(env) > 0 ? atoi(env) : b;   // MTG_MEASURE builds only
    const int64_t fit = ((int64_t)1 << 31) / (16 * (nfft / 2 + 1) + 8 * nfft);
    if (b > fit) b = fit;
    if (b > S) b = S;   // fewer series than a full batch: no transforms of empty slots
    return (int)(b < 1 ? 1 : b);
}

// the context's C2R plan of length nfft for a call of S series (made, or remade for another length or batch, under
// sim_mu); no fail(): may run on a helper thread
static int sim_plan_get(mtg_ctx *ctx, int64_t nfft, int64_t S, hipfftHandle *plan)
{
    std::lock_guard<std::mutex> lock(ctx->sim_mu);
    const int batch = sim_batch_for(nfft, S);
    mtg_ctx::SimPlan &sp = ctx->sim_plans[batch == sim_batch_for(nfft) ? 0 : 1];
    if (!(sp.have && sp.nfft == nfft && sp.batch == batch)) {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        if (sp.have) { (void)hipfftDestroy(sp.h); sp.have = false; }
        if (hipfftPlan1d(&sp.h, (int)nfft, HIPFFT_Z2D, batch) != HIPFFT_SUCCESS) return MTG_E_HIP;
        sp.have = true;
        sp.nfft = nfft;
        sp.batch = batch;
    }
    if (plan) *plan = sp.h;
    return MTG_OK;
}

// ---- the chirp-z path (mtg_simulate.hip) ----
static int64_t czt_length(int64_t nfft)
{
    int64_t m = 1;
    while (m < 2 * nfft - 1) m <<= 1;
    return m;
}
// lengths hipFFT transforms natively (radices 2 .. 13) keep its Z2D plan; anything with a larger prime factor goes through
// power-of-two transforms -- while one pair's work area (16 m bytes) stays within 2 GiB.  mtg_set_simulate_transform
// forces one or the other (mode 1: the library's plan, 2: chirp-z).
static bool sim_wants_czt(const mtg_ctx *ctx, int64_t nfft)
{
    if (ctx->sim_transform == 1) return false;
    if (ctx->sim_transform == 2) return czt_length(nfft) * 16 <= ((int64_t)1 << 31);
    int64_t r = nfft;
    for (int64_t f : {2, 3, 5, 7, 11, 13})
        while (r % f == 0) r /= f;
    return r > 1 && czt_length(nfft) * 16 <= ((int64_t)1 << 31);
}
// complex transforms per execution (each carries `per` = 2 series, or 1 with pairing off): up to 1 GiB of work area, no more
// than the call needs
static int czt_pairs_for(int64_t m, int64_t S = INT64_MAX, int per = 2)
{
    int64_t pairs = ((int64_t)1 << 30) / (m * 16);
    pairs = pairs < 1 ? 1 : pairs > 128 ? 128 : pairs;
    const int64_t need = S == INT64_MAX ? pairs : (S + per - 1) / per;
    return (int)(pairs < need ? pairs : need);
}
static int czt_plan_get(mtg_ctx *ctx, int64_t m, int pairs, hipfftHandle *plan)
{
    std::lock_guard<std::mutex> lock(ctx->sim_mu);
    auto &sp = ctx->czt.plans[pairs == czt_pairs_for(m) ? 0 : 1];
    if (!(sp.have && sp.m == m && sp.pairs == pairs)) {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        if (sp.have) { (void)hipfftDestroy(sp.h); sp.have = false; }
        if (hipfftPlan1d(&sp.h, (int)m, HIPFFT_Z2Z, pairs) != HIPFFT_SUCCESS) return MTG_E_HIP;
        sp.have = true;
        sp.m = m;
        sp.pairs = pairs;
    }
    if (plan) *plan = sp.h;
    return MTG_OK;
}
// chirp and transformed wrapped chirp of length nfft, made on `s` the first time a length is used
static int czt_tables_get(mtg_ctx *ctx, int64_t nfft, hipStream_t s)
{
    mtg_ctx::SimCzt &z = ctx->czt;
    const int64_t m = czt_length(nfft);
    if (z.tables && z.nfft == nfft) return MTG_OK;
    z.tables = false;
    if (z.chirp.reserve((size_t)nfft * 16) != hipSuccess || z.bhat.reserve((size_t)m * 16) != hipSuccess) return MTG_E_HIP;
    mtg_launch_czt_tables(nfft, m, z.chirp.as<double2>(), z.bhat.as<double2>(), s);
    hipfftHandle one = 0;
    {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        if (hipfftPlan1d(&one, (int)m, HIPFFT_Z2Z, 1) != HIPFFT_SUCCESS) return MTG_E_HIP;
    }
    bool ok = hipfftSetStream(one, s) == HIPFFT_SUCCESS &&
              hipfftExecZ2Z(one, (hipfftDoubleComplex *)z.bhat.p, (hipfftDoubleComplex *)z.bhat.p, HIPFFT_FORWARD) == HIPFFT_SUCCESS;
    ok = ok && hipStreamSynchronize(s) == hipSuccess;    // (the plan's own work area goes with the plan)
    {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        (void)hipfftDestroy(one);
    }
    if (!ok) return MTG_E_HIP;
    z.tables = true;
    z.nfft = nfft;
    z.m = m;
    return MTG_OK;
}

MTG_API int mtg_simulate_plan(mtg_ctx *ctx, int64_t nfft)
{
    if (!ctx || nfft < 4 || nfft > ((int64_t)1 << 30)) return MTG_E_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return MTG_E_HIP;   // (HIP's current device is per thread)
    if (sim_wants_czt(ctx, nfft)) {   // the bulk plan of the power-of-two transforms (milliseconds); the tables at first use
        const int64_t m = czt_length(nfft);
        return czt_plan_get(ctx, m, czt_pairs_for(m), nullptr);
    }
    return sim_plan_get(ctx, nfft, INT64_MAX, nullptr);   // the bulk plan
}

// The E13 flux-PDF adjustment (mtg_e13.hip) of the `sc` segments e13.seg[sc][n] of one simulation chunk (global indices
// s0 .. s0 + sc): on return e13.x[sc][n] holds the adjusted series.  `chunk` = the batch of the plans (sc <= chunk).
static int e13_adjust_chunk(mtg_ctx *ctx, int64_t sc, int64_t chunk, int64_t s0, int64_t n, double mean_rate, uint64_t seed, hipStream_t s)
{
    mtg_ctx::E13 &E = ctx->e13;
    const int64_t nk = n / 2 + 1;
    if (chunk * n >= ((int64_t)1 << 31)) return fail(ctx, MTG_E_ARG, "E13 adjustment: %lld segments of %lld samples per chunk exceed 2^31 elements", (long long)chunk, (long long)n);
    const size_t temp_bytes = mtg_e13_sort_temp_bytes(chunk, n);
    HIP_TRY(ctx, E.x.reserve((size_t)chunk * n * 8));
    HIP_TRY(ctx, E.fresh.reserve((size_t)chunk * n * 8));
    HIP_TRY(ctx, E.values.reserve((size_t)chunk * n * 8));
    HIP_TRY(ctx, E.adj.reserve((size_t)chunk * n * 8));
    HIP_TRY(ctx, E.keys.reserve((size_t)chunk * n * 8));
    HIP_TRY(ctx, E.amp.reserve((size_t)chunk * nk * 8));
    HIP_TRY(ctx, E.spec.reserve((size_t)chunk * nk * 16));
    HIP_TRY(ctx, E.idx.reserve((size_t)chunk * n * 4));
    HIP_TRY(ctx, E.order.reserve((size_t)chunk * n * 4));
    HIP_TRY(ctx, E.order_tmp.reserve((size_t)chunk * n * 4));
    HIP_TRY(ctx, E.segment.reserve((size_t)chunk * n * 4));
    HIP_TRY(ctx, E.segment_out.reserve((size_t)chunk * n * 4));
    HIP_TRY(ctx, E.flags.reserve((size_t)(2 * chunk + 1) * 4));
    HIP_TRY(ctx, E.stdv.reserve((size_t)chunk * 8));
    HIP_TRY(ctx, E.temp.reserve(temp_bytes > 0 ? temp_bytes : 16));
    if (!E.have || E.n != n || E.batch != chunk) {
        std::lock_guard<std::mutex> plans(g_fft_plan_mu);
        if (E.have) { (void)hipfftDestroy(E.fwd); (void)hipfftDestroy(E.inv); E.have = false; }
        int len = (int)n;
        if (hipfftPlanMany(&E.fwd, 1, &len, nullptr, 1, (int)n, nullptr, 1, (int)nk, HIPFFT_D2Z, (int)chunk) != HIPFFT_SUCCESS)
            return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftPlanMany (forward, n = %lld, batch = %lld) failed", (long long)n, (long long)chunk);
        if (hipfftPlanMany(&E.inv, 1, &len, nullptr, 1, (int)nk, nullptr, 1, (int)n, HIPFFT_Z2D, (int)chunk) != HIPFFT_SUCCESS) {
            (void)hipfftDestroy(E.fwd);
            return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftPlanMany (inverse, n = %lld, batch = %lld) failed", (long long)n, (long long)chunk);
        }
        E.have = true; E.n = n; E.batch = chunk;
    }
    if (hipfftSetStream(E.fwd, s) != HIPFFT_SUCCESS || hipfftSetStream(E.inv, s) != HIPFFT_SUCCESS)
        return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftSetStream failed");
    double *seg = E.seg.as<double>(), *x = E.x.as<double>(), *fresh = E.fresh.as<double>(), *values = E.values.as<double>();
    double *adj = E.adj.as<double>(), *keys = E.keys.as<double>(), *amp = E.amp.as<double>();
    double2 *spec = E.spec.as<double2>();
    int32_t *idx = E.idx.as<int32_t>(), *order = E.order.as<int32_t>(), *order_tmp = E.order_tmp.as<int32_t>();
    uint32_t *segment = E.segment.as<uint32_t>(), *segment_out = E.segment_out.as<uint32_t>();
    int32_t *done = E.flags.as<int32_t>(), *notconv = done + chunk, *running = done + 2 * chunk;
    HIP_TRY(ctx, hipMemsetAsync(E.flags.p, 0, (size_t)(2 * chunk + 1) * 4, s));
    if (sc < chunk) {   // the plans transform `chunk` slots: the unused ones hold zeros
        HIP_TRY(ctx, hipMemsetAsync(seg + sc * n, 0, (size_t)(chunk - sc) * n * 8, s));
        HIP_TRY(ctx, hipMemsetAsync(x + sc * n, 0, (size_t)(chunk - sc) * n * 8, s));
    }
    mtg_launch_e13_iota(sc, n, idx, done, s);
    // the target: amplitudes of the TK95 segment; the white series and its sorted values
    if (hipfftExecD2Z(E.fwd, seg, (hipfftDoubleComplex *)spec) != HIPFFT_SUCCESS) return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftExecD2Z failed");
    mtg_launch_e13_abs(sc * nk, spec, amp, s);
    if (E.given_S) {
        const int64_t gS = E.given_S, gn = E.given_n;
        if (gn != n || s0 + sc > gS) return fail(ctx, MTG_E_ARG, "E13 adjustment: the draws of mtg_set_simulate_pdf_draws are [%lld][%lld], this call needs series %lld..%lld of %lld samples",
                                                  (long long)gS, (long long)gn, (long long)s0, (long long)(s0 + sc), (long long)n);
        HIP_TRY(ctx, hipMemcpyAsync(x, E.given.as<double>() + s0 * n, (size_t)sc * n * 8, hipMemcpyDeviceToDevice, s));
    } else {
        mtg_launch_e13_std(sc, n, seg, E.stdv.as<double>(), s);
        mtg_launch_e13_draw(sc, s0, ctx->stream_base, n, E.kind, mean_rate, E.stdv.as<double>(), seed, x, s);
    }
    HIP_TRY(ctx, mtg_launch_e13_sort_values(sc, n, x, keys, idx, order_tmp, segment, segment_out, order, values, E.temp.p, temp_bytes, s));
    HIP_TRY(ctx, hipGetLastError());
    int it = 0;
    int32_t still = (int32_t)sc;
    for (; it <= E.max_iter && still > 0; ++it) {
        if (hipfftExecD2Z(E.fwd, x, (hipfftDoubleComplex *)spec) != HIPFFT_SUCCESS) return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftExecD2Z failed");
        mtg_launch_e13_phase(sc * nk, amp, spec, s);
        if (hipfftExecZ2D(E.inv, (hipfftDoubleComplex *)spec, adj) != HIPFFT_SUCCESS) return fail(ctx, MTG_E_HIP, "E13 adjustment: hipfftExecZ2D failed");
        HIP_TRY(ctx, mtg_launch_e13_rank(sc, n, adj, keys, idx, order_tmp, segment, segment_out, order, E.temp.p, temp_bytes, s));
        HIP_TRY(ctx, hipMemsetAsync(running, 0, 4, s));
        mtg_launch_e13_step(sc, n, order, values, x, fresh, done, notconv, running, s);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(&still, running, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    if (it > E.iterations) E.iterations = it;
    E.not_converged += still;
    return MTG_OK;
}

MTG_API int mtg_simulate_tk95(mtg_ctx *ctx, int64_t S, const double *theta, const double *psd_table, int64_t psd_rows,
                              uint64_t seed, int64_t nfft, double sim_dt, double mean_rate, int64_t seg_len,
                              const int32_t *win_lo, const int32_t *win_hi, int noise_kind, double sigma_noise,
                              const double *exposures, double *clean, double *rates, double *dy, double *lc_means,
                              double *segments, int make_resident)
{
    int rc = check_ready(ctx, psd_table == nullptr);   // a tabulated spectrum needs no model
    if (rc) return rc;
    if (psd_table && psd_rows != 1 && psd_rows != S)
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: psd_rows must be 1 or S");
    const int64_t N = ctx->N;
    if (nfft > ((int64_t)1 << 30)) return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: nfft above 2^30");
    if (S <= 0 || nfft < 4 || !(sim_dt > 0.0) || seg_len <= 0 || seg_len > nfft || !win_lo || !win_hi || !rates || !dy ||
        (!psd_table && !theta && ctx->model.P > 0))
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: bad arguments");
    if (noise_kind < 0 || noise_kind > 3 || (noise_kind >= 2 && !exposures) || (noise_kind == 1 && !(sigma_noise >= 0.0)))
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: bad noise specification");
    if (noise_kind == 3 && ctx->kraft.N != N)
        return fail(ctx, MTG_E_STATE, "mtg_simulate_tk95: noise_kind 3 (Kraft) needs mtg_set_simulate_kraft for the %lld epochs of the resident sampling", (long long)N);
    MtgKraftTables kraft;
    if (noise_kind == 3) {
        kraft.bkg_counts = ctx->kraft.bkg.as<double>(); kraft.bkg_rate_err = ctx->kraft.err.as<double>();
        kraft.median = ctx->kraft.med.as<double>(); kraft.half = ctx->kraft.half.as<double>();
        kraft.K = ctx->kraft.K; kraft.threshold = ctx->kraft.threshold;
    }
    for (int64_t n = 0; n < N; ++n)
        if (win_lo[n] < 0 || win_hi[n] < win_lo[n] || win_hi[n] > seg_len)
            return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: window %lld = [%d, %d) outside the segment of %lld samples",
                        (long long)n, win_lo[n], win_hi[n], (long long)seg_len);
    if (make_resident && ctx->t_per_lc)
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: make_resident needs a shared sampling");
    // draws handed in for this call (consumed whatever happens next)
    const int64_t given_S = ctx->given.S, given_nk = ctx->given.nk;
    ctx->given.S = 0;
    if (given_S && (given_S != S || given_nk != nfft / 2 + 1))
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: the draws of mtg_set_simulate_draws are for %lld series of %lld frequencies, "
                    "this call simulates %lld of %lld", (long long)given_S, (long long)given_nk, (long long)S, (long long)(nfft / 2 + 1));
    for (int64_t i = 0; i < given_S; ++i)  // a start is an index into the series: the kernels do not check it
        if (ctx->given.starts_host[i] + seg_len > nfft)
            return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: given start %lld + segment %lld beyond the series of %lld samples",
                        (long long)ctx->given.starts_host[i], (long long)seg_len, (long long)nfft);
    const double *given_normals = given_S ? ctx->given.normals.as<double>() : nullptr;
    const int64_t *given_starts = given_S ? ctx->given.starts.as<int64_t>() : nullptr;
    rc = use_device(ctx);
    if (rc) return rc;
    struct E13Given {   // the E13 draws handed in for this call are consumed whatever happens next; the report starts afresh
        mtg_ctx *c;
        ~E13Given() { c->e13.given_S = 0; }
    } e13_given{ctx};
    ctx->e13.iterations = 0;
    ctx->e13.not_converged = 0;
    if (ctx->e13.kind != 0 && !(mean_rate > 0.0) && ctx->e13.kind == 1)
        return fail(ctx, MTG_E_ARG, "mtg_simulate_tk95: a lognormal flux PDF needs a positive mean rate");
    mtg_trace::Range range("mtg:simulate_tk95");
    MtgModel m0;
    memset(&m0, 0, sizeof m0);
    const MtgModel &m = psd_table ? m0 : ctx->model;
    const int P = m.P;
    MtgCoefLayout lay{m.nr_max, m.nc_max};
    rc = reserve_workspace(ctx, S, lay.nslots() > 4 ? lay.nslots() : 4, 1);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    const int64_t nk = nfft / 2 + 1;
    // the simulations go through the context's plan `chunk` at a time (the last group may be short: the transforms of
    // the unused slots run on whatever the buffer holds and are not looked at)
    const bool czt = sim_wants_czt(ctx, nfft);
    const int64_t czt_m = czt ? czt_length(nfft) : 0;
    const int czt_per = ctx->czt.pairs_on ? 2 : 1;   // series per complex transform
    const int czt_pairs = czt ? czt_pairs_for(czt_m, S, czt_per) : 0;
    const int64_t chunk = czt ? czt_per * (int64_t)czt_pairs : sim_batch_for(nfft, S);
    DevBuf &spec = ctx->sim_spec, &series = ctx->sim_series;
    DevBuf d_lo, d_hi, d_expo, d_clean, d_rates, d_dy, d_sig, d_means, d_psd, d_seg;
    hipError_t e = hipSuccess;
    const char *what = "allocation";
    auto cleanup = [&]() {
        DevBuf *bufs[] = {&d_lo, &d_hi, &d_expo, &d_clean, &d_rates, &d_dy, &d_sig, &d_means, &d_psd, &d_seg};
        for (DevBuf *b : bufs) b->release();
    };
    HIP_TRY(ctx, ctx->theta.reserve((size_t)S * (P > 0 ? P : 1) * 8));
    HIP_TRY(ctx, ctx->out.reserve((size_t)S * 8));
    HIP_TRY(ctx, ctx->status.reserve((size_t)S * 4));
    e = spec.reserve((size_t)chunk * nk * 16);
    if (e == hipSuccess) e = series.reserve((size_t)chunk * nfft * 8);
    if (e == hipSuccess) e = d_lo.reserve((size_t)N * 4);
    if (e == hipSuccess) e = d_hi.reserve((size_t)N * 4);
    if (e == hipSuccess) e = d_expo.reserve((size_t)N * 8);
    if (e == hipSuccess && clean) e = d_clean.reserve((size_t)S * N * 8);
    if (e == hipSuccess) e = d_rates.reserve((size_t)S * N * 8);
    if (e == hipSuccess) e = d_dy.reserve((size_t)S * N * 8);
    if (e == hipSuccess) e = d_sig.reserve((size_t)S * 4);
    if (e == hipSuccess) e = d_means.reserve((size_t)S * 8);
    if (e == hipSuccess && psd_table) e = d_psd.reserve((size_t)psd_rows * nk * 8);
    if (e == hipSuccess && psd_table) e = hipMemcpyAsync(d_psd.p, psd_table, (size_t)psd_rows * nk * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && segments) e = d_seg.reserve((size_t)S * seg_len * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo.p, win_lo, (size_t)N * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hi.p, win_hi, (size_t)N * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && exposures) e = hipMemcpyAsync(d_expo.p, exposures, (size_t)N * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && P > 0) e = hipMemcpyAsync(ctx->theta.p, theta, (size_t)S * P * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && !psd_table) {
        // theta -> celerite coefficients (no prior: the samples come from the posterior itself)
        MtgPrepArgs pa;
        pa.model = m; pa.theta = ctx->theta.as<double>(); pa.B = S; pa.add_prior = 0;
        pa.coef = ctx->coef.as<double>(); pa.cstride = ctx->cstride; pa.nsig = 1;
        pa.lists = ctx->lists.as<int>(); pa.counts = ctx->counts.as<int>();
        pa.out = ctx->out.as<double>(); pa.status = ctx->status.as<int32_t>(); pa.sig = d_sig.as<int32_t>();
        mtg_launch_prepare(pa, s);
        e = hipGetLastError();
    }
    hipfftHandle plan = 0;
    if (e == hipSuccess) {
        what = "hipfftPlan1d";
        const int prc = czt ? (czt_tables_get(ctx, nfft, s) != MTG_OK || ctx->czt.work.reserve((size_t)czt_pairs * czt_m * 16) != hipSuccess
                                   ? MTG_E_HIP : czt_plan_get(ctx, czt_m, czt_pairs, &plan))
                            : sim_plan_get(ctx, nfft, S, &plan);
        if (prc != MTG_OK || hipfftSetStream(plan, s) != HIPFFT_SUCCESS) {
            cleanup();
            return fail(ctx, MTG_E_HIP, "mtg_simulate_tk95: hipFFT plan creation failed (nfft = %lld, batch = %lld)",
                        (long long)nfft, (long long)chunk);
        }
    }
    // irfft normalisation (hipFFT C2R is unnormalised) and the reference's power scaling
    const double scale = sqrt((double)nfft * sim_dt * sqrt(2.0 * M_PI)) / (double)nfft;
    for (int64_t s0 = 0; e == hipSuccess && s0 < S; s0 += chunk) {
        const int64_t sc = s0 + chunk <= S ? chunk : S - s0;
        what = "simulation kernels";
        mtg_launch_tk95_spectrum(sc, s0, ctx->stream_base, nfft, sim_dt, ctx->coef.as<double>(), ctx->cstride, lay, m.nr0, m.nc0,
                                 d_sig.as<int32_t>(), psd_table ? d_psd.as<double>() : nullptr, psd_rows, seed, given_normals,
                                 spec.as<double2>(), s);
        if (czt) {
            // pairs of series through two power-of-two complex transforms (a short last group packs zeros into the
            // pairs it does not fill: the plan's batch is fixed)
            double2 *work = ctx->czt.work.as<double2>();
            mtg_launch_czt_pack(sc, czt_per, nfft, czt_m, spec.as<double2>(), ctx->czt.chirp.as<double2>(), work, s);
            const int64_t used = (sc + czt_per - 1) / czt_per;
            if (used < czt_pairs) e = hipMemsetAsync(work + used * czt_m, 0, (size_t)(czt_pairs - used) * czt_m * 16, s);
            if (e != hipSuccess) break;
            bool ok = hipfftExecZ2Z(plan, (hipfftDoubleComplex *)work, (hipfftDoubleComplex *)work, HIPFFT_FORWARD) == HIPFFT_SUCCESS;
            if (ok) mtg_launch_czt_mul(used, czt_m, ctx->czt.bhat.as<double2>(), work, s);
            ok = ok && hipfftExecZ2Z(plan, (hipfftDoubleComplex *)work, (hipfftDoubleComplex *)work, HIPFFT_BACKWARD) == HIPFFT_SUCCESS;
            if (!ok) {
                cleanup();
                return fail(ctx, MTG_E_HIP, "mtg_simulate_tk95: hipfftExecZ2Z failed");  // (the resident set is untouched so far)
            }
            mtg_launch_czt_unpack(sc, czt_per, nfft, czt_m, work, ctx->czt.chirp.as<double2>(), series.as<double>(), s);
        } else {
            if (sc < chunk)  // a short last group: the unused slots transform zeros
                e = hipMemsetAsync((char *)spec.p + (size_t)sc * nk * 16, 0, (size_t)(chunk - sc) * nk * 16, s);
            if (e != hipSuccess) break;
            if (hipfftExecZ2D(plan, (hipfftDoubleComplex *)spec.p, series.as<double>()) != HIPFFT_SUCCESS) {
                cleanup();
                return fail(ctx, MTG_E_HIP, "mtg_simulate_tk95: hipfftExecZ2D failed");  // (the resident set is untouched so far)
            }
        }
        if (ctx->e13.kind != 0) {
            // A non-Gaussian flux PDF (simulator.py:65-140): the cut segments as rates on the fine grid, adjusted on the
            // device (mtg_e13.hip), then averaged into the epochs from the ADJUSTED series (start 0, no rescaling).
            what = "E13 adjustment";
            e = ctx->e13.seg.reserve((size_t)chunk * seg_len * 8);
            if (e != hipSuccess) break;
            mtg_launch_tk95_segment(sc, s0, ctx->stream_base, nfft, seg_len, sim_dt, scale, mean_rate, series.as<double>(), seed,
                                    given_starts, ctx->e13.seg.as<double>(), s, /* out_first = */ s0);
            if (segments)   // (what the caller asked for: the segments as the reference hands them to its adjustment)
                e = hipMemcpyAsync(d_seg.as<double>() + s0 * seg_len, ctx->e13.seg.p, (size_t)sc * seg_len * 8, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) break;
            const int arc = e13_adjust_chunk(ctx, sc, chunk, s0, seg_len, mean_rate, seed, s);
            if (arc) {
                cleanup();
                if (make_resident) { ctx->N = 0; ctx->L = 0; }
                return arc;
            }
            mtg_launch_tk95_observe(sc, s0, ctx->stream_base, N, seg_len, seg_len, sim_dt, sim_dt, 0.0, ctx->e13.x.as<double>(),
                                    d_lo.as<int32_t>(), d_hi.as<int32_t>(), noise_kind, sigma_noise, d_expo.as<double>(),
                                    0, seed, nullptr, clean ? d_clean.as<double>() : nullptr, d_rates.as<double>(), d_dy.as<double>(), s, kraft);
            e = hipGetLastError();
            continue;
        }
        mtg_launch_tk95_observe(sc, s0, ctx->stream_base, N, nfft, seg_len, sim_dt, scale, mean_rate, series.as<double>(),
                                d_lo.as<int32_t>(), d_hi.as<int32_t>(), noise_kind, sigma_noise, d_expo.as<double>(),
                                -1, seed, given_starts, clean ? d_clean.as<double>() : nullptr, d_rates.as<double>(), d_dy.as<double>(), s, kraft);
        if (segments)
            mtg_launch_tk95_segment(sc, s0, ctx->stream_base, nfft, seg_len, sim_dt, scale, mean_rate, series.as<double>(), seed,
                                    given_starts, d_seg.as<double>(), s);
        e = hipGetLastError();
    }
    DevBuf yv_tmp;
    if (e == hipSuccess && (make_resident || lc_means)) {
        what = "resident set";
        DevBuf &target = make_resident ? ctx->yv : yv_tmp;  // without make_resident only the means are wanted
        e = target.reserve((size_t)S * N * 16);
        if (e == hipSuccess) {
            mtg_launch_tk95_resident(S, N, d_rates.as<double>(), d_dy.as<double>(), target.as<double2>(),
                                     d_means.as<double>(), s);
            e = hipGetLastError();
        }
        if (e == hipSuccess && lc_means) e = hipMemcpyAsync(lc_means, d_means.p, (size_t)S * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess && clean) e = hipMemcpyAsync(clean, d_clean.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && segments) e = hipMemcpyAsync(segments, d_seg.p, (size_t)S * seg_len * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(rates, d_rates.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dy, d_dy.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    yv_tmp.release();
    // the plan's buffers stay with the context for the next call of the workflow -- unless they are large enough to be
    // in somebody's way (a fine simulation grid: hundreds of MB per transform)
    if (spec.cap + series.cap + ctx->czt.work.cap > ((size_t)1 << 30)) { spec.release(); series.release(); ctx->czt.work.release(); }
    {   // ... and so do the E13 adjustment's (92 bytes per fine sample and segment of a chunk)
        mtg_ctx::E13 &E = ctx->e13;
        DevBuf *eb[] = {&E.seg, &E.x, &E.fresh, &E.values, &E.adj, &E.keys, &E.amp, &E.spec, &E.idx, &E.order, &E.order_tmp, &E.segment, &E.segment_out, &E.temp};
        size_t held = 0;
        for (DevBuf *b : eb) held += b->cap;
        if (held > ((size_t)1 << 30))
            for (DevBuf *b : eb) b->release();
    }
    if (e != hipSuccess) {
        // the resident set may have been freed or partly overwritten on the way: nothing is resident any more
        if (make_resident) { ctx->N = 0; ctx->L = 0; }
        return fail(ctx, MTG_E_HIP, "mtg_simulate_tk95 (%s): %s", what, hipGetErrorString(e));
    }
    if (make_resident) ctx->L = S;  // the simulated light curves replace the resident set (same sampling)
    return MTG_OK;
}

MTG_API int mtg_tk95_observe_series(mtg_ctx *ctx, int64_t S, int64_t nfft, int64_t seg_len, int64_t start,
                                    const double *series, const int32_t *win_lo, const int32_t *win_hi, double *rates)
{
    int rc = check_ready(ctx, false);
    if (rc) return rc;
    const int64_t N = ctx->N;
    if (S <= 0 || nfft <= 0 || seg_len <= 0 || start < 0 || start + seg_len > nfft || !series || !win_lo || !win_hi || !rates)
        return fail(ctx, MTG_E_ARG, "mtg_tk95_observe_series: bad arguments");
    for (int64_t n = 0; n < N; ++n)
        if (win_lo[n] < 0 || win_hi[n] < win_lo[n] || win_hi[n] > seg_len)
            return fail(ctx, MTG_E_ARG, "mtg_tk95_observe_series: window %lld outside the segment", (long long)n);
    rc = use_device(ctx);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    DevBuf d_series, d_lo, d_hi, d_rates, d_dy;
    hipError_t e = d_series.reserve((size_t)S * nfft * 8);
    if (e == hipSuccess) e = d_lo.reserve((size_t)N * 4);
    if (e == hipSuccess) e = d_hi.reserve((size_t)N * 4);
    if (e == hipSuccess) e = d_rates.reserve((size_t)S * N * 8);
    if (e == hipSuccess) e = d_dy.reserve((size_t)S * N * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_series.p, series, (size_t)S * nfft * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lo.p, win_lo, (size_t)N * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hi.p, win_hi, (size_t)N * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        // scale = dt = 1, mean 0, no noise: the plain window average of the series
        mtg_launch_tk95_observe(S, 0, 0, N, nfft, seg_len, 1.0, 1.0, 0.0, d_series.as<double>(), d_lo.as<int32_t>(),
                                d_hi.as<int32_t>(), 0, 0.0, nullptr, start, 0, nullptr, nullptr, d_rates.as<double>(),
                                d_dy.as<double>(), s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rates, d_rates.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(ctx, MTG_E_HIP, "mtg_tk95_observe_series: %s", hipGetErrorString(e));
    return MTG_OK;
}

MTG_API int mtg_predict(mtg_ctx *ctx, int64_t B, const double *theta, const int32_t *lc_index, double *mu,
                        double *var, int32_t *status)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (B <= 0 || !mu || !var || !status || (!theta && ctx->model.P > 0))
        return fail(ctx, MTG_E_ARG, "mtg_predict: bad arguments");
    if (lc_index)
        for (int64_t b = 0; b < B; ++b)
            if (lc_index[b] < 0 || lc_index[b] >= ctx->L)
                return fail(ctx, MTG_E_ARG, "lc_index[%lld] = %d outside [0, %lld)", (long long)b, lc_index[b],
                            (long long)ctx->L);
    rc = use_device(ctx);
    if (rc) return rc;
    const MtgModel &m = ctx->model;
    const int P = m.P, J = m.nr_max + 2 * m.nc_max, N = (int)ctx->N;
    MtgCoefLayout lay{m.nr_max, m.nc_max};
    rc = reserve_workspace(ctx, B, lay.nslots(), 1);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    DevBuf work, d_mu, d_var, d_sig;
    HIP_TRY(ctx, ctx->theta.reserve((size_t)B * (P > 0 ? P : 1) * 8));
    HIP_TRY(ctx, ctx->out.reserve((size_t)B * 8));
    HIP_TRY(ctx, ctx->status.reserve((size_t)B * 4));
    hipError_t e = work.reserve((size_t)B * N * (3 * J + 2) * 8);
    if (e == hipSuccess) e = d_mu.reserve((size_t)B * N * 8);
    if (e == hipSuccess) e = d_var.reserve((size_t)B * N * 8);
    if (e == hipSuccess) e = d_sig.reserve((size_t)B * 4);
    const int32_t *d_lc = nullptr;
    if (e == hipSuccess && P > 0) e = hipMemcpyAsync(ctx->theta.p, theta, (size_t)B * P * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && lc_index) {
        e = ctx->lc.reserve((size_t)B * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->lc.p, lc_index, (size_t)B * 4, hipMemcpyHostToDevice, s);
        d_lc = ctx->lc.as<int32_t>();
    }
    if (e == hipSuccess) {
        MtgPrepArgs pa;
        pa.model = m; pa.theta = ctx->theta.as<double>(); pa.B = B; pa.add_prior = 1;
        pa.coef = ctx->coef.as<double>(); pa.cstride = ctx->cstride; pa.nsig = 1;
        pa.lists = ctx->lists.as<int>(); pa.counts = ctx->counts.as<int>();
        pa.out = ctx->out.as<double>(); pa.status = ctx->status.as<int32_t>(); pa.sig = d_sig.as<int32_t>();
        mtg_launch_prepare(pa, s);
        MtgPredictArgs qa;
        qa.coef = ctx->coef.as<double>(); qa.cstride = ctx->cstride; qa.lay = lay;
        qa.nr0 = m.nr0; qa.nc0 = m.nc0; qa.sig = d_sig.as<int32_t>(); qa.B = B; qa.lc_index = d_lc;
        qa.status_in = ctx->status.as<int32_t>(); qa.dxt = ctx->dxt.as<double2>(); qa.yv = ctx->yv.as<double2>();
        qa.N = ctx->N; qa.t_stride = ctx->t_per_lc ? ctx->N : 0; qa.work = work.as<double>();
        qa.mu = d_mu.as<double>(); qa.var = d_var.as<double>(); qa.status = ctx->status.as<int32_t>();
        mtg_launch_predict(&qa, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mu, d_mu.p, (size_t)B * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(var, d_var.p, (size_t)B * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(status, ctx->status.p, (size_t)B * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    work.release(); d_mu.release(); d_var.release(); d_sig.release();
    if (e != hipSuccess) return fail(ctx, MTG_E_HIP, "mtg_predict: %s", hipGetErrorString(e));
    return MTG_OK;
}

MTG_API int mtg_apply_inverse(mtg_ctx *ctx, const double *theta, int32_t lc_index, int64_t M, double *x,
                              int32_t *status)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (M <= 0 || !x || !status || (!theta && ctx->model.P > 0))
        return fail(ctx, MTG_E_ARG, "mtg_apply_inverse: bad arguments");
    if (lc_index < 0 || lc_index >= ctx->L)
        return fail(ctx, MTG_E_ARG, "lc_index = %d outside [0, %lld)", lc_index, (long long)ctx->L);
    rc = use_device(ctx);
    if (rc) return rc;
    const MtgModel &m = ctx->model;
    const int P = m.P, Jws = m.nr_max + 2 * m.nc_max, J = m.nr0 + 2 * m.nc0;
    const int64_t N = ctx->N;
    MtgCoefLayout lay{m.nr_max, m.nc_max};
    rc = reserve_workspace(ctx, 1, lay.nslots(), 1);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    DevBuf work, d_mu, d_var, d_sig, d_x;
    HIP_TRY(ctx, ctx->theta.reserve((size_t)(P > 0 ? P : 1) * 8));
    HIP_TRY(ctx, ctx->out.reserve(8));
    HIP_TRY(ctx, ctx->status.reserve(4));
    HIP_TRY(ctx, ctx->lc.reserve(4));
    hipError_t e = work.reserve((size_t)N * (3 * Jws + 2) * 8);
    if (e == hipSuccess) e = d_mu.reserve((size_t)N * 8);
    if (e == hipSuccess) e = d_var.reserve((size_t)N * 8);
    if (e == hipSuccess) e = d_sig.reserve(4);
    if (e == hipSuccess) e = d_x.reserve((size_t)N * M * 8);
    if (e == hipSuccess && P > 0) e = hipMemcpyAsync(ctx->theta.p, theta, (size_t)P * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->lc.p, &lc_index, 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_x.p, x, (size_t)N * M * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        // the factorisation of this parameter vector: the forward sweep of mtg_predict_kernel leaves
        // U_n, W_n, phi_n, D_n of every sample in `work`
        MtgPrepArgs pa;
        pa.model = m; pa.theta = ctx->theta.as<double>(); pa.B = 1; pa.add_prior = 1;
        pa.coef = ctx->coef.as<double>(); pa.cstride = ctx->cstride; pa.nsig = 1;
        pa.lists = ctx->lists.as<int>(); pa.counts = ctx->counts.as<int>();
        pa.out = ctx->out.as<double>(); pa.status = ctx->status.as<int32_t>(); pa.sig = d_sig.as<int32_t>();
        mtg_launch_prepare(pa, s);
        MtgPredictArgs qa;
        qa.coef = ctx->coef.as<double>(); qa.cstride = ctx->cstride; qa.lay = lay;
        qa.nr0 = m.nr0; qa.nc0 = m.nc0; qa.sig = d_sig.as<int32_t>(); qa.B = 1; qa.lc_index = ctx->lc.as<int32_t>();
        qa.status_in = ctx->status.as<int32_t>(); qa.dxt = ctx->dxt.as<double2>(); qa.yv = ctx->yv.as<double2>();
        qa.N = N; qa.t_stride = ctx->t_per_lc ? N : 0; qa.work = work.as<double>();
        qa.mu = d_mu.as<double>(); qa.var = d_var.as<double>(); qa.status = ctx->status.as<int32_t>();
        mtg_launch_predict(&qa, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(status, ctx->status.p, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && *status == MTG_ST_OK) {
        mtg_launch_apply_inverse(work.as<double>(), N, J, M, d_x.as<double>(), s);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(x, d_x.p, (size_t)N * M * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    work.release(); d_mu.release(); d_var.release(); d_sig.release(); d_x.release();
    if (e != hipSuccess) return fail(ctx, MTG_E_HIP, "mtg_apply_inverse: %s", hipGetErrorString(e));
    return MTG_OK;
}

MTG_API int mtg_math_probe(mtg_ctx *ctx, int64_t n, const double *x, double *exp_neg, double *sin_x,
                           double *cos_x, double *rcp_x)
{
    if (!ctx || n <= 0 || !x || !exp_neg || !sin_x || !cos_x || !rcp_x) return MTG_E_ARG;
    int rc = use_device(ctx);
    if (rc) return rc;
    DevBuf buf;
    HIP_TRY(ctx, buf.reserve((size_t)n * 8 * 5));
    double *d = buf.as<double>();
    CTX_STREAM(ctx, s);
    hipError_t e = hipMemcpyAsync(d, x, (size_t)n * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        mtg_launch_math_probe(n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(exp_neg, d + n, (size_t)n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(sin_x, d + 2 * n, (size_t)n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(cos_x, d + 3 * n, (size_t)n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(rcp_x, d + 4 * n, (size_t)n * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    buf.release();
    if (e != hipSuccess) return fail(ctx, MTG_E_HIP, "mtg_math_probe: %s", hipGetErrorString(e));
    return MTG_OK;
}

MTG_API int mtg_set_time_parallel(mtg_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 3) return MTG_E_ARG;
    ctx->tp_mode = mode;
    return MTG_OK;
}

MTG_API int mtg_set_simulate_draws(mtg_ctx *ctx, int64_t S, int64_t nk, const double *normals, const int64_t *starts)
{
    if (!ctx) return MTG_E_ARG;
    ctx->given.S = 0;
    ctx->given.nk = 0;
    if (S == 0) return MTG_OK;  // cleared
    if (S < 0 || nk < 3 || !normals || !starts) return fail(ctx, MTG_E_ARG, "mtg_set_simulate_draws: bad arguments");
    for (int64_t i = 0; i < S; ++i)
        if (starts[i] < 0) return fail(ctx, MTG_E_ARG, "mtg_set_simulate_draws: starts[%lld] = %lld is negative", (long long)i, (long long)starts[i]);
    int rc = use_device(ctx);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    HIP_TRY(ctx, ctx->given.normals.reserve((size_t)S * 2 * nk * 8));
    HIP_TRY(ctx, ctx->given.starts.reserve((size_t)S * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->given.normals.p, normals, (size_t)S * 2 * nk * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->given.starts.p, starts, (size_t)S * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));  // the caller's arrays are free again
    ctx->given.starts_host.assign(starts, starts + S);
    ctx->given.S = S;
    ctx->given.nk = nk;
    return MTG_OK;
}

MTG_API int mtg_set_simulate_pairs(mtg_ctx *ctx, int on)
{
    if (!ctx) return MTG_E_ARG;
    ctx->czt.pairs_on = on != 0;
    return MTG_OK;
}

MTG_API int64_t mtg_chain_autocorr_plans_built(const mtg_ctx *ctx) { return ctx ? ctx->acf_plans_built : -1; }

MTG_API int mtg_set_simulate_kraft(mtg_ctx *ctx, int64_t N, int K, double threshold, const double *bkg_counts, const double *bkg_rate_err,
                                   const double *median, const double *half)
{
    if (!ctx) return MTG_E_ARG;
    if (N == 0) { ctx->kraft.N = 0; return MTG_OK; }
    if (N < 0 || K < 1 || K > 4096 || !(threshold >= 0.0) || threshold > (double)K || !bkg_counts || !bkg_rate_err || !median || !half)
        return fail(ctx, MTG_E_ARG, "mtg_set_simulate_kraft: bad arguments (the tables must cover total counts 0 .. K - 1 >= threshold - 1)");
    int rc = use_device(ctx);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    HIP_TRY(ctx, ctx->kraft.bkg.reserve((size_t)N * 8));
    HIP_TRY(ctx, ctx->kraft.err.reserve((size_t)N * 8));
    HIP_TRY(ctx, ctx->kraft.med.reserve((size_t)N * K * 8));
    HIP_TRY(ctx, ctx->kraft.half.reserve((size_t)N * K * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->kraft.bkg.p, bkg_counts, (size_t)N * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->kraft.err.p, bkg_rate_err, (size_t)N * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->kraft.med.p, median, (size_t)N * K * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->kraft.half.p, half, (size_t)N * K * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->kraft.N = N; ctx->kraft.K = K; ctx->kraft.threshold = threshold;
    return MTG_OK;
}

MTG_API int mtg_set_simulate_pdf(mtg_ctx *ctx, int kind, int max_iter)
{
    if (!ctx || kind < 0 || kind > 2 || max_iter < 0) return MTG_E_ARG;
    ctx->e13.kind = kind;
    ctx->e13.max_iter = max_iter;
    return MTG_OK;
}

MTG_API int mtg_set_simulate_pdf_draws(mtg_ctx *ctx, int64_t S, int64_t n, const double *draws)
{
    if (!ctx) return MTG_E_ARG;
    if (S == 0 || !draws) { ctx->e13.given_S = 0; return MTG_OK; }
    if (S < 0 || n < 2) return fail(ctx, MTG_E_ARG, "mtg_set_simulate_pdf_draws: bad shape");
    int rc = use_device(ctx);
    if (rc) return rc;
    CTX_STREAM(ctx, s);
    HIP_TRY(ctx, ctx->e13.given.reserve((size_t)S * n * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->e13.given.p, draws, (size_t)S * n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->e13.given_S = S; ctx->e13.given_n = n;
    return MTG_OK;
}

MTG_API int mtg_simulate_pdf_report(const mtg_ctx *ctx, int64_t *not_converged, int *iterations)
{
    if (!ctx) return MTG_E_ARG;
    if (not_converged) *not_converged = ctx->e13.not_converged;
    if (iterations) *iterations = ctx->e13.iterations;
    return MTG_OK;
}

MTG_API int mtg_set_simulate_transform(mtg_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MTG_E_ARG;
    ctx->sim_transform = mode;
    return MTG_OK;
}

MTG_API int mtg_set_stream_base(mtg_ctx *ctx, int64_t first_index)
{
    if (!ctx) return MTG_E_ARG;
    if (first_index < 0 || first_index > 0x7fffffffll) return fail(ctx, MTG_E_ARG, "stream base must be in [0, 2^31)");
    ctx->stream_base = first_index;
    return MTG_OK;
}

MTG_API int mtg_set_pipeline(mtg_ctx *ctx, int mode)
{
    if (!ctx) return MTG_E_ARG;
    if (mode < 0 || mode > 2) return fail(ctx, MTG_E_ARG, "pipeline mode must be 0, 1 or 2");
    ctx->pipe_mode = mode;
    return MTG_OK;
}

MTG_API int mtg_set_tp_direct(mtg_ctx *ctx, int enabled)
{
    if (!ctx) return MTG_E_ARG;
    ctx->tp_direct = enabled >= 2 ? enabled : (enabled ? 1 : 0);  // 2 (diagnostic): never fall back to the filter pass
    return MTG_OK;
}

MTG_API int mtg_set_speculation(mtg_ctx *ctx, int mode)
{
    if (!ctx) return MTG_E_ARG;
    if (mode < 0 || mode > 2)
        return fail(ctx, MTG_E_ARG, "mtg_set_speculation: mode must be 0 (never), 1 (where it pays) or 2 (as 1, splits ranked in the sampler kernel)");
    ctx->spec_mode = mode;
    return MTG_OK;
}

MTG_API int mtg_set_sort(mtg_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MTG_E_ARG;
    ctx->sort_mode = mode;
    return MTG_OK;
}

MTG_API const char *mtg_last_solver(const mtg_ctx *ctx) { return ctx ? ctx->last_solver : ""; }

MTG_API int mtg_unpair_contexts(mtg_ctx *ctx)
{
    if (!ctx) return MTG_E_ARG;
    const std::shared_ptr<MtgPair> p = std::atomic_load(&ctx->pair);
    if (!p) return MTG_OK;
    mtg_ctx *members[2];
    {
        // A partner thread may be inside pair_launch right now (waiting for this context's half-step, or about to
        // lock): it holds a reference of its own, sees `broken` and launches alone; the rendezvous is freed by
        // whoever drops the last reference.
        std::lock_guard<std::mutex> lk(p->mu);
        p->broken = true;
        members[0] = p->members[0]; members[1] = p->members[1];
        p->members[0] = p->members[1] = nullptr;
    }
    p->cv.notify_all();
    for (mtg_ctx *m : members)
        if (m) std::atomic_store(&m->pair, std::shared_ptr<MtgPair>());
    return MTG_OK;
}

MTG_API int mtg_pair_contexts(mtg_ctx *a, mtg_ctx *b)
{
    if (!a || !b || a == b) return MTG_E_ARG;
    if (a->device != b->device) return fail(a, MTG_E_ARG, "mtg_pair_contexts: the two contexts are on different devices");
    if (std::atomic_load(&a->pair) || std::atomic_load(&b->pair))
        return fail(a, MTG_E_STATE, "mtg_pair_contexts: a context is paired already (mtg_unpair_contexts first)");
    int rc = use_device(a);
    if (rc) return rc;
    std::shared_ptr<MtgPair> p(new (std::nothrow) MtgPair());
    if (!p) return fail(a, MTG_E_HIP, "mtg_pair_contexts: out of memory");
    for (hipEvent_t *e : {&p->ready[0], &p->ready[1], &p->done})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess)
            return fail(a, MTG_E_HIP, "mtg_pair_contexts: event creation failed");   // (~MtgPair destroys the ones made)
    p->members[0] = a; p->members[1] = b;
    // Two models that are known now and have no kernel in common -- one of them without a pipelined sweep (a DRW alone),
    // or a pair of shapes that is not compiled -- never meet: broken from the start, nobody waits for a partner that has
    // nothing to bring.  (Models set later are looked at when their half-steps meet.)
    if (a->has_model && b->has_model) {
        auto shape = [](const mtg_ctx *c) { return MtgPipeShapeId{c->model.nr0, c->model.nc0, c->model.nsho + 1, c->model.last_b0 ? 1 : 0}; };
        const MtgPipeShapeId sa = shape(a), sb = shape(b);
        if (!mtg_find_pipe_pair_solver(sa, sb) && !mtg_find_pipe_pair_solver(sb, sa)) p->broken = true;
    }
    std::atomic_store(&a->pair, p);
    std::atomic_store(&b->pair, p);
    return MTG_OK;
}

MTG_API int mtg_set_pair_patience(mtg_ctx *ctx, int milliseconds)
{
    if (!ctx || milliseconds < 1) return MTG_E_ARG;
    const std::shared_ptr<MtgPair> p = std::atomic_load(&ctx->pair);
    if (!p) return fail(ctx, MTG_E_STATE, "mtg_set_pair_patience: the context is not paired");
    std::lock_guard<std::mutex> lk(p->mu);
    p->patience_ms = milliseconds;
    return MTG_OK;
}

MTG_API int mtg_pair_stats(const mtg_ctx *ctx, int64_t *paired_launches, int64_t *solo_launches, int *broken)
{
    if (!ctx) return MTG_E_ARG;
    const std::shared_ptr<MtgPair> p = std::atomic_load(&const_cast<mtg_ctx *>(ctx)->pair);
    if (paired_launches) *paired_launches = 0;
    if (solo_launches) *solo_launches = 0;
    if (broken) *broken = 0;
    if (!p) return MTG_OK;
    std::lock_guard<std::mutex> lk(p->mu);
    if (paired_launches) *paired_launches = p->n_pair;
    if (solo_launches) *solo_launches = p->n_solo;
    if (broken) *broken = p->broken ? 1 : 0;
    return MTG_OK;
}

MTG_API int mtg_synchronize(mtg_ctx *ctx)
{
    if (!ctx) return MTG_E_ARG;
    int rc = use_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // ... and on the caller's stream of the last mtg_loglike_batch_device, if nothing has waited for it since
    if (ctx->foreign_pending) HIP_TRY(ctx, hipEventSynchronize(ctx->foreign_done));
    return MTG_OK;
}

MTG_API int mtg_profile_begin(mtg_ctx *ctx, int capacity)
{
    if (!ctx || capacity < 0) return MTG_E_ARG;
    int rc = use_device(ctx);
    if (rc) return rc;
    while ((int)ctx->prof_ev.size() < 3 * capacity) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->prof_ev.push_back(e);
    }
    ctx->prof_cap = capacity;
    ctx->prof_n = 0;
    return MTG_OK;
}

MTG_API int mtg_profile_read(mtg_ctx *ctx, int capacity, double *prepare_ms, double *solve_ms)
{
    if (!ctx || capacity < 0) return MTG_E_ARG;
    const int n = ctx->prof_n < capacity ? ctx->prof_n : capacity;
    for (int i = 0; i < n; ++i) {
        hipEvent_t *pe = &ctx->prof_ev[3 * (size_t)i];
        float a = 0.f, b = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(pe[2]));
        HIP_TRY(ctx, hipEventElapsedTime(&a, pe[0], pe[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, pe[1], pe[2]));
        if (prepare_ms) prepare_ms[i] = a;
        if (solve_ms) solve_ms[i] = b;
    }
    ctx->prof_cap = 0;
    return n;
}

MTG_API double mtg_last_kernel_ms(const mtg_ctx *ctx)
{
    if (!ctx || !ctx->timed) return -1.0;
    float ms = -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

}  // extern "C"
