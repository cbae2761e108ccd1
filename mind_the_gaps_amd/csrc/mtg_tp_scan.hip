// mtg_tp_scan.hip -- scan kernels of the big-J time-parallel path (mtg_tp_scan.h, mtg_tp_big.h):
// up-sweep (mtg_tpb_reduce_kernel), down-sweep (mtg_tpb_down_kernel), final sum
// (mtg_tpb_finish_kernel).  They depend on the rank J only, not on the (real, complex) structure.
#include "mtg_tp_scan.h"

#include <math.h>

#define MTG_LN_2PI 1.8378770664093454835606594728112

namespace {

constexpr int GROUPS = 1;  // one wave = one workgroup = one group of elements (namespace tpw: a wave per combination)

// group -> (evaluation, group index inside the evaluation); false = nothing to do
__device__ __forceinline__ bool tpb_group(const MtgSolveArgs &a, int groups_per_eval, int64_t &ev, int &k)
{
    const int64_t gid = (int64_t)blockIdx.x;
    const int64_t i = gid / groups_per_eval;
    k = (int)(gid % groups_per_eval);
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if (i >= count) return false;
    ev = a.list ? (int64_t)a.list[i] : i;
    if (!a.list && a.status[ev] != MTG_ST_OK) return false;
    return true;
}

// out[ev][k] = in[ev][g k] o in[ev][g k + 1] o ... o in[ev][g k + g - 1].
// KAPPA: the likelihood records travel along (rec_in -> rec_out; level0: rec_in is the composition pass's parts
// array, whose fourth entry is not a magnitude) and every group is reduced; otherwise only prefixes are ever
// applied and the last group's total is not needed.
template <int J, bool KAPPA>
__global__ void __launch_bounds__(64, KAPPA ? 4 : 3) mtg_tpb_reduce_kernel(MtgSolveArgs a, const double *in,
                                                                                           double *out, int n_in, int g,
                                                                                           const double *rec_in, double *rec_out,
                                                                                           int level0, int *zero_me)
{
    // (the redo counter of the evaluation's batch: its readers of the previous half-step are done, the top kernel that
    // appends to it comes after the up-sweep -- one memset launch less per half-step)
    if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0;
    __shared__ tpw::Lds<J> L;
    const int gpe = n_in / g;
    int64_t ev;
    int k;
    if (!tpb_group(a, gpe, ev, k)) return;
    if (!KAPPA && k == gpe - 1) return;
    const int l64 = threadIdx.x;
    const tpw::Lane w = tpw::lane_of<J>(l64);
    const int64_t first = ev * n_in + (int64_t)k * g;
    const double *e = in + first * MTG_TPB_ELEM(J);
    tpw::load_first<J>(L, e, l64);
    tpw::Pre<J> pre;
    tpw::fetch<J>(pre, e + MTG_TPB_ELEM(J), l64);
    // the group's record: (dot, mag) as they come; the determinants as a product (pm 2^pe = prod 1 / det G)
    double dot = 0.0, mag = 0.0, pm = 1.0;
    int pe = 0;
    bool positive = true;
#pragma unroll 1
    for (int i = 1; i < g; ++i) {
        tpw::put_second<J>(L, pre, l64);
        if (i + 1 < g) tpw::fetch<J>(pre, e + (int64_t)(i + 1) * MTG_TPB_ELEM(J), l64);
        tpw::wsync();
        double kap[2], dm;
        int de;
        tpw::combine<J, KAPPA>(L, w, kap, dm, de);
        if (KAPPA) {
            dot -= 2.0 * (kap[0] + kap[1]);
            mag += fabs(kap[0]) + fabs(kap[1]);
            positive = positive && dm > 0.0;
            const double pr = pm * dm;
            pm = __builtin_amdgcn_frexp_mant(pr);
            pe += de + __builtin_amdgcn_frexp_exp(pr);
        }
    }
    tpw::store_first<J>(L, out + (ev * gpe + k) * MTG_TPB_ELEM(J), l64);
    if (KAPPA && l64 == 0) {
        double ld = positive ? -(log(pm) + (double)pe * 0.69314718055994530942) : __builtin_nan("");
        double dmin = positive ? INFINITY : -1.0;
        for (int i = 0; i < g; ++i) {
            const double *q = rec_in + (first + i) * 4;
            dot += q[0]; ld += q[1]; dmin = fmin(dmin, q[2]); mag += level0 ? 0.5 * q[0] : q[3];
        }
        double *q = rec_out + (ev * gpe + k) * 4;
        q[0] = dot; q[1] = ld; q[2] = dmin; q[3] = mag;
    }
}

// The filtered state after sample 0 (update of the stationary prior) into (b1 | C1) of the group's LDS region, that
// sample's terms of the likelihood into head[ev]; false: the evaluation's light curve index is out of range.
template <int J>
__device__ __forceinline__ bool tpb_head_state(const MtgSolveArgs &a, int64_t ev, tpw::Lds<J> &L, int l64, const tpw::Lane w, double *head)
{
    constexpr int M = J * J;
    const int64_t lc = a.lc_index ? (int64_t)a.lc_index[ev] : 0;
    if (lc < 0 || (uint64_t)(lc + 1) * (uint64_t)a.N * 16u > (uint64_t)a.yv_bytes) return false;
    const int nover = a.sig ? a.sig[ev] : 0, nr = a.tp_nr0 + 2 * nover, nc = a.tp_nc0 - nover;  // structure of this evaluation
    for (int i = l64; i < M; i += 64) L.C1[i] = 0.0;
    tpw::wsync();
    if (l64 == 0) {  // stationary covariance P_inf and P_inf h, rank by rank
        const double *cf = a.coef + ev;
        const int64_t cs = a.cstride;
        double D0 = 0.0;
        for (int j = 0; j < nr; ++j) {
            const double aj = cf[a.lay.ar(j) * cs];
            L.C1[j * J + j] = aj; L.v1[j] = aj; D0 += aj;
        }
        for (int q = 0; q < nc; ++q) {
            const double aa = cf[a.lay.ac(q) * cs], bb = cf[a.lay.bc(q) * cs], c = cf[a.lay.cc(q) * cs], d = cf[a.lay.dc(q) * cs];
            const double p = d != 0.0 ? (2.0 * d * (2.0 * c * bb + d * aa) + 4.0 * c * (c * aa - d * bb)) / (2.0 * d * d) : aa;
            const int o = nr + 2 * q;
            L.C1[o * J + o] = aa; L.C1[o * J + o + 1] = -bb; L.C1[(o + 1) * J + o] = -bb; L.C1[(o + 1) * J + o + 1] = p;
            L.v1[o] = aa; L.v1[o + 1] = -bb; D0 += aa;
        }
        const double2 y0 = a.yv[lc * a.N], t0 = a.dxt[lc * a.t_stride];
        D0 += y0.y + cf[a.lay.jit() * cs];
        const double z0 = y0.x - fma(cf[a.lay.mean(0) * cs], t0.y, cf[a.lay.mean(1) * cs]);
        L.v2[0] = z0 / D0; L.v2[1] = 1.0 / D0;
        double *h = head + ev * 4;
        h[0] = z0 * z0 / D0; h[1] = log(D0); h[2] = D0;
    }
    tpw::wsync();
    {
        const double chr = L.v1[w.r];
        double prow[2];
        prow[0] = L.C1[w.r * J + w.c0] - chr * L.v1[w.c0] * L.v2[1];
        prow[1] = L.C1[w.r * J + w.c0 + 1] - chr * L.v1[w.c0 + 1] * L.v2[1];
        const double m = chr * L.v2[0];
        tpw::wsync();   // (lanes beyond the last pair repeat it: everybody reads before anybody writes)
        L.b1[w.r] = m;
        tpw::put2<J>(L.C1, w.r, w.c0, prow);
    }
    tpw::wsync();
    return true;
}

// lnL of an evaluation from its n top-level elements and their likelihood records: the elements applied one after
// the other to the state after sample 0, each leaving its correction (tpg::apply) -- the samples were read once,
// by the composition pass.  The records are computed under a wrong hypothesis (x_in = 0), so the terms can be far
// larger than the result and cancel; an evaluation whose terms exceed 1e3 x the result, with anything not positive or
// not finite on the way, or with a complex term whose power spectrum can go negative (b d > a c: the matrix need not
// be positive definite and only the filter pass sees every pivot's sign, mtg_timeparallel.h) is appended to the redo
// list and goes through the down-sweep and the filter pass, whose pivots are celerite's own.
template <int J>
__global__ void __launch_bounds__(64, 2) mtg_tpb_top_direct_kernel(MtgSolveArgs a, const double *elems, const double *recs,
                                                                double *head, int n, int *redo_list, int *redo_count)
{
    __shared__ tpw::Lds<J> L;
    int64_t ev;
    int k;
    if (!tpb_group(a, 1, ev, k)) return;
    const int l64 = threadIdx.x;
    const tpw::Lane w = tpw::lane_of<J>(l64);
    if (!tpb_head_state<J>(a, ev, L, l64, w, head)) {
        if (l64 == 0) { a.out[ev] = -INFINITY; a.status[ev] = MTG_ST_NONFINITE; }
        return;
    }
    const int64_t first = ev * n;
    tpw::Pre<J> pre;
    tpw::fetch<J>(pre, elems + first * MTG_TPB_ELEM(J), l64);
    double corr = 0.0, mag = 0.0, dot = 0.0, ld = 0.0, dmin = INFINITY;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        tpw::put_second<J>(L, pre, l64);
        if (i + 1 < n) tpw::fetch<J>(pre, elems + (first + i + 1) * MTG_TPB_ELEM(J), l64);
        tpw::wsync();
        const double c = i + 1 < n ? tpw::apply<J, true, true>(L, w) : tpw::apply<J, true, false>(L, w);
        const double *q = recs + (first + i) * 4;
        corr += c; mag += fabs(c) + q[3];
        dot += q[0]; ld += q[1]; dmin = fmin(dmin, q[2]);
    }
    if (l64 == 0) {
        const double *h = head + ev * 4;
        mag += 0.5 * h[0];
        dot += h[0]; ld += h[1]; dmin = fmin(dmin, h[2]);
        const double ll = corr - 0.5 * (dot + ld + (double)a.N * MTG_LN_2PI);
        bool psd = true;
        const int nc = a.tp_nc0 - (a.sig ? a.sig[ev] : 0);
        const double *cf = a.coef + ev;
        for (int q = 0; q < nc; ++q)
            if (!(fabs(cf[a.lay.bc(q) * a.cstride] * cf[a.lay.dc(q) * a.cstride])
                  <= cf[a.lay.ac(q) * a.cstride] * cf[a.lay.cc(q) * a.cstride] * (1.0 + 1.0e-12)))
                psd = false;
        if (a.tp_direct >= 2 || (psd && dmin > 0.0 && isfinite(ll) && mag <= 1.0e3 * fabs(ll))) {  // 2: diagnostic, never redo
            a.out[ev] = a.tp_direct == 3 ? dot : a.tp_direct == 4 ? ld : a.tp_direct == 5 ? corr : a.tp_direct == 6 ? mag : ll;
            a.status[ev] = MTG_ST_OK;
        } else {
            redo_list[atomicAdd(redo_count, 1)] = (int)ev;
        }
    }
}

// Start states of the chunks, for the evaluations that go through the filter pass: ONE launch, one wave per group of
// chunks at the bottom level.  The wave starts from the filtered state after sample 0 (which also leaves that sample's
// terms of the likelihood in head[ev]) and walks down the tree to its first chunk: at the top level it applies the
// elements before its ancestor, at every level below the elements of its ancestor's group that come before the next
// ancestor -- at most (top - 1) + (g - 1) per level applications, ~16 for 1024 chunks --, then writes the start state of
// each of its own chunks, applying them in turn.  (Round 2 swept down level by level, one launch per level, each
// reading the states the level above had written: as many launches as levels in every half-step, for a list that is
// empty as a rule.  Evaluations on the list pay ~5 x the applications here; they are rare, and one launch it is.)
template <int J>
__global__ void __launch_bounds__(64, 2) mtg_tpb_descend_kernel(MtgSolveArgs a, MtgTpBigPlan p, double *ws)
{
    constexpr int M = J * J;
    __shared__ tpw::Lds<J> L;
    const int g0 = p.gl[0] > 0 ? p.gl[0] : p.n[0];   // (a single level: the chunks themselves are the top)
    const int gpe = p.n[0] / g0;
    int64_t ev;
    int k;
    if (!tpb_group(a, gpe, ev, k)) return;
    const int l64 = threadIdx.x;
    const tpw::Lane w = tpw::lane_of<J>(l64);
    if (!tpb_head_state<J>(a, ev, L, l64, w, ws + p.head_off)) return;
    // ancestors of the group's first chunk, level by level
    int idx[MTG_TPB_MAX_LEVELS];
    idx[0] = k * g0;
    for (int l = 0; l + 1 < p.nlev; ++l) idx[l + 1] = idx[l] / p.gl[l];
    tpw::Pre<J> pre;
    auto run = [&](const double *elems, int64_t from, int64_t to) {   // the state through elements [from, to)
        if (to <= from) return;
        tpw::fetch<J>(pre, elems + from * MTG_TPB_ELEM(J), l64);
        for (int64_t e = from; e < to; ++e) {
            tpw::put_second<J>(L, pre, l64);
            if (e + 1 < to) tpw::fetch<J>(pre, elems + (e + 1) * MTG_TPB_ELEM(J), l64);
            tpw::wsync();
            tpw::apply<J>(L, w);
        }
    };
    const int top = p.nlev - 1;
    for (int l = top; l >= 1; --l) {
        const int64_t base = ev * p.n[l];
        const int first = l == top ? 0 : idx[l + 1] * p.gl[l];
        run(ws + p.elem_off[l], base + first, base + idx[l]);
    }
    // the group's own chunks
    const int64_t first = ev * p.n[0] + idx[0];
    double *states = ws + p.state_off[0];
    const double *elems0 = ws + p.elem_off[0];
    tpw::fetch<J>(pre, elems0 + first * MTG_TPB_ELEM(J), l64);
#pragma unroll 1
    for (int i = 0; i < g0; ++i) {
        tpw::gcopy(states + (first + i) * MTG_TPB_STATE(J), L.b1, J + M, l64);
        if (i + 1 == g0) break;
        tpw::put_second<J>(L, pre, l64);
        tpw::fetch<J>(pre, elems0 + (first + i + 1) * MTG_TPB_ELEM(J), l64);
        tpw::wsync();
        tpw::apply<J>(L, w);
    }
}

__global__ void __launch_bounds__(64) mtg_tpb_finish_kernel(MtgSolveArgs a, const double *parts, const double *head, int C)
{
    const int64_t count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
    if ((int64_t)blockIdx.x >= count) return;
    const int64_t ev = a.list ? (int64_t)a.list[blockIdx.x] : (int64_t)blockIdx.x;
    if (!a.list && a.status[ev] != MTG_ST_OK) return;
    const int lane = threadIdx.x;
    const int64_t lc = a.lc_index ? (int64_t)a.lc_index[ev] : 0;
    if (lc < 0 || (uint64_t)(lc + 1) * (uint64_t)a.N * 16u > (uint64_t)a.yv_bytes) {
        if (lane == 0) { a.out[ev] = -INFINITY; a.status[ev] = MTG_ST_NONFINITE; }
        return;
    }
    double dot = 0.0, ld = 0.0, dmin = INFINITY;
    for (int c = lane; c < C; c += 64) {
        const double *p = parts + (ev * C + c) * 4;
        dot += p[0]; ld += p[1]; dmin = fmin(dmin, p[2]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        dot += __shfl_down(dot, off);
        ld += __shfl_down(ld, off);
        dmin = fmin(dmin, __shfl_down(dmin, off));
    }
    if (lane == 0) {
        const double *h = head + ev * 4;
        dot += h[0]; ld += h[1]; dmin = fmin(dmin, h[2]);
        double ll = -0.5 * (dot + ld + (double)a.N * MTG_LN_2PI);
        int st = MTG_ST_OK;
        if (!(dmin > 0.0)) { st = MTG_ST_NOTPD; ll = -INFINITY; }
        else if (!isfinite(ll)) { st = MTG_ST_NONFINITE; ll = -INFINITY; }
        a.out[ev] = ll;
        a.status[ev] = st;
    }
}

template <int J>
void launch_up(const MtgSolveArgs &a, const MtgTpBigPlan &p, int64_t nevals, int kappa, int *zero_me, hipStream_t s)
{
    double *ws = a.tp_ws;
    auto blocks = [&](int64_t groups_per_eval) { return dim3((unsigned)((nevals * groups_per_eval + GROUPS - 1) / GROUPS)); };
    for (int l = 0; l + 1 < p.nlev; ++l) {
        const double *rec_in = ws + (l == 0 ? p.part_off : p.rec_off[l]);
        if (kappa)
            hipLaunchKernelGGL((mtg_tpb_reduce_kernel<J, true>), blocks(p.n[l] / p.gl[l]), dim3(64), 0, s, a, ws + p.elem_off[l],
                               ws + p.elem_off[l + 1], p.n[l], p.gl[l], rec_in, ws + p.rec_off[l + 1], l == 0 ? 1 : 0,
                               l == 0 ? zero_me : (int *)nullptr);
        else
            hipLaunchKernelGGL((mtg_tpb_reduce_kernel<J, false>), blocks(p.n[l] / p.gl[l]), dim3(64), 0, s, a, ws + p.elem_off[l],
                               ws + p.elem_off[l + 1], p.n[l], p.gl[l], (const double *)nullptr, (double *)nullptr, 0, (int *)nullptr);
    }
}

template <int J>
void launch_top_direct(const MtgSolveArgs &a, const MtgTpBigPlan &p, int64_t nevals, int *redo_list, int *redo_count, hipStream_t s)
{
    double *ws = a.tp_ws;
    const int top = p.nlev - 1;  // (the chunk count is at least 64, so the top level is never level 0)
    hipLaunchKernelGGL((mtg_tpb_top_direct_kernel<J>), dim3((unsigned)((nevals + GROUPS - 1) / GROUPS)), dim3(64), 0, s, a,
                       ws + p.elem_off[top], ws + p.rec_off[top], ws + p.head_off, p.n[top], redo_list, redo_count);
}

template <int J>
void launch_down(const MtgSolveArgs &a, const MtgTpBigPlan &p, int64_t nevals, hipStream_t s)
{
    const int g0 = p.gl[0] > 0 ? p.gl[0] : p.n[0];
    const int64_t groups = nevals * (p.n[0] / g0);
    hipLaunchKernelGGL((mtg_tpb_descend_kernel<J>), dim3((unsigned)groups), dim3(64), 0, s, a, p, a.tp_ws);
}

}  // namespace

void mtg_launch_tpb_up(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, int kappa, int *zero_me,
                       hipStream_t stream)
{
    if (J == 10) launch_up<10>(a, plan, nevals, kappa, zero_me, stream);
}

void mtg_launch_tpb_top_direct(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, int *redo_list,
                               int *redo_count, hipStream_t stream)
{
    if (J == 10) launch_top_direct<10>(a, plan, nevals, redo_list, redo_count, stream);
}

void mtg_launch_tpb_down(int J, const MtgSolveArgs &a, const MtgTpBigPlan &plan, int64_t nevals, hipStream_t stream)
{
    if (J == 10) launch_down<10>(a, plan, nevals, stream);
}

void mtg_launch_tpb_finish(const MtgSolveArgs &a, const double *parts, const double *head, int C, int64_t nevals,
                           hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_tpb_finish_kernel, dim3((unsigned)nevals), dim3(64), 0, stream, a, parts, head, C);
}
