// mtg_kernels_pipe.hip -- the serial sweep as a two-wave pipeline (mtg_sweep_pipe.h) for batches that leave the
// one-lane-per-evaluation launch with at most one wave on half of the SIMDs: ~8 000 to 32 768 rows, e.g. one GPU's
// share of the Protassov refits at 8 GPUs (250 light curves x 128 walkers per half-step;
// reference docs/notebooks/tutorial_ppp.ipynb:326-343, gpmodelling.py:247-248).
//
// A workgroup is four waves on the four SIMDs of a CU: waves 0 and 1 produce the generators of 64 evaluations each,
// waves 2 and 3 consume them; 144 KiB of LDS (tables + rings) keep it the only workgroup of its CU, so 256 CUs take
// 32 768 rows in one round.  Every structure of a model (SHO terms on either side of Q = 1/2) in one launch, as
// mtg_kernels_multi.hip does it: rows in the order of the library's sort by (structure, light curve), every structure's
// segment padded to whole workgroups.
#include "mtg_sweep_pipe.h"

namespace {

template <int NR, int NC, int LASTB0>
struct PipeB0 { static constexpr int value = (LASTB0 && NC > 0 && NR < 5 && NC < 4 && NR + 2 * NC <= 6) ? 1 : 0; };

template <int NR, int NC, int NB0, int CH>
__device__ __forceinline__ void pipe_rows(const MtgSolveArgs &a, int64_t e, bool active, int wave, double2 *ring,
                                          const MtgMathTables *tab)
{
    // the row's light curve (inside the one descriptor window: the launcher checks yv_bytes <= window_bytes)
    uint32_t lc = a.lc_index ? (uint32_t)a.lc_index[e] : 0u;
    const uint64_t lc_bytes = (uint64_t)a.N * 16u;
    bool lost = false;
    if (((uint64_t)lc + 1u) * lc_bytes > a.yv_bytes) { lost = active; lc = 0; active = false; }
    const uint32_t yoff = (uint32_t)((uint64_t)lc * lc_bytes);
    const uint32_t toff = a.t_stride ? yoff : 0u;
    if (wave < 2) {
        // table or libm sincos: decided per 64 rows exactly as mtg_solve_row decides it per wave
        double dmax = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) dmax = fmax(dmax, fabs(a.coef[e + a.lay.dc(k) * a.cstride]));
        const bool fast = !__any(active && !(dmax * *a.dxmax <= MTG_TRIG_FAST_MAX));
        if (fast) mtg_pipe_produce<NR, NC, true, CH>(a, e, toff, ring, tab);
        else mtg_pipe_produce<NR, NC, false, CH>(a, e, toff, ring, tab);
    } else {
        if (lost) {  // a device-side lc_index outside the resident set: no likelihood (as mtg_solve_row)
            a.out[e] = -INFINITY;
            a.status[e] = MTG_ST_NONFINITE;
        }
        if (a.has_mean) mtg_pipe_consume<NR, NC, NB0, true, CH>(a, e, active, yoff, toff, ring);
        else mtg_pipe_consume<NR, NC, NB0, false, CH>(a, e, active, yoff, toff, ring);
    }
}

template <int NR0, int NC0, int NSIG, int LASTB0, int CH, int S = 0>
__device__ __forceinline__ void pipe_dispatch(int k, const MtgSolveArgs &a, int64_t e, bool active, int wave,
                                              double2 *ring, const MtgMathTables *tab)
{
    if (k == S) pipe_rows<NR0 + 2 * S, NC0 - S, PipeB0<NR0 + 2 * S, NC0 - S, LASTB0>::value, CH>(a, e, active, wave, ring, tab);
    else if constexpr (S + 1 < NSIG) pipe_dispatch<NR0, NC0, NSIG, LASTB0, CH, S + 1>(k, a, e, active, wave, ring, tab);
}

template <int NR0, int NC0, int NSIG, int LASTB0>
__global__ void __launch_bounds__(MTG_PIPE_BLOCK, 1) mtg_pipe_kernel(MtgSolveArgs a)
{
    constexpr int N2 = MtgPipeShape<NR0, NC0>::N2;   // the widest hand-over: the structure with every SHO under-damped
    // workgroup -> (structure, first row of the workgroup inside the structure's segment)
    int64_t block = blockIdx.x, first = 0, count = 0;
    int k = 0;
    if (NSIG > 1) {
        for (; k < NSIG; ++k) {
            count = a.seg_counts[k];
            const int64_t blocks = (count + MTG_PIPE_ROWS - 1) / MTG_PIPE_ROWS;
            if (block < blocks) break;
            block -= blocks;
            first += count;
        }
        if (k == NSIG) return;  // the grid is sized for the worst padding
    } else {
        count = a.count_ptr ? (int64_t)*a.count_ptr : a.B;
        if (block * MTG_PIPE_ROWS >= count) return;
    }
    __shared__ MtgMathTables tab;
    constexpr int CH = mtg_pipe_chunk(N2);
    __shared__ double2 ring[2][MTG_PIPE_RING * CH * N2 * 64];
    mtg_fill_tables(&tab, threadIdx.x, MTG_PIPE_BLOCK);
    __syncthreads();
    const int wave = threadIdx.x >> 6, pair = wave & 1, lane = threadIdx.x & 63;
    const int64_t gid = block * MTG_PIPE_ROWS + pair * 64 + lane;
    bool active = gid < count;
    // idle lanes walk along on row 0 of the batch (any row with readable coefficients) and store nothing
    int64_t e = 0;
    if (active) e = a.list ? (int64_t)a.list[first + gid] : gid;
    if (active && a.status[e] != MTG_ST_OK) active = false;  // prior said -inf, or another rank's row
    pipe_dispatch<NR0, NC0, NSIG, LASTB0, CH>(k, a, e, active, wave, &ring[pair][lane], &tab);
}

template <int NR0, int NC0, int NSIG, int LASTB0>
void launch_pipe(const MtgSolveArgs &a, int64_t nlanes, hipStream_t stream)
{
    const int64_t blocks = (nlanes + MTG_PIPE_ROWS - 1) / MTG_PIPE_ROWS + (NSIG > 1 ? NSIG : 0);
    if (blocks <= 0) return;
    hipLaunchKernelGGL((mtg_pipe_kernel<NR0, NC0, NSIG, LASTB0>), dim3((unsigned)blocks), dim3(MTG_PIPE_BLOCK), 0, stream, a);
}

// ranks 3 to 6 with one or two complex terms in the widest structure (a real-only model has next to nothing to hand
// over; three complex terms hand over more than the ring holds)
#ifdef MTG_PIPE_FEW   // experiments (scripts/build_pipe_variant.sh): the two models of the Protassov test only
#define MTG_PIPE_WANTED(nr0, nc0, nsig, lastb0) ((nr0) == 1 && (nsig) == 2 && (((nc0) == 1 && !(lastb0)) || ((nc0) == 2 && (lastb0))))
#else
#define MTG_PIPE_WANTED(nr0, nc0, nsig, lastb0) true
#endif
template <int NR0, int NC0, int NSIG, int LASTB0,
          bool OK = (NC0 >= 1 && NSIG - 1 <= NC0 && NR0 + 2 * NC0 >= 3 && NR0 + 2 * NC0 <= 6 && MtgPipeShape<NR0, NC0>::N2 <= 4 && MTG_PIPE_WANTED(NR0, NC0, NSIG, LASTB0))>
struct Sel { static constexpr mtg_solve_launcher fn = launch_pipe<NR0, NC0, NSIG, LASTB0>; };
template <int NR0, int NC0, int NSIG, int LASTB0>
struct Sel<NR0, NC0, NSIG, LASTB0, false> { static constexpr mtg_solve_launcher fn = nullptr; };

#define CELL(nr0, nc0) { { Sel<nr0, nc0, 1, 0>::fn, Sel<nr0, nc0, 1, 1>::fn }, { Sel<nr0, nc0, 2, 0>::fn, Sel<nr0, nc0, 2, 1>::fn }, \
                         { Sel<nr0, nc0, 3, 0>::fn, Sel<nr0, nc0, 3, 1>::fn } }
#define ROW(nr0) { CELL(nr0, 1), CELL(nr0, 2), CELL(nr0, 3) }
const mtg_solve_launcher table[5][3][3][2] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4)};

}  // namespace

// nr0 real + nc0 complex terms in the all-under-damped structure, nsig structures (1: list / count_ptr as for
// mtg_solve_kernel; > 1: a.list = the sorted order, a.seg_counts = rows per structure), last complex term with b = 0
mtg_solve_launcher mtg_find_pipe_solver(int nr0, int nc0, int nsig, int last_b0)
{
    if (nr0 < 0 || nr0 > 4 || nc0 < 1 || nc0 > 3 || nsig < 1 || nsig > 3) return nullptr;
    return table[nr0][nc0 - 1][nsig - 1][last_b0 ? 1 : 0];
}
