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

template <int NR0, int NC0, int NSIG, int LASTB0>
__global__ void __launch_bounds__(MTG_PIPE_BLOCK, 1) mtg_pipe_kernel(MtgSolveArgs a)
{
    constexpr int N2 = MtgPipeShape<NR0, NC0>::N2;   // the widest hand-over: the structure with every SHO under-damped
    constexpr int CH = mtg_pipe_chunk(N2);
    __shared__ MtgMathTables tab;
    __shared__ double2 ring[2 * MTG_PIPE_RING * CH * N2 * 64];
    if (!mtg_pipe_has_block<NSIG>(a, blockIdx.x)) return;
    mtg_fill_tables(&tab, threadIdx.x, MTG_PIPE_BLOCK);
    __syncthreads();
    mtg_pipe_quartet<NR0, NC0, NSIG, LASTB0, CH>(a, blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, ring, &tab);
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
