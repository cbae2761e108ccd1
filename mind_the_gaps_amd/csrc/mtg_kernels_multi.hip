// mtg_kernels_multi.hip -- the serial sweep for a batch whose rows fall into several structures, in ONE launch.
//
// An SHO term is one complex celerite term when under-damped and two real ones when over-damped: the rows of a batch
// then need different instantiations of the sweep.  One launch per structure on one stream costs their latencies in a
// row -- the sweep is N dependent steps whatever the number of rows, and a handful of over-damped walkers added
// 3.2-4.3 ms to a 14.9 ms half-step (profiles/r03_c3_halfstep_trace.txt); on side streams the launches overlap, at the
// price of events to fork and join and of launch-order games to get the rarer structure's waves resident.
//
// Here every workgroup looks up which structure its rows belong to and runs that instantiation: the rows come in
// the order of the library's stable sort by (structure, light curve) (mtg_sort.hip), each structure's segment padded
// to whole workgroups, so a workgroup is uniform and all workgroups ask for the same registers (the maximum over the
// structures, which for the models at hand is the common structure's own count).  Measured against the side streams
// on one box: 0.3 % ahead; the gain is the simpler dispatch.
#include "mtg_sweep.h"

namespace {

// the b = 0 specialisation the one-structure launcher would pick for (NR, NC) (mtg_solver_uses_b0)
template <int NR, int NC, int LASTB0>
struct B0Of { static constexpr int value = (LASTB0 && NC > 0 && NR < 5 && NC < 4 && NR + 2 * NC <= 6) ? 1 : 0; };

template <int NR0, int NC0, int NSIG, int LASTB0, int K = 0>
__device__ __forceinline__ void multi_dispatch(int k, const MtgSolveArgs &a, int64_t e, const MtgMathTables *tab)
{
    if (k == K) mtg_solve_row<NR0 + 2 * K, NC0 - K, B0Of<NR0 + 2 * K, NC0 - K, LASTB0>::value>(a, e, tab);
    else if constexpr (K + 1 < NSIG) multi_dispatch<NR0, NC0, NSIG, LASTB0, K + 1>(k, a, e, tab);
}

constexpr int multi_waves(int nr0, int nc0)
{
    return mtg_waves_for(nr0 + 2 * nc0);   // every structure of a model has the same rank J
}

template <int NR0, int NC0, int NSIG, int LASTB0>
__global__ void __launch_bounds__(MTG_BLOCK, multi_waves(NR0, NC0)) mtg_solve_kernel_multi(MtgSolveArgs a)
{
    // workgroup -> (structure, first row of the workgroup inside the structure's segment)
    int64_t block = blockIdx.x, first = 0, count = 0;
    int k = 0;
    for (; k < NSIG; ++k) {
        count = a.seg_counts[k];
        const int64_t blocks = (count + MTG_BLOCK - 1) / MTG_BLOCK;
        if (block < blocks) break;
        block -= blocks;
        first += count;
    }
    if (k == NSIG) return;  // the grid is sized for the worst padding
    __shared__ MtgMathTables tab;
    mtg_fill_tables(&tab, threadIdx.x, MTG_BLOCK);
    __syncthreads();
    const int64_t gid = block * MTG_BLOCK + threadIdx.x;
    if (gid >= count) return;
    const int64_t e = a.list[first + gid];
    if (a.status[e] != MTG_ST_OK) return;
    multi_dispatch<NR0, NC0, NSIG, LASTB0>(k, a, e, &tab);
}

template <int NR0, int NC0, int NSIG, int LASTB0>
void launch_multi(const MtgSolveArgs &a, int64_t nlanes, hipStream_t stream)
{
    const int64_t blocks = (nlanes + MTG_BLOCK - 1) / MTG_BLOCK + NSIG;
    hipLaunchKernelGGL((mtg_solve_kernel_multi<NR0, NC0, NSIG, LASTB0>), dim3((unsigned)blocks), dim3(MTG_BLOCK), 0, stream, a);
}

// models of rank J <= 6 with one or two SHO terms: NR0 real and NC0 complex terms when every SHO is under-damped
template <int NR0, int NC0, int NSIG, int LASTB0, bool OK = (NSIG - 1 <= NC0 && NR0 + 2 * NC0 <= 6)>
struct Sel { static constexpr mtg_solve_launcher fn = launch_multi<NR0, NC0, NSIG, LASTB0>; };
template <int NR0, int NC0, int NSIG, int LASTB0>
struct Sel<NR0, NC0, NSIG, LASTB0, false> { static constexpr mtg_solve_launcher fn = nullptr; };

#define CELL(nr0, nc0) { { Sel<nr0, nc0, 2, 0>::fn, Sel<nr0, nc0, 2, 1>::fn }, { Sel<nr0, nc0, 3, 0>::fn, Sel<nr0, nc0, 3, 1>::fn } }
#define ROW(nr0) { CELL(nr0, 1), CELL(nr0, 2), CELL(nr0, 3) }
const mtg_solve_launcher table[5][3][2][2] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4)};

}  // namespace

// nr0 real + nc0 complex terms in the all-under-damped structure, nsig structures, last complex term with b = 0 or not
mtg_solve_launcher mtg_find_multi_solver(int nr0, int nc0, int nsig, int last_b0)
{
    if (nr0 < 0 || nr0 > 4 || nc0 < 1 || nc0 > 3 || nsig < 2 || nsig > 3) return nullptr;
    return table[nr0][nc0 - 1][nsig - 2][last_b0 ? 1 : 0];
}
