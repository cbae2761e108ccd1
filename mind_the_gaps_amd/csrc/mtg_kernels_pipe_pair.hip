// mtg_kernels_pipe_pair.hip -- the pipelined sweeps of TWO models in one launch: the null and the alternative kernel of
// the Protassov test, refitted side by side on one GPU's share of the simulated light curves (250 x 128 rows per
// half-step and model at 8 GPUs; reference docs/notebooks/tutorial_ppp.ipynb:326-343: both kernels are fitted to every
// simulated light curve).
//
// Why.  mtg_pipe_kernel (mtg_kernels_pipe.hip) is one workgroup of four waves per CU -- its tables and rings take 144 KiB
// of the CU's 160 KiB of LDS -- so the two models' launches, though they come from two contexts on two hardware queues,
// cannot share a CU: they take turns on the compute units, each leaving every SIMD with ONE wave that issues 62 % of
// the time (profiles/r04_midbatch_sq_counters.txt).  Here a workgroup is EIGHT waves: a quartet of model A and a quartet of
// model B (mtg_pipe_quartet, the body of mtg_pipe_kernel), ONE table set between them, two samples per hand-over instead
// of four (the rings halve: 48 + 48 + 24 KiB for the configs[3] pair) -- two waves per SIMD, the issue slots one
// model's wave leaves are what the other's needs.  Waves w and w + 4 of a workgroup share a SIMD
// (scripts/micro/simd_probe.hip), so model B's roles are rotated by two: every SIMD runs a producer of one model and a
// consumer of the other.  The eight waves meet at the same s_barrier: both models walk the same N samples in chunks of
// two, so they pass the same number of barriers; a quartet without rows (the two models' grids differ in size) ends
// at once, and waves that have ended no longer count at a barrier.
// A row's arithmetic is that of mtg_pipe_kernel (the chunk size changes where the barriers fall, not what is computed):
// bit-identical results (tests/test_pipe_gpu.py).
#include "mtg_sweep_pipe.h"

namespace {

constexpr int PAIR_CH = 2;   // samples per hand-over (even: the pivot product is renormalised in pairs)

template <int NR0, int NC0, int NSIG, int LASTB0> struct Shape {
    static constexpr int nr0 = NR0, nc0 = NC0, nsig = NSIG, lastb0 = LASTB0;
    static constexpr int N2 = MtgPipeShape<NR0, NC0>::N2;
    static constexpr int ring = 2 * MTG_PIPE_RING * PAIR_CH * N2 * 64;   // double2 slots of the quartet's two rings
};

template <class A, class B>
__global__ void __launch_bounds__(2 * MTG_PIPE_BLOCK, 1) mtg_pipe_pair_kernel(MtgSolveArgs a, MtgSolveArgs b)
{
    __shared__ MtgMathTables tab;
    __shared__ double2 ring_a[A::ring];
    __shared__ double2 ring_b[B::ring];
    const bool has_a = mtg_pipe_has_block<A::nsig>(a, blockIdx.x), has_b = mtg_pipe_has_block<B::nsig>(b, blockIdx.x);
    if (!has_a && !has_b) return;
    mtg_fill_tables(&tab, threadIdx.x, 2 * MTG_PIPE_BLOCK);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (has_a) mtg_pipe_quartet<A::nr0, A::nc0, A::nsig, A::lastb0, PAIR_CH>(a, blockIdx.x, wave, lane, ring_a, &tab);
    } else {
        // (roles rotated by two: waves 4, 5 -- on the SIMDs of A's producers -- are B's consumers)
        if (has_b) mtg_pipe_quartet<B::nr0, B::nc0, B::nsig, B::lastb0, PAIR_CH>(b, blockIdx.x, (wave + 2) & 3, lane, ring_b, &tab);
    }
}

template <class A, class B>
void launch_pair(const MtgSolveArgs &a, int64_t na, const MtgSolveArgs &b, int64_t nb, hipStream_t stream)
{
    const int64_t ba = (na + MTG_PIPE_ROWS - 1) / MTG_PIPE_ROWS + (A::nsig > 1 ? A::nsig : 0);
    const int64_t bb = (nb + MTG_PIPE_ROWS - 1) / MTG_PIPE_ROWS + (B::nsig > 1 ? B::nsig : 0);
    const int64_t blocks = ba > bb ? ba : bb;
    if (blocks <= 0) return;
    hipLaunchKernelGGL((mtg_pipe_pair_kernel<A, B>), dim3((unsigned)blocks), dim3(2 * MTG_PIPE_BLOCK), 0, stream, a, b);
}

struct Entry { MtgPipeShapeId a, b; mtg_pipe_pair_launcher fn; };
#define PAIR(a0, a1, a2, a3, b0, b1, b2, b3) { {a0, a1, a2, a3}, {b0, b1, b2, b3}, launch_pair<Shape<a0, a1, a2, a3>, Shape<b0, b1, b2, b3>> }
// A null model and the alternative that adds one term to it -- the pairs the posterior-predictive test is made of -- for
// nulls that have a pipelined sweep at all (rank 3 to 6 with a complex term; a DRW, an SHO or a Lorentzian alone hands
// nothing over or is below rank 3):
//   shape = <real terms, complex terms with every SHO under-damped, structures = SHO terms + 1, last complex term has b = 0>
const Entry pairs[] = {
    PAIR(1, 1, 2, 0, 1, 2, 2, 1),   // DRW + SHO            | + Lorentzian        (BASELINE configs[3])
    PAIR(1, 1, 2, 0, 1, 2, 3, 0),   // DRW + SHO            | + SHO
    PAIR(1, 1, 2, 0, 2, 1, 2, 0),   // DRW + SHO            | + real term
    PAIR(1, 1, 1, 1, 1, 2, 1, 1),   // DRW + Lorentzian     | + Lorentzian
    PAIR(1, 1, 1, 1, 1, 2, 2, 0),   // DRW + Lorentzian     | + SHO
    PAIR(0, 2, 2, 1, 1, 2, 2, 1),   // SHO + Lorentzian     | + DRW
    PAIR(0, 2, 3, 0, 1, 2, 3, 0),   // SHO + SHO            | + DRW
    PAIR(2, 1, 2, 0, 2, 2, 2, 1),   // DRW + real + SHO     | + Lorentzian
};

bool same(const MtgPipeShapeId &x, const MtgPipeShapeId &y)
{
    return x.nr0 == y.nr0 && x.nc0 == y.nc0 && x.nsig == y.nsig && (x.last_b0 != 0) == (y.last_b0 != 0);
}

}  // namespace

// the launcher for (a, b) in THIS order, or NULL (the caller tries (b, a) with the arguments swapped)
mtg_pipe_pair_launcher mtg_find_pipe_pair_solver(const MtgPipeShapeId &a, const MtgPipeShapeId &b)
{
    for (const Entry &e : pairs)
        if (same(e.a, a) && same(e.b, b)) return e.fn;
    return nullptr;
}
