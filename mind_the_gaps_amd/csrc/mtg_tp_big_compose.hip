// mtg_tp_big_compose.hip -- the composition kernel of the rank-10 time-parallel path (mtg_tp_big.h),
// every structure in one kernel.
// 2048 + 1024-entry tables (16 + 16 KiB): next to the 37 KiB of rings two composition workgroups still fit a CU
#define MTG_EXP_BITS 11
#define MTG_TRIG_BITS 10
#include "mtg_tp_big.h"

namespace {

// grid (C / 64, evaluations), 128 lanes
__global__ void __launch_bounds__(128, 1) mtg_tpb_compose2_kernel(MtgSolveArgs a, double *elems, double *parts, int C)
{
    __shared__ TpbRing<10> ring;
    __shared__ MtgMathTables tab;
    const int64_t ev = tpb_evaluation(a, blockIdx.y);
    if (ev < 0) return;
    mtg_fill_tables(&tab, threadIdx.x, 128);
    __syncthreads();
    tpb_dispatch<TpbComposeF>(tpb_nr(a, ev), a, ev, elems, parts, C, ring, &tab);
}

}  // namespace

void mtg_launch_tpb_compose(const MtgSolveArgs &a, double *elems, double *parts, int C, int64_t nevals, hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_tpb_compose2_kernel, dim3((unsigned)(C / 64), (unsigned)nevals), dim3(128), 0, stream, a, elems, parts, C);
}
