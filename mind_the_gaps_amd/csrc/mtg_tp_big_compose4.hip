// mtg_tp_big_compose4.hip -- the composition kernel of the rank-10 time-parallel path with FOUR waves per 64 chunks
// (mtg_tp_big.h, round 3), every structure in one kernel.
// 2048 + 1024-entry tables (16 + 16 KiB) shared by the two quartets of a workgroup, next to 2 x 62 KiB of rings: one
// workgroup of eight waves per CU
#define MTG_EXP_BITS 11
#define MTG_TRIG_BITS 10
#include "mtg_tp_big.h"

namespace {

// 512 lanes = two quartets of waves; work item i = (evaluation i / (C / 64), chunk block i % (C / 64)); workgroup b takes
// items 2 b and 2 b + 1.  Waves w and w + 4 share a SIMD: the second quartet's roles are rotated by two.
__global__ void __launch_bounds__(512, 2) mtg_tpb_compose4_kernel(MtgSolveArgs a, double *elems, double *parts, int C, int64_t nitems)
{
    __shared__ TpbRing4<10> ring[2];
    __shared__ MtgMathTables tab;
    const int quartet = threadIdx.x >> 8, wave = (threadIdx.x >> 6) & 3;
    const int64_t item = (int64_t)blockIdx.x * 2 + quartet;
    const uint32_t cbs = (uint32_t)C / 64u;
    mtg_fill_tables(&tab, threadIdx.x, 512);
    __syncthreads();
    if (item >= nitems) return;
    const int64_t ev = tpb_evaluation(a, item / cbs);
    if (ev < 0) return;
    tpb_dispatch<TpbCompose4F>(tpb_nr(a, ev), a, ev, elems, parts, C, ring[quartet], &tab, (uint32_t)(item % cbs),
                               (wave + 2 * quartet) & 3);
}

}  // namespace

void mtg_launch_tpb_compose4(const MtgSolveArgs &a, double *elems, double *parts, int C, int64_t nevals, hipStream_t stream)
{
    const int64_t nitems = nevals * (C / 64);
    hipLaunchKernelGGL(mtg_tpb_compose4_kernel, dim3((unsigned)((nitems + 1) / 2)), dim3(512), 0, stream, a, elems, parts, C, nitems);
}
