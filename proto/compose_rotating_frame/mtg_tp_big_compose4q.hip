// mtg_tp_big_compose4q.hip -- the four-wave composition kernel of the rank-10 time-parallel path (mtg_tp_big.h) as
// workgroups of ONE quartet (256 lanes), two workgroups per CU with independent barriers, and the roles handed out by
// SIMD so that every SIMD still runs one "columns" and one "filter" wave.
//
// With both quartets of a CU in one eight-wave workgroup (round 3) they stop at the same barrier: every wave of the CU
// drains its LDS writes, waits and then fetches its first operands at the same moment, and the FP64 pipe idles
// meanwhile (35 % of the SIMD cycles, profiles/r03_compose_pmc.txt).  Two independent workgroups drift apart and fill
// each other's waits -- but left to the dispatcher their waves pair up at random, "columns" with "columns" as often as
// not (the first version of the kernel: no gain).  So a wave takes its role from the SIMD it runs on: the four waves of
// a quartet land on four different SIMDs (scripts/micro/simd_probe.hip; checked at run time, wave order otherwise),
// role = SIMD id in one workgroup of a CU and 3 - SIMD id in the other -- which of the two a workgroup is, it learns
// from a counter of its CU (XCC, shader engine, array, CU id of HW_REG_HW_ID / HW_REG_XCC_ID).
// 256 + 256-entry tables (2 + 4 KiB: the same polynomials as 1024 + 512 entries, mtg_math.h) next to the 72 KiB of rings:
// two workgroups fit a CU (2 x 78 KiB).
#define MTG_EXP_BITS 8
#define MTG_TRIG_BITS 8
#include "mtg_tp_big.h"

namespace {

__device__ unsigned int tpb_cu_arrivals[8 * 8 * 2 * 16];   // [xcc][se][sh][cu]: workgroups that have started there

__global__ void __launch_bounds__(256, 2) mtg_tpb_compose4q_kernel(MtgSolveArgs a, double *elems, double *parts, int C)
{
    __shared__ TpbRing4<10> ring;
    __shared__ MtgMathTables tab;
    __shared__ int s_simd[4];
    __shared__ int s_flip;
    const int64_t ev = tpb_evaluation(a, blockIdx.y);
    if (ev < 0) return;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = threadIdx.x >> 6, simd = (hw >> 4) & 3;
    if ((threadIdx.x & 63) == 0) s_simd[wave] = simd;
    if (threadIdx.x == 0) {
        const unsigned cu = ((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 32u + ((hw >> 12) & 1u) * 16u + ((hw >> 8) & 15u);
        s_flip = (int)(atomicAdd(&tpb_cu_arrivals[cu], 1u) & 1u);
    }
    mtg_fill_tables(&tab, threadIdx.x, 256);
    __syncthreads();
    // roles by SIMD when the quartet sits on four different SIMDs (a permutation of 0..3), by wave order otherwise
    const bool spread = (1 << s_simd[0] | 1 << s_simd[1] | 1 << s_simd[2] | 1 << s_simd[3]) == 15;
    // (the second workgroup of a CU mirrors the roles, 3 - simd: every SIMD runs one "columns" and one "filter" wave, the
    // heavier columns wave -- role 0, three of the five terms' generators -- with the lighter filter wave, role 3)
    const int r0 = spread ? simd : wave;
    // (wave-uniform by construction; said so, the four roles are four scalar branches, not four masked passes)
    const int role = __builtin_amdgcn_readfirstlane(s_flip ? 3 - r0 : r0);
    tpb_dispatch<TpbCompose4F>(tpb_nr(a, ev), a, ev, elems, parts, C, ring, &tab, blockIdx.x, role);
}

}  // namespace

void mtg_launch_tpb_compose4q(const MtgSolveArgs &a, double *elems, double *parts, int C, int64_t nevals, hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_tpb_compose4q_kernel, dim3((unsigned)(C / 64), (unsigned)nevals), dim3(256), 0, stream, a, elems, parts, C);
}
