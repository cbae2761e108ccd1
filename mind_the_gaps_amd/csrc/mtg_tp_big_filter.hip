// mtg_tp_big_filter.hip -- the filter pass of the rank-10 time-parallel path (mtg_tp_big.h), every
// structure in one kernel, and the launch sequence of the path.
#define MTG_EXP_BITS 11
#define MTG_TRIG_BITS 10
#include "mtg_tp_big.h"

namespace {

// grid (ceil(C / 256), evaluations), 256 lanes: lane = chunk; four waves share one set of tables
__global__ void __launch_bounds__(256, 1) mtg_tpb_filter_kernel(MtgSolveArgs a, const double *states, double *parts, int C)
{
    __shared__ MtgMathTables tab;
    const int64_t ev = tpb_evaluation(a, blockIdx.y);
    if (ev < 0) return;
    mtg_fill_tables(&tab, threadIdx.x, 256);
    __syncthreads();
    tpb_dispatch<TpbFilterF>(tpb_nr(a, ev), a, ev, states, parts, C, &tab);
}

}  // namespace

void mtg_launch_tpb_filter(const MtgSolveArgs &a, const double *states, double *parts, int C, int64_t nevals, hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_tpb_filter_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)nevals), dim3(256), 0, stream, a, states,
                       parts, C);
}

// Every prepared evaluation of a rank-10 model (status OK; its structure in a.sig) in one sequence of
// launches.  a.tp_direct: the likelihood from the composition pass and the up-sweep alone (mtg_tp_scan.h:
// the elements carry their likelihood records); the down-sweep and the filter pass then run for the
// evaluations on the redo list only -- as a rule none, and their workgroups leave at once.  Otherwise the filter
// pass runs for everybody.
void mtg_launch_tp_big(const MtgSolveArgs &a, int64_t nevals, hipStream_t s)
{
    constexpr int J = 10;
    if (nevals <= 0 || !a.tp_ws) return;
    const int C = a.tp_chunks;
    const MtgTpBigPlan plan = mtg_tp_big_plan(J, a.B, C, a.tp_gsize);
    double *ws = a.tp_ws;
    mtg_launch_tpb_compose4q(a, ws + plan.elem_off[0], ws + plan.part_off, C, nevals, s);
    int *redo_list = (int *)(ws + plan.redo_off), *redo_count = redo_list + a.B;
    mtg_launch_tpb_up(J, a, plan, nevals, a.tp_direct, a.tp_direct ? redo_count : nullptr, s);
    MtgSolveArgs f = a;
    if (a.tp_direct) {
        mtg_launch_tpb_top_direct(J, a, plan, nevals, redo_list, redo_count, s);
        f.list = redo_list;
        f.count_ptr = redo_count;
    }
    mtg_launch_tpb_down(J, f, plan, nevals, s);
    mtg_launch_tpb_filter(f, ws + plan.state_off[0], ws + plan.part_off, C, nevals, s);
    mtg_launch_tpb_finish(f, ws + plan.part_off, ws + plan.head_off, C, nevals, s);
}
