// mtg_timeparallel_big.hip -- ONE rank-10 structure of the big-J time-parallel path per compilation:
//   hipcc -DMTG_TP_BIG_NR=<nr> -DMTG_TP_BIG_NC=<nc> -c mtg_timeparallel_big.hip
// (compose and filter kernels of mtg_tp_big.h; the scan kernels, which depend on the rank only, are
// compiled once in mtg_tp_scan.hip).
// 2048 + 1024-entry tables (16 + 16 KiB): next to the 37 KiB of rings two composition workgroups still fit a CU
#define MTG_EXP_BITS 11
#define MTG_TRIG_BITS 10
#include "mtg_tp_big.h"

#define MTG_CAT2(a, b, c, d) a##b##c##d
#define MTG_CAT(a, b, c, d) MTG_CAT2(a, b, c, d)

void MTG_CAT(mtg_launch_tp_big_, MTG_TP_BIG_NR, _, MTG_TP_BIG_NC)(const MtgSolveArgs &a, int64_t nevals, hipStream_t s)
{
    mtg_launch_tp_big<MTG_TP_BIG_NR, MTG_TP_BIG_NC>(a, nevals, s);
}
