// mtg_timeparallel_big.hip -- ONE J = 10 structure of the time-parallel kernel per compilation:
//   hipcc -DMTG_TP_BIG_NR=<nr> -DMTG_TP_BIG_NC=<nc> -c mtg_timeparallel_big.hip
// in two shapes: 64 chunks per evaluation with the elements in LDS (118 KiB), and 256 chunks with
// the elements exchanged through a.tp_ws in global memory -- a quarter of the serial depth per
// pass against two more, slower scan rounds; the launcher picks by the length of the light curves.
#include "mtg_timeparallel.h"

#define MTG_CAT2(a, b, c, d) a##b##c##d
#define MTG_CAT(a, b, c, d) MTG_CAT2(a, b, c, d)

void MTG_CAT(mtg_launch_tp_big_, MTG_TP_BIG_NR, _, MTG_TP_BIG_NC)(const MtgSolveArgs &a, int64_t nevals, hipStream_t s)
{
    if (a.tp_ws && a.N >= MTG_TP_BIG_WIDE_MIN_N)
        mtg_launch_tp<MTG_TP_BIG_NR, MTG_TP_BIG_NC, MTG_TP_BIG_LANES>(a, nevals, s);
    else
        mtg_launch_tp<MTG_TP_BIG_NR, MTG_TP_BIG_NC, 64>(a, nevals, s);
}
