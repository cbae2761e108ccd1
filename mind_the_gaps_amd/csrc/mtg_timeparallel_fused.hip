// mtg_timeparallel_fused.hip -- the time-parallel kernel for models with SHOTerms, every
// signature in one launch (mtg_tp_fused_kernel, mtg_timeparallel.h).  Built once per number of
// complex terms of the all-under-damped structure (-DMTG_TPF_NC0=1|2|3) to keep the
// translation units parallel; J <= 6 with 64 chunks per evaluation, J <= 5 with 256.
#include "mtg_timeparallel.h"

#ifndef MTG_TPF_NC0
#error "compile with -DMTG_TPF_NC0=<complex terms of the base structure>"
#endif

namespace {

template <int NR0, int NSIG, int LANES,
          bool OK = (NSIG - 1 <= MTG_TPF_NC0 && NR0 + 2 * MTG_TPF_NC0 <= (LANES == 64 ? 6 : 5))>
struct Sel { static constexpr mtg_solve_launcher fn = mtg_launch_tp_fused<NR0, MTG_TPF_NC0, NSIG, LANES>; };
template <int NR0, int NSIG, int LANES>
struct Sel<NR0, NSIG, LANES, false> { static constexpr mtg_solve_launcher fn = nullptr; };

#define ROW(nr0, lanes) { Sel<(nr0), 2, (lanes)>::fn, Sel<(nr0), 3, (lanes)>::fn, Sel<(nr0), 4, (lanes)>::fn }
const mtg_solve_launcher table64[5][3] = {ROW(0, 64), ROW(1, 64), ROW(2, 64), ROW(3, 64), ROW(4, 64)};
const mtg_solve_launcher table256[5][3] = {ROW(0, 256), ROW(1, 256), ROW(2, 256), ROW(3, 256), ROW(4, 256)};
// two waves per evaluation: the elements of a rank-4 or rank-5 evaluation then take half a CU's LDS, two workgroups
// share a CU, and 257 ... 512 evaluations are resident at once
const mtg_solve_launcher table128[5][3] = {ROW(0, 128), ROW(1, 128), ROW(2, 128), ROW(3, 128), ROW(4, 128)};

}  // namespace

#define MTG_TPF_CAT2(a, b) a##b
#define MTG_TPF_CAT(a, b) MTG_TPF_CAT2(a, b)

mtg_solve_launcher MTG_TPF_CAT(mtg_find_tp_fused_nc, MTG_TPF_NC0)(int nr0, int nsig, int lanes)
{
    if (nr0 < 0 || nr0 > 4 || nsig < 2 || nsig > 4) return nullptr;
    return (lanes == 256 ? table256 : lanes == 128 ? table128 : table64)[nr0][nsig - 2];
}
