// mtg_timeparallel.hip -- instantiations of the time-parallel kernel (mtg_timeparallel.h).
// J <= 6: the filtering element lives in registers; compiled here.
// J = 10 structures of a five-SHOTerm model (BASELINE configs[4]): their own path, mtg_tp_big.h
// (mtg_tp_big_compose.hip, mtg_tp_big_filter.hip, mtg_tp_scan.hip).
#include "mtg_timeparallel.h"
#include "mtg_tp_scan.h"

template <int NR, int NC, bool OK = (NR + NC > 0 && NR + 2 * NC <= 6)>
struct MtgTpSel { static constexpr mtg_solve_launcher fn = mtg_launch_tp<NR, NC>; };
template <int NR, int NC>
struct MtgTpSel<NR, NC, false> { static constexpr mtg_solve_launcher fn = nullptr; };
#define MTG_TP_ROW(nr) { MtgTpSel<(nr), 0>::fn, MtgTpSel<(nr), 1>::fn, MtgTpSel<(nr), 2>::fn, MtgTpSel<(nr), 3>::fn }
static const mtg_solve_launcher mtg_tp_table[7][4] = {MTG_TP_ROW(0), MTG_TP_ROW(1), MTG_TP_ROW(2), MTG_TP_ROW(3),
                                                      MTG_TP_ROW(4), MTG_TP_ROW(5), MTG_TP_ROW(6)};

// 256 chunks per evaluation (four waves) for the smallest batches: J <= 5 (LDS: 256 elements)
template <int NR, int NC, bool OK = (NR + NC > 0 && NR + 2 * NC <= 5)>
struct MtgTpWideSel { static constexpr mtg_solve_launcher fn = mtg_launch_tp<NR, NC, 256>; };
template <int NR, int NC>
struct MtgTpWideSel<NR, NC, false> { static constexpr mtg_solve_launcher fn = nullptr; };
#define MTG_TPW_ROW(nr) { MtgTpWideSel<(nr), 0>::fn, MtgTpWideSel<(nr), 1>::fn, MtgTpWideSel<(nr), 2>::fn }
static const mtg_solve_launcher mtg_tp_wide_table[6][3] = {MTG_TPW_ROW(0), MTG_TPW_ROW(1), MTG_TPW_ROW(2),
                                                           MTG_TPW_ROW(3), MTG_TPW_ROW(4), MTG_TPW_ROW(5)};

// the context's resident copy of the tables: mtg_fill_tables itself, once (same entries as a workgroup's own fill)
__global__ void __launch_bounds__(256) mtg_tables_kernel(MtgMathTables *tab) { mtg_fill_tables(tab, (int)threadIdx.x, 256); }

void mtg_launch_tables(void *tables, hipStream_t stream)
{
    hipLaunchKernelGGL(mtg_tables_kernel, dim3(1), dim3(256), 0, stream, static_cast<MtgMathTables *>(tables));
}

size_t mtg_tables_bytes() { return sizeof(MtgMathTables); }

mtg_solve_launcher mtg_find_tp_wide_solver(int nr, int nc)
{
    if (nr < 0 || nc < 0 || nr > 5 || nc > 2) return nullptr;
    return mtg_tp_wide_table[nr][nc];
}

mtg_solve_launcher mtg_find_tp_fused_nc1(int nr0, int nsig, int lanes);
mtg_solve_launcher mtg_find_tp_fused_nc2(int nr0, int nsig, int lanes);
mtg_solve_launcher mtg_find_tp_fused_nc3(int nr0, int nsig, int lanes);

mtg_solve_launcher mtg_find_tp_fused_solver(int nr0, int nc0, int nsig, int lanes)
{
    switch (nc0) {
    case 1: return mtg_find_tp_fused_nc1(nr0, nsig, lanes);
    case 2: return mtg_find_tp_fused_nc2(nr0, nsig, lanes);
    case 3: return mtg_find_tp_fused_nc3(nr0, nsig, lanes);
    }
    return nullptr;
}

mtg_solve_launcher mtg_find_tp_solver(int nr, int nc)
{
    // the rank-10 structures of a five-SHO family: ONE launch sequence serves every structure of a model
    // (mtg_tp_big.h); the caller passes the whole batch, not a structure's list
    if (nr + 2 * nc == 10 && nr % 2 == 0 && nc >= 0 && nc <= 5) return mtg_launch_tp_big;
    if (nr < 0 || nc < 0 || nr > 6 || nc > 3) return nullptr;
    return mtg_tp_table[nr][nc];
}
