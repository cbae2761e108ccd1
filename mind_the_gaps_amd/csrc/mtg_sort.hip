// mtg_sort.hip -- order of the serial sweep: evaluations sorted by (structure, light curve).
//
// mtg_solve_kernel gives every evaluation a lane and reads the samples of that lane's light curve: 64 lanes on
// one light curve read ONE address per step (a broadcast), 64 lanes on 64 light curves read 64 cache lines.  A
// (walker x light curve) sweep handed over grouped by light curve is the first case; the same evaluations in
// any other order -- a caller's own batch, the (P + 1)-point finite-difference blocks of many light curves
// interleaved -- used to run up to 11 x slower (round 2, gpurun_out/s2_bigmem.log).  So the sweep no longer
// takes the caller's order: a stable radix sort of the evaluation indices by key = structure * L + light curve
// (rejected rows last) gives every wave one or two light curves whatever the order was.  Stable, hence a
// deterministic lane assignment for a given batch.  The sort itself is rocPRIM's device radix sort over the
// key's significant bits -- plumbing, like hipFFT in mtg_chain_autocorr; ~0.2 % of the sweep it reorders.
#include "mtg_device.h"

#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

// rocPRIM's default falls back to a merge sort below 2^20 items (nine merge passes, ~20 launches of ~5 us for the
// 512 000 evaluations of the bench); the few significant bits of the key make the one-sweep radix sort two passes
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;

__global__ void __launch_bounds__(256) mtg_sort_keys_kernel(int64_t B, const int32_t *__restrict__ status,
                                                            const int32_t *__restrict__ sig,
                                                            const int32_t *__restrict__ lc, uint32_t L, uint32_t reject_key,
                                                            uint32_t *__restrict__ keys)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    uint32_t key = reject_key;
    if (status[e] == MTG_ST_OK) {
        const uint32_t l = lc ? (uint32_t)lc[e] : 0u;
        key = (sig ? (uint32_t)sig[e] : 0u) * L + (l < L ? l : L - 1u);   // (an index outside the set is reported by the sweep)
    }
    keys[e] = key;
}

}  // namespace

int mtg_sort_key_bits(int64_t L, int nsig)
{
    const uint64_t top = (uint64_t)L * (uint64_t)nsig;   // the key of rejected rows
    int bits = 1;
    while (bits < 32 && (top >> bits) != 0) ++bits;
    return bits;
}

size_t mtg_sort_temp_bytes(int64_t B, int bits)
{
    size_t bytes = 0;
    uint32_t *k = nullptr;
    int *v = nullptr;
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, k, k, rocprim::counting_iterator<int>(0), v, (size_t)B, 0u, (unsigned)bits,
                                    (hipStream_t) nullptr);
    return bytes;
}

hipError_t mtg_launch_sort_by_lightcurve(int64_t B, const int32_t *status, const int32_t *sig, const int32_t *lc, int64_t L,
                                         int nsig, uint32_t *keys_in, uint32_t *keys_out, int *order, void *temp,
                                         size_t temp_bytes, hipStream_t stream)
{
    const int bits = mtg_sort_key_bits(L, nsig);
    hipLaunchKernelGGL(mtg_sort_keys_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, B, status, sig, lc,
                       (uint32_t)L, (uint32_t)((uint64_t)L * (uint64_t)nsig), keys_in);
    return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<int>(0), order, (size_t)B,
                                     0u, (unsigned)bits, stream);
}
