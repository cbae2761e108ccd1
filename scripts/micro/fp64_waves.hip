// FP64 FMA throughput of one SIMD against the number of waves sharing it (gfx950): a workgroup of W waves on one CU,
// every wave 4096 independent v_fma_f64 (8 accumulators); s_memtime ticks per instruction per wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fp64_waves.hip -o gpurun_bin/fp64_waves && ./gpurun_bin/fp64_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(1024) probe(double *out, long long *cycles, double seed)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-3;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 64; ++rep) {
        if (KIND == 0) {
        REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                          "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        } else {   // dependent pairs: two chains of four
        REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n"
                          "v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8"
                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[threadIdx.x >> 6] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h[16];
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
    for (int kind = 0; kind < 2; ++kind)
        for (int waves : {1, 4, 8, 12, 16}) {
            for (int r = 0; r < 2; ++r) {
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0000001);
                else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0000001);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, waves * 8, hipMemcpyDeviceToHost);
            long long mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
            printf("%s: %2d waves in the workgroup (%.1f per SIMD): %.2f ticks per v_fma_f64 per wave -> %.2f per SIMD\n",
                   kind ? "two dependent chains" : "independent", waves, waves / 4.0, mx / 4096.0, mx / 4096.0 / (waves < 4 ? 1 : waves / 4.0));
        }
    return 0;
}
