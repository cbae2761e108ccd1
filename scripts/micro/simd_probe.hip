// Which SIMD does wave w of a workgroup land on?  (gfx950; HW_REG_HW_ID: wave [3:0], simd [5:4], pipe [7:6], cu [11:8],
// sh [12], se [15:13])  hipcc --offload-arch=gfx950 -O2 simd_probe.hip -o simd_probe && ./simd_probe [threads] [lds_bytes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ char lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if (spin) { lds[threadIdx.x] = 1; __syncthreads(); for (volatile int i = 0; i < spin; ++i) {} }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 512, lds = argc > 2 ? atoi(argv[2]) : 140000, blocks = 16;
    unsigned *d, h[16 * 16];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 2000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int wpb = threads / 64;
    for (int b = 0; b < blocks; ++b) {
        printf("wg %2d:", b);
        for (int w = 0; w < wpb; ++w) { unsigned id = h[b * wpb + w]; printf("  w%d simd %u cu %u se %u", w, (id >> 4) & 3, (id >> 8) & 15, (id >> 13) & 7); }
        printf("\n");
    }
    return 0;
}
