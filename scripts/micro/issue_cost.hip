// Issue cost of the FP64-path instructions the sweeps are made of, one wave on one SIMD (gfx950):
// 64 independent instructions of one kind between two s_memtime reads, repeated; prints cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(64) probe(double *out, long long *cycles, double seed)
{
    double a[8];
    int ia[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3; ia[i] = i + (int)threadIdx.x; }
    long long best = 1ll << 60;
    for (int rep = 0; rep < 20; ++rep) {
        const long long t0 = __builtin_readcyclecounter();
        if (KIND == 0) { REP8(REP8(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[0]) : "v"(a[1]));)) }   // dependent chain
        if (KIND == 1) {
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                              "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        }
        if (KIND == 2) {
            REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        }
        if (KIND == 3) {
            REP8(asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n"
                              "v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(ia[0] & 1));)
        }
        if (KIND == 4) {
            REP8(asm volatile("v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11\n"
                              "v_cvt_i32_f64 %4, %12\n v_cvt_i32_f64 %5, %13\n v_cvt_i32_f64 %6, %14\n v_cvt_i32_f64 %7, %15"
                              : "=v"(ia[0]), "=v"(ia[1]), "=v"(ia[2]), "=v"(ia[3]), "=v"(ia[4]), "=v"(ia[5]), "=v"(ia[6]), "=v"(ia[7])
                              : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));)
        }
        if (KIND == 5) {
            REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                              "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        }
        if (KIND == 6) {
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        }
        if (KIND == 7) {
            REP8(asm volatile("v_lshl_add_u32 %0, %0, 4, %0\n v_lshl_add_u32 %1, %1, 4, %1\n v_lshl_add_u32 %2, %2, 4, %2\n v_lshl_add_u32 %3, %3, 4, %3\n"
                              "v_lshl_add_u32 %4, %4, 4, %4\n v_lshl_add_u32 %5, %5, 4, %5\n v_lshl_add_u32 %6, %6, 4, %6\n v_lshl_add_u32 %7, %7, 4, %7"
                              : "+v"(ia[0]), "+v"(ia[1]), "+v"(ia[2]), "+v"(ia[3]), "+v"(ia[4]), "+v"(ia[5]), "+v"(ia[6]), "+v"(ia[7]));)
        }
        if (KIND == 8) {
            REP8(asm volatile("v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3\n"
                              "v_frexp_mant_f64 %4, %4\n v_frexp_mant_f64 %5, %5\n v_frexp_mant_f64 %6, %6\n v_frexp_mant_f64 %7, %7"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        }
        if (KIND == 9) {   // v_fma_f64 with an SGPR operand and a literal-free form, as the sweeps use
            REP8(asm volatile("v_fmac_f64 %0, %8, %8\n v_fmac_f64 %1, %8, %8\n v_fmac_f64 %2, %8, %8\n v_fmac_f64 %3, %8, %8\n"
                              "v_fmac_f64 %4, %8, %8\n v_fmac_f64 %5, %8, %8\n v_fmac_f64 %6, %8, %8\n v_fmac_f64 %7, %8, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        }
        if (KIND == 10) {  // ds_read_b64 from distinct addresses
            extern __shared__ double lds[];
            lds[threadIdx.x] = seed;
            REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                              "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                              : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]) : "v"((int)threadIdx.x * 8));)
        }
        const long long t1 = __builtin_readcyclecounter();
        if (t1 - t0 < best) best = t1 - t0;
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += a[i] + ia[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = best;
}

template <int KIND> void run(const char *name, double *out, long long *cyc)
{
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 8192, 0, out, cyc, 1.000001);
    long long h = 0;
    hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-34s %6.2f shader-clock-counter ticks per instruction (64 in a row)\n", name, (double)h / 64.0);
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    run<0>("v_fma_f64, dependent chain", out, cyc);
    run<1>("v_fma_f64, independent", out, cyc);
    run<9>("v_fmac_f64, independent", out, cyc);
    run<2>("v_mul_f64, independent", out, cyc);
    run<6>("v_add_f64, independent", out, cyc);
    run<3>("v_ldexp_f64, independent", out, cyc);
    run<4>("v_cvt_i32_f64, independent", out, cyc);
    run<5>("v_rcp_f64, independent", out, cyc);
    run<8>("v_frexp_mant_f64, independent", out, cyc);
    run<7>("v_lshl_add_u32, independent", out, cyc);
    run<10>("ds_read_b64 x8 + wait", out, cyc);
    return 0;
}
