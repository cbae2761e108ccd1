// persist_probe.hip -- what would ONE persistent launch per convergence block save a small chain?
//
// The device sampler's iteration is  solve (R workgroups, one per proposal row)  ->  sampler step (ONE workgroup per
// ensemble)  -> next solve ..., two kernels per iteration on one stream (csrc/mtg_sampler.hip, mtg_ensemble_run).  A
// persistent kernel would keep the R + 1 workgroups resident and replace the two kernel boundaries of an iteration by two
// hand-overs through memory: "fan-in" (every solver row publishes its result; the sampler workgroup waits for all R) and
// "fan-out" (the sampler publishes the next proposals; every solver waits for them).  This probe runs the SKELETON of both
// forms -- no arithmetic, only the hand-overs, with the data travelling the way it would (results and proposals carry
// the wait; a row's result IS the flag the sampler polls: one trip to memory, not flag-then-data) -- and prints microseconds
// per iteration.  The difference is the most a persistent launch can take off an iteration.
//
//     hipcc -O3 --offload-arch=gfx950 scripts/persist_probe.hip -o /tmp/persist_probe && /tmp/persist_probe
//
// Every spin is bounded (SPIN_LIMIT polls, then the workgroup gives up and says so): no wave can wait for ever.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
#define SPIN_LIMIT (1u << 22)

struct Slot { double value; uint64_t tag; };   // (the two-kernel form's rows; the persistent form uses the words below)

// Hand-over through memory between workgroups on different XCDs (each XCD has its own L2): agent-scope atomics go to the
// memory side; plain data is made visible by a release fence (L2 write-back) before the flag and read after an acquire.
#define SENTINEL 0x7ff8dead0000beefull   // a NaN payload no arithmetic produces: "row not evaluated yet"
#define PD 24                            // doubles a solver row reads per iteration (theta + coefficient columns)

// ---- persistent form: workgroups 0 .. R-1 are the solver rows, workgroup R is the sampler ------------------------------
__global__ void __launch_bounds__(256) persistent_kernel(int R, int iterations, uint64_t *results, double *proposals, uint32_t *flag,
                                                         int *gave_up, double *sink)
{
    const int wg = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_ok;
    double acc = 0.0;
    if (wg < R) {
        for (int it = 1; it <= iterations; ++it) {
            // fan-out: wait for the proposals of iteration `it` (thread 0 polls, the workgroup follows), then read this row's
            if (tid == 0) {
                unsigned spins = 0;
                uint32_t seen = 0;
                while ((seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != (uint32_t)it && seen != 0xffffffffu &&
                       ++spins < SPIN_LIMIT)
                    __builtin_amdgcn_s_sleep(1);
                s_ok = seen == (uint32_t)it;
            }
            __syncthreads();
            if (!s_ok) { if (tid == 0) atomicAdd(gave_up, 1); return; }
            if (tid < PD) acc += proposals[(int64_t)wg * PD + tid];
            // (the solve would run here) ... and its result IS the flag the sampler waits for: one trip
            for (int off = 16; off > 0; off >>= 1) acc += __shfl_down(acc, off);
            if (tid == 0) {
                uint64_t bits = (uint64_t)__double_as_longlong(acc);
                if (bits == SENTINEL) bits ^= 1;
                __hip_atomic_store(results + wg, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    } else {
        for (int it = 1; it <= iterations; ++it) {
            // the proposals of iteration `it` (the sampler's expansion would run before this): data, release, flag
            for (int j = tid; j < R * PD; j += blockDim.x) proposals[j] = (double)it + acc * 1e-300;
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(flag, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // fan-in: thread r waits for row r's result, takes it and puts the sentinel back
            int ok = 1;
            for (int r = tid; r < R; r += blockDim.x) {
                uint64_t bits = SENTINEL; unsigned spins = 0;
                while ((bits = __hip_atomic_load(results + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == SENTINEL && ++spins < SPIN_LIMIT)
                    __builtin_amdgcn_s_sleep(1);
                ok = ok && spins < SPIN_LIMIT;
                acc += __longlong_as_double((long long)bits);
                __hip_atomic_store(results + r, (uint64_t)SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) s_ok = 1;
            __syncthreads();
            if (!ok) s_ok = 0;
            __syncthreads();
            if (!s_ok) {    // let the solvers go (they would spin to their own limit otherwise) and leave
                if (tid == 0) { atomicAdd(gave_up, 1); __hip_atomic_store(flag, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                return;
            }
        }
    }
    if (tid == 0) sink[wg] = acc;
}

// ---- two kernels per iteration, as shipped: the argument block is ~1.5 KB by value (MtgEnsembleArgs + MtgPrepArgs) -----
struct FatArgs { int R; int it; Slot *results; Slot *proposals; double *sink; int pad[368]; };

__global__ void __launch_bounds__(256) solve_kernel(FatArgs a)
{
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) {
        const Slot p = a.proposals[wg];
        Slot r; r.value = p.value + wg + a.pad[wg & 255]; r.tag = (uint64_t)a.it;
        a.results[wg] = r;
    }
}
__global__ void __launch_bounds__(256) sampler_kernel(FatArgs a)
{
    double acc = 0.0;
    for (int r = threadIdx.x; r < a.R; r += blockDim.x) acc += a.results[r].value + a.pad[r & 255];
    for (int r = threadIdx.x; r < a.R; r += blockDim.x) { Slot p; p.value = acc; p.tag = (uint64_t)a.it + 1; a.proposals[r] = p; }
}

int main(int argc, char **argv)
{
    const int iterations = argc > 1 ? atoi(argv[1]) : 2000;
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("# %s, %d CUs; %d iterations per measurement; us per iteration (= one fan-out + one fan-in)\n", prop.gcnArchName, prop.multiProcessorCount, iterations);
    printf("# rows R | persistent launch | two kernels per iteration (as shipped) | difference\n");
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int R : {45, 96, 192, 384}) {
        Slot *results, *proposals; int *gave_up; double *sink;
        uint64_t *p_results; double *p_proposals; uint32_t *flag;
        CHECK(hipMalloc(&results, sizeof(Slot) * R)); CHECK(hipMalloc(&proposals, sizeof(Slot) * R));
        CHECK(hipMalloc(&p_results, 8 * R)); CHECK(hipMalloc(&p_proposals, 8 * R * PD)); CHECK(hipMalloc(&flag, 4));
        CHECK(hipMalloc(&gave_up, 4)); CHECK(hipMalloc(&sink, 8 * (R + 1)));
        std::vector<uint64_t> sentinels(R, SENTINEL);
        double best_p = 1e30, best_k = 1e30;
        int bad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemcpy(p_results, sentinels.data(), 8 * R, hipMemcpyHostToDevice));
            CHECK(hipMemsetAsync(flag, 0, 4, s));
            CHECK(hipMemsetAsync(gave_up, 0, 4, s));
            CHECK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            // R + 1 workgroups of 256 threads on 256 CUs: all resident at once (R <= 384: at most two per CU)
            hipLaunchKernelGGL(persistent_kernel, dim3(R + 1), dim3(256), 0, s, R, iterations, p_results, p_proposals, flag, gave_up, sink);
            CHECK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iterations;
            int g = 0;
            CHECK(hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost));
            bad += g;
            if (us < best_p) best_p = us;

            CHECK(hipMemsetAsync(results, 0, sizeof(Slot) * R, s)); CHECK(hipMemsetAsync(proposals, 0, sizeof(Slot) * R, s));
            CHECK(hipStreamSynchronize(s));
            FatArgs a{}; a.R = R; a.results = results; a.proposals = proposals; a.sink = sink;
            t0 = std::chrono::steady_clock::now();
            for (int it = 1; it <= iterations; ++it) {
                a.it = it;
                hipLaunchKernelGGL(solve_kernel, dim3(R), dim3(256), 0, s, a);
                hipLaunchKernelGGL(sampler_kernel, dim3(1), dim3(256), 0, s, a);
            }
            CHECK(hipStreamSynchronize(s));
            us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iterations;
            if (us < best_k) best_k = us;
        }
        printf("%8d | %17.2f | %38.2f | %10.2f%s\n", R, best_p, best_k, best_k - best_p, bad ? "   (a persistent workgroup GAVE UP: numbers invalid)" : "");
        CHECK(hipFree(results)); CHECK(hipFree(proposals)); CHECK(hipFree(gave_up)); CHECK(hipFree(sink));
        CHECK(hipFree(p_results)); CHECK(hipFree(p_proposals)); CHECK(hipFree(flag));
    }
    return 0;
}
