// What does one wave64 VALU instruction cost a SIMD of this chip: 4 cycles (a SIMD-16 pass four times), or 2 (SIMD-32)?
// Every wave runs a long unrolled loop of independent 32-bit VALU instructions (v_add_u32, v_sad_u16, v_addc_co_u32 chains
// as the NMS pair filter issues them, or v_fma_f32) with W waves per SIMD on every CU, and reports
//   wave-instructions per cycle per SIMD  (from the shader clock each wave reads around its loop), and
//   wave-instructions per second for the chip (from the event time of the launch).
// Build:  hipcc --offload-arch=gfx950 -O3 -o exp_valu tools/bw/exp_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 64;     // instructions per loop trip (8 independent chains, 8 deep)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(int trips, unsigned seed, unsigned long long *cycles, unsigned *sink) {
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
    const unsigned x = seed ^ 0x9e3779b9u, y = threadIdx.x | 1u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                else if (KIND == 1) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                else asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= a[i];
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char *name, int insts_per_slot, int cus) {
    const int trips = 2000;
    unsigned long long *cyc; unsigned *sink;
    CK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int W : {1, 2, 4, 5, 8}) {
        const int blocks = cus * W;                      // 256-thread blocks: one wave on each SIMD of a CU
        k_valu<KIND><<<blocks, 256>>>(trips, 1u, cyc, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_valu<KIND><<<blocks, 256>>>(trips, 2u, cyc, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)blocks * 4);
        CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double winst = (double)trips * UNROLL * insts_per_slot;          // wave-instructions per wave
        // W waves share a SIMD: in `med` cycles the SIMD issued W * winst wave-instructions
        printf("%-28s W=%d waves/SIMD: %.3f wave-inst/cycle/SIMD (%.2f cycles each), chip %.2f T wave-inst/s, kernel %.1f us, "
               "clock %.0f MHz\n", name, W, W * winst / med, med / (W * winst), (double)blocks * 4 * winst / (ms * 1e-3) / 1e12,
               ms * 1e3, med / (ms * 1e3));
    }
    CK(hipFree(cyc)); CK(hipFree(sink));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    run<0>("v_add_u32", 1, cus);
    run<1>("v_sad_u16", 1, cus);
    run<2>("v_fma_f32", 1, cus);
    run<3>("v_cmp + v_addc (pair)", 2, cus);
    return 0;
}
