// Where does one radix-select round of a 256-thread workgroup spend its time?  (LDS histogram with 16 copies, barrier,
// atomics, barrier, sum of the copies, barrier, one wave scans the 256 bins with shuffles, barrier.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define CLK() ((long long)__builtin_readcyclecounter())

__device__ inline uint32_t suffix_scan64(uint32_t t, int lane) {
    uint32_t suf = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_down(suf, o); if (lane + o < 64) suf += up; }
    return suf;
}
__device__ inline uint32_t suffix_scan64_dpp(uint32_t v) {
    // suffix sum = total - exclusive prefix; inclusive prefix with DPP row shifts and row broadcasts
    uint32_t x = v;
#define ADD(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false)
    ADD(0x111, 0xf); ADD(0x112, 0xf); ADD(0x114, 0xf); ADD(0x118, 0xf); ADD(0x142, 0xa); ADD(0x143, 0xc);
#undef ADD
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return total - x + v;
}

__global__ __launch_bounds__(256) void k_round(const uint32_t *__restrict__ keys, int n, int rounds, int variant, unsigned long long *clk, uint32_t *out) {
    __shared__ uint32_t s_k[4096];
    __shared__ __align__(16) uint32_t s_h[256];
    __shared__ uint32_t s_h16[16 * 257];
    __shared__ uint32_t s_pref, s_need;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < n; i += 256) s_k[i] = keys[blockIdx.x * n + i];
    if (tid == 0) { s_pref = 0; s_need = n / 2; }
    __syncthreads();
    long long c_zero = 0, c_atom = 0, c_sum = 0, c_scan = 0;
    for (int r = 0; r < rounds; ++r) {
        const int shift = 24 - 8 * (r & 3);
        const long long t0 = CLK();
        for (int i = tid; i < 16 * 257; i += 256) s_h16[i] = 0u;
        __syncthreads();
        const long long t1 = CLK();
        const uint32_t pref = s_pref;
        uint32_t *mine = s_h16 + (lane & 15) * 257;
        for (int i = tid; i < n; i += 256) {
            const uint32_t d = s_k[i];
            if (shift == 24 || (d >> (shift + 8)) == (pref >> (shift + 8))) atomicAdd(&mine[(d >> shift) & 255u], 1u);
        }
        __syncthreads();
        const long long t2 = CLK();
        {
            uint32_t acc = 0u;
#pragma unroll
            for (int cpy = 0; cpy < 16; ++cpy) acc += s_h16[cpy * 257 + tid];
            s_h[tid] = acc;
        }
        __syncthreads();
        const long long t3 = CLK();
        if (tid < 64) {
            const uint4 q = reinterpret_cast<const uint4 *>(s_h)[lane];
            const uint32_t t = q.x + q.y + q.z + q.w;
            const uint32_t suf = variant ? suffix_scan64_dpp(t) : suffix_scan64(t, lane);
            const unsigned long long ge = __ballot(suf >= s_need);
            const int L = 63 - __clzll((long long)ge);
            uint32_t b_ = 0, a_ = 0;
            if (lane == L) {
                uint32_t acc = suf - t;
                const uint32_t hb[4] = {q.x, q.y, q.z, q.w};
                for (int i = 3; i >= 0; --i) { if (acc + hb[i] >= s_need) { b_ = 4 * L + i; a_ = acc; break; } acc += hb[i]; }
            }
            const uint32_t digit = variant ? (uint32_t)__builtin_amdgcn_readlane((int)b_, L) : __shfl(b_, L);
            const uint32_t above = variant ? (uint32_t)__builtin_amdgcn_readlane((int)a_, L) : __shfl(a_, L);
            if (lane == 0) { s_pref = (r & 3) ? (pref | (digit << shift)) : (digit << shift); s_need = (r & 3) == 3 ? n / 2 : s_need - above; }
        }
        __syncthreads();
        const long long t4 = CLK();
        c_zero += t1 - t0; c_atom += t2 - t1; c_sum += t3 - t2; c_scan += t4 - t3;
    }
    if (tid == 0) {
        atomicAdd(&clk[0], (unsigned long long)c_zero); atomicAdd(&clk[1], (unsigned long long)c_atom);
        atomicAdd(&clk[2], (unsigned long long)c_sum); atomicAdd(&clk[3], (unsigned long long)c_scan);
        out[blockIdx.x] = s_pref;
    }
}

int main() {
    const int n = 2048, blocks = 1280, rounds = 64;
    uint32_t *keys, *out; unsigned long long *clk;
    CK(hipMalloc(&keys, (size_t)blocks * n * 4)); CK(hipMalloc(&out, blocks * 4)); CK(hipMalloc(&clk, 64));
    uint32_t *h = (uint32_t *)malloc((size_t)blocks * n * 4);
    for (size_t i = 0; i < (size_t)blocks * n; ++i) h[i] = 0x3c000000u + (uint32_t)(rand() % 0x2000000);   // a few exponents, like scores
    CK(hipMemcpy(keys, h, (size_t)blocks * n * 4, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 2; ++variant)
        for (int nb : {160, 1280}) {
            CK(hipMemset(clk, 0, 64));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            k_round<<<nb, 256>>>(keys, n, rounds, variant, clk, out);
            CK(hipDeviceSynchronize()); CK(hipMemset(clk, 0, 64));
            CK(hipEventRecord(a)); k_round<<<nb, 256>>>(keys, n, rounds, variant, clk, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            unsigned long long c[4]; CK(hipMemcpy(c, clk, 32, hipMemcpyDeviceToHost));
            const double d = (double)nb * rounds;
            printf("variant %d, %4d workgroups: %.1f us for %d rounds = %.2f us per round; cycles per round: zero %.0f atomics %.0f sum %.0f scan %.0f\n",
                   variant, nb, ms * 1e3, rounds, ms * 1e3 / rounds, c[0] / d, c[1] / d, c[2] / d, c[3] / d);
        }
    return 0;
}
