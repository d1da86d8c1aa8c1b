// How long do the scattered 16-byte reads of a decode pass take?  1280 workgroups x 256 threads x 4 rows, each row reads
// locs[b][p] (30.7 MB table, ascending p inside a workgroup, about 120 rows apart) and anchors[p] (1.9 MB table) and
// writes one 16-byte row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ locs, const float4 *__restrict__ anchors, const int *__restrict__ idx,
                                                int P, int per_wg, int both, int xcd_affine, float4 *__restrict__ out) {
    int wg = blockIdx.x;
    if (xcd_affine) {                                    // workgroups of one image (80 consecutive) onto one XCD
        const int n = gridDim.x, x = wg % 8, slot = wg / 8;
        wg = x * (n / 8) + slot;
    }
    const int b = wg / 80;
    for (int r = threadIdx.x; r < per_wg; r += 256) {
        const int p = idx[(size_t)wg * per_wg + r];
        float4 l = locs[(size_t)b * P + p];
        if (both) { const float4 a = anchors[p]; l.x += a.x; l.y += a.y; l.z += a.z; l.w += a.w; }
        out[(size_t)wg * per_wg + r] = l;
    }
}

int main() {
    const int B = 16, P = 120087, C = 80, per = 1000, nwg = B * C;
    float4 *locs, *anchors, *out; int *idx;
    CK(hipMalloc(&locs, (size_t)B * P * 16)); CK(hipMalloc(&anchors, (size_t)P * 16)); CK(hipMalloc(&out, (size_t)nwg * per * 16));
    CK(hipMalloc(&idx, (size_t)nwg * per * 4));
    CK(hipMemset(locs, 0, (size_t)B * P * 16)); CK(hipMemset(anchors, 0, (size_t)P * 16));
    std::vector<int> h((size_t)nwg * per);
    for (int w = 0; w < nwg; ++w) {
        std::vector<int> v(per);
        for (auto &x : v) x = rand() % P;
        std::sort(v.begin(), v.end());
        std::copy(v.begin(), v.end(), h.begin() + (size_t)w * per);
    }
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int xa = 0; xa < 2; ++xa)
        for (int both = 0; both < 2; ++both) {
            for (int i = 0; i < 3; ++i) k_gather<<<nwg, 256>>>(locs, anchors, idx, P, per, both, xa, out);
            CK(hipDeviceSynchronize());
            float sum = 0;
            for (int i = 0; i < 10; ++i) {
                CK(hipEventRecord(a)); k_gather<<<nwg, 256>>>(locs, anchors, idx, P, per, both, xa, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms;
            }
            printf("xcd-affine %d, %s: %.1f us\n", xa, both ? "offsets + anchor rows" : "offsets only", sum / 10 * 1e3);
        }
    return 0;
}
