// Per-row top-k membership mask: out[b,p] = 1 iff the descending rank of values[b,p] inside row b is
// below k[b], equal values ranking by ascending index.  Replaces the double argsort of SSD's
// hard-negative mining, `loss_neg.argsort(descending=True).argsort() < 3 * num_pos`
// (src/models/ssd300.py:277-278; ssd512 inherits it).
//
// One workgroup per row: an exact radix select (four 8-bit rounds over the monotone uint32 image of the
// floats, histograms in LDS) finds the k-th largest key T and how many elements equal to T still belong
// to the top k; a last pass in index order writes the mask, admitting the first of the ties.
#include "detbox_internal.h"

using namespace detbox;

namespace {

constexpr int RANK_BLOCK = 1024;

// -0 and +0 compare equal in a sort; NaN sorts as the largest value (torch's ordering)
__device__ inline uint32_t rank_key(float v) {
    if (v != v) return 0xffffffffu;
    return okey_from_bits(__float_as_uint(v + 0.0f));
}

__global__ __launch_bounds__(RANK_BLOCK) void topk_mask_kernel(const float *__restrict__ values,
                                                               const int64_t *__restrict__ k, int P,
                                                               uint8_t *__restrict__ out) {
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_pref, s_rem;
    __shared__ uint32_t s_wcount[RANK_BLOCK / 64];
    __shared__ uint32_t s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *row = values + (size_t)b * P;
    uint8_t *orow = out + (size_t)b * P;
    long long kk = k[b];
    if (kk <= 0 || kk >= P) {                                   // nothing or everything
        const uint8_t v = kk >= P ? 1 : 0;
        for (int i = tid; i < P; i += RANK_BLOCK) orow[i] = v;
        return;
    }
    if (tid == 0) { s_pref = 0u; s_rem = (uint32_t)kk; }
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) s_h[tid] = 0u;
        __syncthreads();
        const uint32_t pref = s_pref;
        for (int i = tid; i < P; i += RANK_BLOCK) {
            const uint32_t key = rank_key(row[i]);
            if (shift == 24 || (key >> (shift + 8)) == (pref >> (shift + 8))) atomicAdd(&s_h[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {                                         // highest digit d with count(digits >= d) >= rem
            const uint4 q = reinterpret_cast<const uint4 *>(s_h)[lane];       // bins 4*lane .. 4*lane+3
            const uint32_t t = q.x + q.y + q.z + q.w;
            const uint32_t inc = wave_inclusive_add(t);                        // suffix sum over lanes >= lane
            const uint32_t suf = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63) - inc + t;
            const uint32_t rem = s_rem;
            const unsigned long long ge = __ballot(suf >= rem);
            const int L = 63 - __clzll((long long)ge);                         // the lane whose bins hold the answer
            if (lane == L) {
                uint32_t acc = suf - t;                                        // count in the bins above this lane's
                const uint32_t hb[4] = {q.x, q.y, q.z, q.w};
                int d = 0;
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    if (acc + hb[i] >= rem) { d = i; break; }
                    acc += hb[i];
                }
                s_pref = pref | ((uint32_t)(4 * L + d) << shift);
                s_rem = rem - acc;                              // still wanted among the keys sharing the new prefix
            }
        }
        __syncthreads();
    }
    const uint32_t T = s_pref;
    const uint32_t want_eq = s_rem;                             // >= 1: ties at T that belong to the top k
    if (tid == 0) s_base = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < P; i0 += RANK_BLOCK) {                // index order: tile by tile, lane by lane
        const int i = i0 + tid;
        const uint32_t key = i < P ? rank_key(row[i]) : 0u;
        const bool eq = i < P && key == T;
        const unsigned long long em = __ballot(eq);
        if (lane == 0) s_wcount[wv] = (uint32_t)__popcll(em);
        __syncthreads();
        uint32_t before = s_base;
        for (int w = 0; w < wv; ++w) before += s_wcount[w];
        before += __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
        if (i < P) orow[i] = (key > T || (eq && before < want_eq)) ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = s_base;
            for (int w = 0; w < RANK_BLOCK / 64; ++w) t += s_wcount[w];
            s_base = t;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int detbox_topk_mask(const float *values, const int64_t *k, int64_t B, int64_t P, uint8_t *out_mask,
                                void *stream) {
    DETBOX_CHECK_ARG(B >= 0 && P >= 0, "negative size");
    DETBOX_CHECK_ARG(P < (1ll << 31) && B < (1ll << 31), "size too large");
    if (B == 0 || P == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(values && k && out_mask, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope ps("topk_mask", s);
    topk_mask_kernel<<<dim3((unsigned)B), dim3(RANK_BLOCK), 0, s>>>(values, k, (int)P, out_mask);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
