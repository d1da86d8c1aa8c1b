// Per-class (segmented) greedy NMS for a whole batch, plus the result-row packing.
// Replaces torchvision.ops.batched_nms as called at src/utilities/predictor.py:55
// and the per-detection Python loop of src/utilities/predictor.py:56-73.
//
// nms_segments   one workgroup (or, for thousands of short segments, one wave) per (image, class
//                segment): stable descending sort of the segment on 64-bit keys held in registers,
//                boxes gathered into LDS in that order, then a blocked greedy sweep: for every
//                64-box chunk the mutual suppression words, a serial resolve by one wave, and the
//                boxes it kept knock out later boxes (integer reject on quantised boxes first, the
//                reference's fp32 test for what is left).  Only kept boxes ever act as
//                suppressors, exactly like the sequential CPU algorithm.  A generic kernel covers
//                presorted, very long and negative-threshold segments.
// nms_scan       per image: offsets of the kept runs.
// nms_merge      stable merge of the per-class kept runs into one list per image in descending
//                score order -- the order torchvision returns: the first levels inside LDS, the
//                rest as merge-path tiles.
// pack           result rows (float64, src/utilities/predictor.py:60-73) and the compact message
//                ranks exchange.
//
// IoU arithmetic follows torchvision's CPU kernel: areas (x2-x1)*(y2-y1),
// inter / (area_i + area_j - inter) in fp32, suppressed iff (double)iou > threshold;
// the host passes the largest fp32 not above the double threshold, which makes the
// fp32 comparison equivalent.
#include "detbox_internal.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

using namespace detbox;

namespace {

constexpr int NMS_BLOCK = 256;
constexpr int NMS_LDS_BOXES = 4096;       // segments up to this length keep their boxes in LDS
constexpr int NMS_MAX_SEGMENT = 262144;   // bitmap limit for presorted segments

struct NmsParams {
    int S, max_seg, n2, in_lds, presorted, thr_neg, stats, network_sort;
    long long stride;
    float thr;
};

struct NmsLds { size_t keys, removed, maskq, keepw, wprefix, box, total; };

// Byte offsets of the dynamic-LDS regions, every one 16-byte aligned.
__host__ __device__ inline NmsLds nms_lds(int max_seg, int n2, int presorted, int in_lds) {
    const size_t words_cap = (size_t)((max_seg + 63) >> 6);
    NmsLds l;
    size_t off = 0;
    l.keys = off;    off += presorted ? 0 : (size_t)n2 * 8;            // sort keys, later the order
    l.removed = off; off += ((words_cap * 8 + 15) & ~(size_t)15);       // suppression bitmap
    l.maskq = off;   off += 256 * 8;                                    // 4 x 64 partial chunk words
    l.keepw = off;   off += 16;                                         // kept word of the chunk
    l.wprefix = off; off += ((words_cap * 4 + 15) & ~(size_t)15);       // kept-count prefix per word
    l.box = off;     off += in_lds ? (size_t)max_seg * 16 : 0;          // boxes in sorted order
    l.total = off;
    return l;
}

__device__ inline bool iou_gt(const float4 a, const float area_a, const float4 b, const float thr, const int thr_neg) {
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
    const float inter = w * h;
    if (!(inter > 0.0f) && !thr_neg) return false;      // iou is 0 (or NaN): never above a threshold >= 0
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float ovr = inter / ((area_a + area_b) - inter);
    return ovr > thr;
}

__device__ inline void bitonic_desc_u64(unsigned long long *s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int x = i ^ j;
                if (x > i) {
                    const unsigned long long a = s[i], c = s[x];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < c) : (a > c)) { s[i] = c; s[x] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Bitonic sort (descending) of 256*R keys held R per thread: element e = m*256 + tid lives in K[m] of
// thread tid.  Strides >= 256 are exchanges between registers of one thread, strides < 64 are wave
// shuffles; only strides 64 and 128 go through LDS (`xch`, 256*R keys) and need barriers.  Keys at
// e >= n2 must be zero: only the passes up to n2 run and they never move a key across an n2 boundary.
// One compare-exchange pass at distance J of the stage that builds sorted runs of KK keys, then the passes below it.
template <int R, int KK, int J, int BLK>
__device__ inline void bitonic_pass_u64(unsigned long long (&K)[R], unsigned long long *xch, const int tid) {
    if constexpr (J >= BLK) {
        constexpr int dm = J / BLK;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int pm = m ^ dm;
            if (pm > m) {
                const unsigned long long a = K[m], c = K[pm];
                const bool desc = ((m * BLK) & KK) == 0;           // KK >= 2*BLK here: the bit lies in m
                const bool swap = (a > c) != desc && a != c;
                K[m] = swap ? c : a;
                K[pm] = swap ? a : c;
            }
        }
    } else {
        if constexpr (J >= 64) {
#pragma unroll
            for (int m = 0; m < R; ++m) xch[m * BLK + tid] = K[m];
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int e = m * BLK + tid;
            const unsigned long long a = K[m];
            unsigned long long c;
            if constexpr (J >= 64) c = xch[e ^ J];
            else c = lane_xor_u64<J>(a);
            const bool lower = (tid & J) == 0;
            const bool desc = (e & KK) == 0;
            // the lower element of a descending pair keeps the larger key: one compare, one select per word
            K[m] = ((a > c) == (lower == desc)) ? a : c;
        }
        if constexpr (J >= 64) __syncthreads();
    }
    if constexpr (J > 1) bitonic_pass_u64<R, KK, J / 2, BLK>(K, xch, tid);
}

template <int R, int KK, int BLK>
__device__ inline void bitonic_stage_u64(unsigned long long (&K)[R], unsigned long long *xch, const int tid) {
    bitonic_pass_u64<R, KK, KK / 2, BLK>(K, xch, tid);
}

template <int R, int KK, int BLK>
__device__ inline void bitonic_stages_u64(unsigned long long (&K)[R], unsigned long long *xch, const int n2, const int tid) {
    if constexpr (KK <= BLK * R) {
        if (KK <= n2) {                                      // uniform over the workgroup
            bitonic_stage_u64<R, KK, BLK>(K, xch, tid);
            bitonic_stages_u64<R, KK * 2, BLK>(K, xch, n2, tid);
        }
    }
}

// Bitonic sort (descending) of BLK*R keys held R per thread: element e = m*BLK + tid lives in K[m] of
// thread tid.  Strides >= BLK are exchanges between registers of one thread, strides < 64 are wave
// shuffles; only the strides in between (64 and 128 for BLK = 256, none for a single wave) go through
// LDS (`xch`, BLK*R keys) and need barriers.  Keys at e >= n2 must be zero: only the stages up to n2 run
// and they never move a key across an n2 boundary.
template <int R, int BLK>
__device__ inline void bitonic_desc_u64_regs(unsigned long long (&K)[R], unsigned long long *xch, const int n2, const int tid) {
    bitonic_stages_u64<R, 2, BLK>(K, xch, n2, tid);
}

// Minimum of lx, ly and maximum of hx, hy over the wave, left in every lane.
template <int O>
__device__ inline void wave_extent(float &lx, float &ly, float &hx, float &hy) {
    lx = fminf(lx, lane_xor_f32<O>(lx)); ly = fminf(ly, lane_xor_f32<O>(ly));
    hx = fmaxf(hx, lane_xor_f32<O>(hx)); hy = fmaxf(hy, lane_xor_f32<O>(hy));
    if constexpr (O > 1) wave_extent<O / 2>(lx, ly, hx, hy);
}

__device__ inline unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ inline float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ inline unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(NMS_BLOCK) void nms_segments_generic_kernel(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
    NmsParams np, float2 *__restrict__ kept, int32_t *__restrict__ kcount) {
    extern __shared__ unsigned long long s_dyn[];
    const int s = blockIdx.x, b = blockIdx.y;
    const int32_t *so = seg_off + (size_t)b * (np.S + 1);
    const int seg0 = so[s];
    const int n = so[s + 1] - seg0;
    if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = 0;
    if (n <= 0) return;
    if (n > np.max_seg || (!np.presorted && n > DETBOX_NMS_MAX_SORT)) {
        if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = -1;      // reported as keep_counts[b] = -1
        return;
    }
    // LDS carve-up (sizes fixed by the host from max_seg)
    const NmsLds ld = nms_lds(np.max_seg, np.n2, np.presorted, np.in_lds);
    char *lds_base = reinterpret_cast<char *>(s_dyn);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(lds_base + ld.keys);
    unsigned long long *s_removed = reinterpret_cast<unsigned long long *>(lds_base + ld.removed);
    unsigned long long *s_maskq = reinterpret_cast<unsigned long long *>(lds_base + ld.maskq);
    unsigned long long *s_keepw = reinterpret_cast<unsigned long long *>(lds_base + ld.keepw);
    uint32_t *s_wprefix = reinterpret_cast<uint32_t *>(lds_base + ld.wprefix);
    float4 *s_box = reinterpret_cast<float4 *>(lds_base + ld.box);

    const size_t row0 = (size_t)b * np.stride + seg0;
    const int tid = threadIdx.x;
    const int words = (n + 63) >> 6;

    if (!np.presorted) {
        int n2 = 2;
        while (n2 < n) n2 <<= 1;
        for (int i = tid; i < n2; i += NMS_BLOCK) {
            unsigned long long k = 0ull;
            if (i < n) {
                // every real key is above the zero padding: its low word is 0xffffffff - i > 0
                k = ((unsigned long long)okey_from_bits(__float_as_uint(scores[row0 + i] + 0.0f)) << 32) |
                    (unsigned long long)(0xffffffffu - (uint32_t)i);
            }
            s_keys[i] = k;
        }
        __syncthreads();
        bitonic_desc_u64(s_keys, n2);
    }
    auto order_of = [&](int i) -> int {
        return np.presorted ? i : (int)(0xffffffffu - (uint32_t)(s_keys[i] & 0xffffffffull));
    };
    if (np.in_lds)
        for (int i = tid; i < n; i += NMS_BLOCK) s_box[i] = boxes[row0 + order_of(i)];
    for (int i = tid; i < words; i += NMS_BLOCK) s_removed[i] = 0ull;
    __syncthreads();
    auto box_of = [&](int i) -> float4 { return np.in_lds ? s_box[i] : boxes[row0 + order_of(i)]; };

    const int lane = tid & 63, q = tid >> 6;
    for (int ch = 0; ch < words; ++ch) {
        const int c0 = ch << 6;
        const int cn = min(64, n - c0);
        const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
        const unsigned long long alive0 = ~s_removed[ch] & valid;
        // 64x64 suppression words of this chunk: thread (lane, q) covers columns q*16..q*16+15
        {
            unsigned long long part = 0ull;
            if (lane < cn && ((alive0 >> lane) & 1ull)) {
                const float4 bi = box_of(c0 + lane);
                const float ai = (bi.z - bi.x) * (bi.w - bi.y);
                const int j0 = max(q * 16, lane + 1), j1 = min(q * 16 + 16, cn);
                for (int j = j0; j < j1; ++j)
                    if (iou_gt(bi, ai, box_of(c0 + j), np.thr, np.thr_neg)) part |= 1ull << j;
            }
            s_maskq[q * 64 + lane] = part;
        }
        __syncthreads();
        if (q == 0) {
            const unsigned long long m = s_maskq[lane] | s_maskq[64 + lane] | s_maskq[128 + lane] | s_maskq[192 + lane];
            unsigned long long alive = uniform64(alive0);
            unsigned long long keep = 0ull;
            while (alive) {
                const int i = __ffsll((long long)alive) - 1;
                keep |= 1ull << i;
                alive &= ~readlane64(m, i);
                alive &= ~(1ull << i);
            }
            if (lane == 0) { s_keepw[0] = keep; s_removed[ch] = valid & ~keep; }
        }
        __syncthreads();
        const unsigned long long keep = s_keepw[0];
        if (keep != 0ull) {
            for (int j = c0 + 64 + tid; j < n; j += NMS_BLOCK) {
                if ((s_removed[j >> 6] >> (j & 63)) & 1ull) continue;
                const float4 bj = box_of(j);
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                unsigned long long kk = keep;
                while (kk) {
                    const int i = __ffsll((long long)kk) - 1;
                    kk &= kk - 1ull;
                    const float4 bi = box_of(c0 + i);
                    // same operand roles as the sequential algorithm: i is the kept box, j the candidate
                    const float ai = (bi.z - bi.x) * (bi.w - bi.y);
                    const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.0f);
                    const float h = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y), 0.0f);
                    const float inter = w * h;
                    if (!(inter > 0.0f) && !np.thr_neg) continue;
                    const float ovr = inter / ((ai + aj) - inter);
                    if (ovr > np.thr) { atomicOr(&s_removed[j >> 6], 1ull << (j & 63)); break; }
                }
            }
        }
        __syncthreads();
    }
    // kept = not removed; compact in sorted order
    if (q == 0) {
        uint32_t running = 0;
        for (int base = 0; base < words; base += 64) {
            const int w = base + lane;
            uint32_t pc = 0;
            if (w < words) {
                const int cn = min(64, n - (w << 6));
                const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
                pc = (uint32_t)__popcll(~s_removed[w] & valid);
            }
            const uint32_t inc = wave_inclusive_add(pc);
            if (w < words) s_wprefix[w] = running + inc - pc;
            running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) kcount[(size_t)b * np.S + s] = (int32_t)running;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NMS_BLOCK) {
        const unsigned long long rem = s_removed[i >> 6];
        if ((rem >> (i & 63)) & 1ull) continue;
        const uint32_t rank = s_wprefix[i >> 6] + (uint32_t)__popcll(~rem & ((1ull << (i & 63)) - 1ull));
        const int pos = seg0 + order_of(i);
        kept[row0 + rank] = make_float2(__uint_as_float(okey_from_bits(__float_as_uint(scores[(size_t)b * np.stride + pos] + 0.0f))), __int_as_float(pos));
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path: unsorted segments of at most 256*JPT boxes, threshold >= 0.
//
// Every thread keeps its JPT boxes (sorted positions tid, tid+256, ...) in registers for the whole
// sweep.  Per 64-box chunk (in sorted order): the mutual suppression words of the chunk, a wave-serial
// resolve over the rows that suppress anything, then the boxes the chunk keeps are broadcast with
// readlane against every later box still alive (wave-uniform 64-bit alive masks per register slot).
//
// The pair test is two-staged.  Stage 1 is a conservative integer reject on quantised boxes
// over the segment's extent (outer boxes on a 15-bit grid: low edges rounded down, high edges up), two coordinates per
// packed instruction:  inter_up = w_q * h_q  is an upper bound of the intersection in grid units, and
//     iou > thr  =>  inter * (1 + thr) > thr * (area_i + area_j)  =>  inter_up > T_i + T_j,
// T = floor(thr' / (1 + thr') * area * (1 - eps)) per box with thr' a hair below thr; so
// inter_up <= T_i + T_j proves "not suppressed".  Stage 2 runs only while some lane is undecided and is
// the reference's own fp32 arithmetic (division-free where the margins decide, the division itself in
// the sliver that is left).  The results are the reference's, bit for bit.
// ---------------------------------------------------------------------------------------------

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ inline u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// `i` is the kept (earlier) box.  k_hi / k_lo: thr widened by 2^-20 either way; u_min guards tiny unions.
__device__ inline bool suppresses_nodiv(const float4 bi, const float ai, const float4 bj, const float thr,
                                        const float k_hi, const float k_lo, const float u_min) {
    const float w = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x), 0.0f);
    const float h = fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y);     // unclamped: h < 0 gives inter <= 0, a "no"
    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
    const float inter = w * h;
    const float u = (ai + aj) - inter;
    if (u > u_min) {
        if (inter > u * k_hi) return true;
        if (inter < u * k_lo) return false;
    }
    const float it = w * fmaxf(h, 0.0f);
    return it / ((ai + aj) - it) > thr;
}

// Quantisation of one box: outer box on a 15-bit grid and its threshold share T (< 2^29, so that
// w_q * h_q + (2^30 - T_j) and T_i + 2^30 both stay below 2^31).
struct QBox { uint32_t lo, hi, T; };
constexpr uint32_t Q_BIAS = 1u << 30;
constexpr float Q_MAX = 32767.0f;

// inter_up + (2^30 - T_j):  d = (h_q << 16) | w_q, one multiply-add on the two halves of d
__device__ inline uint32_t inter_up_biased(uint32_t d, uint32_t bias_minus_t) {
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %1, %2 op_sel:[0,1,0,0]" : "=v"(r) : "v"(d), "v"(bias_minus_t));
    return r;
}

__device__ inline QBox quantise_box(const float4 bx, const float area, const float x0, const float y0, const float scale,
                                    const float t_scale, const bool ok) {
    QBox q;
    if (!ok) { q.lo = 0u; q.hi = 0x7fff7fffu; q.T = 0u; return q; }        // never rejects anything
    const float fx1 = floorf((bx.x - x0) * scale - 0.0625f), fy1 = floorf((bx.y - y0) * scale - 0.0625f);
    const float fx2 = ceilf((bx.z - x0) * scale + 0.0625f), fy2 = ceilf((bx.w - y0) * scale + 0.0625f);
    const uint32_t x1 = (uint32_t)fminf(fmaxf(fx1, 0.0f), Q_MAX), y1 = (uint32_t)fminf(fmaxf(fy1, 0.0f), Q_MAX);
    const uint32_t x2 = (uint32_t)fminf(fmaxf(fx2, 0.0f), Q_MAX), y2 = (uint32_t)fminf(fmaxf(fy2, 0.0f), Q_MAX);
    q.lo = (y1 << 16) | x1;
    q.hi = (y2 << 16) | x2;
    const float t = area * t_scale;                                         // NaN or <= 0 -> 0
    q.T = t > 0.0f ? (uint32_t)fminf(t, 536870848.0f) : 0u;
    return q;
}

struct NmsFastLds { size_t keys, box, qbox, qt, kbox, kq, removed, mask, misc, wprefix, total; };

__host__ __device__ inline NmsFastLds nms_fast_lds(int max_seg, int n2, int blk) {
    const size_t words_cap = (size_t)((max_seg + 63) >> 6);
    NmsFastLds l;
    size_t off = 0;
    l.keys = off;                                                     // sort exchange buffer, dead before the boxes arrive
    const size_t xch = blk > 64 ? (size_t)(n2 > blk ? n2 : blk) * 8 : 0;           // a single wave sorts without LDS
    l.box = off;     off += (size_t)max_seg * 16 > xch ? (size_t)max_seg * 16 : xch;
    l.qbox = off;    off += (size_t)max_seg * 8;                       // quantised outer boxes (lo, hi)
    l.qt = off;      off += (((size_t)max_seg * 4 + 15) & ~(size_t)15); // threshold shares
    l.kbox = off;    off += 64 * 16;
    l.kq = off;      off += 65 * 16;                                   // quantised kept boxes (+1: the sweep prefetches one ahead)
    l.removed = off; off += ((words_cap * 8 + 15) & ~(size_t)15);
    l.mask = off;    off += 64 * 8;
    l.misc = off;    off += 96;                                       // keep word; 4 x 4 floats of the extent reduction
    l.wprefix = off; off += ((words_cap * 4 + 15) & ~(size_t)15);
    l.total = off;
    return l;
}

// Five workgroups per CU (five waves per SIMD) keep a whole 80-class x 16-image launch resident in one round.
// BLK = 256: up to 4096 boxes per segment; BLK = 64: one wave per segment (up to 256 boxes), no barriers at all.
template <int JPT, int BLK>
__device__ inline void nms_chunked_body(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
    const NmsParams &np, float2 *__restrict__ kept, int32_t *__restrict__ kcount, unsigned long long *s_dyn) {
    const int s = blockIdx.x, b = blockIdx.y;
    const int32_t *so = seg_off + (size_t)b * (np.S + 1);
    const int seg0 = so[s];
    const int n = so[s + 1] - seg0;
    if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = 0;
    if (n <= 0) return;
    if (n > np.max_seg) {
        if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = -1;      // reported as keep_counts[b] = -1
        return;
    }
    const NmsFastLds ld = nms_fast_lds(np.max_seg, np.n2, BLK);
    char *lds_base = reinterpret_cast<char *>(s_dyn);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(lds_base + ld.keys);
    float4 *s_box = reinterpret_cast<float4 *>(lds_base + ld.box);
    uint2 *s_qbox = reinterpret_cast<uint2 *>(lds_base + ld.qbox);
    uint32_t *s_qt = reinterpret_cast<uint32_t *>(lds_base + ld.qt);
    float4 *s_kbox = reinterpret_cast<float4 *>(lds_base + ld.kbox);
    uint4 *s_kq = reinterpret_cast<uint4 *>(lds_base + ld.kq);
    unsigned long long *s_removed = reinterpret_cast<unsigned long long *>(lds_base + ld.removed);
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(lds_base + ld.mask);
    unsigned long long *s_keepw = reinterpret_cast<unsigned long long *>(lds_base + ld.misc);
    float *s_red = reinterpret_cast<float *>(lds_base + ld.misc + 16);
    uint32_t *s_wprefix = reinterpret_cast<uint32_t *>(lds_base + ld.wprefix);

    const size_t row0 = (size_t)b * np.stride + seg0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int words = (n + 63) >> 6;
    const float thr = np.thr;
    // Division-free decision: with u = (area_i + area_j) - inter > u_min,
    //   inter > u * k_hi  =>  fl(inter / u) > thr      and      inter < u * k_lo  =>  fl(inter / u) <= thr,
    // k_hi / k_lo being thr widened by 2^-20 either way (the roundings of the products stay below 2^-22).
    const float k_hi = thr * 1.000001f, k_lo = thr * 0.999999f;
    const float u_min = thr > 0.0f ? 1e-30f / thr : __builtin_inff();

    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    unsigned long long K[JPT];
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int i = tid + m * BLK;
        K[m] = 0ull;
        if (i < n)   // every real key is above the zero padding: its low word is 0xffffffff - i > 0
            K[m] = ((unsigned long long)okey_from_bits(__float_as_uint(scores[row0 + i] + 0.0f)) << 32) |
                   (unsigned long long)(0xffffffffu - (uint32_t)i);
    }
    bitonic_desc_u64_regs<JPT, BLK>(K, s_keys, n2, tid);
    __syncthreads();                                     // the exchange buffer overlaps the box array
    // boxes in sorted order: LDS copy for the chunk phases, register copy of the thread's own boxes
    int orig[JPT];
    float4 bj[JPT];
    float aj[JPT];
    float lx = __builtin_inff(), ly = __builtin_inff(), hx = -__builtin_inff(), hy = -__builtin_inff();
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int j = tid + m * BLK;
        bj[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        aj[m] = 0.f;
        orig[m] = 0;
        if (j < n) {
            const int o = (int)(0xffffffffu - (uint32_t)(K[m] & 0xffffffffull));
            orig[m] = o;
            bj[m] = boxes[row0 + o];
            aj[m] = (bj[m].z - bj[m].x) * (bj[m].w - bj[m].y);
            s_box[j] = bj[m];
            lx = fminf(lx, bj[m].x); ly = fminf(ly, bj[m].y); hx = fmaxf(hx, bj[m].z); hy = fmaxf(hy, bj[m].w);
        }
    }
    for (int i = tid; i < words; i += BLK) s_removed[i] = 0ull;
    if (tid < 64) s_mask[tid] = 0ull;
    wave_extent<32>(lx, ly, hx, hy);
    if (lane == 0) { s_red[wv * 4 + 0] = lx; s_red[wv * 4 + 1] = ly; s_red[wv * 4 + 2] = hx; s_red[wv * 4 + 3] = hy; }
    __syncthreads();
    lx = s_red[0]; ly = s_red[1]; hx = s_red[2]; hy = s_red[3];
#pragma unroll
    for (int w = 1; w < BLK / 64; ++w) {
        lx = fminf(lx, s_red[4 * w]); ly = fminf(ly, s_red[4 * w + 1]);
        hx = fmaxf(hx, s_red[4 * w + 2]); hy = fmaxf(hy, s_red[4 * w + 3]);
    }
    // 16-bit grid over the extent of the segment; switched off (nothing is ever rejected early) when the
    // extent is not a sane positive number
    const float E = fmaxf(hx - lx, hy - ly);
    const bool q_ok = E > 1e-30f && E < 1e30f;
    const float scale = q_ok ? Q_MAX / E : 0.0f;
    const float thr_q = fminf(thr * 0.9999f, 1.0f);
    const float t_scale = (thr_q / (1.0f + thr_q)) * scale * scale * 0.9999f;
    uint32_t qlo[JPT], qhi[JPT], qTn[JPT];               // qTn = 2^30 - T
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const QBox q = quantise_box(bj[m], aj[m], lx, ly, scale, t_scale, q_ok);
        qlo[m] = q.lo; qhi[m] = q.hi; qTn[m] = Q_BIAS - q.T;
        const int j = tid + m * BLK;
        if (j < n) { s_qbox[j] = make_uint2(q.lo, q.hi); s_qt[j] = q.T; }
    }
    __syncthreads();

    unsigned rem_mask = 0u;                               // bit m: own box m is suppressed
    for (int ch = 0; ch < words; ++ch) {
        const int c0 = ch << 6;
        const int cn = min(64, n - c0);
        const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
        const unsigned long long alive0 = ~s_removed[ch] & valid;
        {   // mutual suppression words of the chunk; rows rp and 63-rp together hold 63 pairs, TPR threads share them
            constexpr int TPR = BLK / 32;
            const int rp = tid / TPR, s8 = tid % TPR;
            for (int c = s8; c < 63; c += TPR) {
                const bool first = c < 63 - rp;
                const int i = first ? rp : 63 - rp;
                const int j = first ? rp + 1 + c : c + 1;
                if (j < cn && ((alive0 >> i) & 1ull) && ((alive0 >> j) & 1ull)) {
                    const uint2 qi = s_qbox[c0 + i], qj = s_qbox[c0 + j];
                    const u16x2 d = __builtin_elementwise_sub_sat(__builtin_elementwise_min(as_u16x2(qi.y), as_u16x2(qj.y)),
                                                                  __builtin_elementwise_max(as_u16x2(qi.x), as_u16x2(qj.x)));
                    if ((uint32_t)d.x * (uint32_t)d.y > s_qt[c0 + i] + s_qt[c0 + j]) {      // not rejected: the exact test
                        const float4 bi = s_box[c0 + i];
                        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
                        if (suppresses_nodiv(bi, ai, s_box[c0 + j], thr, k_hi, k_lo, u_min)) atomicOr(&s_mask[i], 1ull << j);
                    }
                }
            }
        }
        __syncthreads();
        if (wv == 0) {
            const unsigned long long m = s_mask[lane];
            s_mask[lane] = 0ull;
            unsigned long long rows = __ballot(m != 0ull);
            unsigned long long keep = uniform64(alive0);
            while (rows) {                                   // only rows that suppress anything matter, in order
                const int i = __ffsll((long long)rows) - 1;
                rows &= rows - 1ull;
                if ((keep >> i) & 1ull) keep &= ~readlane64(m, i);
            }
            if ((keep >> lane) & 1ull) {                     // compact the kept boxes: fp32 and quantised (share biased by 2^30)
                const int rk = __popcll(keep & ((1ull << lane) - 1ull));
                const uint2 q = s_qbox[c0 + lane];
                s_kbox[rk] = s_box[c0 + lane];
                s_kq[rk] = make_uint4(q.x, q.y, s_qt[c0 + lane] + Q_BIAS, 0u);
            }
            if (lane == 0) { s_keepw[0] = keep; s_removed[ch] = valid & ~keep; }
        }
        __syncthreads();
        const int nk = __popcll(s_keepw[0]);
        // The kept boxes of this chunk against every later box still alive.  The later boxes are the
        // lanes' own register boxes; kept box t is broadcast with readlane from a register image of the
        // kept list; which lanes are still alive is a wave-uniform 64-bit mask per register slot, so
        // all bookkeeping is scalar and a slot whose lanes are all settled is skipped.
        if (nk > 0 && ch + 1 < words) {
            unsigned long long act[JPT];
            bool any_act = false;
#pragma unroll
            for (int m = 0; m < JPT; ++m) {
                const int j = tid + m * BLK;
                act[m] = __ballot(j >= c0 + 64 && j < n && !((rem_mask >> m) & 1u));
                any_act |= act[m] != 0ull;
            }
            if (any_act) {
                const float4 kb = lane < nk ? s_kbox[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float ka = (kb.z - kb.x) * (kb.w - kb.y);
                unsigned long long hit[JPT];
#pragma unroll
                for (int m = 0; m < JPT; ++m) hit[m] = 0ull;
                // Slots below the first live one lie entirely before the chunk; the sweep is compiled for every
                // possible first slot (in steps of GS), so no slot is visited just to find its mask empty.
                constexpr int GS = JPT > 4 ? JPT / 4 : 1;
                int m_first = 0;
#pragma unroll
                for (int m = JPT - 1; m >= 0; --m)
                    if (act[m]) m_first = m;
                auto sweep = [&](auto first_slot) {
                    constexpr int M0 = decltype(first_slot)::value;
                    auto against = [&](const int t, const uint4 kq) {
                        const uint32_t klo = kq.x, khi = kq.y, ktb = kq.z;
                        // stage 1 for every live slot first (independent chains, one branch for all of them):
                        // integer upper bound of the intersection against the threshold shares
                        unsigned long long open_m[JPT];
                        unsigned long long any_open = 0ull;
#pragma unroll
                        for (int m = M0; m < JPT; ++m) {
                            const u16x2 lo = __builtin_elementwise_max(as_u16x2(klo), as_u16x2(qlo[m]));
                            const u16x2 hi = __builtin_elementwise_min(as_u16x2(khi), as_u16x2(qhi[m]));
                            const uint32_t d = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(hi, lo));
                            open_m[m] = act[m] & ~__ballot(inter_up_biased(d, qTn[m]) <= ktb);
                            any_open |= open_m[m];
                        }
                        if (!any_open) return;
#pragma unroll
                        for (int m = M0; m < JPT; ++m) {
                            const unsigned long long open = open_m[m];
                            if (open) {                                  // stage 2: the reference's arithmetic
                                const float ix1 = readlane_f(kb.x, t), iy1 = readlane_f(kb.y, t);
                                const float ix2 = readlane_f(kb.z, t), iy2 = readlane_f(kb.w, t);
                                const float ai = readlane_f(ka, t);
                                // One clamp is enough: with w >= 0, a negative h makes inter <= 0, which is a "no".
                                const float w = fmaxf(fminf(ix2, bj[m].z) - fmaxf(ix1, bj[m].x), 0.0f);
                                const float h = fminf(iy2, bj[m].w) - fmaxf(iy1, bj[m].y);
                                const float inter = w * h;
                                const float u = (ai + aj[m]) - inter;
                                const unsigned long long sane = __ballot(u > u_min);
                                unsigned long long yes = __ballot(inter > u * k_hi) & sane;
                                const unsigned long long no = __ballot(inter < u * k_lo) & sane;
                                const unsigned long long amb = open & ~(yes | no);
                                if (amb) {                               // the sliver the margins leave: the reference's own test
                                    const float it = w * fmaxf(h, 0.0f);
                                    yes |= __ballot(it / ((ai + aj[m]) - it) > thr) & amb;
                                }
                                yes &= open;
                                act[m] &= ~yes;
                                hit[m] |= yes;
                            }
                        }
                    };
                    // broadcast LDS reads one kept box ahead; two per trip so that the prefetched box needs no register copy
                    uint4 k0 = s_kq[0];
                    for (int t = 0; t < nk; t += 2) {
                        const uint4 k1 = s_kq[t + 1];
                        against(t, k0);
                        if (t + 1 >= nk) break;
                        k0 = s_kq[t + 2];
                        against(t + 1, k1);
                    }
                };
                switch (m_first / GS) {
                    case 0: sweep(std::integral_constant<int, 0>{}); break;
                    case 1: sweep(std::integral_constant<int, (JPT > 1 ? GS : 0)>{}); break;
                    case 2: sweep(std::integral_constant<int, (JPT > 2 ? 2 * GS : 0)>{}); break;
                    default: sweep(std::integral_constant<int, (JPT > 3 ? 3 * GS : 0)>{}); break;
                }
#pragma unroll
                for (int m = 0; m < JPT; ++m) {
                    if ((hit[m] >> lane) & 1ull) {
                        const int j = tid + m * BLK;
                        rem_mask |= 1u << m;
                        atomicOr(&s_removed[j >> 6], 1ull << (j & 63));
                    }
                }
            }
        }
        __syncthreads();
    }
    // kept = not removed; compact in sorted order
    if (wv == 0) {
        uint32_t running = 0;
        for (int base = 0; base < words; base += 64) {
            const int w = base + lane;
            uint32_t pc = 0;
            if (w < words) {
                const int cn = min(64, n - (w << 6));
                const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
                pc = (uint32_t)__popcll(~s_removed[w] & valid);
            }
            const uint32_t inc = wave_inclusive_add(pc);
            if (w < words) s_wprefix[w] = running + inc - pc;
            running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) kcount[(size_t)b * np.S + s] = (int32_t)running;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int i = tid + m * BLK;
        if (i >= n) continue;
        const unsigned long long rem = s_removed[i >> 6];
        if ((rem >> (i & 63)) & 1ull) continue;
        const uint32_t rank = s_wprefix[i >> 6] + (uint32_t)__popcll(~rem & ((1ull << (i & 63)) - 1ull));
        const int pos = seg0 + orig[m];
        kept[row0 + rank] = make_float2(__uint_as_float(okey_from_bits(__float_as_uint(scores[(size_t)b * np.stride + pos] + 0.0f))), __int_as_float(pos));
    }
}

template <int JPT, int BLK>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == 256 && JPT <= 4 ? 5 : 1))) void nms_segments_fast_kernel(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
    NmsParams np, float2 *__restrict__ kept, int32_t *__restrict__ kcount) {
    extern __shared__ unsigned long long s_dyn[];
    nms_chunked_body<JPT, BLK>(boxes, scores, seg_off, np, kept, kcount, s_dyn);
}

// ---------------------------------------------------------------------------------------------
// Pair-list form of the same greedy NMS (unsorted segments of at most 256*JPT boxes, JPT <= 4, threshold >= 0).
//
// Greedy NMS keeps box j iff no KEPT box i before it (in descending score order) has iou(i, j) > thr.  That
// is a function of the set of edges {(i, j): i before j, iou(i, j) > thr} alone, so the kernel first finds
// all edges -- an order-free all-pairs join with no dependency between pairs -- and only then walks them
// in score order:
//   1. sort (registers, as in the chunked kernel): every box learns its rank in score order.  The four-wave
//      form then lays the boxes out in a *spatial* order for the join, by a counting sort in LDS: one bin for
//      the large boxes (radius r below), then 256 rows of the centre for the medium and 256 for the small ones.
//      Which of two boxes comes first in that order is irrelevant to the join (every unordered pair is met
//      once); 32 consecutive boxes now share a narrow band of the image.  Every wave owns JPT 64-box chunks of
//      the order, dealt boustrophedon so that the four waves (one per SIMD) carry equal shares of the triangle;
//   2. filter: box i (broadcast from LDS) against a wave's 64-box slots with three VALU instructions per 64
//      pairs, window by window of 32 boxes i.  A window and a slot each carry the bounding intervals of their
//      centres -+ radii; when those do not meet in x or in y no pair of the two can pass, and the window is
//      skipped for that slot (one specialised sweep per set of active slots).  With p = |cx_i - cx_j|, q = |cy_i - cy_j|, W = (w_i + w_j) / 2, H = (h_i + h_j) / 2, the overlaps
//      satisfy  overlap_x <= W - p,  overlap_y <= H - q,  and  iou > thr  means
//          overlap_x * overlap_y > t * (A_i + A_j),  t = thr / (1 + thr).
//      By AM-GM  (W - p) + (H - q) >= 2 sqrt((W - p)(H - q)) > 2 sqrt(t (A_i + A_j)) >= sqrt(2 t) (sqrt A_i + sqrt A_j),
//      hence     p + q < r_i + r_j,   r = (w + h) / 2 - sqrt(2 t w h)   (>= 0 for thr <= 1).
//      Centres on a 15-bit grid, radii rounded up: v_sad_u16 gives p + q + (BIAS - r_j) per slot, the compare
//      against BIAS + r_i goes through VCC into a 32-bit history per lane and slot (one bit per box i of a window
//      of 32).  After each window the set bits (a few pairs per thousand) become entries of the wave's own queue
//      in LDS: one prefix sum over the lanes' bit counts hands every lane its places, no atomics, no ballots;
//   3. exact test: the wave drains its queue 64 pairs at a time with the reference's fp32 arithmetic
//      (division-free where margins decide, the IEEE division in the sliver; the test is symmetric in the two
//      boxes, bit for bit) on the copy of the boxes in LDS, compacting the true edges in place at the front of
//      the queue as (rank of the higher score, rank of the lower score);
//   4. resolve: a few rounds of  removed' = { j : some edge (i, j) with i not removed }  from the empty set,
//      one pass over the edges each, until a round changes nothing; long chains of edges take the
//      chunk-by-chunk form instead (edges from earlier chunks remove their targets if the source was kept,
//      edges inside the chunk form the 64 x 64 words one wave resolves serially).
// No barrier inside the all-pairs phase.  If a wave's queue cannot hold its edges (heavily clustered boxes:
// many true suppressions) the segment is redone by the chunked algorithm above, whose cost shrinks exactly
// when suppressions are many.  The results are the reference's, bit for bit: the filter only ever discards
// pairs that cannot suppress, and every edge is decided by the exact test.
// ---------------------------------------------------------------------------------------------

constexpr uint32_t PAIR_BIAS = 1u << 24;          // radii stay below 2^17 grid units

// Measurement aid (detbox_debug_nms_stats): segments, segments redone by the chunked algorithm, pairs that passed
// the filter, true edges, queue drains -- summed over all launches since the last reset.
__device__ unsigned long long g_pair_stats[8];

struct NmsPairLds { size_t box, q, irec, removed, mask, misc, wprefix, rank, win, total; };

// Score order of a segment by a distribution sort instead of the bitonic network.  The keys of a class's
// candidates lie in a narrow, smoothly filled range, so a linear map of the score key onto RS_NB buckets leaves about one
// key per bucket: one LDS atomic per key counts the buckets, a prefix sum places them, the keys are scattered, and a key's
// rank is its bucket's start plus the number of larger keys in its bucket.  Some dozens of instructions per key and seven
// barriers, against the 55 compare-exchange stages of the network (27 k of an 1000-box segment's 50 k cycles up to the
// records).  Keys that pile up (equal scores: more than RS_BUCKET_MAX in a bucket) leave the segment to the network.
constexpr int RS_NB = 1024;            // buckets of the four-wave form (up to 1024 keys); the single-wave form (up to 256) takes 256
constexpr int RS_BUCKET_MAX = 32;
__host__ __device__ constexpr int rs_buckets(int blk) { return blk > 64 ? RS_NB : 256; }

template <int O>
__device__ inline uint32_t rs_wave_min(uint32_t v) {
    v = min(v, lane_xor_u32<O>(v));
    if constexpr (O > 1) return rs_wave_min<O / 2>(v);
    else return v;
}

// rank[m] = descending rank of K[m] (the key of element tid + m * BLK, distinct keys; elements beyond n are ignored).
// s_b: n words of scratch; s_hist: RS_NB + 16 words.  Returns false (block-uniform) when a bucket is too full.
template <int JPT, int BLK>
__device__ inline bool rank_by_buckets(const unsigned long long (&K)[JPT], const int n, unsigned long long *__restrict__ s_b,
                                       uint32_t *__restrict__ s_hist, uint32_t (&rank)[JPT]) {
    constexpr int RS_NB = rs_buckets(BLK);           // (shadows the four-wave constant on purpose)
    static_assert(RS_NB % BLK == 0, "buckets per thread");
    constexpr int BPT = RS_NB / BLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *s_min = s_hist + RS_NB, *s_max = s_hist + RS_NB + 1, *s_big = s_hist + RS_NB + 2, *s_w = s_hist + RS_NB + 4;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int m = 0; m < JPT; ++m)
        if (tid + m * BLK < n) { kmin = min(kmin, (uint32_t)(K[m] >> 32)); kmax = max(kmax, (uint32_t)(K[m] >> 32)); }
#pragma unroll
    for (int q = 0; q < BPT; ++q) s_hist[tid + q * BLK] = 0u;
    if (tid == 0) { *s_min = 0xffffffffu; *s_max = 0u; *s_big = 0u; }
    __syncthreads();
    kmin = rs_wave_min<32>(kmin);
    kmax = ~rs_wave_min<32>(~kmax);
    if (lane == 0) { atomicMin(s_min, kmin); atomicMax(s_max, kmax); }
    __syncthreads();
    const uint32_t hi = *s_max, width = hi - *s_min;
    // monotone in the key: int -> float rounding, the product with a positive scale and the truncation all are
    const float scale = width ? (float)(RS_NB - 1) / (float)width : 0.0f;
    int bucket[JPT];
    uint32_t slot[JPT];
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        bucket[m] = min(RS_NB - 1, (int)((float)(hi - (uint32_t)(K[m] >> 32)) * scale));
        slot[m] = 0u;
        if (tid + m * BLK < n) slot[m] = atomicAdd(&s_hist[bucket[m]], 1u);
        else bucket[m] = RS_NB - 1;
    }
    __syncthreads();
    {   // exclusive prefix sums of the bucket counts, in place; the largest count
        uint32_t c[BPT], sum = 0u, big = 0u;
#pragma unroll
        for (int q = 0; q < BPT; ++q) { c[q] = s_hist[tid * BPT + q]; sum += c[q]; big = max(big, c[q]); }
        const uint32_t inc = wave_inclusive_add(sum);
        if (lane == 63) s_w[wave] = inc;
        if (big > (uint32_t)RS_BUCKET_MAX) *s_big = 1u;
        __syncthreads();
        uint32_t base = inc - sum;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) base += w < wave ? s_w[w] : 0u;
#pragma unroll
        for (int q = 0; q < BPT; ++q) { s_hist[tid * BPT + q] = base; base += c[q]; }
    }
    __syncthreads();
    if (*s_big) return false;
#pragma unroll
    for (int m = 0; m < JPT; ++m)
        if (tid + m * BLK < n) s_b[s_hist[bucket[m]] + slot[m]] = K[m];
    __syncthreads();
    int bs[JPT], len[JPT], ahead[JPT], longest = 0;
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        bs[m] = (int)s_hist[bucket[m]];
        len[m] = tid + m * BLK < n ? (bucket[m] + 1 < RS_NB ? (int)s_hist[bucket[m] + 1] : n) - bs[m] : 0;
        ahead[m] = 0;
        longest = max(longest, len[m]);
    }
    for (int j = 0; j < longest; ++j) {              // the thread's keys walk their buckets together: the LDS reads overlap
#pragma unroll
        for (int m = 0; m < JPT; ++m) {
            const unsigned long long x = s_b[min(bs[m] + j, n - 1)];
            ahead[m] += (j < len[m] && x > K[m]) ? 1 : 0;
        }
    }
#pragma unroll
    for (int m = 0; m < JPT; ++m) rank[m] = (uint32_t)(bs[m] + ahead[m]);
    return true;
}

// Entries per wave queue.  A queue that fills up is drained (tested pairs leave, true edges stay at its front); a window
// that passes more pairs than are free is appended one box i at a time (at most 64 * JPT pairs each); the segment goes to
// the chunked algorithm once the true edges of a wave leave no room for that.  With 1000-box segments the whole layout is
// 31.0 KB: five workgroups share a CU (at 32 KB to the byte only four were resident).
constexpr int PAIR_QCAP = 256;
// Spatial order of the all-pairs phase: boxes are binned by size class and by the row of their centre (a counting sort:
// one bin for the large boxes, 256 rows each for the medium and the small ones), so that 32 consecutive boxes share a
// narrow band of the image and whole windows of pairs are skipped by their bounding intervals.
constexpr int PAIR_BINS = 513;
static_assert(PAIR_BINS <= 576, "bins");
constexpr int PAIR_BINS_PAD = 576;           // nine bins per lane of the wave that turns the counts into offsets
constexpr uint32_t PAIR_R_MEDIUM = 1024u;    // radii (grid units of the 15-bit grid) from which a box is medium / large
constexpr uint32_t PAIR_R_LARGE = 4096u;
constexpr int PAIR_RESOLVE_ROUNDS = 8;        // fixed-point rounds of the resolve before it takes the chunk-by-chunk form

__host__ __device__ inline NmsPairLds nms_pair_lds(int max_seg, int n2, int blk) {
    const size_t words_cap = (size_t)((max_seg + 63) >> 6);
    NmsPairLds l;
    size_t off = 0;
    const size_t xch = blk > 64 ? (size_t)(n2 > blk ? n2 : blk) * 8 : 0;          // sort exchange buffer, dead before the boxes arrive
    const size_t boxes = (size_t)max_seg * 16;                                    // the boxes in sorted order, for the exact test
    // (the rank sort scatters the keys into the first max_seg words and keeps its bucket counters behind them)
    const size_t rsort = (((size_t)max_seg * 8 + 15) & ~(size_t)15) + (size_t)(rs_buckets(blk) + 16) * 4;
    size_t box_bytes = boxes > xch ? boxes : xch;
    if (rsort > box_bytes) box_bytes = rsort;
    l.box = off;     off += box_bytes;
    const size_t queues = (size_t)(blk / 64) * PAIR_QCAP * 4;
    // rank -> row inside the segment (2048 bytes: at most 1024 boxes), then the bins of the counting sort; both dead before the queues fill
    const size_t orig = blk > 64 ? 2048 + (size_t)PAIR_BINS_PAD * 4 : (((size_t)max_seg * 2 + 15) & ~(size_t)15);
    l.q = off;       off += queues > orig ? queues : orig;
    l.irec = off;    off += (size_t)((max_seg + 63) & ~63) * 8 + 64;              // per box: packed centre, BIAS + radius (+ read-ahead slack)
    l.removed = off; off += ((words_cap * 8 + 15) & ~(size_t)15);
    l.mask = off;    off += 64 * 8;
    l.misc = off;    off += 128;                                                 // keep word; extent reduction; edge counts; overflow flag
    l.wprefix = off; off += ((words_cap * 4 + 15) & ~(size_t)15);
    l.rank = off;    off += blk > 64 ? (((size_t)max_seg * 2 + 15) & ~(size_t)15) : 0;   // spatial position -> rank in score order
    l.win = off;     off += blk > 64 ? 32 * 16 : 0;                                     // bounding intervals of the 32-box windows
    l.total = off;
    return l;
}

// Returns false when a queue overflowed (block-uniform): nothing was written, the caller falls back.
template <int JPT, int BLK>
__device__ inline bool nms_pairs_body(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
    const NmsParams &np, float2 *__restrict__ kept, int32_t *__restrict__ kcount, unsigned long long *s_dyn) {
    static_assert(JPT <= 4, "the sweep is specialised for at most four slots per wave");
    constexpr int W = BLK / 64;
    constexpr int QCAP = PAIR_QCAP;
    // The four-wave form (up to 1024 boxes) orders the boxes spatially for the all-pairs phase; the single-wave form (up to
    // 256 boxes: at most eight windows) keeps the score order, where the re-ordering would cost more than it skips.
    constexpr bool SPATIAL = BLK > 64;
    const int s = blockIdx.x, b = blockIdx.y;
    const int32_t *so = seg_off + (size_t)b * (np.S + 1);
    const int seg0 = so[s];
    const int n = so[s + 1] - seg0;
    if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = 0;
    if (n <= 0) return true;
    if (n > np.max_seg) {
        if (threadIdx.x == 0) kcount[(size_t)b * np.S + s] = -1;      // reported as keep_counts[b] = -1
        return true;
    }
    const NmsPairLds ld = nms_pair_lds(np.max_seg, np.n2, BLK);
    char *lds_base = reinterpret_cast<char *>(s_dyn);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(lds_base + ld.box);
    float4 *s_box = reinterpret_cast<float4 *>(lds_base + ld.box);
    uint32_t *s_queue = reinterpret_cast<uint32_t *>(lds_base + ld.q);
    uint16_t *s_orig = reinterpret_cast<uint16_t *>(lds_base + ld.q);
    uint2 *s_irec = reinterpret_cast<uint2 *>(lds_base + ld.irec);
    unsigned long long *s_removed = reinterpret_cast<unsigned long long *>(lds_base + ld.removed);
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(lds_base + ld.mask);
    float *s_red = reinterpret_cast<float *>(lds_base + ld.misc + 16);            // 4 x 4 floats
    uint32_t *s_ne = reinterpret_cast<uint32_t *>(lds_base + ld.misc + 80);       // edges per wave
    uint32_t *s_ovf = reinterpret_cast<uint32_t *>(lds_base + ld.misc + 96);
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(lds_base + ld.misc + 100);    // two words: "this round changed something"
    uint32_t *s_wprefix = reinterpret_cast<uint32_t *>(lds_base + ld.wprefix);
    uint32_t *s_bin = reinterpret_cast<uint32_t *>(lds_base + ld.q + 2048);
    uint16_t *s_rank = reinterpret_cast<uint16_t *>(lds_base + ld.rank);
    float4 *s_win = reinterpret_cast<float4 *>(lds_base + ld.win);

    const size_t row0 = (size_t)b * np.stride + seg0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and known to be
    const int words = (n + 63) >> 6;
    const float thr = np.thr;
    const float k_hi = thr * 1.000001f, k_lo = thr * 0.999999f;
    const float u_min = thr > 0.0f ? 1e-30f / thr : __builtin_inff();

    // ---- 1. sort
    const long long t_begin = np.stats ? (long long)__builtin_readcyclecounter() : 0ll;
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    {
        unsigned long long K[JPT];
#pragma unroll
        for (int m = 0; m < JPT; ++m) {
            const int i = tid + m * BLK;
            K[m] = 0ull;
            if (i < n)   // every real key is above the zero padding: its low word is 0xffffffff - i > 0
                K[m] = ((unsigned long long)okey_from_bits(__float_as_uint(scores[row0 + i] + 0.0f)) << 32) |
                       (unsigned long long)(0xffffffffu - (uint32_t)i);
        }
        bool ranked = false;
        if (!np.network_sort) {
            uint32_t rk[JPT];
            uint32_t *s_hist = reinterpret_cast<uint32_t *>(lds_base + ld.box + ((((size_t)np.max_seg * 8 + 15) & ~(size_t)15)));
            ranked = rank_by_buckets<JPT, BLK>(K, n, s_keys, s_hist, rk);
            if (ranked) {
#pragma unroll
                for (int m = 0; m < JPT; ++m) {
                    const int i = tid + m * BLK;
                    if (i < n) s_orig[rk[m]] = (uint16_t)i;
                }
            }
        }
        if (!ranked) {
            bitonic_desc_u64_regs<JPT, BLK>(K, s_keys, n2, tid);
#pragma unroll
            for (int m = 0; m < JPT; ++m) {
                const int i = tid + m * BLK;
                if (i < n) s_orig[i] = (uint16_t)(0xffffffffu - (uint32_t)(K[m] & 0xffffffffull));
            }
        }
    }
    for (int i = tid; i < words; i += BLK) s_removed[i] = 0ull;
    if (tid < 64) s_mask[tid] = 0ull;
    if (tid == 0) { *s_ovf = 0u; s_flag[0] = 0u; s_flag[1] = 0u; }
    if (SPATIAL)
        for (int i = tid; i < PAIR_BINS_PAD; i += BLK) s_bin[i] = 0u;
    __syncthreads();
    // this thread's boxes, for the records and (at the very end) for the output: ranks tid, tid + BLK, ... of the score order
    int orig[JPT];
    float4 ba[JPT];
    float lx = __builtin_inff(), ly = __builtin_inff(), hx = -__builtin_inff(), hy = -__builtin_inff();
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int rk = tid + m * BLK;
        ba[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        orig[m] = 0;
        if (rk < n) {
            orig[m] = (int)s_orig[rk];
            ba[m] = boxes[row0 + orig[m]];
            lx = fminf(lx, ba[m].x); ly = fminf(ly, ba[m].y); hx = fmaxf(hx, ba[m].z); hy = fmaxf(hy, ba[m].w);
        }
    }
    wave_extent<32>(lx, ly, hx, hy);
    if (lane == 0) { s_red[wv * 4 + 0] = lx; s_red[wv * 4 + 1] = ly; s_red[wv * 4 + 2] = hx; s_red[wv * 4 + 3] = hy; }
    __syncthreads();
    lx = s_red[0]; ly = s_red[1]; hx = s_red[2]; hy = s_red[3];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        lx = fminf(lx, s_red[4 * w]); ly = fminf(ly, s_red[4 * w + 1]);
        hx = fmaxf(hx, s_red[4 * w + 2]); hy = fmaxf(hy, s_red[4 * w + 3]);
    }
    // ---- 2a. centres and radii on a 15-bit grid over the extent of the segment.  With a grid that is not a sane
    // positive number the filter passes every pair (and the queues overflow into the chunked algorithm).
    const float E = fmaxf(hx - lx, hy - ly);
    const bool q_ok = E > 1e-30f && E < 1e30f;
    const float scale = q_ok ? Q_MAX / E : 0.0f;
    const float thr_q = thr * 0.999f;
    const float two_t = fminf(2.0f * thr_q / (1.0f + thr_q), 1.0f);      // 2 * thr / (1 + thr), at most 1
    // A box's record: packed centre, BIAS + radius.  In the j role the same box enters as BIAS - radius, which is
    // 2 * BIAS - (BIAS + radius); a box without positive width and height has the record (0, 0) and meets nobody in
    // either role.  Bins of the spatial order: 0 for the large boxes, then 256 rows of medium and 256 rows of small ones.
    uint32_t c_a[JPT], ri_a[JPT], place_a[JPT];
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int rk = tid + m * BLK;
        const float w = ba[m].z - ba[m].x, h = ba[m].w - ba[m].y;
        uint32_t c = 0u, ri = 0u, bin = 0u, place = 0u;
        if (rk < n) {
            if (!q_ok) ri = PAIR_BIAS;                     // radius 0 on a grid of one point: passes every real box
            else if (w > 0.0f && h > 0.0f) {
                const float fx = rintf(((ba[m].x + ba[m].z) * 0.5f - lx) * scale), fy = rintf(((ba[m].y + ba[m].w) * 0.5f - ly) * scale);
                const uint32_t x = (uint32_t)fminf(fmaxf(fx, 0.0f), Q_MAX), y = (uint32_t)fminf(fmaxf(fy, 0.0f), Q_MAX);
                c = (y << 16) | x;
                // rounding of the two centres moves |dcx| + |dcy| by at most 2 grid units (+ fp32 slop): 1.25 per box
                const float rad = fmaxf((w + h) * 0.5f - sqrtf(two_t * w * h), 0.0f) * 1.001f * scale;
                const uint32_t r = (uint32_t)fminf(ceilf(rad + 1.25f), 131000.0f);
                ri = PAIR_BIAS + r;
                if (r < PAIR_R_LARGE) bin = (r >= PAIR_R_MEDIUM ? 1u : 257u) + (y >> 7);
            }
            if (SPATIAL) place = atomicAdd(&s_bin[bin], 1u);             // place inside the bin, in no particular order
        }
        c_a[m] = c; ri_a[m] = ri;
        place_a[m] = (bin << 16) | place;
    }
    if (SPATIAL) {
        __syncthreads();
        if (wv == 0) {                                          // bin counts -> bin offsets, nine bins per lane
            uint32_t cnts[9], sum = 0u;
#pragma unroll
            for (int k = 0; k < 9; ++k) { cnts[k] = s_bin[lane * 9 + k]; sum += cnts[k]; }
            uint32_t run = wave_inclusive_add(sum) - sum;
#pragma unroll
            for (int k = 0; k < 9; ++k) { s_bin[lane * 9 + k] = run; run += cnts[k]; }
        }
    }
    __syncthreads();                                          // (single-wave form: everybody has read the row table)
    // records, boxes and ranks by spatial position (the sort buffer is dead: the boxes may land); padding behind them
    const int cap_pos = (np.max_seg + 63) & ~63;
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int rk = tid + m * BLK;
        if (rk < n) {
            const uint32_t sp = SPATIAL ? s_bin[place_a[m] >> 16] + (place_a[m] & 0xffffu) : (uint32_t)rk;
            s_irec[sp] = make_uint2(c_a[m], ri_a[m]);
            s_box[sp] = ba[m];
            if (SPATIAL) s_rank[sp] = (uint16_t)rk;
        }
    }
    for (int i = n + tid; i < cap_pos; i += BLK) s_irec[i] = make_uint2(0u, 0u);
    __syncthreads();                                      // records and boxes visible; the row table and the bins are dead: queues may fill
    // slot m of this wave is chunk m*W + wv (even m) or m*W + W-1-wv (odd m) of the spatial order
    int pos[JPT];
    uint32_t cq[JPT], nj[JPT];                            // packed centre; BIAS - radius (j role)
    float sb_lx[JPT], sb_ly[JPT], sb_hx[JPT], sb_hy[JPT];  // wave-uniform: bounding intervals of the slot's centres -+ radii
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int ch = m * W + ((m & 1) ? (W - 1 - wv) : wv);
        pos[m] = (ch << 6) + lane;
        uint2 rec = make_uint2(0u, 0u);
        if (pos[m] < cap_pos) rec = s_irec[pos[m]];
        cq[m] = rec.x; nj[m] = 2u * PAIR_BIAS - rec.y;
        sb_lx[m] = 0.f; sb_ly[m] = 0.f; sb_hx[m] = 0.f; sb_hy[m] = 0.f;
        if (!SPATIAL) continue;
        float x0 = __builtin_inff(), y0 = __builtin_inff(), x1 = -__builtin_inff(), y1 = -__builtin_inff();
        if (rec.y >= PAIR_BIAS) {
            const float r = (float)(rec.y - PAIR_BIAS), cx = (float)(rec.x & 0xffffu), cy = (float)(rec.x >> 16);
            x0 = cx - r; x1 = cx + r; y0 = cy - r; y1 = cy + r;
        }
        wave_extent<16>(x0, y0, x1, y1);                    // the two windows of 32 boxes of this chunk
        if ((lane & 31) == 0 && ch < 16) s_win[2 * ch + (lane >> 5)] = make_float4(x0, y0, x1, y1);
        x0 = fminf(x0, lane_xor_f32<32>(x0)); y0 = fminf(y0, lane_xor_f32<32>(y0));
        x1 = fmaxf(x1, lane_xor_f32<32>(x1)); y1 = fmaxf(y1, lane_xor_f32<32>(y1));
        sb_lx[m] = readlane_f(x0, 0); sb_ly[m] = readlane_f(y0, 0); sb_hx[m] = readlane_f(x1, 0); sb_hy[m] = readlane_f(y1, 0);
    }
    __syncthreads();                                      // window intervals visible
    const long long t_sorted = np.stats ? (long long)__builtin_readcyclecounter() : 0ll;

    // ---- 2b / 3. all pairs (i before j), wave-local queue
    uint32_t *q = s_queue + wv * QCAP;
    uint32_t nq = 0, ne = 0;                              // wave-uniform: entries in the queue; of them, tested edges at the front
    uint32_t n_passed = 0, n_drains = 0;
    bool ovf = false;
    auto drain = [&]() {
        for (uint32_t base = ne; base < nq; base += 64) {
            const uint32_t idx = base + lane;
            uint32_t e = 0u;
            bool yes = false;
            if (idx < nq) {
                e = q[idx];
                const int i = (int)(e >> 16), j = (int)(e & 0xffffu);
                const float4 bi = s_box[i], bk = s_box[j];
                const float ai = (bi.z - bi.x) * (bi.w - bi.y);
                yes = suppresses_nodiv(bi, ai, bk, thr, k_hi, k_lo, u_min);
            }
            const unsigned long long ym = __ballot(yes);
            if (yes) {      // a true edge leaves the spatial positions behind: (rank of the earlier box, rank of the later one)
                const uint32_t ra = SPATIAL ? (uint32_t)s_rank[e >> 16] : (e >> 16), rb = SPATIAL ? (uint32_t)s_rank[e & 0xffffu] : (e & 0xffffu);
                q[ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(ym >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ym, 0u))] =
                    ra < rb ? ((ra << 16) | rb) : ((rb << 16) | ra);
            }
            ne += (uint32_t)__popcll(ym);
        }
        nq = ne;
    };
    // The filter proper touches no scalar state: for a window of 32 boxes i, every lane shifts the outcome of each
    // test into a 32-bit history per slot (compare into VCC, add-with-carry: three VALU per 64 pairs, no branch).
    // The histories are then expanded into queue entries, one round per set bit of the busiest lane.
    for (int ic = 0; ic < words && !ovf; ++ic) {
        int m0 = JPT;
#pragma unroll
        for (int m = JPT - 1; m >= 0; --m)
            if (m * W + ((m & 1) ? (W - 1 - wv) : wv) >= ic) m0 = m;
        if (m0 == JPT) break;                              // every slot of this wave lies before chunk ic
        const bool intra = (m0 * W + ((m0 & 1) ? (W - 1 - wv) : wv)) == ic;
        const int c0 = ic << 6, cn = min(64, n - c0);
        for (int t0 = 0; t0 < cn && !ovf; t0 += 32) {
            // slots (from m0 on) whose bounding intervals meet those of this window of 32 boxes: the others cannot pass the filter
            int act = 0;
            if (SPATIAL) {
                const float4 wb4 = s_win[(ic << 1) + (t0 >> 5)];
                const float w_lx = readlane_f(wb4.x, 0), w_ly = readlane_f(wb4.y, 0), w_hx = readlane_f(wb4.z, 0), w_hy = readlane_f(wb4.w, 0);
#pragma unroll
                for (int m = 0; m < JPT; ++m)
                    if (m >= m0 && w_lx <= sb_hx[m] && sb_lx[m] <= w_hx && w_ly <= sb_hy[m] && sb_ly[m] <= w_hy) act |= 1 << m;
                if (act == 0) continue;
            } else act = ((1 << JPT) - 1) & ~((1 << m0) - 1);
            uint32_t hist[JPT];
#pragma unroll
            for (int m = 0; m < JPT; ++m) hist[m] = 0u;
            auto sweep = [&](auto slots) {
                constexpr int MASK = decltype(slots)::value;
                auto step = [&](const uint32_t rc, const uint32_t rr) {
#pragma unroll
                    for (int m = 0; m < JPT; ++m) {
                        if (!((MASK >> m) & 1)) continue;
                        const uint32_t d = __builtin_amdgcn_sad_u16(rc, cq[m], nj[m]);
                        // hist = 2 * hist + (d <= rr)
                        uint32_t hv = hist[m];
                        asm volatile("v_cmp_le_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                                     : "+v"(hv) : "v"(d), "v"(rr) : "vcc");
                        hist[m] = hv;
                    }
                };
                // four records per trip, the next four already in flight (the same address in every lane: broadcast
                // reads); records beyond n are padding that passes nothing.  Two trips per turn of the loop, so that
                // the two register sets swap roles without moves.
                const uint4 *rp = reinterpret_cast<const uint4 *>(s_irec + c0 + t0);
                uint4 ra = rp[0], rb = rp[1];
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    const uint4 na = rp[2], nb = rp[3];
                    step(ra.x, ra.y);
                    step(ra.z, ra.w);
                    step(rb.x, rb.y);
                    step(rb.z, rb.w);
                    rp += 4;
                    ra = rp[0]; rb = rp[1];
                    step(na.x, na.y);
                    step(na.z, na.w);
                    step(nb.x, nb.y);
                    step(nb.z, nb.w);
                }
            };
            constexpr int ALL = (1 << JPT) - 1;
            switch (act) {
#define DETBOX_SWEEP(K) case K: sweep(std::integral_constant<int, (K) & ALL>{}); break;
                DETBOX_SWEEP(1) DETBOX_SWEEP(2) DETBOX_SWEEP(3) DETBOX_SWEEP(4) DETBOX_SWEEP(5) DETBOX_SWEEP(6) DETBOX_SWEEP(7)
                DETBOX_SWEEP(8) DETBOX_SWEEP(9) DETBOX_SWEEP(10) DETBOX_SWEEP(11) DETBOX_SWEEP(12) DETBOX_SWEEP(13) DETBOX_SWEEP(14)
                default: sweep(std::integral_constant<int, ALL>{}); break;
#undef DETBOX_SWEEP
            }
            // bit 31 - k of a history is box i = c0 + t0 + k.  Inside box i's own chunk only the later boxes count.
            if (intra) {
                const int later = lane - t0;                // boxes c0 + t0 .. c0 + lane - 1 come before this lane's box
                const uint32_t keepbits = later <= 0 ? 0u : (later >= 32 ? 0xffffffffu : (0xffffffffu << (32 - later)));
#pragma unroll
                for (int m = 0; m < JPT; ++m)
                    if (m == m0) hist[m] &= keepbits;
            }
            // every lane appends the pairs of its set bits: one prefix sum over the per-lane counts gives the places
            uint32_t cnt = 0u;
#pragma unroll
            for (int m = 0; m < JPT; ++m) cnt += (uint32_t)__popc(hist[m]);
            if (__ballot(cnt != 0u) == 0ull) continue;
            const uint32_t inc = wave_inclusive_add(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (nq + total > (uint32_t)QCAP) {
                drain();
                ++n_drains;
            }
            if (ne + total > (uint32_t)QCAP) {
                // A window of large boxes against a slot of large boxes can pass more pairs than the queue holds: it is
                // appended one box i at a time (at most 64 * JPT pairs each), drained in between; only true edges can
                // overflow the queue.
                for (int k = 0; k < 32; ++k) {
                    const uint32_t bitm = 0x80000000u >> k;                 // bit 31 - k: box c0 + t0 + k
                    uint32_t ck = 0u;
#pragma unroll
                    for (int m = 0; m < JPT; ++m) ck += (hist[m] & bitm) ? 1u : 0u;
                    if (__ballot(ck != 0u) == 0ull) continue;
                    const uint32_t inck = wave_inclusive_add(ck);
                    const uint32_t totk = (uint32_t)__builtin_amdgcn_readlane((int)inck, 63);
                    if (nq + totk > (uint32_t)QCAP) {
                        drain();
                        ++n_drains;
                        if (ne + totk > (uint32_t)QCAP) { ovf = true; break; }
                    }
                    uint32_t *qk = q + (nq + inck - ck);
#pragma unroll
                    for (int m = 0; m < JPT; ++m)
                        if (hist[m] & bitm) *qk++ = ((uint32_t)(c0 + t0 + k) << 16) | (uint32_t)pos[m];
                    nq += totk;
                    n_passed += totk;
                }
                continue;
            }
            uint32_t *qp = q + (nq + inc - cnt);
#pragma unroll
            for (int m = 0; m < JPT; ++m) {
                uint32_t hm = __brev(hist[m]);              // bit k: box c0 + t0 + k
                const uint32_t tag = ((uint32_t)(c0 + t0) << 16) | (uint32_t)pos[m];
                while (hm != 0u) {
                    const uint32_t bit = (uint32_t)__builtin_ctz(hm);
                    hm &= hm - 1u;
                    *qp++ = (bit << 16) + tag;
                }
            }
            nq += total;
            n_passed += total;
        }
    }
    if (!ovf) { drain(); ++n_drains; }
    if (lane == 0) {
        s_ne[wv] = ne;
        if (ovf) *s_ovf = 1u;
        if (np.stats) {
            atomicAdd(&g_pair_stats[2], (unsigned long long)n_passed);
            atomicAdd(&g_pair_stats[3], (unsigned long long)ne);
            atomicAdd(&g_pair_stats[4], (unsigned long long)n_drains);
            if (wv == 0) atomicAdd(&g_pair_stats[0], 1ull);
        }
    }
    __syncthreads();
    if (*s_ovf) {
        if (np.stats && tid == 0) atomicAdd(&g_pair_stats[1], 1ull);
        return false;
    }

    // ---- 4. resolve in score order
    const long long t_swept = np.stats ? (long long)__builtin_readcyclecounter() : 0ll;
    uint32_t nev[W];
#pragma unroll
    for (int w = 0; w < W; ++w) nev[w] = s_ne[w];
    // Box j goes iff some kept earlier box i has an edge to it.  Iterating  removed' = { j : some edge (i, j) with i not in
    // removed }  from the empty set settles every box whose longest chain of edges has k links after k rounds, and a round
    // that changes nothing is the greedy result (the equation has one solution: by induction over the score order).  Sparse
    // edge lists settle in a handful of rounds of one pass over the edges each; the rest take the chunk-by-chunk resolve.
    bool settled = false;
    for (int it = 0; it < PAIR_RESOLVE_ROUNDS; ++it) {
        if (tid < words) s_mask[tid] = 0ull;
        __syncthreads();
        if (tid == 0) s_flag[(it + 1) & 1] = 0u;           // everybody has read it (the round before) by now
#pragma unroll
        for (int w = 0; w < W; ++w)
            for (uint32_t e = tid; e < nev[w]; e += BLK) {
                const uint32_t en = s_queue[w * QCAP + e];
                const int i = (int)(en >> 16), j = (int)(en & 0xffffu);
                if (!((s_removed[i >> 6] >> (i & 63)) & 1ull)) atomicOr(&s_mask[j >> 6], 1ull << (j & 63));
            }
        __syncthreads();
        if (tid < words) {
            const unsigned long long next = s_mask[tid];
            if (next != s_removed[tid]) s_flag[it & 1] = 1u;
            s_removed[tid] = next;
        }
        __syncthreads();
        if (s_flag[it & 1] == 0u) { settled = true; break; }
    }
    if (!settled) {
        for (int i = tid; i < words; i += BLK) s_removed[i] = 0ull;
        if (tid < 64) s_mask[tid] = 0ull;
        __syncthreads();
        for (int ch = 0; ch < words; ++ch) {
            const int c0 = ch << 6;
            const int cn = min(64, n - c0);
            const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
#pragma unroll
            for (int w = 0; w < W; ++w)
                for (uint32_t e = tid; e < nev[w]; e += BLK) {
                    const uint32_t en = s_queue[w * QCAP + e];
                    const int i = (int)(en >> 16), j = (int)(en & 0xffffu);
                    if ((j >> 6) != ch) continue;
                    if (i >= c0) atomicOr(&s_mask[i - c0], 1ull << (j - c0));
                    else if (!((s_removed[i >> 6] >> (i & 63)) & 1ull)) atomicOr(&s_removed[ch], 1ull << (j & 63));
                }
            __syncthreads();
            if (wv == 0) {
                const unsigned long long m = s_mask[lane];
                s_mask[lane] = 0ull;
                unsigned long long rows = __ballot(m != 0ull);
                unsigned long long keep = uniform64(~s_removed[ch] & valid);
                while (rows) {                                   // only rows that suppress anything matter, in order
                    const int i = __ffsll((long long)rows) - 1;
                    rows &= rows - 1ull;
                    if ((keep >> i) & 1ull) keep &= ~readlane64(m, i);
                }
                if (lane == 0) s_removed[ch] = valid & ~keep;
            }
            __syncthreads();
        }
    }
    // kept = not removed; compact in sorted order
    if (wv == 0) {
        uint32_t running = 0;
        for (int base = 0; base < words; base += 64) {
            const int w = base + lane;
            uint32_t pc = 0;
            if (w < words) {
                const int cn = min(64, n - (w << 6));
                const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
                pc = (uint32_t)__popcll(~s_removed[w] & valid);
            }
            const uint32_t inc = wave_inclusive_add(pc);
            if (w < words) s_wprefix[w] = running + inc - pc;
            running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (lane == 0) kcount[(size_t)b * np.S + s] = (int32_t)running;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < JPT; ++m) {
        const int i = tid + m * BLK;
        if (i >= n) continue;
        const unsigned long long rem = s_removed[i >> 6];
        if ((rem >> (i & 63)) & 1ull) continue;
        const uint32_t rank = s_wprefix[i >> 6] + (uint32_t)__popcll(~rem & ((1ull << (i & 63)) - 1ull));
        const int p = seg0 + orig[m];
        kept[row0 + rank] = make_float2(__uint_as_float(okey_from_bits(__float_as_uint(scores[(size_t)b * np.stride + p] + 0.0f))), __int_as_float(p));
    }
    if (np.stats && tid == 0) {                            // cycles of wave 0 per phase: sort + records, all pairs, resolve + output
        atomicAdd(&g_pair_stats[5], (unsigned long long)(t_sorted - t_begin));
        atomicAdd(&g_pair_stats[6], (unsigned long long)(t_swept - t_sorted));
        atomicAdd(&g_pair_stats[7], (unsigned long long)((long long)__builtin_readcyclecounter() - t_swept));
    }
    return true;
}

template <int JPT, int BLK>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == 256 ? 5 : 1))) void nms_segments_pairs_kernel(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
    NmsParams np, float2 *__restrict__ kept, int32_t *__restrict__ kcount) {
    extern __shared__ unsigned long long s_dyn[];
    if (nms_pairs_body<JPT, BLK>(boxes, scores, seg_off, np, kept, kcount, s_dyn)) return;
    __syncthreads();
    nms_chunked_body<JPT, BLK>(boxes, scores, seg_off, np, kept, kcount, s_dyn);      // heavily clustered segment
}

// Exclusive scan of the kept counts of an image's segments; a negative count marks a segment the NMS
// kernel refused (longer than max_segment) and turns the image's keep count into -1.
__global__ __launch_bounds__(64) void nms_scan_kernel(const int32_t *__restrict__ kcount, int S,
                                                      int32_t *__restrict__ koff, int32_t *__restrict__ keep_counts) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t *kc = kcount + (size_t)b * S;
    int32_t *ko = koff + (size_t)b * (S + 1);
    int acc = 0, bad = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {              // one wave: 64 counts a trip, prefix sums by shuffles
        const int s = s0 + lane;
        int c = s < S ? kc[s] : 0;
        if (c < 0) { bad = 1; c = 0; }
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (s < S) ko[s] = acc + inc - c;
        acc += __shfl(inc, 63);
    }
    bad = __any(bad);
    if (lane == 0) {
        ko[S] = acc;
        keep_counts[b] = bad ? -1 : acc;
    }
}

// One stable merge level: neighbouring runs (each `group` class segments of one image) are merged
// pairwise.  A run element is (ordered score key, row index); the left run wins ties, which keeps equal
// scores in ascending row order.  Every workgroup produces one tile of MP_TILE outputs of one pair: two
// merge-path searches in global memory bound the tile's inputs, which are staged in LDS; every thread
// then finds its own starting point by a merge-path search in LDS and merges MP_VT outputs serially.
constexpr int MP_THREADS = 256;
constexpr int MP_VT = 8;
constexpr int MP_TILE = MP_THREADS * MP_VT;

__device__ inline uint32_t run_key(const float2 v) { return __float_as_uint(v.x); }

__global__ __launch_bounds__(MP_THREADS) void nms_merge_kernel(
    const float2 *__restrict__ src, float2 *__restrict__ dst, int64_t *__restrict__ keep_out,
    const int32_t *__restrict__ koff, const int32_t *__restrict__ seg_off, int S, long long stride, int group,
    int gapped, int last) {
    __shared__ float2 s_in[MP_TILE];
    __shared__ int s_split[2];
    const int b = blockIdx.z, pair = blockIdx.y, tile = blockIdx.x;
    const int32_t *ko = koff + (size_t)b * (S + 1);
    const int32_t *so = seg_off + (size_t)b * (S + 1);
    const int sA = pair * 2 * group;
    if (sA >= S) return;
    const int sB = min(S, sA + group), sE = min(S, sA + 2 * group);
    const int nA = ko[sB] - ko[sA], nB = ko[sE] - ko[sB];
    const int total = nA + nB;
    const int d0 = tile * MP_TILE;
    if (d0 >= total) return;
    const int d1 = min(d0 + MP_TILE, total);
    const size_t base = (size_t)b * stride;
    // gapped (first level only, group == 1): the kept run of segment x starts at seg_off[x]
    const float2 *A = src + base + (gapped ? so[sA] : ko[sA]);
    const float2 *Bp = src + base + (gapped ? so[sB] : ko[sB]);
    const int tid = threadIdx.x;
    if (tid < 128) {        // waves 0 and 1: number of left-run elements among the first d0 / d1 outputs
        const int d = tid >= 64 ? d1 : d0;
        const int lane = tid & 63;
        int lo = max(0, d - nB), hi = min(d, nA);
        // 64-ary search: the predicate "A[p] goes before B[d-1-p]" holds for a prefix of [lo, hi)
        while (hi - lo > 64) {
            const int step = (hi - lo + 63) >> 6;
            const int p = lo + (lane + 1) * step - 1;
            const bool before = p < hi && run_key(A[p]) >= run_key(Bp[d - 1 - p]);
            const int c = __popcll(__ballot(before));
            hi = min(hi, lo + (c + 1) * step - 1);
            lo = lo + c * step;
        }
        const int p = lo + lane;
        const bool before = p < hi && run_key(A[p]) >= run_key(Bp[d - 1 - p]);
        lo += __popcll(__ballot(before));
        if (lane == 0) s_split[tid >> 6] = lo;
    }
    __syncthreads();
    const int a0 = s_split[0], a1 = s_split[1];
    const int b0 = d0 - a0;
    const int na = a1 - a0, nb = (d1 - d0) - na;
    const int nt = na + nb;
    {   // all the loads of a thread are issued before the first one is needed
        float2 v[MP_VT];
#pragma unroll
        for (int q = 0; q < MP_VT; ++q) {
            const int i = tid + q * MP_THREADS;
            v[q] = make_float2(0.f, 0.f);
            if (i < nt) v[q] = i < na ? A[a0 + i] : Bp[b0 + (i - na)];
        }
#pragma unroll
        for (int q = 0; q < MP_VT; ++q) {
            const int i = tid + q * MP_THREADS;
            if (i < nt) s_in[i] = v[q];
        }
    }
    __syncthreads();
    const int t0 = min(tid * MP_VT, nt), t1 = min(t0 + MP_VT, nt);
    int lo = max(0, t0 - nb), hi = min(t0, na);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (run_key(s_in[mid]) >= run_key(s_in[na + t0 - 1 - mid])) lo = mid + 1; else hi = mid;
    }
    int ia = lo, ib = t0 - lo;
    const size_t out0 = base + (last ? 0 : ko[sA]) + d0;
    for (int o = t0; o < t1; ++o) {
        bool take_a = ib >= nb;
        if (!take_a && ia < na) take_a = run_key(s_in[ia]) >= run_key(s_in[na + ib]);
        const float2 v = take_a ? s_in[ia] : s_in[na + ib];
        ia += take_a ? 1 : 0;
        ib += take_a ? 0 : 1;
        if (last) keep_out[out0 + o] = (int64_t)__float_as_int(v.y);
        else dst[out0 + o] = v;
    }
}

// The first merge levels in one launch: a workgroup loads the kept runs of F consecutive class segments
// of one image (F a power of two, F * max_segment elements fit the LDS buffer), merges them pairwise
// log2(F) times between two LDS buffers -- every thread produces a contiguous slice of the output by a
// merge-path search and a serial merge -- and writes one dense run (or, if F covers all segments, the
// final keep list).
constexpr int ML_THREADS = 1024;
constexpr int ML_CAP = 8192;               // elements per LDS buffer (two buffers of 64 KiB)

constexpr int ML_SCAN_MAX = 2048;          // segments per image up to which the kernel scans the kept counts itself

__global__ __launch_bounds__(ML_THREADS) void nms_merge_lds_kernel(
    const float2 *__restrict__ src, float2 *__restrict__ dst, int64_t *__restrict__ keep_out,
    int32_t *__restrict__ koff, const int32_t *__restrict__ seg_off, int S, long long stride, int F, int last,
    const int32_t *__restrict__ kcount, int32_t *__restrict__ keep_counts) {
    extern __shared__ float2 s_ml[];
    __shared__ int s_off[65];                // run boundaries inside the buffer (F <= 64)
    __shared__ int s_koff[ML_SCAN_MAX + 1];
    __shared__ int s_wsum[ML_THREADS / 64];
    __shared__ int s_bad;
    float2 *cur = s_ml, *nxt = s_ml + ML_CAP;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int s0 = blockIdx.x * F;
    if (s0 >= S) return;
    const int s1 = min(S, s0 + F);
    const int32_t *ko = koff + (size_t)b * (S + 1);
    if (kcount) {
        // The exclusive scan of the image's kept counts (what nms_scan_kernel does), by every workgroup for itself; the
        // first one also leaves it in global memory for the merge levels and the packing that follow.  A negative count
        // marks a segment the NMS kernel refused: it counts as empty and turns the image's keep count into -1.
        const int32_t *kc = kcount + (size_t)b * S;
        const int e0 = 2 * tid, lane = tid & 63, wave = tid >> 6;
        int c0 = e0 < S ? kc[e0] : 0, c1 = e0 + 1 < S ? kc[e0 + 1] : 0;
        if (tid == 0) s_bad = 0;
        __syncthreads();
        if (c0 < 0 || c1 < 0) s_bad = 1;
        c0 = max(c0, 0); c1 = max(c1, 0);
        int inc = c0 + c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int base = inc - (c0 + c1);
        for (int w = 0; w < wave; ++w) base += s_wsum[w];
        if (e0 < S) s_koff[e0] = base;
        if (e0 + 1 < S) s_koff[e0 + 1] = base + c0;
        if (e0 < S && e0 + 2 >= S) s_koff[S] = base + c0 + c1;
        __syncthreads();
        if (blockIdx.x == 0) {
            int32_t *kog = koff + (size_t)b * (S + 1);
            for (int i = tid; i <= S; i += ML_THREADS) kog[i] = s_koff[i];
            if (tid == 0) keep_counts[b] = s_bad ? -1 : s_koff[S];
        }
        ko = s_koff;
    }
    const int32_t *so = seg_off + (size_t)b * (S + 1);
    const size_t base = (size_t)b * stride;
    const int out0 = ko[s0];
    const int total = ko[s1] - out0;
    __shared__ int s_src[64];                // where run r starts in src, minus its offset in the buffer
    if (tid <= F) s_off[tid] = ko[min(s1, s0 + tid)] - out0;
    if (tid < F) s_src[tid] = so[min(S, s0 + tid)] - (ko[min(s1, s0 + tid)] - out0);
    __syncthreads();
    for (int e = tid; e < total; e += ML_THREADS) {      // independent loads: the runs sit at their segment starts
        int r = 0;                                        // the run that holds element e: largest r with s_off[r] <= e
        for (int step = 32; step > 0; step >>= 1)
            if (r + step < F && s_off[r + step] <= e) r += step;
        cur[e] = src[base + s_src[r] + e];
    }
    __syncthreads();
    const int per = (total + ML_THREADS - 1) / ML_THREADS;
    for (int g = 1; g < F; g <<= 1) {         // runs of g segments -> runs of 2g segments
        int o = min(tid * per, total);
        const int o_end = min(o + per, total);
        while (o < o_end) {
            int pr = 0;                        // pair containing output position o: largest pr with s_off[pr*2g] <= o
            for (int step = 32; step > 0; step >>= 1)
                if ((pr + step) * 2 * g < F && s_off[(pr + step) * 2 * g] <= o) pr += step;
            const int a_lo = s_off[min(F, pr * 2 * g)], a_hi = s_off[min(F, pr * 2 * g + g)];
            const int b_hi = s_off[min(F, (pr + 1) * 2 * g)];
            const int na = a_hi - a_lo, nb = b_hi - a_hi;
            const float2 *A = cur + a_lo, *Bp = cur + a_hi;
            const int d = o - a_lo;            // diagonal inside the pair
            int lo = max(0, d - nb), hi = min(d, na);
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (run_key(A[mid]) >= run_key(Bp[d - 1 - mid])) lo = mid + 1; else hi = mid;
            }
            int ia = lo, ib = d - lo;
            const int stop = min(o_end, b_hi);
            for (; o < stop; ++o) {
                bool take_a = ib >= nb;
                if (!take_a && ia < na) take_a = run_key(A[ia]) >= run_key(Bp[ib]);
                nxt[o] = take_a ? A[ia] : Bp[ib];
                ia += take_a ? 1 : 0;
                ib += take_a ? 0 : 1;
            }
        }
        __syncthreads();
        float2 *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (last) {
        for (int i = tid; i < total; i += ML_THREADS) keep_out[base + i] = (int64_t)__float_as_int(cur[i].y);
    } else {
        for (int i = tid; i < total; i += ML_THREADS) dst[base + out0 + i] = cur[i];
    }
}

// WIRE: the same rows also leave in the wire form described at pack_wire_kernel (one gather feeds both).
template <bool WIRE>
__global__ __launch_bounds__(256) void pack_detections_kernel(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int64_t *__restrict__ cls,
    const int64_t *__restrict__ keep, const int32_t *__restrict__ keep_counts,
    const int64_t *__restrict__ image_ids, const int32_t *__restrict__ image_sizes, int B, long long stride,
    int64_t *__restrict__ rows_ids, double *__restrict__ rows_vals, int32_t *__restrict__ row_offsets,
    int hdr_rows, uint32_t *__restrict__ wire) {
    __shared__ int s_red[256];
    const TileOfImage wg = xcd_affine_tile();            // the gathers of one image stay in one L2
    const int b = wg.y, bx_ = wg.x;
    if (bx_ != 0 && (int)(bx_ * blockDim.x) >= keep_counts[b]) return;     // no rows here
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) acc += max(keep_counts[i], 0);
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    const int row0 = s_red[0];
    const int n = max(keep_counts[b], 0);
    if (bx_ == 0 && threadIdx.x == 0) {
        row_offsets[b] = row0;
        if (b == B - 1) row_offsets[B] = row0 + n;
        if (WIRE) {
            wire[2 + b] = (uint32_t)row0;
            const unsigned long long id = (unsigned long long)image_ids[b];
            wire[2 + (B + 1) + 2 * b] = (uint32_t)id;
            wire[2 + (B + 1) + 2 * b + 1] = (uint32_t)(id >> 32);
            wire[2 + (B + 1) + 2 * B + 2 * b] = (uint32_t)image_sizes[2 * b];
            wire[2 + (B + 1) + 2 * B + 2 * b + 1] = (uint32_t)image_sizes[2 * b + 1];
            if (b == B - 1) { wire[0] = (uint32_t)(row0 + n); wire[1] = (uint32_t)B; wire[2 + B] = (uint32_t)(row0 + n); }
        }
    }
    // rows of this block are built in LDS and leave as contiguous 8-byte streams
    __shared__ double s_vals[256 * 5];
    __shared__ int64_t s_ids[256 * 2];
    __shared__ uint32_t s_wire[WIRE ? 256 * 6 : 1];
    const int t0 = bx_ * blockDim.x;
    if (t0 >= n) return;
    const int nb = min(256, n - t0);
    const int t = t0 + threadIdx.x;
    if (t < n) {
        const size_t base = (size_t)b * stride;
        const int64_t k = keep[base + t];
        const float4 bx = boxes[base + k];
        const double W = (double)image_sizes[2 * b + 1], H = (double)image_sizes[2 * b];
        const int r = threadIdx.x;
        s_ids[2 * r] = image_ids[b];
        s_ids[2 * r + 1] = cls[base + k];
        s_vals[5 * r + 0] = (double)bx.x * W;
        s_vals[5 * r + 1] = (double)bx.y * H;
        s_vals[5 * r + 2] = (double)(bx.z - bx.x) * W;
        s_vals[5 * r + 3] = (double)(bx.w - bx.y) * H;
        const float sc = scores[base + k];
        s_vals[5 * r + 4] = (double)sc;
        if (WIRE) {
            uint32_t *w = s_wire + 6 * r;
            w[0] = (uint32_t)s_ids[2 * r + 1];
            w[1] = __float_as_uint(bx.x); w[2] = __float_as_uint(bx.y);
            w[3] = __float_as_uint(bx.z); w[4] = __float_as_uint(bx.w);
            w[5] = __float_as_uint(sc);
        }
    }
    __syncthreads();
    const size_t r0 = (size_t)row0 + t0;
    for (int i = threadIdx.x; i < 5 * nb; i += 256) store_nt(&rows_vals[5 * r0 + i], s_vals[i]);
    for (int i = threadIdx.x; i < 2 * nb; i += 256) store_nt(&rows_ids[2 * r0 + i], s_ids[i]);
    if (WIRE) {
        uint32_t *out = wire + 6 * ((size_t)hdr_rows + r0);
        for (int i = threadIdx.x; i < 6 * nb; i += 256) store_nt(&out[i], s_wire[i]);
    }
}

// Wire form of the kept detections for the exchange between ranks: 32-bit words, six per row
// (class id, x1, y1, x2, y2, score as fp32 bits), behind a header that makes the message self-describing:
//   word 0: rows n, word 1: images B, then row_offsets[B+1], then the image ids (low, high word), then the
//   image sizes (H, W); the rows start at word 6 * hdr_rows.
// The float64 products of the result rows are a function of these words and the sizes, so a receiver
// rebuilds exactly the rows pack_detections writes.
__global__ __launch_bounds__(256) void pack_wire_kernel(
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int64_t *__restrict__ cls,
    const int64_t *__restrict__ keep, const int32_t *__restrict__ keep_counts,
    const int64_t *__restrict__ image_ids, const int32_t *__restrict__ image_sizes, int B, long long stride,
    int hdr_rows, uint32_t *__restrict__ wire) {
    __shared__ int s_red[256];
    __shared__ uint32_t s_rows[256 * 6];
    const TileOfImage wg = xcd_affine_tile();
    const int b = wg.y, bx_ = wg.x;
    const int n = max(keep_counts[b], 0);
    if (bx_ != 0 && (int)(bx_ * blockDim.x) >= n) return;
    int acc = 0;
    for (int i = threadIdx.x; i < b; i += blockDim.x) acc += max(keep_counts[i], 0);
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    const int row0 = s_red[0];
    if (bx_ == 0 && threadIdx.x == 0) {
        wire[2 + b] = (uint32_t)row0;
        const unsigned long long id = (unsigned long long)image_ids[b];
        wire[2 + (B + 1) + 2 * b] = (uint32_t)id;
        wire[2 + (B + 1) + 2 * b + 1] = (uint32_t)(id >> 32);
        wire[2 + (B + 1) + 2 * B + 2 * b] = (uint32_t)image_sizes[2 * b];
        wire[2 + (B + 1) + 2 * B + 2 * b + 1] = (uint32_t)image_sizes[2 * b + 1];
        if (b == B - 1) { wire[0] = (uint32_t)(row0 + n); wire[1] = (uint32_t)B; wire[2 + B] = (uint32_t)(row0 + n); }
    }
    const int t0 = bx_ * blockDim.x;
    if (t0 >= n) return;
    const int nb = min(256, n - t0);
    const int t = t0 + threadIdx.x;
    if (t < n) {
        const size_t base = (size_t)b * stride;
        const int64_t k = keep[base + t];
        const float4 bx = boxes[base + k];
        uint32_t *r = s_rows + 6 * threadIdx.x;
        r[0] = (uint32_t)cls[base + k];
        r[1] = __float_as_uint(bx.x); r[2] = __float_as_uint(bx.y);
        r[3] = __float_as_uint(bx.z); r[4] = __float_as_uint(bx.w);
        r[5] = __float_as_uint(scores[base + k]);
    }
    __syncthreads();
    uint32_t *out = wire + 6 * ((size_t)hdr_rows + row0 + t0);
    for (int i = threadIdx.x; i < 6 * nb; i += 256) out[i] = s_rows[i];
}

// DETBOX_NMS_CHUNKED=1 (measurements, tests): always the chunked kernel, never the pair-list one
const bool g_nms_chunked_only = std::getenv("DETBOX_NMS_CHUNKED") != nullptr;
// DETBOX_NMS_NETWORK_SORT=1 (measurements, tests): the pair-list kernel always orders a segment with the bitonic network
const bool g_nms_network_sort = std::getenv("DETBOX_NMS_NETWORK_SORT") != nullptr;
// DETBOX_NMS_STATS=1: the pair-list kernel adds to the counters detbox_debug_nms_stats reads (a few atomics per wave)
bool g_nms_stats = std::getenv("DETBOX_NMS_STATS") != nullptr;     // also switched by detbox_debug_nms_stats_enable

struct NmsLayout { size_t kept_a, kept_b, kcount, koff, merge, total; };

// DETBOX_NMS_LEVEL_MERGE=1 (measurements, tests): always the level-by-level merge, never the tile merge
bool g_nms_level_merge = std::getenv("DETBOX_NMS_LEVEL_MERGE") != nullptr;     // also switched by detbox_debug_nms_level_merge

MergePlan nms_merge_plan(int64_t B, int64_t stride, int64_t S) {
    MergePlan p = merge_plan(B, stride, S);
    if (g_nms_level_merge) p.ok = false;
    return p;
}

// The tile merge reads the kept runs once and needs no second run buffer; the level-by-level merge ping-pongs.
NmsLayout nms_layout(int64_t B, int64_t stride, int64_t S) {
    NmsLayout L;
    const MergePlan p = nms_merge_plan(B, stride, S);
    size_t off = 0;
    L.kept_a = off; off = align_up(off + (size_t)B * stride * sizeof(float2), 256);
    L.kept_b = off; off = align_up(off + (p.ok ? 0 : (size_t)B * stride * sizeof(float2)), 256);
    L.kcount = off; off = align_up(off + (size_t)B * S * sizeof(int32_t), 256);
    L.koff = off; off = align_up(off + (size_t)B * (S + 1) * sizeof(int32_t), 256);
    L.merge = off; off = align_up(off + (p.ok ? merge_workspace_bytes(p, B, S) : 0), 256);
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t detbox_batched_nms_segments_workspace_bytes(int64_t B, int64_t stride, int64_t S) {
    if (B <= 0 || stride <= 0 || S <= 0) return 256;
    return nms_layout(B, stride, S).total;
}

namespace {
// NMS of every segment, then the order step.  rows == null: the keep lists only.  rows != null: the result rows as
// well (keep may then be null); *rows_done tells the caller whether they were written here (tile merge) or are still
// to be packed from the keep lists (level-by-level merge).
int nms_run(const float *boxes, const float *scores, const int32_t *seg_offsets,
            int64_t B, int64_t stride, int64_t S, int64_t max_segment, double iou_threshold, int presorted,
            int64_t *keep, int32_t *keep_counts, const RowsOut *rows, bool *rows_done,
            void *workspace, size_t workspace_bytes, void *stream) {
    if (rows_done) *rows_done = false;
    DETBOX_CHECK_ARG(B >= 0 && stride >= 0 && S >= 0, "negative size");
    DETBOX_CHECK_ARG(!(iou_threshold != iou_threshold), "iou_threshold is NaN");
    DETBOX_CHECK_ARG(B <= 65535 && S <= (1 << 30) && stride < (1ll << 31), "size too large");
    if (B == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(keep_counts, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stride == 0 || S == 0) {
        DETBOX_HIP(hipMemsetAsync(keep_counts, 0, (size_t)B * sizeof(int32_t), st));
        return DETBOX_OK;
    }
    DETBOX_CHECK_ARG(boxes && scores && seg_offsets && (keep || rows), "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)boxes % 16 == 0, "boxes must be 16-byte aligned");
    if (max_segment <= 0 || max_segment > stride) max_segment = stride;
    if (!presorted && max_segment > DETBOX_NMS_MAX_SORT) max_segment = DETBOX_NMS_MAX_SORT;
    if (max_segment > NMS_MAX_SEGMENT) {
        set_error("detbox_batched_nms_segments: segments longer than %d are not supported", NMS_MAX_SEGMENT);
        return DETBOX_ERR_UNSUPPORTED;
    }
    const NmsLayout L = nms_layout(B, stride, S);
    if (!workspace || (uintptr_t)workspace % 256 || workspace_bytes < L.total) {
        set_error("detbox_batched_nms_segments: workspace missing, misaligned or smaller than %zu bytes", L.total);
        return DETBOX_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(workspace);
    float2 *kept_a = reinterpret_cast<float2 *>(ws + L.kept_a);
    float2 *kept_b = reinterpret_cast<float2 *>(ws + L.kept_b);
    int32_t *kcount = reinterpret_cast<int32_t *>(ws + L.kcount);
    int32_t *koff = reinterpret_cast<int32_t *>(ws + L.koff);

    NmsParams np;
    np.S = (int)S; np.max_seg = (int)max_segment; np.presorted = presorted ? 1 : 0;
    np.stride = stride;
    np.in_lds = max_segment <= NMS_LDS_BOXES ? 1 : 0;
    int n2 = 2;
    while (n2 < max_segment) n2 <<= 1;
    np.n2 = n2;
    // largest fp32 <= threshold: (double)x > threshold  <=>  x > thr for every fp32 x
    float t32 = (float)iou_threshold;
    if ((double)t32 > iou_threshold) t32 = std::nextafterf(t32, -INFINITY);
    np.thr = t32;
    np.thr_neg = iou_threshold < 0.0 ? 1 : 0;
    np.stats = g_nms_stats ? 1 : 0;
    np.network_sort = g_nms_network_sort ? 1 : 0;
    const bool fast = !np.presorted && !np.thr_neg && max_segment <= DETBOX_NMS_MAX_SORT;
    {
        ProfScope ps("nms_segments", st);
        dim3 grid((unsigned)S, (unsigned)B);
        const float4 *b4 = reinterpret_cast<const float4 *>(boxes);
        if (fast) {
            // one wave per segment pays off once there are enough segments to fill the chip with single waves;
            // with few segments the four-wave kernel finishes each of them sooner
            const int blk = (max_segment <= 256 && B * S >= 4096) ? 64 : 256;
            const size_t lds = nms_fast_lds(np.max_seg, np.n2, blk).total;
#define DETBOX_NMS_FAST(J, BLKC)                                                                                 \
    {                                                                                                            \
        if (lds > 64 * 1024)                                                                                     \
            DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_segments_fast_kernel<J, BLKC>),    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        nms_segments_fast_kernel<J, BLKC><<<grid, dim3(BLKC), lds, st>>>(b4, scores, seg_offsets, np, kept_a, kcount); \
    }
            const int jpt = (int)((max_segment + blk - 1) / blk);
            // up to four slots per wave: the pair-list kernel (chunked algorithm inside it as the fall-back)
            const size_t lds_p0 = nms_pair_lds(np.max_seg, np.n2, blk).total;
            const size_t lds_p = lds_p0 > lds ? lds_p0 : lds;
#define DETBOX_NMS_PAIRS(J, BLKC)                                                                                \
    {                                                                                                            \
        if (lds_p > 64 * 1024)                                                                                   \
            DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_segments_pairs_kernel<J, BLKC>),   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));             \
        nms_segments_pairs_kernel<J, BLKC><<<grid, dim3(BLKC), lds_p, st>>>(b4, scores, seg_offsets, np, kept_a, kcount); \
    }
            if (g_nms_chunked_only) {
                if (blk == 64) {
                    if (jpt <= 1) DETBOX_NMS_FAST(1, 64)
                    else if (jpt <= 2) DETBOX_NMS_FAST(2, 64)
                    else DETBOX_NMS_FAST(4, 64)
                } else if (jpt <= 1) DETBOX_NMS_FAST(1, 256)
                else if (jpt <= 2) DETBOX_NMS_FAST(2, 256)
                else if (jpt <= 4) DETBOX_NMS_FAST(4, 256)
                else if (jpt <= 8) DETBOX_NMS_FAST(8, 256)
                else DETBOX_NMS_FAST(16, 256)
            } else if (blk == 64) {
                if (jpt <= 1) DETBOX_NMS_PAIRS(1, 64)
                else if (jpt <= 2) DETBOX_NMS_PAIRS(2, 64)
                else DETBOX_NMS_PAIRS(4, 64)
            } else if (jpt <= 1) DETBOX_NMS_PAIRS(1, 256)
            else if (jpt <= 2) DETBOX_NMS_PAIRS(2, 256)
            else if (jpt <= 4) DETBOX_NMS_PAIRS(4, 256)
            else if (jpt <= 8) DETBOX_NMS_FAST(8, 256)
            else DETBOX_NMS_FAST(16, 256)
#undef DETBOX_NMS_PAIRS
#undef DETBOX_NMS_FAST
        } else {
            const size_t lds = nms_lds(np.max_seg, np.n2, np.presorted, np.in_lds).total;
            if (lds > 64 * 1024)
                DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_segments_generic_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            nms_segments_generic_kernel<<<grid, dim3(NMS_BLOCK), lds, st>>>(b4, scores, seg_offsets, np, kept_a, kcount);
        }
        DETBOX_LAUNCH_CHECK();
    }
    const MergePlan plan = nms_merge_plan(B, stride, S);
    if (plan.ok) {
        if (rows_done) *rows_done = rows != nullptr;
        return merge_rows_launch(plan, kept_a, kcount, seg_offsets, B, stride, S, keep, keep_counts, rows, ws + L.merge, st);
    }
    DETBOX_CHECK_ARG(keep, "null pointer");
    long long F = 1;                       // as many merge levels as fit in LDS at once
    while (F * 2 * max_segment <= ML_CAP && F < 64 && F < S) F *= 2;
    const bool scan_in_merge = F > 1 && S <= ML_SCAN_MAX;
    if (!scan_in_merge) {
        ProfScope ps("nms_scan", st);
        nms_scan_kernel<<<dim3((unsigned)B), dim3(64), 0, st>>>(kcount, (int)S, koff, keep_counts);
        DETBOX_LAUNCH_CHECK();
    }
    float2 *src = kept_a, *dst = kept_b;
    long long group = 1;
    {
        if (F > 1) {
            const int last = F >= S ? 1 : 0;
            const size_t lds = 2 * (size_t)ML_CAP * sizeof(float2);
            DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_merge_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ProfScope ps("nms_merge", st);
            nms_merge_lds_kernel<<<dim3((unsigned)((S + F - 1) / F), (unsigned)B), dim3(ML_THREADS), lds, st>>>(
                src, dst, keep, koff, seg_offsets, (int)S, stride, (int)F, last, scan_in_merge ? kcount : nullptr, keep_counts);
            DETBOX_LAUNCH_CHECK();
            float2 *tmp = src; src = dst; dst = tmp;
            group = F;
        }
    }
    while (group < S || group == 1) {
        const long long runs = (S + group - 1) / group;
        const long long pairs = (runs + 1) / 2;
        long long longest = 2 * group * max_segment;
        if (longest > stride) longest = stride;
        const int last = pairs == 1 ? 1 : 0;
        ProfScope ps("nms_merge", st);
        nms_merge_kernel<<<dim3((unsigned)((longest + MP_TILE - 1) / MP_TILE), (unsigned)pairs, (unsigned)B), dim3(MP_THREADS), 0,
                           st>>>(src, dst, keep, koff, seg_offsets, (int)S, stride, (int)group, group == 1 ? 1 : 0, last);
        DETBOX_LAUNCH_CHECK();
        float2 *tmp = src; src = dst; dst = tmp;
        group *= 2;
    }
    return DETBOX_OK;
}
}  // namespace

extern "C" int detbox_batched_nms_segments(const float *boxes, const float *scores, const int32_t *seg_offsets,
                                           int64_t B, int64_t stride, int64_t S, int64_t max_segment,
                                           double iou_threshold, int presorted,
                                           int64_t *keep, int32_t *keep_counts,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    DETBOX_CHECK_ARG(keep || B == 0 || stride == 0 || S == 0, "null pointer");
    return nms_run(boxes, scores, seg_offsets, B, stride, S, max_segment, iou_threshold, presorted, keep, keep_counts,
                   nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int detbox_debug_nms_level_merge(int on) {
    g_nms_level_merge = on != 0;
    return DETBOX_OK;
}

extern "C" int detbox_debug_nms_stats_enable(int on) {
    g_nms_stats = on != 0;
    return DETBOX_OK;
}

extern "C" int detbox_debug_nms_stats(uint64_t *out8, int reset) {
    DETBOX_CHECK_ARG(out8, "null pointer");
    unsigned long long host[8];
    DETBOX_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pair_stats), sizeof(host)));
    for (int i = 0; i < 8; ++i) out8[i] = host[i];
    if (reset) {
        for (int i = 0; i < 8; ++i) host[i] = 0ull;
        DETBOX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stats), host, sizeof(host)));
    }
    return DETBOX_OK;
}

extern "C" int64_t detbox_wire_header_rows(int64_t images_cap);

namespace {
int pack_rows(const float *boxes, const float *scores, const int64_t *class_ids, const int64_t *keep,
              const int32_t *keep_counts, const int64_t *image_ids, const int32_t *image_sizes, int64_t B, int64_t stride,
              int64_t *rows_ids, double *rows_vals, int32_t *row_offsets, int64_t images_cap, uint32_t *wire, void *stream) {
    DETBOX_CHECK_ARG(B >= 0 && stride >= 0, "negative size");
    DETBOX_CHECK_ARG(B <= 65535 && stride < (1ll << 31), "size too large");
    DETBOX_CHECK_ARG(row_offsets, "null pointer");
    DETBOX_CHECK_ARG(!wire || images_cap >= B, "images_cap smaller than the number of images");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B == 0 || stride == 0) {
        DETBOX_HIP(hipMemsetAsync(row_offsets, 0, (size_t)(B + 1) * sizeof(int32_t), st));
        if (wire) DETBOX_HIP(hipMemsetAsync(wire, 0, (size_t)detbox_wire_header_rows(images_cap) * 6 * sizeof(uint32_t), st));
        return DETBOX_OK;
    }
    DETBOX_CHECK_ARG(boxes && scores && class_ids && keep && keep_counts && image_ids && image_sizes && rows_ids && rows_vals,
                     "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)boxes % 16 == 0, "boxes must be 16-byte aligned");
    ProfScope ps("pack_detections", st);
    const dim3 grid((unsigned)((stride + 255) / 256), (unsigned)B);
    const float4 *b4 = reinterpret_cast<const float4 *>(boxes);
    if (wire)
        pack_detections_kernel<true><<<grid, dim3(256), 0, st>>>(b4, scores, class_ids, keep, keep_counts, image_ids, image_sizes,
                                                                 (int)B, stride, rows_ids, rows_vals, row_offsets,
                                                                 (int)detbox_wire_header_rows(images_cap), wire);
    else
        pack_detections_kernel<false><<<grid, dim3(256), 0, st>>>(b4, scores, class_ids, keep, keep_counts, image_ids, image_sizes,
                                                                  (int)B, stride, rows_ids, rows_vals, row_offsets, 0, nullptr);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
}  // namespace

extern "C" int detbox_pack_detections(const float *boxes, const float *scores, const int64_t *class_ids,
                                      const int64_t *keep, const int32_t *keep_counts,
                                      const int64_t *image_ids, const int32_t *image_sizes,
                                      int64_t B, int64_t stride,
                                      int64_t *rows_ids, double *rows_vals, int32_t *row_offsets, void *stream) {
    return pack_rows(boxes, scores, class_ids, keep, keep_counts, image_ids, image_sizes, B, stride, rows_ids, rows_vals,
                     row_offsets, 0, nullptr, stream);
}

extern "C" int detbox_pack_detections_wire(const float *boxes, const float *scores, const int64_t *class_ids,
                                           const int64_t *keep, const int32_t *keep_counts,
                                           const int64_t *image_ids, const int32_t *image_sizes,
                                           int64_t B, int64_t stride,
                                           int64_t *rows_ids, double *rows_vals, int32_t *row_offsets,
                                           int64_t images_cap, uint32_t *wire, void *stream) {
    DETBOX_CHECK_ARG(wire, "null pointer");
    return pack_rows(boxes, scores, class_ids, keep, keep_counts, image_ids, image_sizes, B, stride, rows_ids, rows_vals,
                     row_offsets, images_cap, wire, stream);
}

extern "C" size_t detbox_nms_rows_workspace_bytes(int64_t B, int64_t stride, int64_t S) {
    return detbox_batched_nms_segments_workspace_bytes(B, stride, S);
}

extern "C" int detbox_nms_rows(const float *boxes, const float *scores, const int64_t *class_ids,
                               const int32_t *seg_offsets, int64_t B, int64_t stride, int64_t S, int64_t max_segment,
                               double iou_threshold, int presorted,
                               const int64_t *image_ids, const int32_t *image_sizes,
                               int64_t *keep, int32_t *keep_counts,
                               int64_t *rows_ids, double *rows_vals, int32_t *row_offsets,
                               int64_t images_cap, uint32_t *wire,
                               void *workspace, size_t workspace_bytes, void *stream) {
    DETBOX_CHECK_ARG(B >= 0 && stride >= 0 && S >= 0, "negative size");
    DETBOX_CHECK_ARG(row_offsets && keep_counts, "null pointer");
    DETBOX_CHECK_ARG(!wire || images_cap >= B, "images_cap smaller than the number of images");
    if (B > 0 && stride > 0)
        DETBOX_CHECK_ARG(class_ids && image_ids && image_sizes && rows_ids && rows_vals && keep, "null pointer");
    RowsOut ro;
    ro.boxes = reinterpret_cast<const float4 *>(boxes); ro.scores = scores; ro.class_ids = class_ids;
    ro.image_ids = image_ids; ro.image_sizes = image_sizes;
    ro.rows_ids = rows_ids; ro.rows_vals = rows_vals; ro.row_offsets = row_offsets;
    ro.hdr_rows = wire ? (int)detbox_wire_header_rows(images_cap) : 0; ro.wire = wire;
    bool rows_done = false;
    const int rc = nms_run(boxes, scores, seg_offsets, B, stride, S, max_segment, iou_threshold, presorted, keep, keep_counts,
                           &ro, &rows_done, workspace, workspace_bytes, stream);
    if (rc != DETBOX_OK || rows_done) return rc;
    return pack_rows(boxes, scores, class_ids, keep, keep_counts, image_ids, image_sizes, B, stride, rows_ids, rows_vals,
                     row_offsets, images_cap, wire, stream);
}

extern "C" int64_t detbox_wire_header_rows(int64_t images_cap) {
    if (images_cap < 0) images_cap = 0;
    return (2 + (images_cap + 1) + 4 * images_cap + 5) / 6;
}

extern "C" int detbox_pack_wire(const float *boxes, const float *scores, const int64_t *class_ids,
                                const int64_t *keep, const int32_t *keep_counts,
                                const int64_t *image_ids, const int32_t *image_sizes,
                                int64_t B, int64_t stride, int64_t images_cap, uint32_t *wire, void *stream) {
    DETBOX_CHECK_ARG(B >= 0 && stride >= 0 && images_cap >= B, "bad size");
    DETBOX_CHECK_ARG(B <= 65535 && stride < (1ll << 31), "size too large");
    DETBOX_CHECK_ARG(wire, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B == 0 || stride == 0) {
        DETBOX_HIP(hipMemsetAsync(wire, 0, (size_t)detbox_wire_header_rows(images_cap) * 6 * sizeof(uint32_t), st));
        return DETBOX_OK;
    }
    DETBOX_CHECK_ARG(boxes && scores && class_ids && keep && keep_counts && image_ids && image_sizes, "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)boxes % 16 == 0, "boxes must be 16-byte aligned");
    ProfScope ps("pack_wire", st);
    pack_wire_kernel<<<dim3((unsigned)((stride + 255) / 256), (unsigned)B), dim3(256), 0, st>>>(
        reinterpret_cast<const float4 *>(boxes), scores, class_ids, keep, keep_counts, image_ids, image_sizes, (int)B,
        stride, (int)detbox_wire_header_rows(images_cap), wire);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
