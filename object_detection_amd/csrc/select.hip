// Inference-time candidate selection: score threshold + exact per-class top-k +
// box decode, for a whole batch, replacing the per-image / per-class Python loop
// of every pre_predict (src/models/ssd300.py:364-382, src/models/retinanet.py:292-310,
// src/models/fcos.py:324-341, src/models/yolov3.py:311-329) and their
// `argsort(descending=True).argsort() < top_k` rank trick (ssd300.py:370).
//
// The score tensor [B,P,C] (class innermost) is streamed twice, fully coalesced:
//   score_hist      per-(image,class) 256-bin histogram of the scores above the
//                   threshold (bins are equal slices of the monotone uint32 image
//                   of fp32 between the threshold and score_hi_hint), built in
//                   LDS per anchor tile and merged with global atomics;
//   select_kth      per (image,class): the bin holding the top_k-th score, the
//                   selection rule, the class segment offsets of the output;
//   select_refine   only for (image,class) pairs whose boundary bin is too full
//                   (massive ties): exact radix search of the 64-bit cut-off key;
//   select_emit     second pass: scores at or above the boundary bin are appended
//                   (unordered) to a small per-(image,class) staging list;
//   select_finalize per (image,class): bitonic sort of the staged keys in LDS,
//                   exact cut at top_k, re-sort by anchor index, decode, write.
//
// Selection key: K = ((okey(score) - okey(thr)) << 32) | (0xffffffff - anchor), so
// "larger K" means higher score, then lower anchor index: boundary ties go to the
// lower anchor index, and all keys of one class are distinct.
#include "detbox_internal.h"

#include <cstring>

using namespace detbox;

namespace {

constexpr int NB = 256;             // histogram bins per (image, class)
constexpr int PASS_BLOCK = 512;
constexpr int MAX_SEL_CLASSES = 128;  // LDS histogram: classes * NB/2 words
constexpr int MAX_TILE_ANCHORS = 16384;

enum { RULE_ALL = 0, RULE_HIST = 1, RULE_EXACT = 2 };

struct SelRule {
    uint32_t mode;      // RULE_*
    uint32_t beta;      // RULE_HIST: boundary bin (bins above are certain)
    uint32_t n_above;   // scores in bins above beta
    uint32_t n_sel;     // rows this (image,class) will output
    unsigned long long cutoff;  // RULE_EXACT: keep K >= cutoff
    float min_score;    // cheap reject: nothing below this can be selected
    uint32_t pad;
};

struct SelParams {
    int P, C, cb, Cs, top_k, cap, shift, tile_anchors;
    float thr;
    uint32_t key_thr;
};

__device__ inline uint32_t okey(float s) { return okey_from_bits(__float_as_uint(s)); }

__device__ inline uint32_t bin_of(uint32_t d, int shift) {
    uint32_t b = d >> shift;
    return b > (uint32_t)(NB - 1) ? (uint32_t)(NB - 1) : b;
}

// Walks the [p0,p1) x C slab of image b with coalesced 16-byte loads; calls f(score, class, element).
template <typename F>
__device__ inline void walk_tile(const float *__restrict__ base, int n, int C, F f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int head = (int)((4u - (unsigned)(((uintptr_t)base >> 2) & 3u)) & 3u);
    if (head > n) head = n;
    if (tid < head) f(base[tid], tid % C, tid);
    const int nvec = (n - head) >> 2;
    const float4 *vb = reinterpret_cast<const float4 *>(base + head);
    int c = (head + 4 * tid) % C;
    const int cstep = (4 * nt) % C;
    for (int v = tid; v < nvec; v += nt) {
        const float4 x = vb[v];
        const int e = head + 4 * v;
        int c1 = c + 1; if (c1 >= C) c1 -= C;
        int c2 = c1 + 1; if (c2 >= C) c2 -= C;
        int c3 = c2 + 1; if (c3 >= C) c3 -= C;
        f(x.x, c, e); f(x.y, c1, e + 1); f(x.z, c2, e + 2); f(x.w, c3, e + 3);
        c += cstep; if (c >= C) c -= C;
    }
    const int done = head + 4 * nvec;
    if (tid < n - done) f(base[done + tid], (done + tid) % C, done + tid);
}

__global__ __launch_bounds__(PASS_BLOCK) void score_hist_kernel(const float *__restrict__ scores, SelParams sp,
                                                                uint32_t *__restrict__ ghist) {
    extern __shared__ uint32_t s_hist[];   // [Cs][NB/2], two 16-bit counters per word
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * sp.tile_anchors;
    const int p1 = min(sp.P, p0 + sp.tile_anchors);
    const int words = sp.Cs * (NB / 2);
    for (int i = threadIdx.x; i < words; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    const float *base = scores + ((size_t)b * sp.P + p0) * sp.C;
    const float thr = sp.thr;
    const uint32_t key_thr = sp.key_thr;
    const int shift = sp.shift, cb = sp.cb;
    walk_tile(base, (p1 - p0) * sp.C, sp.C, [&](float s, int c, int) {
        if (c >= cb && s > thr) {
            const uint32_t bin = bin_of(okey(s) - key_thr, shift);
            atomicAdd(&s_hist[(c - cb) * (NB / 2) + (bin >> 1)], 1u << ((bin & 1u) * 16u));
        }
    });
    __syncthreads();
    uint32_t *gh = ghist + (size_t)b * sp.Cs * NB;
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const uint32_t v = s_hist[i];
        if (v) {
            const uint32_t lo = v & 0xffffu, hi = v >> 16;
            if (lo) atomicAdd(&gh[2 * i], lo);
            if (hi) atomicAdd(&gh[2 * i + 1], hi);
        }
    }
}

__device__ inline float score_from_d(uint32_t key_thr, unsigned long long d) {
    unsigned long long k = (unsigned long long)key_thr + d;
    if (k > 0xffffffffull) k = 0xffffffffull;
    return __uint_as_float(bits_from_okey((uint32_t)k));
}

__global__ __launch_bounds__(256) void select_kth_kernel(const uint32_t *__restrict__ ghist, SelParams sp,
                                                         SelRule *__restrict__ rules, int32_t *__restrict__ seg_off,
                                                         int32_t *__restrict__ counts) {
    __shared__ int s_nsel[MAX_SEL_CLASSES];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t k = (uint32_t)sp.top_k;
    for (int ci = wave; ci < sp.Cs; ci += nwaves) {
        const uint32_t *h = ghist + ((size_t)b * sp.Cs + ci) * NB;
        const uint4 q = reinterpret_cast<const uint4 *>(h)[lane];       // bins 4*lane .. 4*lane+3
        const uint32_t t = q.x + q.y + q.z + q.w;
        uint32_t suf = t;                                               // sum over lanes >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_down(suf, o);
            if (lane + o < 64) suf += up;
        }
        const uint32_t total = __shfl(suf, 0);
        SelRule r;
        r.pad = 0; r.cutoff = 0ull; r.beta = 0; r.n_above = 0;
        if (total <= k) {
            r.mode = RULE_ALL; r.n_sel = total;
            r.min_score = score_from_d(sp.key_thr, 1ull);
        } else {
            const unsigned long long ge = __ballot(suf >= k);           // a prefix of lanes
            const int L = 63 - __clzll((long long)ge);
            uint32_t beta = 0, n_above = 0, n_beta = 0;
            if (lane == L) {
                uint32_t acc = suf - t;
                const uint32_t hb[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    if (acc + hb[i] >= k) { beta = 4 * L + i; n_above = acc; n_beta = hb[i]; break; }
                    acc += hb[i];
                }
            }
            beta = __shfl(beta, L); n_above = __shfl(n_above, L); n_beta = __shfl(n_beta, L);
            r.beta = beta; r.n_above = n_above; r.n_sel = k;
            r.mode = (n_above + n_beta <= (uint32_t)sp.cap) ? RULE_HIST : RULE_EXACT;
            unsigned long long dmin = (unsigned long long)beta << sp.shift;
            if (dmin < 1ull) dmin = 1ull;
            r.min_score = score_from_d(sp.key_thr, dmin);
        }
        if (lane == 0) { rules[(size_t)b * sp.Cs + ci] = r; s_nsel[ci] = (int)r.n_sel; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        int32_t *so = seg_off + (size_t)b * (sp.Cs + 1);
        for (int ci = 0; ci < sp.Cs; ++ci) { so[ci] = acc; acc += s_nsel[ci]; }
        so[sp.Cs] = acc;
        counts[b] = acc;
    }
}

// Exact cut-off key for (image,class) pairs whose boundary bin holds more scores than the staging
// list can take.  One block per pair walks its (strided) score column once per 8 key bits.
__global__ __launch_bounds__(256) void select_refine_kernel(const float *__restrict__ scores, SelParams sp,
                                                            SelRule *__restrict__ rules) {
    __shared__ uint32_t s_h[256];
    __shared__ unsigned long long s_lo, s_hi;
    __shared__ uint32_t s_need;
    const int ci = blockIdx.x, b = blockIdx.y;
    SelRule *rp = rules + (size_t)b * sp.Cs + ci;
    if (rp->mode != RULE_EXACT) return;
    const int c = ci + sp.cb;
    if (threadIdx.x == 0) {
        const uint32_t beta = rp->beta;
        s_lo = (unsigned long long)beta << (sp.shift + 32);
        s_hi = (beta == (uint32_t)(NB - 1)) ? ~0ull : ((((unsigned long long)beta + 1ull) << (sp.shift + 32)) - 1ull);
        s_need = (uint32_t)sp.top_k - rp->n_above;
    }
    __syncthreads();
    const float *col = scores + (size_t)b * sp.P * sp.C + c;
    for (int iter = 0; iter < 16; ++iter) {            // 64 key bits / 8 per round, with slack
        const unsigned long long lo = s_lo, hi = s_hi;
        if (lo >= hi) break;
        const unsigned long long span = hi - lo;
        const int sb = 64 - __clzll((long long)span);
        const int sh = sb > 8 ? sb - 8 : 0;
        s_h[threadIdx.x] = 0u;
        __syncthreads();
        for (int p = threadIdx.x; p < sp.P; p += blockDim.x) {
            const float s = col[(size_t)p * sp.C];
            if (s > sp.thr) {
                const unsigned long long K =
                    ((unsigned long long)(okey(s) - sp.key_thr) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)p);
                if (K >= lo && K <= hi) atomicAdd(&s_h[(uint32_t)((K - lo) >> sh)], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t need = s_need, acc = 0;
            int sbin = 0;
            for (int i = 255; i >= 0; --i) {
                if (acc + s_h[i] >= need) { sbin = i; break; }
                acc += s_h[i];
            }
            const unsigned long long nlo = lo + ((unsigned long long)sbin << sh);
            unsigned long long nhi = nlo + ((1ull << sh) - 1ull);
            if (nhi > hi || nhi < nlo) nhi = hi;
            s_lo = nlo; s_hi = nhi; s_need = need - acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rp->cutoff = s_lo;
        rp->min_score = score_from_d(sp.key_thr, s_lo >> 32);
    }
}

__global__ __launch_bounds__(PASS_BLOCK) void select_emit_kernel(const float *__restrict__ scores, SelParams sp,
                                                                 const SelRule *__restrict__ rules,
                                                                 uint32_t *__restrict__ cnt,
                                                                 unsigned long long *__restrict__ staging) {
    __shared__ float s_min[MAX_SEL_CLASSES];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * sp.tile_anchors;
    const int p1 = min(sp.P, p0 + sp.tile_anchors);
    const SelRule *rb = rules + (size_t)b * sp.Cs;
    for (int i = threadIdx.x; i < sp.Cs; i += blockDim.x)
        s_min[i] = rb[i].n_sel ? rb[i].min_score : __builtin_inff();
    __syncthreads();
    const float *base = scores + ((size_t)b * sp.P + p0) * sp.C;
    const int cb = sp.cb, C = sp.C, shift = sp.shift, cap = sp.cap;
    const uint32_t key_thr = sp.key_thr;
    walk_tile(base, (p1 - p0) * C, C, [&](float s, int c, int e) {
        if (c >= cb && s >= s_min[c - cb]) {
            const int ci = c - cb;
            const uint32_t d = okey(s) - key_thr;
            const uint32_t p = (uint32_t)(p0 + e / C);
            const unsigned long long K = ((unsigned long long)d << 32) | (unsigned long long)(0xffffffffu - p);
            const SelRule r = rb[ci];
            bool take = true;
            if (r.mode == RULE_HIST) take = bin_of(d, shift) >= r.beta;
            else if (r.mode == RULE_EXACT) take = K >= r.cutoff;
            if (take) {
                const uint32_t slot = atomicAdd(&cnt[(size_t)b * sp.Cs + ci], 1u);
                if (slot < (uint32_t)cap) staging[((size_t)b * sp.Cs + ci) * cap + slot] = K;
            }
        }
    });
}

__device__ inline void bitonic_desc(unsigned long long *s, int n) {
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

__device__ inline int pow2_at_least(int n) {
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

template <int DECODE>
__device__ inline float4 decode_box(const float *__restrict__ anchors, float4 l, uint32_t p) {
    if (DECODE == DETBOX_DECODE_CXCYWH) {
        const float4 a = reinterpret_cast<const float4 *>(anchors)[p];
        const float cx = a.x + (0.1f * l.x) * a.z;
        const float cy = a.y + (0.1f * l.y) * a.w;
        const float w = a.z * expf(0.2f * l.z);
        const float h = a.w * expf(0.2f * l.w);
        return make_float4(cx - 0.5f * w, cy - 0.5f * h, cx + 0.5f * w, cy + 0.5f * h);
    } else if (DECODE == DETBOX_DECODE_YOLO) {
        const float *a = anchors + (size_t)p * 5;
        const float x = a[0] + (0.1f * l.x) * a[4];
        const float y = a[1] + (0.1f * l.y) * a[4];
        const float w = a[2] * expf(0.2f * l.z);
        const float h = a[3] * expf(0.2f * l.w);
        return make_float4(x, y, x + w, y + h);
    } else {
        const float2 pt = reinterpret_cast<const float2 *>(anchors)[p];
        return make_float4(pt.x - expf(l.x), pt.y - expf(l.y), pt.x + expf(l.z), pt.y + expf(l.w));
    }
}

template <int DECODE>
__global__ __launch_bounds__(256) void select_finalize_kernel(
    SelParams sp, const SelRule *__restrict__ rules, const uint32_t *__restrict__ cnt,
    const unsigned long long *__restrict__ staging, const int32_t *__restrict__ seg_off,
    const float4 *__restrict__ locs, const float *__restrict__ anchors, int class_id_offset,
    float4 *__restrict__ out_boxes, float *__restrict__ out_scores, int64_t *__restrict__ out_cls) {
    extern __shared__ unsigned long long s_k[];
    const int ci = blockIdx.x, b = blockIdx.y;
    const size_t bc = (size_t)b * sp.Cs + ci;
    const int n_sel = (int)rules[bc].n_sel;
    if (n_sel == 0) return;
    int n_stage = (int)cnt[bc];
    if (n_stage > sp.cap) n_stage = sp.cap;
    const int n2 = pow2_at_least(n_stage);
    const unsigned long long *st = staging + bc * sp.cap;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) s_k[i] = i < n_stage ? st[i] : 0ull;
    __syncthreads();
    bitonic_desc(s_k, n2);                      // best (score desc, anchor asc) first
    const int take = min(n_sel, n_stage);
    const int m2 = pow2_at_least(take);
    // swap the key halves so that the same descending sort orders by ascending anchor index
    for (int i = threadIdx.x; i < m2; i += blockDim.x) {
        unsigned long long v = 0ull;
        if (i < take) { const unsigned long long K = s_k[i]; v = (K << 32) | (K >> 32); }
        s_k[i] = v;
    }
    __syncthreads();
    bitonic_desc(s_k, m2);
    const size_t row0 = (size_t)b * ((size_t)sp.Cs * sp.top_k) + (size_t)seg_off[(size_t)b * (sp.Cs + 1) + ci];
    for (int i = threadIdx.x; i < take; i += blockDim.x) {
        const unsigned long long K2 = s_k[i];
        const uint32_t p = 0xffffffffu - (uint32_t)(K2 >> 32);
        const uint32_t d = (uint32_t)K2;
        const float4 l = locs[(size_t)b * sp.P + p];
        out_boxes[row0 + i] = decode_box<DECODE>(anchors, l, p);
        out_scores[row0 + i] = __uint_as_float(bits_from_okey(sp.key_thr + d));
        out_cls[row0 + i] = (int64_t)(ci + sp.cb + class_id_offset);
    }
}

int staging_cap(int top_k) {
    int slack = top_k / 2 > 256 ? top_k / 2 : 256;
    int need = top_k + slack, cap = 2;
    while (cap < need) cap <<= 1;
    return cap;
}

struct SelLayout { size_t hist, cnt, rules, staging, total, zero_bytes; };

SelLayout layout(int64_t B, int64_t Cs, int cap) {
    SelLayout L;
    size_t off = 0;
    L.hist = off; off = align_up(off + (size_t)B * Cs * NB * sizeof(uint32_t), 256);
    L.cnt = off; off = align_up(off + (size_t)B * Cs * sizeof(uint32_t), 256);
    L.zero_bytes = off;                      // histogram + counters are cleared every call
    L.rules = off; off = align_up(off + (size_t)B * Cs * sizeof(SelRule), 256);
    L.staging = off; off = align_up(off + (size_t)B * Cs * cap * sizeof(unsigned long long), 256);
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t detbox_select_decode_workspace_bytes(int64_t B, int64_t P, int64_t C, int class_begin, int top_k) {
    (void)P;
    if (B <= 0 || C <= class_begin || top_k <= 0) return 256;
    return layout(B, C - class_begin, staging_cap(top_k)).total;
}

extern "C" int detbox_select_decode(int decode_mode, const float *scores, const float *locs, const float *anchors,
                                    int64_t B, int64_t P, int64_t C, int class_begin, int class_id_offset,
                                    float conf_thresh, int top_k, float score_hi_hint,
                                    float *out_boxes, float *out_scores, int64_t *out_class_ids,
                                    int32_t *out_seg_offsets, int32_t *out_counts,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    DETBOX_CHECK_ARG(decode_mode >= 0 && decode_mode <= 2, "unknown decode mode");
    DETBOX_CHECK_ARG(B >= 0 && P >= 0 && C >= 1, "bad size");
    DETBOX_CHECK_ARG(class_begin >= 0 && class_begin < C, "class_begin out of range");
    DETBOX_CHECK_ARG(!(conf_thresh != conf_thresh), "conf_thresh is NaN");
    if (top_k < 1 || top_k > DETBOX_MAX_TOP_K) {
        set_error("detbox_select_decode: top_k %d outside 1..%d", top_k, DETBOX_MAX_TOP_K);
        return DETBOX_ERR_UNSUPPORTED;
    }
    const int64_t Cs = C - class_begin;
    if (Cs > MAX_SEL_CLASSES) {
        set_error("detbox_select_decode: %lld selected classes, limit %d", (long long)Cs, MAX_SEL_CLASSES);
        return DETBOX_ERR_UNSUPPORTED;
    }
    DETBOX_CHECK_ARG(P < (1ll << 31) && P * C < (1ll << 40), "P too large");
    if (B == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(out_seg_offsets && out_counts, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P == 0) {
        DETBOX_HIP(hipMemsetAsync(out_seg_offsets, 0, (size_t)B * (Cs + 1) * sizeof(int32_t), s));
        DETBOX_HIP(hipMemsetAsync(out_counts, 0, (size_t)B * sizeof(int32_t), s));
        return DETBOX_OK;
    }
    DETBOX_CHECK_ARG(scores && locs && anchors && out_boxes && out_scores && out_class_ids, "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)scores % 4 == 0 && (uintptr_t)locs % 16 == 0 && (uintptr_t)out_boxes % 16 == 0,
                     "misaligned array");
    DETBOX_CHECK_ARG(decode_mode != DETBOX_DECODE_CXCYWH || (uintptr_t)anchors % 16 == 0, "anchors must be 16-byte aligned");
    DETBOX_CHECK_ARG(decode_mode != DETBOX_DECODE_FCOS || (uintptr_t)anchors % 8 == 0, "points must be 8-byte aligned");
    const int cap = staging_cap(top_k);
    const SelLayout L = layout(B, Cs, cap);
    if (!workspace || (uintptr_t)workspace % 256 || workspace_bytes < L.total) {
        set_error("detbox_select_decode: workspace missing, misaligned or smaller than %zu bytes", L.total);
        return DETBOX_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(workspace);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(ws + L.hist);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(ws + L.cnt);
    SelRule *rules = reinterpret_cast<SelRule *>(ws + L.rules);
    unsigned long long *staging = reinterpret_cast<unsigned long long *>(ws + L.staging);

    SelParams sp;
    sp.P = (int)P; sp.C = (int)C; sp.cb = class_begin; sp.Cs = (int)Cs; sp.top_k = top_k; sp.cap = cap;
    sp.thr = conf_thresh;
    uint32_t tb, hb;
    std::memcpy(&tb, &conf_thresh, 4);
    std::memcpy(&hb, &score_hi_hint, 4);
    sp.key_thr = okey_from_bits(tb);
    uint32_t key_hi = okey_from_bits(hb);
    if (!(score_hi_hint > conf_thresh)) key_hi = 0xffffffffu;
    const uint32_t range = key_hi - sp.key_thr;
    int shift = 0;
    while ((range >> shift) >= (uint32_t)(NB - 1)) ++shift;
    sp.shift = shift;
    int64_t tile = (P * B + 2047) / 2048;
    if (tile < 256) tile = 256;
    if (tile > MAX_TILE_ANCHORS) tile = MAX_TILE_ANCHORS;
    if (tile > P) tile = P;
    sp.tile_anchors = (int)tile;
    const int64_t tiles = (P + tile - 1) / tile;
    DETBOX_CHECK_ARG(B <= 65535, "B > 65535");

    DETBOX_HIP(hipMemsetAsync(ws, 0, L.zero_bytes, s));
    {
        ProfScope ps("score_hist", s);
        score_hist_kernel<<<dim3((unsigned)tiles, (unsigned)B), dim3(PASS_BLOCK), (size_t)Cs * (NB / 2) * 4, s>>>(
            scores, sp, ghist);
        DETBOX_LAUNCH_CHECK();
    }
    {
        ProfScope ps("select_kth", s);
        select_kth_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(ghist, sp, rules, out_seg_offsets, out_counts);
        DETBOX_LAUNCH_CHECK();
    }
    {
        ProfScope ps("select_refine", s);
        select_refine_kernel<<<dim3((unsigned)Cs, (unsigned)B), dim3(256), 0, s>>>(scores, sp, rules);
        DETBOX_LAUNCH_CHECK();
    }
    {
        ProfScope ps("select_emit", s);
        select_emit_kernel<<<dim3((unsigned)tiles, (unsigned)B), dim3(PASS_BLOCK), 0, s>>>(scores, sp, rules, cnt, staging);
        DETBOX_LAUNCH_CHECK();
    }
    {
        ProfScope ps("select_finalize", s);
        dim3 grid((unsigned)Cs, (unsigned)B);
        const size_t lds = (size_t)cap * sizeof(unsigned long long);
        const float4 *l4 = reinterpret_cast<const float4 *>(locs);
        float4 *ob = reinterpret_cast<float4 *>(out_boxes);
        if (decode_mode == DETBOX_DECODE_CXCYWH)
            select_finalize_kernel<DETBOX_DECODE_CXCYWH><<<grid, dim3(256), lds, s>>>(
                sp, rules, cnt, staging, out_seg_offsets, l4, anchors, class_id_offset, ob, out_scores, out_class_ids);
        else if (decode_mode == DETBOX_DECODE_YOLO)
            select_finalize_kernel<DETBOX_DECODE_YOLO><<<grid, dim3(256), lds, s>>>(
                sp, rules, cnt, staging, out_seg_offsets, l4, anchors, class_id_offset, ob, out_scores, out_class_ids);
        else
            select_finalize_kernel<DETBOX_DECODE_FCOS><<<grid, dim3(256), lds, s>>>(
                sp, rules, cnt, staging, out_seg_offsets, l4, anchors, class_id_offset, ob, out_scores, out_class_ids);
        DETBOX_LAUNCH_CHECK();
    }
    return DETBOX_OK;
}
