// Inference-time candidate selection: activation + score threshold + exact per-class
// top-k + box decode for a whole batch, replacing the per-image / per-class Python loop
// of every pre_predict (src/models/ssd300.py:354-382, src/models/retinanet.py:282-310,
// src/models/fcos.py:314-341, src/models/yolov3.py:297-329) and their
// `argsort(descending=True).argsort() < top_k` rank trick (ssd300.py:370).
//
// The head output [B,P,C] (class innermost) is streamed from HBM ONCE on the fast path:
//   select_sample   activates every sample_rate-th 64-anchor sub-tile and writes the sampled
//                   scores class-major;
//   select_plan     per (image,class): the exact sample_m-th largest sampled score (LDS radix
//                   select) = a score t_lo low enough that, with overwhelming probability, at
//                   least top_k scores lie at or above it (or "stage all");
//   select_scan     the one full pass: 64 anchors x C logits go HBM -> LDS by 16-byte LDS-DMA,
//                   double buffered; four lanes share an anchor row for the soft-max; every
//                   score >= t_lo is appended to the wave's own LDS list (no atomics), and a full
//                   list goes to the per-(image,class) staging lists with one global atomic per
//                   class present;
//   select_check    verifies the guess (enough staged, not too many); pairs that fail are
//                   handed to the exact two-pass fallback below and a flag is raised;
//   select_finalize per (image,class): exact top_k of the staged keys by radix select,
//                   ordering by anchor index through an LDS bitmap + popcount scan, decode.
// The focal-loss kernels at the end of the file stream the same tensor through the same
// sub-tile machinery (training side).
// Fallback (kernels exit at once unless the flag is up; also the only path for tiny P):
//   score_hist / select_kth / select_refine / select_emit find the exact boundary bin (or, on
//   massive ties, the exact 64-bit cut-off key) from a full histogram and re-stage; with a fused
//   activation they recompute the scores from the head output (no copy of the scores is kept).
//
// Selection key: K = ((okey(score) - okey(thr)) << 32) | (0xffffffff - anchor): larger K
// means higher score, then lower anchor index (boundary ties go to the lower anchor), and
// all keys of one class are distinct.
#include "detbox_internal.h"

#include <cmath>
#include <cstring>

using namespace detbox;

namespace {

constexpr int NB = 256;               // histogram bins per (image, class)
constexpr int PASS_BLOCK = 512;       // fallback streaming passes
constexpr int MAX_SEL_CLASSES = 128;  // LDS tables are sized for this many selected classes
constexpr int MAX_TILE_ANCHORS = 16384;
constexpr int SCAN_BLOCK = 256;
constexpr int LPR = 4;                // lanes that share one anchor row (8, with 512 threads, was measured: no faster)
constexpr int SUB_ROWS = SCAN_BLOCK / LPR;   // anchors per LDS sub-tile
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;  // every wave of a scan workgroup keeps its own candidate list
constexpr int SCAN_EBUF = 256;        // most entries of one of a scan wave's two candidate lists in LDS (sp.ebuf: 128 or 256)
constexpr int SAMPLE_SUBS = 4;        // sampled sub-tiles per workgroup of select_sample
constexpr int BITMAP_MAX_WORDS = 4096;  // finalize orders by anchor through a P-bit LDS bitmap up to this
constexpr int CNT_STRIDE = 1;           // words between two (image, class) staging counters (a line each, 32: no faster)

enum { ACT_NONE = 0, ACT_SOFTMAX = 1, ACT_SOFTMAX_SIGMOID = 2, ACT_SIGMOID_SIGMOID = 3 };
enum { RULE_ALL = 0, RULE_HIST = 1, RULE_EXACT = 2, RULE_STAGED = 3, RULE_PENDING = 4 };

struct SelRule {
    uint32_t mode;      // RULE_*
    uint32_t beta;      // RULE_HIST: boundary bin; RULE_STAGED: 1 when every score above the threshold is staged
    uint32_t n_above;   // RULE_HIST/EXACT: scores in bins above beta
    uint32_t n_sel;     // rows this (image,class) will output
    unsigned long long cutoff;  // RULE_EXACT: keep K >= cutoff
    float min_score;    // nothing below this is staged
    uint32_t pad;
};

struct SelParams {
    int P, C, cb, Cs, top_k, cap, shift, tile_anchors;
    int use_sample, sample_rate, sample_m, subs_per_block, n_sub, tile_floats;
    int ns_cap, scan_gx, ebuf, flush_at;
    long long total_floats;
    float thr;
    uint32_t key_thr;
};

__device__ inline uint32_t okey(float s) { return okey_from_bits(__float_as_uint(s)); }

__device__ inline uint32_t bin_of(uint32_t d, int shift) {
    uint32_t b = d >> shift;
    return b > (uint32_t)(NB - 1) ? (uint32_t)(NB - 1) : b;
}

__device__ inline float score_from_d(uint32_t key_thr, unsigned long long d) {
    unsigned long long k = (unsigned long long)key_thr + d;
    if (k > 0xffffffffull) k = 0xffffffffull;
    return __uint_as_float(bits_from_okey((uint32_t)k));
}

__device__ inline unsigned long long make_key(float s, uint32_t key_thr, uint32_t p) {
    return ((unsigned long long)(okey(s) - key_thr) << 32) | (unsigned long long)(0xffffffffu - p);
}

// ---------------------------------------------------------------------------------------------
// Activation of one sub-tile: rows x C head outputs -> scores, four lanes per anchor row.
// ---------------------------------------------------------------------------------------------

// Reductions over the LPR lanes of a row: in-row moves of the VALU, nothing goes through the LDS crossbar.
template <int O>
__device__ inline float quad_max_from(float v) {
    if constexpr (O < LPR) return quad_max_from<O * 2>(fmaxf(v, lane_xor_f32<O>(v)));
    else return v;
}
template <int O>
__device__ inline float quad_sum_from(float v) {
    if constexpr (O < LPR) return quad_sum_from<O * 2>(v + lane_xor_f32<O>(v));
    else return v;
}
__device__ inline float quad_max(float v) { return quad_max_from<1>(v); }
__device__ inline float quad_sum(float v) { return quad_sum_from<1>(v); }
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// exp of the soft-max: the hardware's exp2 with a scaled argument (2 ulp), or, built with DETBOX_SOFTMAX_LIBM_EXP, the
// library's expf (1 ulp, about three times the instructions)
#ifdef DETBOX_SOFTMAX_LIBM_EXP
__device__ inline float softmax_exp(float x) { return expf(x); }
#else
__device__ inline float softmax_exp(float x) { return __expf(x); }
#endif

// One sub-tile = rows [row_base, row_base+rows) x C of the head output.  It is brought into LDS by
// LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers); the LDS
// image starts `shift` floats in, so that every 16-byte piece is aligned in both memories.  Four lanes
// share an anchor row, so a wave works on 16 rows of the sub-tile -- and it brings in exactly those 16
// rows itself, into two LDS buffers of its own that alternate: the DMA of its part of sub-tile s+1 is in
// flight while it consumes its part of sub-tile s, and all it ever waits for is its own DMA.  No barrier
// inside the loop: the four waves of a workgroup drift apart freely, and a wave that is held up (handing
// its candidates over, say) holds up nobody else.  A lane keeps its EPT = ceil(C/4) elements in registers
// for the whole soft-max.
//   ACT_SOFTMAX          score = exp(x - max) / sum                    (ssd300.py:354, retinanet.py:282)
//   ACT_SOFTMAX_SIGMOID  score = softmax(x) * sigmoid(extra[row])      (fcos.py:314)
//   ACT_SIGMOID_SIGMOID  score = sigmoid(x) * sigmoid(extra[row])      (yolov3.py:298-300)
constexpr int WAVE_ROWS = 64 / LPR;      // anchor rows a wave works on

struct SubGeom {          // this wave's part of a sub-tile
    const float *src;     // 16-byte aligned start of the pieces
    int shift, nvec;      // floats between src and the first row; 16-byte pieces (0: the wave has no row)
};

__device__ inline SubGeom sub_geom(const float *in, long long row_base, int rows, int C) {
    SubGeom g;
    const int wave = (int)(threadIdx.x >> 6);
    const int wr = min(WAVE_ROWS, rows - wave * WAVE_ROWS);
    const long long f0 = (row_base + wave * WAVE_ROWS) * C;
    g.shift = (int)(f0 & 3);
    g.src = in + (f0 - g.shift);
    g.nvec = wr > 0 ? (wr * C + g.shift + 3) >> 2 : 0;
    return g;
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Every piece is a 16-byte aligned 16-byte unit that overlaps the tensor, so it lies in a page the tensor owns even when
// its last bytes are past the tensor's end (those land in slack floats nobody reads).
// Cache policy of the sub-tile DMA: non-temporal.  Every sub-tile is read exactly once; without the hint the stream
// is allocated in the L2 and the memory-side cache like reusable data (the scan alone: 155 -> 136 us on
// 16 x 120087 x 81 logits, 125 -> 105 us when next to nothing is emitted).
constexpr int DMA_CPOL = 2;      // 'nt' on gfx950
template <int EPT>
__device__ inline void dma_issue(const SubGeom &g, float *s_wave) {       // s_wave: the wave's own buffer
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < EPT / 4 + 1; ++j) {                              // 16 rows of C <= 4 EPT floats: at most 16 EPT + 2 pieces
        const int v = lane + j * 64;
        if (v < g.nvec)                                                 // LDS destination = wave-uniform base + lane * 16
            __builtin_amdgcn_global_load_lds((gptr_t)(g.src + 4 * v), (lptr_t)(s_wave + 4 * (j * 64)), 16, 0, DMA_CPOL);
    }
}

// What a lane needs to recompute one score of its row from the LDS tile, bit for bit.
struct RowCtx {
    const float *row;   // &tile[row][sub]; element i sits at row[LPR*i]
    float m, inv, sg;   // soft-max maximum, 1/sum, sigmoid of the extra input
    float sum;          // soft-max denominator
};

template <int ACT>
__device__ inline float rescore(const RowCtx &c, int i) {
    const float v = c.row[LPR * i];
    if (ACT == ACT_SOFTMAX) return softmax_exp(v - c.m) * c.inv;
    if (ACT == ACT_SOFTMAX_SIGMOID) return (softmax_exp(v - c.m) * c.inv) * c.sg;
    if (ACT == ACT_SIGMOID_SIGMOID) return sigmoidf_(v) * c.sg;
    return v;
}

// Scores of this lane's elements of the sub-tile: x[i] = score of class sub + 4*i of local row threadIdx.x / 4
// (meaningless for classes >= C), read from the wave's own buffer s_wave.  Returns false for lanes beyond `rows`.
template <int ACT, int EPT>
__device__ inline bool subtile_scores(const float *s_wave, const float *__restrict__ extra, long long row_base,
                                      int rows, int C, int shift, float (&x)[EPT], RowCtx &ctx) {
    const int r = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    ctx.row = s_wave; ctx.m = 0.f; ctx.inv = 1.f; ctx.sg = 1.f; ctx.sum = 1.f;
    if (r >= rows) return false;
    const float *row = s_wave + shift + (r % WAVE_ROWS) * C + sub;     // the buffer has LPR*EPT floats of slack behind it
    ctx.row = row;
    const float fill = (ACT == ACT_SOFTMAX || ACT == ACT_SOFTMAX_SIGMOID) ? -__builtin_inff() : 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const float v = row[LPR * i];
        x[i] = (sub + LPR * i < C) ? v : fill;
    }
    if (ACT == ACT_SOFTMAX || ACT == ACT_SOFTMAX_SIGMOID) {
        float m = x[0];
#pragma unroll
        for (int i = 1; i < EPT; ++i) m = fmaxf(m, x[i]);
        m = quad_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            x[i] = softmax_exp(x[i] - m);
            sum += x[i];
        }
        sum = quad_sum(sum);
        const float inv = 1.0f / sum;
        ctx.m = m; ctx.inv = inv; ctx.sum = sum;
        if (ACT == ACT_SOFTMAX_SIGMOID) {
            const float sg = sigmoidf_(extra[row_base + r]);
            ctx.sg = sg;
#pragma unroll
            for (int i = 0; i < EPT; ++i) x[i] = (x[i] * inv) * sg;
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) x[i] = x[i] * inv;
        }
    } else if (ACT == ACT_SIGMOID_SIGMOID) {
        const float so = sigmoidf_(extra[row_base + r]);
        ctx.sg = so;
#pragma unroll
        for (int i = 0; i < EPT; ++i) x[i] = sigmoidf_(x[i]) * so;
    }
    return true;
}

// Runs f(x, valid, first_anchor_of_subtile, ctx) over sub-tiles first, first+step, ... (< last) of image b
// with the double-buffered DMA pipeline of every wave; after(s) runs once per sub-tile and wave behind f, with no
// vector-memory operation of the wave in flight.  s_tiles holds two buffers of tile_floats floats, a quarter
// of each per wave.  The waves are NOT synchronised inside the loop (nor behind it).
struct NoStart { __device__ void operator()() const {} };

// `start` runs once, after the DMA of the first sub-tile has been issued and before anybody waits for it: the place
// for a kernel's own set-up loads, which would otherwise leave the memory pipe idle at the head of every workgroup.
template <int ACT, int EPT, typename F, typename G, typename H = NoStart, typename H2 = NoStart>
__device__ inline void for_each_subtile(const float *__restrict__ in, const float *__restrict__ extra, int b, int P,
                                        int C, long long total_floats, long long first, long long last,
                                        long long step, float *s_tiles, int tile_floats, F f, G after, H start = H(),
                                        H2 started = H2()) {
    if (first >= last) { start(); __syncthreads(); started(); return; }
    auto geom = [&](long long s) {
        const int r0 = (int)s * SUB_ROWS;
        return sub_geom(in, (long long)b * P + r0, min(SUB_ROWS, P - r0), C);
    };
    float *s_wave = s_tiles + (threadIdx.x >> 6) * (tile_floats / SCAN_WAVES);      // this wave's quarter of buffer 0
    SubGeom g = geom(first);
    int buf = 0;
    dma_issue<EPT>(g, s_wave);
    start();
    __syncthreads();             // (the one barrier: it also waits for the first DMA)
    started();                   // what start() left in LDS is visible
    for (long long s = first; s < last; s += step) {
        const bool more = s + step < last;
        SubGeom gn = g;
        if (more) {
            gn = geom(s + step);
            dma_issue<EPT>(gn, s_wave + (buf ^ 1) * tile_floats);
        }
        const int r0 = (int)s * SUB_ROWS;
        const int rows = min(SUB_ROWS, P - r0);
        float x[EPT];
        RowCtx ctx;
        const bool valid = subtile_scores<ACT, EPT>(s_wave + buf * tile_floats, extra, (long long)b * P + r0, rows, C,
                                                    g.shift, x, ctx);
        f(x, valid, r0, ctx);
        // the wave's part of the next sub-tile has landed, and nothing of sub-tile s is read from LDS behind this point
        // (the builtin, not inline assembly: the compiler's own wait-count bookkeeping then knows that no LDS-DMA is pending and
        // does not wait again in front of later LDS accesses)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0) lgkmcnt(0)
        asm volatile("" ::: "memory");
        after(s);                 // what it issues has the whole next sub-tile to complete before the wave waits again
        g = gn;
        buf ^= 1;
    }
}

template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void activate_scores_kernel(const float *__restrict__ in,
                                                                     const float *__restrict__ extra, SelParams sp,
                                                                     const uint32_t *__restrict__ flag,
                                                                     float *__restrict__ out) {
    extern __shared__ float s_dynf[];
    if (flag && *flag == 0u) return;
    const int b = blockIdx.y;
    const long long sub0 = (long long)blockIdx.x * sp.subs_per_block;
    const long long sub1 = min((long long)sp.n_sub, sub0 + sp.subs_per_block);
    float *o = out + (size_t)b * sp.P * sp.C;
    const int C = sp.C;
    const int r = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    for_each_subtile<ACT, EPT>(in, extra, b, sp.P, C, sp.total_floats, sub0, sub1, 1, s_dynf, sp.tile_floats,
                               [&](float (&x)[EPT], bool valid, int r0, const RowCtx &) {
                                   if (!valid) return;
                                   float *orow = o + (size_t)(r0 + r) * C + sub;
#pragma unroll
                                   for (int i = 0; i < EPT; ++i)
                                       if (sub + LPR * i < C) orow[LPR * i] = x[i];
                               },
                               [](long long) {});
}

// ---------------------------------------------------------------------------------------------
// Focal loss over the anchors that take part in the confidence loss (label >= 0), summed; replaces the
// boolean-mask gather + softmax + log_softmax + nll_loss of src/models/losses.py:5-9 as called at
// src/models/retinanet.py:215 and src/models/fcos.py:271:
//     loss_row = -((alpha * (1 - pt)^gamma) * log_pt),  pt = softmax(x)[t],  log_pt = (x_t - max) - log(sum)
// The rows stream through the same LDS sub-tiles as the selection; the [rows, C] gather never exists.
// The sum is accumulated in double.  Backward: d loss_row / d x_j = dL/dlog_pt * ((j == t) - p_j) with
// dL/dlog_pt = -alpha * ((1 - pt)^gamma - gamma * log_pt * pt * (1 - pt)^(gamma - 1)); rows that do not
// take part get zero gradients.
// ---------------------------------------------------------------------------------------------

__device__ inline float pow_gamma(float base, float gamma) {
    if (gamma == 2.0f) return base * base;                 // what torch's pow does for an exponent of 2
    if (gamma == 1.0f) return base;
    if (gamma == 0.0f) return 1.0f;
    return powf(base, gamma);
}

template <int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void focal_forward_kernel(const float *__restrict__ in,
                                                                   const int64_t *__restrict__ labels, int rows, int C,
                                                                   int subs_per_block, int n_sub, int tile_floats,
                                                                   float alpha, float gamma, double *__restrict__ loss_sum) {
    extern __shared__ float s_dynf[];
    __shared__ double s_part[SCAN_BLOCK / 64];
    const int tid = threadIdx.x;
    const int r = tid / LPR, sub = tid % LPR;
    const long long sub0 = (long long)blockIdx.x * subs_per_block;
    const long long sub1 = min((long long)n_sub, sub0 + subs_per_block);
    double acc = 0.0;
    for_each_subtile<ACT_SOFTMAX, EPT>(in, nullptr, 0, rows, C, (long long)rows * C, sub0, sub1, 1, s_dynf, tile_floats,
                                       [&](float (&x)[EPT], bool valid, int r0, const RowCtx &ctx) {
                                           if (!valid) return;
                                           const long long t = labels[r0 + r];
                                           if (t < 0 || t >= C || (int)(t % LPR) != sub) return;   // one lane per row
                                           const float d = ctx.row[LPR * (int)(t / LPR)] - ctx.m;
                                           const float pt = __expf(d) * ctx.inv;
                                           const float log_pt = d - logf(ctx.sum);
                                           acc += (double)(-((alpha * pow_gamma(1.0f - pt, gamma)) * log_pt));
                                       },
                                       [](long long) {});
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += s_part[w];
        if (t != 0.0) atomicAdd(loss_sum, t);
    }
}

template <int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void focal_backward_kernel(const float *__restrict__ in,
                                                                    const int64_t *__restrict__ labels, int rows, int C,
                                                                    int subs_per_block, int n_sub, int tile_floats,
                                                                    float alpha, float gamma,
                                                                    const float *__restrict__ grad_scale,
                                                                    float *__restrict__ grad) {
    extern __shared__ float s_dynf[];
    const int tid = threadIdx.x;
    const int r = tid / LPR, sub = tid % LPR;
    const long long sub0 = (long long)blockIdx.x * subs_per_block;
    const long long sub1 = min((long long)n_sub, sub0 + subs_per_block);
    const float gs = *grad_scale;
    for_each_subtile<ACT_SOFTMAX, EPT>(in, nullptr, 0, rows, C, (long long)rows * C, sub0, sub1, 1, s_dynf, tile_floats,
                                       [&](float (&x)[EPT], bool valid, int r0, const RowCtx &ctx) {
                                           if (!valid) return;
                                           const long long t = labels[r0 + r];
                                           float *orow = grad + (size_t)(r0 + r) * C + sub;
                                           const bool takes_part = t >= 0 && t < C;
                                           float g = 0.0f;                       // gs * dL/dlog_pt, the same in the row's lanes
                                           if (takes_part) {
                                               const float d = ctx.row[LPR * (int)(t / LPR) - sub + (int)(t % LPR)] - ctx.m;
                                               const float pt = __expf(d) * ctx.inv;
                                               const float log_pt = d - logf(ctx.sum);
                                               const float om = 1.0f - pt;
                                               const float dw = gamma == 0.0f ? 0.0f : gamma * pow_gamma(om, gamma - 1.0f);
                                               g = gs * (-alpha * (pow_gamma(om, gamma) - (log_pt * pt) * dw));
                                           }
#pragma unroll
                                           for (int i = 0; i < EPT; ++i) {
                                               const int c = sub + LPR * i;
                                               if (c < C) orow[LPR * i] = takes_part ? g * ((c == t ? 1.0f : 0.0f) - x[i]) : 0.0f;
                                           }
                                       },
                                       [](long long) {});
}

// ---------------------------------------------------------------------------------------------
// Fast path
// ---------------------------------------------------------------------------------------------

// Activates every sample_rate-th sub-tile and writes its scores class-major:
// samp[(b*Cs + ci) * ns_cap + q*64 + local_row], q = ordinal of the sampled sub-tile.
template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void select_sample_kernel(const float *__restrict__ in,
                                                                   const float *__restrict__ extra, SelParams sp,
                                                                   float *__restrict__ samp) {
    extern __shared__ float s_dynf[];
    const int b = blockIdx.y;
    const int cb = sp.cb, C = sp.C;
    const int r = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    const long long first = (long long)blockIdx.x * SAMPLE_SUBS * sp.sample_rate;
    const long long last = min((long long)sp.n_sub, first + (long long)SAMPLE_SUBS * sp.sample_rate);
    float *sb = samp + (size_t)b * sp.Cs * sp.ns_cap;
    const int rate = sp.sample_rate, ns_cap = sp.ns_cap;
    for_each_subtile<ACT, EPT>(in, extra, b, sp.P, C, sp.total_floats, first, last, rate, s_dynf, sp.tile_floats,
                               [&](float (&x)[EPT], bool valid, int r0, const RowCtx &) {
                                   if (!valid) return;
                                   const int q = (r0 / SUB_ROWS) / rate;
                                   float *o = sb + (size_t)q * SUB_ROWS + r;
#pragma unroll
                                   for (int i = 0; i < EPT; ++i) {
                                       const int c = sub + LPR * i;
                                       if (c >= cb && c < C) o[(size_t)(c - cb) * ns_cap] = x[i];
                                   }
                               },
                               [](long long) {});
}

// Suffix sums of a 256-bin histogram row held four bins per lane; returns the lane's suffix sum.
__device__ inline uint32_t suffix_scan64(uint32_t t, int lane) {
    (void)lane;
    const uint32_t inc = wave_inclusive_add(t);                   // whole wave active
    return (uint32_t)__builtin_amdgcn_readlane((int)inc, 63) - inc + t;
}

// Finds the highest bin beta with count(bins >= beta) >= k.  Needs total >= k.  All lanes return the same.
__device__ inline void find_kth_bin(uint4 q, uint32_t suf, uint32_t t, int lane, uint32_t k, uint32_t &beta,
                                    uint32_t &n_above, uint32_t &n_beta) {
    const unsigned long long ge = __ballot(suf >= k);
    const int L = 63 - __clzll((long long)ge);
    uint32_t b_ = 0, a_ = 0, nb_ = 0;
    if (lane == L) {
        uint32_t acc = suf - t;
        const uint32_t hb[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if (acc + hb[i] >= k) { b_ = 4 * L + i; a_ = acc; nb_ = hb[i]; break; }
            acc += hb[i];
        }
    }
    beta = (uint32_t)__builtin_amdgcn_readlane((int)b_, L);      // L is the same in every lane
    n_above = (uint32_t)__builtin_amdgcn_readlane((int)a_, L);
    n_beta = (uint32_t)__builtin_amdgcn_readlane((int)nb_, L);
}

// Per (image, class): the sample_m-th largest sampled score, found exactly by a 4-round radix search
// on the monotone uint32 image of the scores; it becomes the staging threshold t_lo of the scan.
// With no more than sample_m sampled scores above the threshold, everything above it is staged.
// BLK = 64: one wave per (image, class) -- its barriers cost nothing, which is what pays when there are many
// short sample lists (thousands of segments); BLK = 256 shares a long list between four waves.
static_assert(NB == 256, "the histogram of the radix round doubles as the 256-entry gather buffer");
template <int BLK>
__global__ __launch_bounds__(BLK) void select_plan_kernel(const float *__restrict__ samp, SelParams sp,
                                                          SelRule *__restrict__ rules, uint32_t *__restrict__ flag,
                                                          uint32_t *__restrict__ gcnt, uint32_t *__restrict__ img_ovf,
                                                          uint32_t *__restrict__ ghist) {
    extern __shared__ __align__(16) uint32_t s_key[];  // [ns_cap] keys of the sampled scores above the threshold (0 = none)
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_pref, s_rem, s_total, s_max;
    const int ci = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    SelRule r;
    r.pad = 0; r.cutoff = 0ull; r.n_above = 0; r.n_sel = 0;
    // first kernel of a call: the counters and flags the later kernels add to start from zero
    if (tid == 0) {
        gcnt[((size_t)b * sp.Cs + ci) * CNT_STRIDE] = 0u;
        if (ci == 0) img_ovf[b] = 0u;
        if (ci == 0 && b == 0) *flag = sp.use_sample ? 0u : 1u;
    }
    if (!sp.use_sample) {
        r.mode = RULE_PENDING; r.beta = 0; r.min_score = __builtin_inff();
        if (tid == 0) rules[(size_t)b * sp.Cs + ci] = r;
        for (int i = tid; i < NB; i += BLK) ghist[((size_t)b * sp.Cs + ci) * NB + i] = 0u;
        return;
    }
    if (tid == 0) { s_total = 0u; s_pref = 0u; s_rem = (uint32_t)sp.sample_m; s_max = 0u; }
    __syncthreads();
    const float *sv = samp + ((size_t)b * sp.Cs + ci) * sp.ns_cap;
    uint32_t mine = 0, kmax = 0;
    // 16-byte loads, two in flight per thread: the list is read once and the read is all latency
    // (ns_cap is a multiple of SUB_ROWS, the list starts on a 256-byte boundary, rows beyond P hold no sample)
#pragma unroll 2
    for (int i = tid * 4; i < sp.ns_cap; i += BLK * 4) {
        const long long row = (long long)(i / SUB_ROWS) * sp.sample_rate * SUB_ROWS + (i % SUB_ROWS);
        const float4 v4 = row < sp.P ? *reinterpret_cast<const float4 *>(sv + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        uint32_t k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            k[q] = 0u;
            if (row + q < sp.P && v[q] > sp.thr) { k[q] = okey(v[q]) - sp.key_thr; ++mine; }     // k >= 1
            kmax = max(kmax, k[q]);
        }
        *reinterpret_cast<uint4 *>(s_key + i) = make_uint4(k[0], k[1], k[2], k[3]);
    }
    if (mine) { atomicAdd(&s_total, mine); atomicMax(&s_max, kmax); }
    __syncthreads();
    r.mode = RULE_STAGED;
    unsigned long long dmin = 1ull;
    if (s_total > (uint32_t)sp.sample_m) {              // block-uniform
        // radix search from the highest bit any key has, eight bits a round
        int width = 32 - __clz((int)s_max);            // s_max >= 1 here
        while (width > 0) {
            const int bits = width < 8 ? width : 8;
            const int shift = width - bits;
            const uint32_t pref = s_pref;
            for (int i = tid; i < 256; i += BLK) s_h[i] = 0u;
            __syncthreads();
            for (int i = tid; i < sp.ns_cap; i += BLK) {
                const uint32_t k = s_key[i];
                if (k && (width == 32 || (k >> width) == (pref >> width))) atomicAdd(&s_h[(k >> shift) & ((1u << bits) - 1u)], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                const uint4 q = reinterpret_cast<const uint4 *>(s_h)[lane];
                const uint32_t t = q.x + q.y + q.z + q.w;
                const uint32_t suf = suffix_scan64(t, lane);
                uint32_t digit, above, nd;
                find_kth_bin(q, suf, t, lane, s_rem, digit, above, nd);
                if (lane == 0) { s_pref = pref | (digit << shift); s_rem -= above; s_total = nd; s_max = 0u; }
            }
            __syncthreads();
            width = shift;
            const uint32_t nd = s_total;
            if (width > 0 && nd <= 256u) {
                // Few keys are left in the chosen bin: gather them into the histogram array and rank them by
                // counting, instead of sweeping all the keys once more per remaining digit.
                const uint32_t want = s_pref >> width, rem = s_rem;
                uint32_t nf = 0;
                for (int i = tid; i < sp.ns_cap; i += BLK) {
                    const uint32_t k = s_key[i];
                    nf += (k != 0u && (k >> width) == want);
                }
                uint32_t pos = nf ? atomicAdd(&s_max, nf) : 0u;
                if (nf)
                    for (int i = tid; i < sp.ns_cap; i += BLK) {
                        const uint32_t k = s_key[i];
                        if (k != 0u && (k >> width) == want) s_h[pos++] = k;
                    }
                __syncthreads();
                for (uint32_t e = (uint32_t)tid; e < nd; e += BLK) {
                    const uint32_t mine_k = s_h[e];
                    uint32_t rk = 0;
                    for (uint32_t j = 0; j < nd; ++j) {
                        const uint32_t v = s_h[j];
                        rk += (v > mine_k || (v == mine_k && j < e));
                    }
                    if (rk == rem - 1u) s_pref = mine_k;                    // rem <= nd: exactly one entry
                }
                __syncthreads();
                width = 0;
            }
        }
        dmin = s_pref;
    }
    r.beta = dmin == 1ull ? 1u : 0u;            // 1: everything above the threshold gets staged
    r.min_score = score_from_d(sp.key_thr, dmin);
    if (tid == 0) rules[(size_t)b * sp.Cs + ci] = r;
}

// Staged candidate as the scan writes it: score key (32) | class slot (8) | anchor (24).
__device__ inline unsigned long long pack_entry(uint32_t d, uint32_t ci, uint32_t p) {
    return ((unsigned long long)d << 32) | ((unsigned long long)ci << 24) | (unsigned long long)p;
}

template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void select_scan_kernel(const float *__restrict__ in,
                                                                 const float *__restrict__ extra, SelParams sp,
                                                                 const SelRule *__restrict__ rules,
                                                                 uint32_t *__restrict__ gcnt,
                                                                 unsigned long long *__restrict__ staging) {
    extern __shared__ float s_dynf[];
    // The workgroups of an image share an XCD: their partial writes to the image's staging lists meet in one L2 (and fill
    // whole lines there before they leave it), the per-class counters live in that L2, and select_finalize, mapped the
    // same way, finds the lists where they were written.
    const TileOfImage wg = xcd_affine_tile();
    const int b = wg.y, tid = threadIdx.x, lane = tid & 63;
    const int Cs = sp.Cs, cb = sp.cb, C = sp.C;
    const uint32_t key_thr = sp.key_thr;
    const SelRule *rb = rules + (size_t)b * Cs;
    // Every wave collects its candidates in an LDS list of its own (no atomics, a wave-uniform count) and hands full lists
    // straight to the per-(image, class) staging lists: the entries of a list are counted per class in LDS, one global
    // atomic per class present reserves their places, and the entries are scattered there.  The hand-over is split over
    // two sub-tiles -- the atomics are issued behind the wave's wait for one sub-tile's DMA, their results used behind its
    // wait for the next, which they have returned by -- and the wave fills its second list meanwhile.
    const int wv = tid >> 6;
    const uint32_t ebuf_n = (uint32_t)sp.ebuf;            // entries per list: 256, or 128 where the sub-tiles are small and
                                                          // more workgroups share a CU
    const bool two_rows = Cs > 64;                        // classes 64.. exist: a lane looks after class lane and class lane + 64
    unsigned long long *ebuf0 = reinterpret_cast<unsigned long long *>(s_dynf + 2 * (size_t)sp.tile_floats + 4) + (size_t)wv * 2 * ebuf_n;
    unsigned long long *ebuf = ebuf0, *pbuf = ebuf0 + ebuf_n;
    // per wave, behind the lists: entries per class of the list being handed over, then their bases
    uint32_t *wc = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned long long *>(s_dynf + 2 * (size_t)sp.tile_floats + 4) + (size_t)SCAN_WAVES * 2 * ebuf_n)
                   + (size_t)wv * (two_rows ? 128 : 64);
    unsigned long long *stg = staging + (size_t)b * Cs * sp.cap;
    uint32_t *gc = gcnt + (size_t)b * Cs * CNT_STRIDE;
    const uint32_t cap = (uint32_t)sp.cap;
    uint32_t ln = 0, pn = 0;                              // entries waiting in the current list; entries of the list being handed over
    uint32_t pmeta[SCAN_EBUF / 64], gbase0 = 0, gbase1 = 0;
    // Both halves are written without a branch around their LDS accesses (a lane beyond the list's end reads a slot of the
    // list all the same and adds zero to counter 0): the four rounds' reads, and then their atomics, are independent
    // instructions in flight together -- one LDS round trip per step instead of one per round and step.
    auto flush_begin = [&](uint32_t &g0, uint32_t &g1) {  // pn == 0; g0 / g1 receive the bases of the lane's classes
        wc[lane] = 0u;
        if (two_rows) wc[lane + 64] = 0u;
        uint32_t ci[SCAN_EBUF / 64], old[SCAN_EBUF / 64];
#pragma unroll
        for (int j = 0; j < SCAN_EBUF / 64; ++j) {
            const uint32_t k = (uint32_t)lane + 64u * j;                  // (k < 256: inside the wave's two lists either way)
            const uint32_t lo = reinterpret_cast<const uint32_t *>(ebuf)[2u * k];
            ci[j] = k < ln ? (lo >> 24) : 0u;
        }
#pragma unroll
        for (int j = 0; j < SCAN_EBUF / 64; ++j)
            old[j] = atomicAdd(&wc[ci[j]], (uint32_t)lane + 64u * j < ln ? 1u : 0u);   // place among the list's entries of that class
#pragma unroll
        for (int j = 0; j < SCAN_EBUF / 64; ++j)
            pmeta[j] = (uint32_t)lane + 64u * j < ln ? (ci[j] << 16) | old[j] : 0u;
        const uint32_t c0 = wc[lane], c1 = two_rows ? wc[lane + 64] : 0u;
        g0 = c0 ? atomicAdd(&gc[lane * CNT_STRIDE], c0) : 0u;             // c0 != 0 only for lane < Cs
        g1 = c1 ? atomicAdd(&gc[(lane + 64) * CNT_STRIDE], c1) : 0u;
        unsigned long long *t = ebuf; ebuf = pbuf; pbuf = t;
        pn = ln;
        ln = 0u;
    };
    auto flush_end = [&](const uint32_t g0, const uint32_t g1) {
        if (pn == 0u) return;
        wc[lane] = g0;
        if (two_rows) wc[lane + 64] = g1;
        unsigned long long en[SCAN_EBUF / 64];
        uint32_t base[SCAN_EBUF / 64];
#pragma unroll
        for (int j = 0; j < SCAN_EBUF / 64; ++j) {
            en[j] = pbuf[(uint32_t)lane + 64u * j];
            base[j] = wc[pmeta[j] >> 16];
        }
#pragma unroll
        for (int j = 0; j < SCAN_EBUF / 64; ++j) {
            const uint32_t k = (uint32_t)lane + 64u * j;
            const uint32_t ci = pmeta[j] >> 16;
            const uint32_t pos = base[j] + (pmeta[j] & 0xffffu);
            const uint32_t p = (uint32_t)en[j] & 0xffffffu;
            // a list that overflows keeps counting: select_check sees the count and hands the pair to the exact path
            if (k < pn && pos < cap) stg[(size_t)ci * cap + pos] = (en[j] & 0xffffffff00000000ull) | (unsigned long long)(0xffffffffu - p);
        }
        pn = 0u;
    };
    // Inside a sub-tile (a list that cannot take the sub-tile's entries: rare): completes the hand-over in flight and makes
    // one of its own, waiting for its atomics.  It writes to bases of its own: gbase0 / gbase1, which an atomic in flight
    // may still be about to write, are assigned behind the wave's wait for its DMA only -- were they assigned here as
    // well, the copies the compiler then makes of them on the way into this code would each wait for every load in flight,
    // the sub-tile's DMA included, in every sub-tile.
    auto flush = [&]() {
        flush_end(gbase0, gbase1);
        uint32_t t0 = 0u, t1 = 0u;
        flush_begin(t0, t1);
        flush_end(t0, t1);
    };
    // the lane's classes never change: their staging thresholds live in registers (fetched into LDS by the first Cs
    // threads while the first sub-tile's DMA is in flight, then picked up by every lane)
    const int r = tid / LPR, sub = tid % LPR;
    float tlo[EPT];
    __shared__ float s_tlo[MAX_SEL_CLASSES];
    auto fetch_thresholds = [&]() {              // one round of loads for the whole workgroup
        if (tid < Cs) s_tlo[tid] = rb[tid].mode == RULE_STAGED ? rb[tid].min_score : __builtin_inff();
    };
    auto load_thresholds = [&]() {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int c = sub + LPR * i;
            tlo[i] = (c >= cb && c < C) ? s_tlo[c - cb] : __builtin_inff();
        }
    };

    // Workgroup j of an image takes sub-tiles j, j + gx, j + 2 gx, ...: at any moment the workgroups of an image read
    // one compact window of it (3 % faster than a contiguous run per workgroup).
    for_each_subtile<ACT, EPT>(
        in, extra, b, sp.P, C, sp.total_floats, (long long)wg.x, (long long)sp.n_sub, (long long)gridDim.x, s_dynf, sp.tile_floats,
        [&](float (&x)[EPT], bool valid, int r0, const RowCtx &ctx) {
            // bit i: element i is staged.  Built high element first with compare-into-VCC + add-with-carry
            // (pm = 2 * pm + (x >= t)): two VALU per element, no select chains
            uint32_t pm_lo = 0u, pm_hi = 0u;
#pragma unroll
            for (int i = EPT - 1; i >= 0; --i) {
                if (i >= 32)
                    asm volatile("v_cmp_ge_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                                 : "+v"(pm_hi) : "v"(x[i]), "v"(tlo[i]) : "vcc");
                else
                    asm volatile("v_cmp_ge_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                                 : "+v"(pm_lo) : "v"(x[i]), "v"(tlo[i]) : "vcc");
            }
            const uint32_t p = (uint32_t)(r0 + r);
            if (!valid) { pm_lo = 0u; pm_hi = 0u; }
            // Every lane's entries get consecutive slots of the wave's LDS buffer: popcount per lane, prefix sum over the wave
            // (the order of the entries inside the list is irrelevant), then each lane writes its own entries.  A global store
            // inside the loop would be waited for with the sub-tile's DMA (the counter that tells the DMA has landed counts
            // stores as well), once per sub-tile; the buffer is flushed once per dozen sub-tiles instead.
            uint32_t cnt = (uint32_t)__popc(pm_lo) + (EPT > 32 ? (uint32_t)__popc(pm_hi) : 0u);
            if (__ballot(cnt != 0u) != 0ull) {
                const uint32_t inc = wave_inclusive_add(cnt);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if (total <= ebuf_n) {
                    if (ln + total > ebuf_n) flush();
                    uint32_t slot = ln + inc - cnt;
                    auto put = [&](int i) {
                        const float sc = rescore<ACT>(ctx, i);               // same arithmetic as x[i], bit for bit
                        ebuf[slot++] = pack_entry(okey(sc) - key_thr, (uint32_t)(sub + LPR * i - cb), p);
                    };
                    while (pm_lo) {
                        const int i = __ffs((int)pm_lo) - 1;
                        pm_lo &= pm_lo - 1u;
                        put(i);
                    }
                    if (EPT > 32)
                        while (pm_hi) {
                            const int i = __ffs((int)pm_hi) - 1;
                            pm_hi &= pm_hi - 1u;
                            put(32 + i);
                        }
                    ln += total;
                } else {
                    // more entries than the buffer holds (nearly every score of the sub-tile is staged): round k appends the
                    // k-th entry of every lane that still has one
                    unsigned long long pm = ((unsigned long long)pm_hi << 32) | pm_lo;
                    unsigned long long has = __ballot(pm != 0ull);
                    while (has) {
                        if (ln + 64u > ebuf_n) flush();
                        if (pm) {
                            const uint32_t slot = ln + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u));
                            const int i = __ffsll((long long)pm) - 1;
                            pm &= pm - 1ull;
                            const float sc = rescore<ACT>(ctx, i);
                            ebuf[slot] = pack_entry(okey(sc) - key_thr, (uint32_t)(sub + LPR * i - cb), p);
                        }
                        ln += (uint32_t)__popcll(has);
                        has = __ballot(pm != 0ull);
                    }
                }
            }
        },
        // behind the wave's wait: the atomics and the stores have the whole next sub-tile to complete before it waits again
        [&](long long) {
            flush_end(gbase0, gbase1);
            if (ln >= (uint32_t)sp.flush_at) flush_begin(gbase0, gbase1);
        }, fetch_thresholds, load_thresholds);
    flush();
}

__global__ __launch_bounds__(128) void select_check_kernel(SelParams sp, SelRule *__restrict__ rules,
                                                           uint32_t *__restrict__ gcnt, const uint32_t *__restrict__ img_ovf,
                                                           uint32_t *__restrict__ flag, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t s_pending[MAX_SEL_CLASSES];
    const int b = blockIdx.x, ci = threadIdx.x;
    bool pending = false;
    if (ci < sp.Cs) {
        SelRule *r = rules + (size_t)b * sp.Cs + ci;
        if (r->mode == RULE_STAGED) {
            const uint32_t n = gcnt[((size_t)b * sp.Cs + ci) * CNT_STRIDE];
            const bool ok = !img_ovf[b] && n <= (uint32_t)sp.cap && (r->beta != 0u || n >= (uint32_t)sp.top_k);
            if (ok) {
                r->n_sel = n < (uint32_t)sp.top_k ? n : (uint32_t)sp.top_k;
            } else {
                r->mode = RULE_PENDING;
                gcnt[((size_t)b * sp.Cs + ci) * CNT_STRIDE] = 0u;
                atomicOr(flag, 1u);
                pending = true;
            }
        }
    }
    s_pending[threadIdx.x] = pending ? 1u : 0u;
    if (!__syncthreads_or(pending ? 1 : 0)) return;
    // a segment handed to the exact path starts from an empty histogram
    for (int c = 0; c < sp.Cs; ++c)
        if (s_pending[c]) {
            uint32_t *h = ghist + ((size_t)b * sp.Cs + c) * NB;
            for (int i = threadIdx.x; i < NB; i += 128) h[i] = 0u;
        }
}

// ---------------------------------------------------------------------------------------------
// Exact two-pass fallback (works on activated scores)
// ---------------------------------------------------------------------------------------------

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
                                                                const SelRule *__restrict__ rules,
                                                                const uint32_t *__restrict__ flag,
                                                                uint32_t *__restrict__ ghist) {
    extern __shared__ uint32_t s_hist[];   // [Cs][NB/2] packed 16-bit counters, then [Cs] pending flags
    if (*flag == 0u) return;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * sp.tile_anchors;
    const int p1 = min(sp.P, p0 + sp.tile_anchors);
    const int words = sp.Cs * (NB / 2);
    uint32_t *s_pending = s_hist + words;
    for (int i = threadIdx.x; i < words; i += blockDim.x) s_hist[i] = 0u;
    for (int i = threadIdx.x; i < sp.Cs; i += blockDim.x)
        s_pending[i] = rules[(size_t)b * sp.Cs + i].mode == RULE_PENDING ? 1u : 0u;
    __syncthreads();
    const float *base = scores + ((size_t)b * sp.P + p0) * sp.C;
    const float thr = sp.thr;
    const uint32_t key_thr = sp.key_thr;
    const int shift = sp.shift, cb = sp.cb;
    walk_tile(base, (p1 - p0) * sp.C, sp.C, [&](float s, int c, int) {
        if (c >= cb && s > thr && s_pending[c - cb]) {
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

// Always runs: turns every PENDING rule into ALL / HIST / EXACT from the full histogram, then lays
// out the class segments of the output.
__global__ __launch_bounds__(256) void select_kth_kernel(const uint32_t *__restrict__ ghist, SelParams sp,
                                                         SelRule *__restrict__ rules, int32_t *__restrict__ seg_off,
                                                         int32_t *__restrict__ counts) {
    __shared__ int s_nsel[MAX_SEL_CLASSES];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t k = (uint32_t)sp.top_k;
    __shared__ uint32_t s_pending[MAX_SEL_CLASSES];
    if (threadIdx.x < sp.Cs) {                       // Cs <= MAX_SEL_CLASSES <= blockDim
        const SelRule *rp = rules + (size_t)b * sp.Cs + threadIdx.x;
        const uint32_t mode = rp->mode;
        s_pending[threadIdx.x] = mode == RULE_PENDING ? 1u : 0u;
        s_nsel[threadIdx.x] = (int)rp->n_sel;
    }
    __syncthreads();
    for (int ci = wave; ci < sp.Cs; ci += nwaves) {
        if (!s_pending[ci]) continue;
        SelRule *rp = rules + (size_t)b * sp.Cs + ci;
        const uint32_t *h = ghist + ((size_t)b * sp.Cs + ci) * NB;
        const uint4 q = reinterpret_cast<const uint4 *>(h)[lane];       // bins 4*lane .. 4*lane+3
        const uint32_t t = q.x + q.y + q.z + q.w;
        const uint32_t suf = suffix_scan64(t, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)suf);
        SelRule r;
        r.pad = 0; r.cutoff = 0ull; r.beta = 0; r.n_above = 0;
        if (total <= k) {
            r.mode = RULE_ALL; r.n_sel = total;
            r.min_score = score_from_d(sp.key_thr, 1ull);
        } else {
            uint32_t beta, n_above, n_beta;
            find_kth_bin(q, suf, t, lane, k, beta, n_above, n_beta);
            r.beta = beta; r.n_above = n_above; r.n_sel = k;
            r.mode = (n_above + n_beta <= (uint32_t)sp.cap) ? RULE_HIST : RULE_EXACT;
            unsigned long long dmin = (unsigned long long)beta << sp.shift;
            if (dmin < 1ull) dmin = 1ull;
            r.min_score = score_from_d(sp.key_thr, dmin);
        }
        if (lane == 0) { *rp = r; s_nsel[ci] = (int)r.n_sel; }
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
// list can take (massive ties).  One block per pair walks its (strided) score column once per 8 key bits.
__global__ __launch_bounds__(256) void select_refine_kernel(const float *__restrict__ scores, SelParams sp,
                                                            SelRule *__restrict__ rules,
                                                            const uint32_t *__restrict__ flag) {
    __shared__ uint32_t s_h[256];
    __shared__ unsigned long long s_lo, s_hi;
    __shared__ uint32_t s_need;
    if (*flag == 0u) return;
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
                const unsigned long long K = make_key(s, sp.key_thr, (uint32_t)p);
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
                                                                 const uint32_t *__restrict__ flag,
                                                                 uint32_t *__restrict__ cnt,
                                                                 unsigned long long *__restrict__ staging) {
    __shared__ float s_min[MAX_SEL_CLASSES];
    if (*flag == 0u) return;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * sp.tile_anchors;
    const int p1 = min(sp.P, p0 + sp.tile_anchors);
    const SelRule *rb = rules + (size_t)b * sp.Cs;
    for (int i = threadIdx.x; i < sp.Cs; i += blockDim.x)
        s_min[i] = (rb[i].mode != RULE_STAGED && rb[i].n_sel) ? rb[i].min_score : __builtin_inff();
    __syncthreads();
    const float *base = scores + ((size_t)b * sp.P + p0) * sp.C;
    const int cb = sp.cb, C = sp.C, shift = sp.shift, cap = sp.cap;
    const uint32_t key_thr = sp.key_thr;
    walk_tile(base, (p1 - p0) * C, C, [&](float s, int c, int e) {
        if (c >= cb && s >= s_min[c - cb]) {
            const int ci = c - cb;
            const uint32_t p = (uint32_t)(p0 + e / C);
            const unsigned long long K = make_key(s, key_thr, p);
            const SelRule r = rb[ci];
            bool take = true;
            if (r.mode == RULE_HIST) take = bin_of((uint32_t)(K >> 32), shift) >= r.beta;
            else if (r.mode == RULE_EXACT) take = K >= r.cutoff;
            if (take) {
                const uint32_t slot = atomicAdd(&cnt[((size_t)b * sp.Cs + ci) * CNT_STRIDE], 1u);
                if (slot < (uint32_t)cap) staging[((size_t)b * sp.Cs + ci) * cap + slot] = K;
            }
        }
    });
}

// The same three passes when the activation is fused: the scores are not in memory, so every pass streams the head
// output through the scan's sub-tile pipeline and recomputes them -- with the scan's own arithmetic, bit for bit, which
// is what makes the counts of the passes agree with one another and with the scan's staged keys.

// lds_hist: the per-class counters are packed 16-bit pairs in LDS behind the two sub-tile buffers (a workgroup sees
// fewer than 65536 anchors); otherwise (very wide heads: the counters do not fit beside the sub-tiles) they are
// added to the global histogram directly.
template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void score_hist_fused_kernel(const float *__restrict__ in,
                                                                      const float *__restrict__ extra, SelParams sp,
                                                                      const SelRule *__restrict__ rules,
                                                                      const uint32_t *__restrict__ flag,
                                                                      uint32_t *__restrict__ ghist, int lds_hist) {
    extern __shared__ float s_dynf[];
    __shared__ uint32_t s_pending[MAX_SEL_CLASSES];
    if (*flag == 0u) return;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Cs = sp.Cs, cb = sp.cb, C = sp.C;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dynf + 2 * (size_t)sp.tile_floats);
    const int words = lds_hist ? Cs * (NB / 2) : 0;
    for (int i = tid; i < words; i += SCAN_BLOCK) s_hist[i] = 0u;
    for (int i = tid; i < Cs; i += SCAN_BLOCK) s_pending[i] = rules[(size_t)b * Cs + i].mode == RULE_PENDING ? 1u : 0u;
    __syncthreads();
    uint32_t *gh = ghist + (size_t)b * Cs * NB;
    const int sub = tid % LPR;
    const float thr = sp.thr;
    const uint32_t key_thr = sp.key_thr;
    const int shift = sp.shift;
    for_each_subtile<ACT, EPT>(
        in, extra, b, sp.P, C, sp.total_floats, (long long)blockIdx.x, (long long)sp.n_sub, (long long)gridDim.x, s_dynf, sp.tile_floats,
        [&](float (&x)[EPT], bool valid, int, const RowCtx &) {
            if (!valid) return;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = sub + LPR * i;
                if (c >= cb && c < C && x[i] > thr && s_pending[c - cb]) {
                    const uint32_t bin = bin_of(okey(x[i]) - key_thr, shift);
                    if (lds_hist) atomicAdd(&s_hist[(c - cb) * (NB / 2) + (bin >> 1)], 1u << ((bin & 1u) * 16u));
                    else atomicAdd(&gh[(size_t)(c - cb) * NB + bin], 1u);
                }
            }
        },
        [](long long) {});
    __syncthreads();
    for (int i = tid; i < words; i += SCAN_BLOCK) {
        const uint32_t v = s_hist[i];
        if (v) {
            const uint32_t lo = v & 0xffffu, hi = v >> 16;
            if (lo) atomicAdd(&gh[2 * i], lo);
            if (hi) atomicAdd(&gh[2 * i + 1], hi);
        }
    }
}

// Exact cut-off keys of the image's RULE_EXACT pairs: one workgroup per image streams the image once per 8 key bits
// and narrows the key interval of up to `group` pairs at a time (256 LDS counters each, behind the sub-tile buffers).
// Only massive ties at the boundary bring a pair here; one workgroup per image needs no synchronisation between the
// rounds, which is what keeps this a single launch.
template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void select_refine_fused_kernel(const float *__restrict__ in,
                                                                         const float *__restrict__ extra, SelParams sp,
                                                                         SelRule *__restrict__ rules,
                                                                         const uint32_t *__restrict__ flag, int group) {
    extern __shared__ float s_dynf[];
    __shared__ unsigned long long s_lo[MAX_SEL_CLASSES], s_hi[MAX_SEL_CLASSES];
    __shared__ uint32_t s_need[MAX_SEL_CLASSES];
    __shared__ int s_sh[MAX_SEL_CLASSES], s_slot[MAX_SEL_CLASSES], s_cls[MAX_SEL_CLASSES];
    __shared__ int s_n, s_active;
    if (*flag == 0u) return;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Cs = sp.Cs, cb = sp.cb, C = sp.C;
    SelRule *rb = rules + (size_t)b * Cs;
    if (tid == 0) {
        int n = 0;
        for (int ci = 0; ci < Cs; ++ci)
            if (rb[ci].mode == RULE_EXACT) s_cls[n++] = ci;
        s_n = n;
    }
    __syncthreads();
    const int n_exact = s_n;
    if (n_exact == 0) return;
    uint32_t *s_h = reinterpret_cast<uint32_t *>(s_dynf + 2 * (size_t)sp.tile_floats);
    const int r = tid / LPR, sub = tid % LPR;
    const float thr = sp.thr;
    const uint32_t key_thr = sp.key_thr;
    for (int g0 = 0; g0 < n_exact; g0 += group) {
        const int gn = min(group, n_exact - g0);
        for (int i = tid; i < Cs; i += SCAN_BLOCK) s_slot[i] = -1;
        __syncthreads();
        if (tid < gn) {
            const int ci = s_cls[g0 + tid];
            s_slot[ci] = tid;
            const uint32_t beta = rb[ci].beta;
            s_lo[tid] = (unsigned long long)beta << (sp.shift + 32);
            s_hi[tid] = (beta == (uint32_t)(NB - 1)) ? ~0ull : ((((unsigned long long)beta + 1ull) << (sp.shift + 32)) - 1ull);
            s_need[tid] = (uint32_t)sp.top_k - rb[ci].n_above;
        }
        __syncthreads();
        for (int iter = 0; iter < 16; ++iter) {            // 64 key bits / 8 per round, with slack
            if (tid == 0) s_active = 0;
            __syncthreads();
            if (tid < gn) {
                const unsigned long long lo = s_lo[tid], hi = s_hi[tid];
                int sh = -1;
                if (lo < hi) {
                    const int sb = 64 - __clzll((long long)(hi - lo));
                    sh = sb > 8 ? sb - 8 : 0;
                    s_active = 1;
                }
                s_sh[tid] = sh;
            }
            for (int i = tid; i < gn * 256; i += SCAN_BLOCK) s_h[i] = 0u;
            __syncthreads();
            if (!s_active) break;
            for_each_subtile<ACT, EPT>(
                in, extra, b, sp.P, C, sp.total_floats, 0ll, (long long)sp.n_sub, 1ll, s_dynf, sp.tile_floats,
                [&](float (&x)[EPT], bool valid, int r0, const RowCtx &) {
                    if (!valid) return;
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        const int c = sub + LPR * i;
                        if (c >= cb && c < C && x[i] > thr) {
                            const int slot = s_slot[c - cb];
                            if (slot >= 0 && s_sh[slot] >= 0) {
                                const unsigned long long K = make_key(x[i], key_thr, (uint32_t)(r0 + r));
                                const unsigned long long lo = s_lo[slot];
                                if (K >= lo && K <= s_hi[slot]) atomicAdd(&s_h[slot * 256 + (int)((K - lo) >> s_sh[slot])], 1u);
                            }
                        }
                    }
                },
                [](long long) {});
            __syncthreads();
            if (tid < gn && s_sh[tid] >= 0) {
                const uint32_t *h = s_h + tid * 256;
                const unsigned long long lo = s_lo[tid], hi = s_hi[tid];
                const int sh = s_sh[tid];
                const uint32_t need = s_need[tid];
                uint32_t acc = 0;
                int sbin = 0;
                for (int i = 255; i >= 0; --i) {
                    if (acc + h[i] >= need) { sbin = i; break; }
                    acc += h[i];
                }
                const unsigned long long nlo = lo + ((unsigned long long)sbin << sh);
                unsigned long long nhi = nlo + ((1ull << sh) - 1ull);
                if (nhi > hi || nhi < nlo) nhi = hi;
                s_lo[tid] = nlo; s_hi[tid] = nhi; s_need[tid] = need - acc;
            }
            __syncthreads();
        }
        if (tid < gn) {
            SelRule *rp = rb + s_cls[g0 + tid];
            rp->cutoff = s_lo[tid];
            rp->min_score = score_from_d(key_thr, s_lo[tid] >> 32);
        }
        __syncthreads();
    }
}

template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void select_emit_fused_kernel(const float *__restrict__ in,
                                                                       const float *__restrict__ extra, SelParams sp,
                                                                       const SelRule *__restrict__ rules,
                                                                       const uint32_t *__restrict__ flag,
                                                                       uint32_t *__restrict__ cnt,
                                                                       unsigned long long *__restrict__ staging) {
    extern __shared__ float s_dynf[];
    __shared__ float s_min[MAX_SEL_CLASSES];
    if (*flag == 0u) return;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Cs = sp.Cs, cb = sp.cb, C = sp.C, shift = sp.shift, cap = sp.cap;
    const SelRule *rb = rules + (size_t)b * Cs;
    for (int i = tid; i < Cs; i += SCAN_BLOCK)
        s_min[i] = (rb[i].mode != RULE_STAGED && rb[i].n_sel) ? rb[i].min_score : __builtin_inff();
    __syncthreads();
    const int r = tid / LPR, sub = tid % LPR;
    const uint32_t key_thr = sp.key_thr;
    for_each_subtile<ACT, EPT>(
        in, extra, b, sp.P, C, sp.total_floats, (long long)blockIdx.x, (long long)sp.n_sub, (long long)gridDim.x, s_dynf, sp.tile_floats,
        [&](float (&x)[EPT], bool valid, int r0, const RowCtx &) {
            if (!valid) return;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = sub + LPR * i;
                if (c >= cb && c < C && x[i] >= s_min[c - cb]) {
                    const int ci = c - cb;
                    const unsigned long long K = make_key(x[i], key_thr, (uint32_t)(r0 + r));
                    const SelRule rl = rb[ci];
                    bool take = true;
                    if (rl.mode == RULE_HIST) take = bin_of((uint32_t)(K >> 32), shift) >= rl.beta;
                    else if (rl.mode == RULE_EXACT) take = K >= rl.cutoff;
                    if (take) {
                        const uint32_t slot = atomicAdd(&cnt[((size_t)b * Cs + ci) * CNT_STRIDE], 1u);
                        if (slot < (uint32_t)cap) staging[((size_t)b * Cs + ci) * cap + slot] = K;
                    }
                }
            }
        },
        [](long long) {});
}

// ---------------------------------------------------------------------------------------------
// Finalize: exact top-k of the staged keys, anchor order, decode
// ---------------------------------------------------------------------------------------------

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

// Finalize: one workgroup per (image, class).  The staged keys live in registers (KPT per thread).
//   1. exact cut: the take-th largest key by an interval-narrowing radix search (256 buckets over
//      the current [lo,hi] key interval per round, LDS histogram); keys are distinct, so the search
//      ends as soon as the chosen bucket holds one key -- typically after two rounds;
//   2. anchor order: selected anchors set bits of a P-bit LDS bitmap; a popcount scan gives every
//      selected anchor its rank, i.e. its output row inside the class segment;
//   3. decode and write.
// BLK = 64: one wave per segment (no barrier ever waits), for thousands of short staging lists.
template <int DECODE, int KPT, int BLK>
__global__ __launch_bounds__(BLK) void select_finalize_kernel(
    SelParams sp, const SelRule *__restrict__ rules, const uint32_t *__restrict__ cnt,
    const unsigned long long *__restrict__ staging, const int32_t *__restrict__ seg_off,
    const float4 *__restrict__ locs, const float *__restrict__ anchors, int class_id_offset, int use_bitmap,
    float4 *__restrict__ out_boxes, float *__restrict__ out_scores, int64_t *__restrict__ out_cls) {
    extern __shared__ unsigned long long s_dyn64[];
    __shared__ uint32_t s_h[256];
    __shared__ unsigned long long s_lo, s_hi, s_cut;
    __shared__ uint32_t s_rem, s_state;
    __shared__ uint32_t s_wsum[BLK / 64];
    const TileOfImage wg = xcd_affine_tile();         // the classes of one image share an L2: its locs are gathered once
    const int ci = wg.x, b = wg.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const size_t bc = (size_t)b * sp.Cs + ci;
    const int n_sel = (int)rules[bc].n_sel;
    if (n_sel == 0) return;
    int n_stage = (int)cnt[bc * CNT_STRIDE];
    if (n_stage > sp.cap) n_stage = sp.cap;
    const int take = min(n_sel, n_stage);
    const unsigned long long *st = staging + bc * sp.cap;
    const size_t row0 = (size_t)b * ((size_t)sp.Cs * sp.top_k) + (size_t)seg_off[(size_t)b * (sp.Cs + 1) + ci];

    auto emit = [&](int rank, unsigned long long K) {
        const uint32_t p = 0xffffffffu - (uint32_t)(K & 0xffffffffull);
        const uint32_t d = (uint32_t)(K >> 32);
        const float4 l = locs[(size_t)b * sp.P + p];
        out_boxes[row0 + rank] = decode_box<DECODE>(anchors, l, p);
        out_scores[row0 + rank] = __uint_as_float(bits_from_okey(sp.key_thr + d));
        out_cls[row0 + rank] = (int64_t)(ci + sp.cb + class_id_offset);
    };

    if (use_bitmap) {
        unsigned long long k[KPT];                    // 0 marks an empty slot: every real key is >= 2^32
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = tid + j * BLK;
            k[j] = i < n_stage ? st[i] : 0ull;
        }
        unsigned long long cutoff = 0ull;
        if (n_stage > take) {                          // block-uniform
            if (tid == 0) { s_lo = ~0ull; s_hi = 0ull; s_rem = (uint32_t)take; s_state = 0u; }
            __syncthreads();
            unsigned long long mn = ~0ull, mx = 0ull;
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if (k[j]) { mn = k[j] < mn ? k[j] : mn; mx = k[j] > mx ? k[j] : mx; }
            if (mx) { atomicMin(&s_lo, mn); atomicMax(&s_hi, mx); }
            __syncthreads();
            for (int round = 0; round < 12; ++round) {
                const unsigned long long lo = s_lo, hi = s_hi;
                const unsigned long long span = hi - lo;
                const int sb = span ? 64 - __clzll((long long)span) : 0;
                const int sh = sb > 8 ? sb - 8 : 0;
                for (int i = tid; i < 256; i += BLK) s_h[i] = 0u;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < KPT; ++j)
                    if (k[j] >= lo && k[j] <= hi && k[j]) atomicAdd(&s_h[(uint32_t)((k[j] - lo) >> sh)], 1u);
                __syncthreads();
                if (wave == 0) {
                    const uint4 q = reinterpret_cast<const uint4 *>(s_h)[lane];
                    const uint32_t t = q.x + q.y + q.z + q.w;
                    const uint32_t suf = suffix_scan64(t, lane);
                    uint32_t digit, above, nd;
                    find_kth_bin(q, suf, t, lane, s_rem, digit, above, nd);
                    if (lane == 0) {
                        const unsigned long long nlo = lo + ((unsigned long long)digit << sh);
                        unsigned long long nhi = nlo + ((1ull << sh) - 1ull);
                        if (nhi > hi || nhi < nlo) nhi = hi;
                        s_lo = nlo; s_hi = nhi; s_rem -= above;
                        s_state = (nd == 1u || nlo == nhi) ? 1u : 0u;   // one key left in [nlo, nhi]: it is the cut
                    }
                }
                __syncthreads();
                if (s_state) break;
            }
            {
                const unsigned long long lo = s_lo, hi = s_hi;
#pragma unroll
                for (int j = 0; j < KPT; ++j)
                    if (k[j] >= lo && k[j] <= hi && k[j]) s_cut = k[j];
            }
            __syncthreads();
            cutoff = s_cut;
        }
        const int words = (sp.P + 63) >> 6;
        unsigned long long *s_bm = s_dyn64;                             // [words]
        uint32_t *s_wp = reinterpret_cast<uint32_t *>(s_bm + words);    // [words] exclusive popcount prefix
        for (int w = tid; w < words; w += BLK) s_bm[w] = 0ull;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            if (k[j] && k[j] >= cutoff) {
                const uint32_t p = 0xffffffffu - (uint32_t)(k[j] & 0xffffffffull);
                atomicOr(&s_bm[p >> 6], 1ull << (p & 63u));
            }
        }
        __syncthreads();
        // block-wide exclusive scan of per-word popcounts: each thread owns a contiguous run of words
        const int wpt = (words + BLK - 1) / BLK;
        const int w0 = tid * wpt, w1 = min(words, w0 + wpt);
        uint32_t local = 0;
        for (int w = w0; w < w1; ++w) local += (uint32_t)__popcll(s_bm[w]);
        const uint32_t inc = wave_inclusive_add(local);
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t base = inc - local;
        for (int w = 0; w < wave; ++w) base += s_wsum[w];
        for (int w = w0; w < w1; ++w) { s_wp[w] = base; base += (uint32_t)__popcll(s_bm[w]); }
        __syncthreads();
        // ranks first, then the winners go through LDS in rank order so that the output rows leave coalesced
        int rank[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rank[j] = -1;
            if (k[j] && k[j] >= cutoff) {
                const uint32_t p = 0xffffffffu - (uint32_t)(k[j] & 0xffffffffull);
                rank[j] = (int)(s_wp[p >> 6] + (uint32_t)__popcll(s_bm[p >> 6] & ((1ull << (p & 63u)) - 1ull)));
            }
        }
        __syncthreads();                                                // bitmap and prefix are dead from here
        unsigned long long *s_out = s_dyn64;                            // [take]
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (rank[j] >= 0 && rank[j] < take) s_out[rank[j]] = k[j];
        __syncthreads();
        for (int r = tid; r < take; r += BLK) emit(r, s_out[r]);
    } else {
        // very large P: sort the staged keys, cut, then sort the survivors by anchor index
        unsigned long long *s_k = s_dyn64;                              // [cap]
        const int n2 = pow2_at_least(n_stage);
        for (int i = tid; i < n2; i += blockDim.x) s_k[i] = i < n_stage ? st[i] : 0ull;
        __syncthreads();
        bitonic_desc(s_k, n2);
        const int m2 = pow2_at_least(take);
        for (int i = tid; i < m2; i += blockDim.x) {
            unsigned long long v = 0ull;
            if (i < take) { const unsigned long long K = s_k[i]; v = (K << 32) | (K >> 32); }
            s_k[i] = v;
        }
        __syncthreads();
        bitonic_desc(s_k, m2);
        for (int i = tid; i < take; i += blockDim.x) {
            const unsigned long long K2 = s_k[i];
            emit(i, (K2 << 32) | (K2 >> 32));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Direct path for small problems (P <= DIRECT_MAX_P anchors, a few million scores in all): two kernels
// instead of the dozen of the streaming path, whose fixed launch chain is what such a call would spend
// its time on.
//   direct_stats   one streaming pass (the same sub-tile machinery, hence the same score bits): the activated
//                  scores of the selected classes, written class-major so that a class's column is contiguous,
//                  and per (image, class) the number of scores above the threshold, which fixes the class
//                  segments of the output;
//   direct_select  one workgroup per (image, class): the class's P scores go into LDS as ordered keys, the
//                  top_k-th largest key is found by an exact 8-bit radix select (ties at the cut go to the
//                  lower anchor), the winners are ranked in anchor order by wave-local ballots, decoded and
//                  written.
// Exact for any input: no sampling, no capacities, no fallback.
// ---------------------------------------------------------------------------------------------

constexpr int DIRECT_MAX_P = 16384;
constexpr long long DIRECT_MAX_SCORES = 1ll << 23;

template <int ACT, int EPT>
__global__ __launch_bounds__(SCAN_BLOCK) void direct_stats_kernel(const float *__restrict__ in,
                                                                  const float *__restrict__ extra, SelParams sp,
                                                                  float *__restrict__ colscores,
                                                                  uint32_t *__restrict__ npos) {
    extern __shared__ float s_dynf[];
    __shared__ uint32_t s_cnt[MAX_SEL_CLASSES];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cb = sp.cb, C = sp.C;
    const int r = tid / LPR, sub = tid % LPR;
    const long long sub0 = (long long)blockIdx.x * sp.subs_per_block;
    const long long sub1 = min((long long)sp.n_sub, sub0 + sp.subs_per_block);
    if (tid < MAX_SEL_CLASSES) s_cnt[tid] = 0u;
    __syncthreads();
    uint32_t cnt[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) cnt[i] = 0u;
    const float thr = sp.thr;
    float *cs = colscores + (size_t)b * sp.Cs * sp.P;
    const size_t P = (size_t)sp.P;
    for_each_subtile<ACT, EPT>(in, extra, b, sp.P, C, sp.total_floats, sub0, sub1, 1, s_dynf, sp.tile_floats,
                               [&](float (&x)[EPT], bool valid, int r0, const RowCtx &) {
                                   if (!valid) return;
                                   float *o = cs + r0 + r;
#pragma unroll
                                   for (int i = 0; i < EPT; ++i) {
                                       const int c = sub + LPR * i;
                                       if (c >= cb && c < C) {
                                           o[(size_t)(c - cb) * P] = x[i];
                                           if (x[i] > thr) ++cnt[i];
                                       }
                                   }
                               },
                               [](long long) {});
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int c = sub + LPR * i;
        if (cnt[i] && c >= cb && c < C) atomicAdd(&s_cnt[c - cb], cnt[i]);
    }
    __syncthreads();
    if (tid < sp.Cs && s_cnt[tid]) atomicAdd(&npos[(size_t)b * sp.Cs + tid], s_cnt[tid]);
}

template <int DECODE, int BLK>
__global__ __launch_bounds__(BLK) void direct_select_kernel(
    const float *__restrict__ colscores, SelParams sp, const uint32_t *__restrict__ npos,
    const float4 *__restrict__ locs, const float *__restrict__ anchors, int class_id_offset,
    float4 *__restrict__ out_boxes, float *__restrict__ out_scores, int64_t *__restrict__ out_cls,
    int32_t *__restrict__ seg_off, int32_t *__restrict__ counts) {
    extern __shared__ uint32_t s_d[];                  // [P] ordered score keys above the threshold, 0 = not a candidate
    __shared__ __align__(16) uint32_t s_h[256];
    __shared__ uint32_t s_h16[16 * 257];
    __shared__ uint32_t s_lo, s_hi, s_state, s_need, s_row0;
    constexpr int W = BLK / 64;
    __shared__ uint32_t s_wgt[W], s_weq[W];
    const int ci = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = sp.P, Cs = sp.Cs;
    const uint32_t k = (uint32_t)sp.top_k;
    const uint32_t *np = npos + (size_t)b * Cs;
    // class segments of the image: class c contributes min(top_k, scores above the threshold) rows
    __shared__ uint32_t s_n[MAX_SEL_CLASSES];
    if (tid < Cs) s_n[tid] = min(k, np[tid]);              // all the counts in one round of loads
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0u;
        for (int t = 0; t < ci; ++t) acc += s_n[t];
        s_row0 = acc;
        s_lo = 0xffffffffu; s_hi = 0u; s_state = 0u; s_need = k;
        if (ci == 0) {
            int32_t *so = seg_off + (size_t)b * (Cs + 1);
            for (int t = 0; t < Cs; ++t) { so[t] = (int32_t)acc; acc += s_n[t]; }
            so[Cs] = (int32_t)acc;
            counts[b] = (int32_t)acc;
        }
    }
    __syncthreads();
    const uint32_t n_pos = np[ci];
    const uint32_t n_sel = min(k, n_pos);
    if (n_sel == 0u) return;                           // block-uniform
    // the class's scores as keys (eight loads in flight per thread: the column is read once and the read is all latency)
    const int c = ci + sp.cb;
    const float *col = colscores + ((size_t)b * Cs + ci) * P;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (int base = tid; base < P; base += BLK * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + u * BLK;
            v[u] = p < P ? col[p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + u * BLK;
            if (p < P) {
                const uint32_t d = v[u] > sp.thr ? okey(v[u]) - sp.key_thr : 0u;
                s_d[p] = d;
                if (d) { kmin = min(kmin, d); kmax = max(kmax, d); }
            }
        }
    }
    if (kmax) { atomicMin(&s_lo, kmin); atomicMax(&s_hi, kmax); }
    __syncthreads();
    // the top_k-th largest key d* and how many keys equal to it are taken (lowest anchors first): a radix search that
    // narrows the interval [lo, hi] of candidate keys, 256 bins over what is left of it per round (the scores of a class
    // crowd into a few exponents: bins over the whole key range would put nearly all of them into two or three).  It
    // ends when a bin is a single key value, or holds a single key -- typically after two rounds.
    uint32_t dstar = 0u, need_eq = 0u;
    if (n_pos > k) {
        for (int round = 0; round < 6; ++round) {
            // sixteen copies of the histogram (one per lane modulo 16, 257 words apart so that equal digits fall into
            // different banks) keep the LDS atomics from serialising
            for (int i = tid; i < 16 * 257; i += BLK) s_h16[i] = 0u;
            __syncthreads();
            const uint32_t lo = s_lo, hi = s_hi;
            const uint32_t span = hi - lo;
            const int sb = span ? 32 - __clz((int)span) : 0;
            const int sh = sb > 8 ? sb - 8 : 0;
            uint32_t *mine = s_h16 + (lane & 15) * 257;
            for (int base = tid; base < P; base += BLK * 8) {          // eight keys per trip: the reads overlap
                uint32_t d8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = base + u * BLK;
                    d8[u] = p < P ? s_d[p] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t d = d8[u];
                    if (d >= lo && d <= hi) atomicAdd(&mine[(d - lo) >> sh], 1u);       // lo >= 1: a zero is never counted
                }
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t acc = 0u;
#pragma unroll
                for (int cpy = 0; cpy < 16; ++cpy) acc += s_h16[cpy * 257 + tid];
                s_h[tid] = acc;
            }
            __syncthreads();
            if (tid < 64) {
                const uint4 q = reinterpret_cast<const uint4 *>(s_h)[lane];
                const uint32_t t = q.x + q.y + q.z + q.w;
                const uint32_t suf = suffix_scan64(t, lane);
                uint32_t digit, above, nd;
                find_kth_bin(q, suf, t, lane, s_need, digit, above, nd);
                if (lane == 0) {
                    const uint32_t nlo = lo + (digit << sh);
                    uint32_t nhi = nlo + ((1u << sh) - 1u);
                    if (nhi > hi || nhi < nlo) nhi = hi;
                    s_lo = nlo; s_hi = nhi; s_need -= above;
                    s_state = sh == 0 ? 1u : (nd == 1u ? 2u : 0u);   // 1: the bin is one key value; 2: it holds one key
                }
            }
            __syncthreads();
            if (s_state) break;
        }
        if (s_state == 2u) {                               // the one key inside [lo, hi]
            const uint32_t lo = s_lo, hi = s_hi;
            for (int p = tid; p < P; p += BLK) {
                const uint32_t d = s_d[p];
                if (d >= lo && d <= hi) s_lo = d;
            }
            __syncthreads();
        }
        dstar = s_lo;
        need_eq = s_need;                                 // >= 1: the key d* itself is among the top_k
    }
    // wave w walks anchors [w*Q, (w+1)*Q): first the totals per wave, then ranks and output
    const int Q = ((P + BLK - 1) / BLK) * 64;
    const int p_lo = wv * Q, p_hi = min(P, p_lo + Q);
    uint32_t gt = 0u, eq = 0u;
    for (int p0 = p_lo; p0 < p_hi; p0 += 64 * 4) {
        uint32_t d4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * 64 + lane;
            d4[u] = p < p_hi ? s_d[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gt += (uint32_t)__popcll(__ballot(d4[u] > dstar));
            eq += (uint32_t)__popcll(__ballot(d4[u] == dstar && d4[u] != 0u));
        }
    }
    if (lane == 0) { s_wgt[wv] = gt; s_weq[wv] = eq; }
    __syncthreads();
    uint32_t eq_before = 0u, sel_before = 0u;
    for (int w = 0; w < wv; ++w) {
        const uint32_t e = s_weq[w];
        const uint32_t take = eq_before >= need_eq ? 0u : min(e, need_eq - eq_before);
        sel_before += s_wgt[w] + take;
        eq_before += e;
    }
    // winners into LDS at their rank first; the decode and the output rows then go out in one dense pass (the gathers
    // of a sparse, lane-masked loop would each wait out their own latency)
    unsigned long long *s_win = reinterpret_cast<unsigned long long *>(s_d + ((P + 1) & ~1));
    for (int q0 = p_lo; q0 < p_hi; q0 += 64 * 4) {
        uint32_t d4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = q0 + u * 64 + lane;
            d4[u] = p < p_hi ? s_d[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = q0 + u * 64 + lane;
            const uint32_t d = d4[u];
            const bool is_eq = d == dstar && d != 0u;
            const unsigned long long em = __ballot(is_eq);
            const uint32_t my_eq = eq_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
            const bool sel = d > dstar || (is_eq && my_eq < need_eq);
            const unsigned long long sm = __ballot(sel);
            if (sel) {
                const uint32_t rank = sel_before + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
                if (rank < n_sel) s_win[rank] = ((unsigned long long)d << 32) | (uint32_t)p;
            }
            eq_before += (uint32_t)__popcll(em);
            sel_before += (uint32_t)__popcll(sm);
        }
    }
    __syncthreads();
    const size_t row0 = (size_t)b * ((size_t)Cs * sp.top_k) + s_row0;
    for (uint32_t rank = (uint32_t)tid; rank < n_sel; rank += (uint32_t)BLK) {
        const unsigned long long w = s_win[rank];
        const uint32_t p = (uint32_t)w, d = (uint32_t)(w >> 32);
        const float4 l = locs[(size_t)b * P + p];
        out_boxes[row0 + rank] = decode_box<DECODE>(anchors, l, p);
        out_scores[row0 + rank] = __uint_as_float(bits_from_okey(sp.key_thr + d));
        out_cls[row0 + rank] = (int64_t)(c + class_id_offset);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------

int pow2_ge(long long need) {
    int cap = 2;
    while (cap < need) cap <<= 1;
    return cap;
}

// Everything the host derives from the problem size: the sampling plan, the staging capacities and
// the geometry of the scan.
struct SelPlan {
    int use_sample, rate, m, cap;   // cap: entries of one (image,class) staging list
    int n_sub, ns_cap;              // sub-tiles per image; sampled rows per (image,class) (multiple of SUB_ROWS)
    int subs_per_block, scan_gx;    // scan: sub-tiles per workgroup, workgroups per image
};

// With a 1-in-rate sample, the score whose sample rank is m has, with probability about 1 - 3e-6
// (z = 4.5 standard deviations of the binomial), at least top_k scores at or above it; `upper` bounds
// the number of scores at or above it by the same margin on the other side.
int ept_for(int64_t C) { return (C <= 32 ? 32 : (C <= 96 ? 96 : (C <= 128 ? 128 : 256))) / LPR; }

// LDS floats of one sub-tile: the data, the alignment shift, and 4*EPT floats of slack so that the
// unconditional register loads of the activation stay inside the buffer.
// (a quarter per wave: its 16 rows, the alignment shift and the slack, in whole 16-byte pieces)
int tile_floats_for(int64_t C) { return SCAN_WAVES * (int)(((WAVE_ROWS * C + LPR * ept_for(C) + 8) / 4 + 1) * 4); }

// Entries of each of a scan wave's two candidate lists: 256, or 128 where the sub-tiles are small (few classes), many
// workgroups share a CU and the lists would otherwise be most of a workgroup's LDS.
int scan_ebuf_for(int64_t C) { return (size_t)tile_floats_for(C) * 4 >= 16 * 1024 ? 256 : 128; }

// LDS bytes of a scan workgroup: two sub-tile buffers, the waves' candidate lists, their per-class counters.
size_t scan_lds_bytes(int64_t C, int64_t Cs) {
    return 2 * (size_t)tile_floats_for(C) * 4 + 16 + (size_t)SCAN_WAVES * 2 * scan_ebuf_for(C) * 8 +
           (size_t)SCAN_WAVES * (Cs > 64 ? 128 : 64) * sizeof(uint32_t);
}

// Workgroups of the scan that are resident at once: LDS-limited, on every CU of the device.
int64_t scan_resident_workgroups(int64_t C, int64_t Cs) {
    static int cus[64] = {0};                                     // per device; a racing first call writes the same value
    // One sub-tile per workgroup is in flight, and about 40 KB in flight per CU keep the DMA pipeline at its rate (two
    // workgroups at C = 81, eight at C = 21; a third workgroup at C = 81 adds 1 % to the scan on its own).  The LDS left
    // over lets workgroups of another batch's kernels (NMS, on another stream) run beside the scan: 1.5 % more
    // images/s with three batches in flight.  DETBOX_SCAN_PER_CU overrides.
    static const int per_cu_env = [] { const char *e = std::getenv("DETBOX_SCAN_PER_CU"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 0; }();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    const size_t tile = (size_t)tile_floats_for(C) * 4;
    const size_t lds = scan_lds_bytes(C, Cs) + sizeof(float) * MAX_SEL_CLASSES;      // + the thresholds (static)
    int64_t per_cu = (int64_t)((160 * 1024) / lds);
    const int64_t want = per_cu_env ? per_cu_env : (int64_t)((40 * 1024 + tile - 1) / tile);
    if (per_cu > want) per_cu = want;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    return per_cu * cus[dev];
}

SelPlan make_plan(int64_t B, int64_t P, int64_t C, int64_t Cs, int top_k) {
    SelPlan sp;
    std::memset(&sp, 0, sizeof(sp));
    const double z = 4.5;
    sp.n_sub = (int)((P + SUB_ROWS - 1) / SUB_ROWS);
    // Whole rounds of resident workgroups: a partly filled round costs as much as a full one.  With gx
    // workgroups per image the scan takes ceil(B*gx / resident) rounds of ceil(n_sub / gx) sub-tiles each;
    // take the gx that minimises the product.
    const int64_t resident = scan_resident_workgroups(C, Cs);
    int64_t best_cost = -1, spb = sp.n_sub;
    for (int64_t r = 1; r <= 8; ++r) {
        int64_t gx = (r * resident) / (B > 0 ? B : 1);
        if (gx < 1) gx = 1;
        if (gx > sp.n_sub) gx = sp.n_sub;
        const int64_t per = (sp.n_sub + gx - 1) / gx;
        const int64_t gx2 = (sp.n_sub + per - 1) / per;
        const int64_t rounds = (B * gx2 + resident - 1) / resident;
        const int64_t cost = rounds * per;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; spb = per; }
    }
    sp.subs_per_block = (int)(spb < 1 ? 1 : spb);
    sp.scan_gx = (sp.n_sub + sp.subs_per_block - 1) / sp.subs_per_block;
    if (P < 4096 || P > (1 << 24)) {                     // tiny inputs, or anchors beyond the 24-bit entry field
        int slack = top_k / 2 > 256 ? top_k / 2 : 256;
        sp.use_sample = 0; sp.rate = 1; sp.m = 0; sp.cap = pow2_ge(top_k + slack);
        if (sp.cap < 512) sp.cap = 512;
        sp.ns_cap = SUB_ROWS;
        return sp;
    }
    static const int rate_big = [] { const char *e = std::getenv("DETBOX_SAMPLE_RATE"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 32; }();
    // Staging capacity a sampling rate asks for: the m-th largest of the samples is the threshold, the true top-k clears
    // it with z standard deviations to spare, and the list must hold everything above it.
    auto cap_at = [&](int r) {
        const double kr = (double)top_k / r;
        const double sq = 0.5 * (z + std::sqrt(z * z + 4.0 * kr));
        const int m = (int)std::ceil(sq * sq);
        return pow2_ge((long long)((double)m * r + z * r * std::sqrt((double)m)) + 64);
    };
    // Below 16384 anchors every fourth sub-tile is sampled, or every eighth where that asks for no larger a staging
    // list: the sample pass and the planning halve (SSD300 at batch 512: +7 % images/s dense, +13 % trained-like).
    int rate = P >= 65536 ? rate_big : (P >= 16384 || cap_at(8) <= (cap_at(4) < 512 ? 512 : cap_at(4)) ? 8 : 4);
    double upper = 0.0;
    for (;;) {
        const double kr = (double)top_k / rate;
        const double sq = 0.5 * (z + std::sqrt(z * z + 4.0 * kr));
        const int m = (int)std::ceil(sq * sq);
        upper = (double)m * rate + z * rate * std::sqrt((double)m);
        const int cap = pow2_ge((long long)upper + 64);
        if (cap <= 8192 || rate <= 2) {
            sp.use_sample = 1; sp.rate = rate; sp.m = m; sp.cap = cap < 8192 ? cap : 8192;
            if (sp.cap < 512) sp.cap = 512;
            break;
        }
        rate /= 2;
    }
    sp.ns_cap = ((sp.n_sub + sp.rate - 1) / sp.rate) * SUB_ROWS;
    return sp;
}

struct SelLayout { size_t hist, cnt, flag, img_ovf, rules, staging, samp, total; };

bool use_direct(int64_t B, int64_t P, int64_t C) { return P <= DIRECT_MAX_P && B * P * C <= DIRECT_MAX_SCORES; }

// direct path: the activated scores class-major [B, Cs, P], then the per-(image, class) counts [B*Cs]
size_t direct_bytes(int64_t B, int64_t P, int64_t Cs) {
    return align_up((size_t)B * Cs * P * sizeof(float), 256) + align_up((size_t)B * Cs * sizeof(uint32_t), 256);
}

SelLayout layout(int64_t B, int64_t P, int64_t C, int64_t Cs, const SelPlan &pl) {
    SelLayout L;
    size_t off = 0;
    if (use_direct(B, P, C)) {
        std::memset(&L, 0, sizeof(L));
        L.total = direct_bytes(B, P, Cs);
        return L;
    }
    L.hist = off; off = align_up(off + (size_t)B * Cs * NB * sizeof(uint32_t), 256);
    L.cnt = off; off = align_up(off + (size_t)B * Cs * CNT_STRIDE * sizeof(uint32_t), 256);
    L.flag = off; off = align_up(off + sizeof(uint32_t), 256);
    L.img_ovf = off; off = align_up(off + (size_t)B * sizeof(uint32_t), 256);
    L.rules = off; off = align_up(off + (size_t)B * Cs * sizeof(SelRule), 256);
    L.staging = off; off = align_up(off + (size_t)B * Cs * pl.cap * sizeof(unsigned long long), 256);
    L.samp = off; off = align_up(off + (size_t)B * Cs * pl.ns_cap * sizeof(float), 256);
    L.total = off;          // the exact path keeps no copy of the scores: with a fused activation it recomputes them
    return L;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024)
        DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return DETBOX_OK;
}

struct FastArgs {
    const float *in, *extra;
    SelParams sp;
    int64_t B;
    uint32_t *flag, *cnt, *img_ovf, *hist;
    SelRule *rules;
    unsigned long long *staging;
    float *samp;
    int32_t *seg_off, *counts;
    hipStream_t s;
};

// LDS of the exact path's passes when they recompute the scores: two sub-tile buffers, then the pass's own counters.
constexpr size_t FUSED_LDS_BUDGET = 150 * 1024;

template <int ACT, int EPT>
struct LaunchFast {
    static int run(const FastArgs &a) {
        const SelParams &sp = a.sp;
        hipStream_t s = a.s;
        const size_t tile_bytes = (size_t)sp.tile_floats * 4;
        if (sp.use_sample) {
            const int64_t n_sampled = (sp.n_sub + sp.sample_rate - 1) / sp.sample_rate;
            if (int rc = set_lds(select_sample_kernel<ACT, EPT>, 2 * tile_bytes)) return rc;
            ProfScope ps("select_sample", s);
            select_sample_kernel<ACT, EPT><<<dim3((unsigned)((n_sampled + SAMPLE_SUBS - 1) / SAMPLE_SUBS), (unsigned)a.B),
                                             dim3(SCAN_BLOCK), 2 * tile_bytes, s>>>(a.in, a.extra, sp, a.samp);
            DETBOX_LAUNCH_CHECK();
        }
        {
            const size_t lds = (size_t)sp.ns_cap * 4;
            ProfScope ps("select_plan", s);
            // one wave per list once there are thousands of lists of at most a few thousand samples
            if ((int64_t)sp.Cs * a.B >= 4096 && sp.ns_cap <= 4096) {
                if (int rc = set_lds(select_plan_kernel<64>, lds)) return rc;
                select_plan_kernel<64><<<dim3((unsigned)sp.Cs, (unsigned)a.B), dim3(64), lds, s>>>(a.samp, sp, a.rules, a.flag, a.cnt, a.img_ovf, a.hist);
            } else {
                if (int rc = set_lds(select_plan_kernel<256>, lds)) return rc;
                select_plan_kernel<256><<<dim3((unsigned)sp.Cs, (unsigned)a.B), dim3(256), lds, s>>>(a.samp, sp, a.rules, a.flag, a.cnt, a.img_ovf, a.hist);
            }
            DETBOX_LAUNCH_CHECK();
        }
        const unsigned gx = (unsigned)sp.scan_gx;
        if (sp.use_sample) {
            const size_t lds = scan_lds_bytes(sp.C, sp.Cs);
            if (int rc = set_lds(select_scan_kernel<ACT, EPT>, lds)) return rc;
            {
                ProfScope ps("select_scan", s);
                select_scan_kernel<ACT, EPT><<<dim3(gx, (unsigned)a.B), dim3(SCAN_BLOCK), lds, s>>>(
                    a.in, a.extra, sp, a.rules, a.cnt, a.staging);
                DETBOX_LAUNCH_CHECK();
            }
            ProfScope ps("select_check", s);
            select_check_kernel<<<dim3((unsigned)a.B), dim3(128), 0, s>>>(sp, a.rules, a.cnt, a.img_ovf, a.flag, a.hist);
            DETBOX_LAUNCH_CHECK();
        }
        if (ACT != ACT_NONE) {
            // the exact path on recomputed scores; every launch returns at once unless the flag is up
            // (a workgroup's packed 16-bit counters hold fewer than 65536 anchors: at most 1023 sub-tiles each)
            const unsigned gxh = (unsigned)std::max<int64_t>(gx, (sp.n_sub + 1022) / 1023);
            {
                // Behind the scan these launches almost always return at once -- but a workgroup must be given its LDS before it
                // can look at the flag, and with other batches' scan or NMS workgroups on every CU a large request waits for
                // them to drain (12-17 us per launch instead of 4.5, with three batches in flight).  So the passes ask for
                // little there: counters added in global memory, 16 tied classes at a time.  Where the exact path is the main
                // path (short tables: no scan) they take the LDS they can use.
                const size_t hist_bytes = (size_t)sp.Cs * (NB / 2) * 4;
                const int lds_hist = !sp.use_sample && 2 * tile_bytes + hist_bytes <= FUSED_LDS_BUDGET ? 1 : 0;
                const size_t lds = 2 * tile_bytes + (lds_hist ? hist_bytes : 0);
                if (int rc = set_lds(score_hist_fused_kernel<ACT, EPT>, lds)) return rc;
                ProfScope ps("score_hist", s);
                score_hist_fused_kernel<ACT, EPT><<<dim3(gxh, (unsigned)a.B), dim3(SCAN_BLOCK), lds, s>>>(
                    a.in, a.extra, sp, a.rules, a.flag, a.hist, lds_hist);
                DETBOX_LAUNCH_CHECK();
            }
            {
                ProfScope ps("select_kth", s);
                select_kth_kernel<<<dim3((unsigned)a.B), dim3(256), 0, s>>>(a.hist, sp, a.rules, a.seg_off, a.counts);
                DETBOX_LAUNCH_CHECK();
            }
            {
                int group = sp.use_sample ? 16 : (int)((FUSED_LDS_BUDGET - 2 * tile_bytes) / 1024);
                if (2 * tile_bytes + (size_t)group * 1024 > FUSED_LDS_BUDGET) group = (int)((FUSED_LDS_BUDGET - 2 * tile_bytes) / 1024);
                if (group > sp.Cs) group = sp.Cs;
                if (group < 1) group = 1;
                const size_t lds = 2 * tile_bytes + (size_t)group * 1024;
                if (int rc = set_lds(select_refine_fused_kernel<ACT, EPT>, lds)) return rc;
                ProfScope ps("select_refine", s);
                select_refine_fused_kernel<ACT, EPT><<<dim3((unsigned)a.B), dim3(SCAN_BLOCK), lds, s>>>(
                    a.in, a.extra, sp, a.rules, a.flag, group);
                DETBOX_LAUNCH_CHECK();
            }
            {
                if (int rc = set_lds(select_emit_fused_kernel<ACT, EPT>, 2 * tile_bytes)) return rc;
                ProfScope ps("select_emit", s);
                select_emit_fused_kernel<ACT, EPT><<<dim3(gxh, (unsigned)a.B), dim3(SCAN_BLOCK), 2 * tile_bytes, s>>>(
                    a.in, a.extra, sp, a.rules, a.flag, a.cnt, a.staging);
                DETBOX_LAUNCH_CHECK();
            }
        }
        return DETBOX_OK;
    }
};

struct ActArgs {
    const float *in, *extra;
    SelParams sp;
    int64_t B;
    float *out;
    hipStream_t s;
};

template <int ACT, int EPT>
struct LaunchActivate {
    static int run(const ActArgs &a) {
        const size_t lds = 2 * (size_t)a.sp.tile_floats * 4;
        const unsigned gx = (unsigned)((a.sp.n_sub + a.sp.subs_per_block - 1) / a.sp.subs_per_block);
        if (int rc = set_lds(activate_scores_kernel<ACT, EPT>, lds)) return rc;
        ProfScope ps("activate_scores", a.s);
        activate_scores_kernel<ACT, EPT><<<dim3(gx, (unsigned)a.B), dim3(SCAN_BLOCK), lds, a.s>>>(a.in, a.extra, a.sp,
                                                                                                  nullptr, a.out);
        DETBOX_LAUNCH_CHECK();
        return DETBOX_OK;
    }
};

template <template <int, int> class L, typename A>
int dispatch_act_ept(int act, int ept, const A &a) {
#define DETBOX_ROW(ACT)                                        \
    switch (ept) {                                             \
        case 32 / LPR: return L<ACT, 32 / LPR>::run(a);        \
        case 96 / LPR: return L<ACT, 96 / LPR>::run(a);        \
        case 128 / LPR: return L<ACT, 128 / LPR>::run(a);      \
        default: return L<ACT, 256 / LPR>::run(a);             \
    }
    switch (act) {
        case ACT_NONE: DETBOX_ROW(ACT_NONE)
        case ACT_SOFTMAX: DETBOX_ROW(ACT_SOFTMAX)
        case ACT_SOFTMAX_SIGMOID: DETBOX_ROW(ACT_SOFTMAX_SIGMOID)
        default: DETBOX_ROW(ACT_SIGMOID_SIGMOID)
    }
#undef DETBOX_ROW
}

struct DirectArgs {
    const float *in, *extra;
    SelParams sp;
    int64_t B;
    float *colscores;
    uint32_t *npos;
    hipStream_t s;
};

template <int ACT, int EPT>
struct LaunchDirectStats {
    static int run(const DirectArgs &a) {
        const size_t lds = 2 * (size_t)a.sp.tile_floats * 4;
        if (int rc = set_lds(direct_stats_kernel<ACT, EPT>, lds)) return rc;
        ProfScope ps("select_direct_stats", a.s);
        direct_stats_kernel<ACT, EPT><<<dim3((unsigned)a.sp.scan_gx, (unsigned)a.B), dim3(SCAN_BLOCK), lds, a.s>>>(
            a.in, a.extra, a.sp, a.colscores, a.npos);
        DETBOX_LAUNCH_CHECK();
        return DETBOX_OK;
    }
};

int check_act(int activation, const float *extra) {
    DETBOX_CHECK_ARG(activation >= ACT_NONE && activation <= ACT_SIGMOID_SIGMOID, "unknown activation");
    DETBOX_CHECK_ARG(activation == ACT_NONE || activation == ACT_SOFTMAX || extra, "this activation needs `extra`");
    return DETBOX_OK;
}

}  // namespace

extern "C" int detbox_activate_scores(int activation, const float *in, const float *extra, int64_t B, int64_t P,
                                      int64_t C, float *out, void *stream) {
    if (int rc = check_act(activation, extra)) return rc;
    DETBOX_CHECK_ARG(B >= 0 && P >= 0 && C >= 1 && P < (1ll << 31) && B <= 65535, "bad size");
    if (B == 0 || P == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(in && out, "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)in % 16 == 0, "input must be 16-byte aligned");
    if (C > 2 * MAX_SEL_CLASSES) {
        set_error("detbox_activate_scores: C = %lld beyond the limit %d", (long long)C, 2 * MAX_SEL_CLASSES);
        return DETBOX_ERR_UNSUPPORTED;
    }
    ActArgs a;
    a.in = in; a.extra = extra; a.B = B; a.out = out; a.s = static_cast<hipStream_t>(stream);
    std::memset(&a.sp, 0, sizeof(a.sp));
    a.sp.P = (int)P; a.sp.C = (int)C;
    a.sp.n_sub = (int)((P + SUB_ROWS - 1) / SUB_ROWS);
    a.sp.tile_floats = tile_floats_for(C);
    a.sp.total_floats = (long long)B * P * C;
    int64_t spb = (B * a.sp.n_sub + 2047) / 2048;
    a.sp.subs_per_block = (int)(spb < 1 ? 1 : spb);
    return dispatch_act_ept<LaunchActivate>(activation, ept_for(C), a);
}

extern "C" size_t detbox_select_decode_workspace_bytes(int64_t B, int64_t P, int64_t C, int class_begin, int top_k,
                                                       int activation) {
    if (B <= 0 || C <= class_begin || top_k <= 0) return 256;
    (void)activation;       // the workspace does not depend on it: no pass keeps a copy of the activated scores
    return layout(B, P, C, C - class_begin, make_plan(B, P, C, C - class_begin, top_k)).total;
}

extern "C" int detbox_select_decode(int decode_mode, int activation, const float *scores, const float *extra,
                                    const float *locs, const float *anchors,
                                    int64_t B, int64_t P, int64_t C, int class_begin, int class_id_offset,
                                    float conf_thresh, int top_k, float score_hi_hint,
                                    float *out_boxes, float *out_scores, int64_t *out_class_ids,
                                    int32_t *out_seg_offsets, int32_t *out_counts,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    DETBOX_CHECK_ARG(decode_mode >= 0 && decode_mode <= 2, "unknown decode mode");
    if (int rc = check_act(activation, extra)) return rc;
    DETBOX_CHECK_ARG(B >= 0 && P >= 0 && C >= 1, "bad size");
    DETBOX_CHECK_ARG(class_begin >= 0 && class_begin < C, "class_begin out of range");
    DETBOX_CHECK_ARG(!(conf_thresh != conf_thresh), "conf_thresh is NaN");
    if (top_k < 1 || top_k > DETBOX_MAX_TOP_K) {
        set_error("detbox_select_decode: top_k %d outside 1..%d", top_k, DETBOX_MAX_TOP_K);
        return DETBOX_ERR_UNSUPPORTED;
    }
    const int64_t Cs = C - class_begin;
    if (Cs > MAX_SEL_CLASSES || C > 2 * MAX_SEL_CLASSES) {
        set_error("detbox_select_decode: %lld selected classes of %lld, limits %d / %d", (long long)Cs, (long long)C,
                  MAX_SEL_CLASSES, 2 * MAX_SEL_CLASSES);
        return DETBOX_ERR_UNSUPPORTED;
    }
    DETBOX_CHECK_ARG(P < (1ll << 31) && P * C < (1ll << 40), "P too large");
    DETBOX_CHECK_ARG(B <= 65535, "B > 65535");
    if (B == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(out_seg_offsets && out_counts, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P == 0) {
        DETBOX_HIP(hipMemsetAsync(out_seg_offsets, 0, (size_t)B * (Cs + 1) * sizeof(int32_t), s));
        DETBOX_HIP(hipMemsetAsync(out_counts, 0, (size_t)B * sizeof(int32_t), s));
        return DETBOX_OK;
    }
    DETBOX_CHECK_ARG(scores && locs && anchors && out_boxes && out_scores && out_class_ids, "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)scores % 16 == 0 && (uintptr_t)locs % 16 == 0 && (uintptr_t)out_boxes % 16 == 0,
                     "scores, locs and out_boxes must be 16-byte aligned");
    DETBOX_CHECK_ARG(decode_mode != DETBOX_DECODE_CXCYWH || (uintptr_t)anchors % 16 == 0, "anchors must be 16-byte aligned");
    DETBOX_CHECK_ARG(decode_mode != DETBOX_DECODE_FCOS || (uintptr_t)anchors % 8 == 0, "points must be 8-byte aligned");
    const SelPlan plan = make_plan(B, P, C, Cs, top_k);
    const int cap = plan.cap;
    const SelLayout L = layout(B, P, C, Cs, plan);
    if (!workspace || (uintptr_t)workspace % 256 || workspace_bytes < L.total) {
        set_error("detbox_select_decode: workspace missing, misaligned or smaller than %zu bytes", L.total);
        return DETBOX_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(workspace);
    uint32_t *ghist = reinterpret_cast<uint32_t *>(ws + L.hist);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(ws + L.cnt);
    uint32_t *flag = reinterpret_cast<uint32_t *>(ws + L.flag);
    uint32_t *img_ovf = reinterpret_cast<uint32_t *>(ws + L.img_ovf);
    SelRule *rules = reinterpret_cast<SelRule *>(ws + L.rules);
    unsigned long long *staging = reinterpret_cast<unsigned long long *>(ws + L.staging);
    float *samp = reinterpret_cast<float *>(ws + L.samp);

    SelParams sp;
    std::memset(&sp, 0, sizeof(sp));
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
    sp.use_sample = plan.use_sample; sp.sample_rate = plan.rate; sp.sample_m = plan.m;
    sp.n_sub = plan.n_sub; sp.ns_cap = plan.ns_cap;
    sp.subs_per_block = plan.subs_per_block; sp.scan_gx = plan.scan_gx;
    sp.tile_floats = tile_floats_for(C);
    sp.ebuf = scan_ebuf_for(C);
    {
        // A hand-over costs the wave about 1.5 us whatever it carries (the wave's next wait for its DMA also waits for the
        // hand-over's atomics, which take longer than a sub-tile), so lists are handed over three quarters full rather than
        // half full: 143 -> 137 us on the dense RetinaNet-800 batch (1/8: 190, 1/4: 157, 7/8: 136, where a sub-tile's
        // candidates begin not to fit and are handed over inside the sub-tile).  DETBOX_SCAN_FLUSH_AT=1..8 (eighths) overrides.
        static const int eighths = [] { const char *e = std::getenv("DETBOX_SCAN_FLUSH_AT"); const int v = e ? std::atoi(e) : 0; return v > 0 && v <= 8 ? v : 6; }();
        sp.flush_at = sp.ebuf * eighths / 8;
    }
    sp.total_floats = (long long)B * P * C;

    if (use_direct(B, P, C)) {
        float *colscores = reinterpret_cast<float *>(ws);
        uint32_t *npos = reinterpret_cast<uint32_t *>(ws + align_up((size_t)B * Cs * P * sizeof(float), 256));
        DETBOX_HIP(hipMemsetAsync(npos, 0, (size_t)B * Cs * sizeof(uint32_t), s));
        // enough workgroups to fill the chip, at least one sub-tile each
        int64_t spb = (B * sp.n_sub + 1023) / 1024;
        sp.subs_per_block = (int)(spb < 1 ? 1 : spb);
        sp.scan_gx = (sp.n_sub + sp.subs_per_block - 1) / sp.subs_per_block;
        DirectArgs a;
        a.in = scores; a.extra = extra; a.sp = sp; a.B = B; a.colscores = colscores; a.npos = npos; a.s = s;
        if (int rc = dispatch_act_ept<LaunchDirectStats>(activation, ept_for(C), a)) return rc;
        const size_t lds = (size_t)((P + 1) & ~1ll) * 4 + (size_t)top_k * 8;      // keys, then the winners in rank order
        const float4 *l4 = reinterpret_cast<const float4 *>(locs);
        float4 *ob = reinterpret_cast<float4 *>(out_boxes);
        ProfScope ps("select_direct", s);
        // few lists (fewer than two per CU): four times the threads per list, the kernel is one workgroup's latency
        const bool wide = Cs * B <= 512;
#define DETBOX_DIRECT_BLK(D, BLKC)                                                                                \
    {                                                                                                             \
        if (int rc2 = set_lds(direct_select_kernel<D, BLKC>, lds)) return rc2;                                    \
        direct_select_kernel<D, BLKC><<<dim3((unsigned)Cs, (unsigned)B), dim3(BLKC), lds, s>>>(                   \
            colscores, sp, npos, l4, anchors, class_id_offset, ob, out_scores, out_class_ids, out_seg_offsets, out_counts); \
    }
#define DETBOX_DIRECT(D) { if (wide) DETBOX_DIRECT_BLK(D, 1024) else DETBOX_DIRECT_BLK(D, 256) }
        if (decode_mode == DETBOX_DECODE_CXCYWH) DETBOX_DIRECT(DETBOX_DECODE_CXCYWH)
        else if (decode_mode == DETBOX_DECODE_YOLO) DETBOX_DIRECT(DETBOX_DECODE_YOLO)
        else DETBOX_DIRECT(DETBOX_DECODE_FCOS)
#undef DETBOX_DIRECT_BLK
#undef DETBOX_DIRECT
        DETBOX_LAUNCH_CHECK();
        return DETBOX_OK;
    }
    {
        FastArgs a;
        a.in = scores; a.extra = extra; a.sp = sp; a.B = B; a.flag = flag; a.cnt = cnt; a.img_ovf = img_ovf; a.hist = ghist;
        a.rules = rules; a.staging = staging; a.samp = samp;
        a.seg_off = out_seg_offsets; a.counts = out_counts; a.s = s;
        if (int rc = dispatch_act_ept<LaunchFast>(activation, ept_for(C), a)) return rc;
    }
    if (activation == ACT_NONE) {            // the exact path on the scores as given (fused: launched above, on recomputed scores)
      {
        ProfScope ps("score_hist", s);
        const size_t lds = (size_t)Cs * (NB / 2) * 4 + (size_t)Cs * 4;
        if (int rc2 = set_lds(score_hist_kernel, lds)) return rc2;
        score_hist_kernel<<<dim3((unsigned)tiles, (unsigned)B), dim3(PASS_BLOCK), lds, s>>>(scores, sp, rules, flag, ghist);
        DETBOX_LAUNCH_CHECK();
      }
      {
        ProfScope ps("select_kth", s);
        select_kth_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(ghist, sp, rules, out_seg_offsets, out_counts);
        DETBOX_LAUNCH_CHECK();
      }
      {
        ProfScope ps("select_refine", s);
        select_refine_kernel<<<dim3((unsigned)Cs, (unsigned)B), dim3(256), 0, s>>>(scores, sp, rules, flag);
        DETBOX_LAUNCH_CHECK();
      }
      {
        ProfScope ps("select_emit", s);
        select_emit_kernel<<<dim3((unsigned)tiles, (unsigned)B), dim3(PASS_BLOCK), 0, s>>>(scores, sp, rules, flag, cnt, staging);
        DETBOX_LAUNCH_CHECK();
      }
    }
    {
        ProfScope ps("select_finalize", s);
        dim3 grid((unsigned)Cs, (unsigned)B);
        const int words = (int)((P + 63) / 64);
        const int use_bitmap = words <= BITMAP_MAX_WORDS ? 1 : 0;
        size_t lds = use_bitmap ? (size_t)words * 12 + 16 : (size_t)cap * 8;
        if (use_bitmap && lds < (size_t)top_k * 8) lds = (size_t)top_k * 8;      // the winners in rank order reuse the bitmap
        const float4 *l4 = reinterpret_cast<const float4 *>(locs);
        float4 *ob = reinterpret_cast<float4 *>(out_boxes);
        // one wave per segment once there are thousands of short staging lists (up to 16 keys per lane)
        const bool one_wave = use_bitmap && cap <= 1024 && Cs * B >= 4096;
#define DETBOX_FIN(D, K, BLKC)                                                                                    \
    {                                                                                                             \
        if (int rc2 = set_lds(select_finalize_kernel<D, K, BLKC>, lds)) return rc2;                               \
        select_finalize_kernel<D, K, BLKC><<<grid, dim3(BLKC), lds, s>>>(sp, rules, cnt, staging, out_seg_offsets, l4, anchors, \
                                                                         class_id_offset, use_bitmap, ob, out_scores, out_class_ids); \
    }
#define DETBOX_FIN_K(D)                                        \
    if (one_wave) {                                            \
        if (cap <= 512) DETBOX_FIN(D, 8, 64)                   \
        else DETBOX_FIN(D, 16, 64)                             \
    } else switch (cap / 256) {                                \
        case 2: DETBOX_FIN(D, 2, 256) break;                   \
        case 4: DETBOX_FIN(D, 4, 256) break;                   \
        case 8: DETBOX_FIN(D, 8, 256) break;                   \
        case 16: DETBOX_FIN(D, 16, 256) break;                 \
        default: DETBOX_FIN(D, 32, 256) break;                 \
    }
        if (decode_mode == DETBOX_DECODE_CXCYWH) { DETBOX_FIN_K(DETBOX_DECODE_CXCYWH) }
        else if (decode_mode == DETBOX_DECODE_YOLO) { DETBOX_FIN_K(DETBOX_DECODE_YOLO) }
        else { DETBOX_FIN_K(DETBOX_DECODE_FCOS) }
#undef DETBOX_FIN_K
#undef DETBOX_FIN
        DETBOX_LAUNCH_CHECK();
    }
    return DETBOX_OK;
}

namespace {

struct FocalArgs {
    const float *in;
    const int64_t *labels;
    int64_t rows, C;
    float alpha, gamma;
    double *loss_sum;
    const float *grad_scale;
    float *grad;
    hipStream_t s;
};

template <int ACT, int EPT>
struct LaunchFocal {
    static int run(const FocalArgs &a) {
        const int n_sub = (int)((a.rows + SUB_ROWS - 1) / SUB_ROWS);
        const int64_t resident = scan_resident_workgroups(a.C, a.C);
        int spb = (int)((n_sub + resident - 1) / resident);
        if (spb < 1) spb = 1;
        const unsigned gx = (unsigned)((n_sub + spb - 1) / spb);
        const int tile_floats = tile_floats_for(a.C);
        const size_t lds = 2 * (size_t)tile_floats * 4;
        if (a.grad == nullptr) {
            if (int rc = set_lds(focal_forward_kernel<EPT>, lds)) return rc;
            DETBOX_HIP(hipMemsetAsync(a.loss_sum, 0, sizeof(double), a.s));
            ProfScope ps("focal_forward", a.s);
            focal_forward_kernel<EPT><<<dim3(gx), dim3(SCAN_BLOCK), lds, a.s>>>(a.in, a.labels, (int)a.rows, (int)a.C, spb, n_sub,
                                                                               tile_floats, a.alpha, a.gamma, a.loss_sum);
            DETBOX_LAUNCH_CHECK();
        } else {
            if (int rc = set_lds(focal_backward_kernel<EPT>, lds)) return rc;
            ProfScope ps("focal_backward", a.s);
            focal_backward_kernel<EPT><<<dim3(gx), dim3(SCAN_BLOCK), lds, a.s>>>(a.in, a.labels, (int)a.rows, (int)a.C, spb, n_sub,
                                                                                tile_floats, a.alpha, a.gamma, a.grad_scale, a.grad);
            DETBOX_LAUNCH_CHECK();
        }
        return DETBOX_OK;
    }
};

int focal_check(const float *logits, const int64_t *labels, int64_t rows, int64_t C) {
    DETBOX_CHECK_ARG(rows >= 0 && C >= 1, "bad size");
    DETBOX_CHECK_ARG(C <= 256, "more than 256 classes are not supported");
    DETBOX_CHECK_ARG(rows < (1ll << 31) - SUB_ROWS && rows * C < (1ll << 40), "size too large");
    DETBOX_CHECK_ARG(rows == 0 || (logits && labels), "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)logits % 16 == 0, "logits must be 16-byte aligned");
    return DETBOX_OK;
}

}  // namespace

extern "C" int detbox_focal_loss_forward(const float *logits, const int64_t *labels, int64_t rows, int64_t C,
                                         float alpha, float gamma, double *loss_sum, void *stream) {
    if (int rc = focal_check(logits, labels, rows, C)) return rc;
    DETBOX_CHECK_ARG(loss_sum, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows == 0) {
        DETBOX_HIP(hipMemsetAsync(loss_sum, 0, sizeof(double), s));
        return DETBOX_OK;
    }
    FocalArgs a{logits, labels, rows, C, alpha, gamma, loss_sum, nullptr, nullptr, s};
    return dispatch_act_ept<LaunchFocal>(ACT_SOFTMAX, ept_for(C), a);
}

extern "C" int detbox_focal_loss_backward(const float *logits, const int64_t *labels, int64_t rows, int64_t C,
                                          float alpha, float gamma, const float *grad_scale, float *grad_logits,
                                          void *stream) {
    if (int rc = focal_check(logits, labels, rows, C)) return rc;
    DETBOX_CHECK_ARG(rows == 0 || (grad_scale && grad_logits), "null pointer");
    if (rows == 0) return DETBOX_OK;
    FocalArgs a{logits, labels, rows, C, alpha, gamma, nullptr, grad_scale, grad_logits, static_cast<hipStream_t>(stream)};
    return dispatch_act_ept<LaunchFocal>(ACT_SOFTMAX, ept_for(C), a);
}
