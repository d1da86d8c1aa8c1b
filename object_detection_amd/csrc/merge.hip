// Order step of the batched NMS and the result rows, in two launches.
//
// The NMS kernels leave, per (image, class segment), a run of kept boxes in descending score order as
// (ordered score key, row index) pairs.  torchvision.ops.batched_nms (src/utilities/predictor.py:55) returns the kept
// boxes of ALL classes in one stable descending score order, and src/utilities/predictor.py:56-73 turns them into
// result rows in that order.  Both happen here:
//
//   nms_split   one workgroup per image: scans the kept counts; if the image keeps more rows than one workgroup can
//               hold, it takes every SP-th element of every run as a sample, merges the samples (they are sorted runs
//               themselves), makes every Q-th merged sample a splitter and counts, per splitter and run, the samples
//               of that run in front of it (a histogram over the merged samples).
//   nms_tile    one workgroup per (splitter interval, image): turns the sample counts of its two splitters into exact
//               positions with one look at the SP run elements behind the last sample in front (cs samples of a run in
//               front of a splitter: its elements 0 .. SP*(cs-1) are in front, element SP*cs is not), loads its piece
//               of every run into LDS, merges them pairwise in log2(S) levels, and writes its slice of the keep list
//               and of the result rows (and of the message ranks exchange) directly.
//               A splitter has at most SP*Q*t and at least SP*Q*t - (SP-1)*S elements in front of it, so a tile
//               holds at most SP*Q + (SP-1)*S <= MG_CAP elements.
//
// Inside a workgroup the elements are put in order by a distribution sort in LDS: the score keys of a tile (or of an
// image's samples) lie in a narrow, smoothly filled range, so a linear map of the key onto NB buckets leaves about one
// element per bucket; a counting sort by bucket (one LDS atomic per element, one prefix sum) followed by a rank inside
// the bucket (compare with the few other elements of the bucket) orders them with some dozens of instructions per
// element, whatever the number of runs.  Keys that pile up (many equal scores: more than MG_BUCKET_MAX elements in one
// bucket) send the workgroup to the general method instead: log2(S) merge-path levels over the runs, where every
// thread owns VT consecutive outputs, finds its pair of neighbouring groups and its starting point on the pair's
// diagonal by binary searches, and merges serially.
//
// Order: a run element is compared as the 64-bit word (score key << 32) | ~row.  Rows are unique inside an image and
// class segments hold ascending rows, so these words are distinct, every run is strictly descending in them, and
// their descending order is exactly "descending score, ties by ascending row" -- the stable order of the reference.
#include "detbox_internal.h"

using namespace detbox;

namespace {

constexpr int MG_CAP = 4096;            // run elements one tile workgroup holds
constexpr int MG_TILE_THREADS = 512;
constexpr int MG_TILE_VT = 9;           // 512 * 9 >= MG_CAP
constexpr int MG_NS_CAP = 6144;         // samples one split workgroup holds
constexpr int MG_SPLIT_THREADS = 1024;
constexpr int MG_SPLIT_VT = 7;          // 1024 * 7 >= MG_NS_CAP
constexpr int MG_MAX_S = 256;           // segments per image
constexpr int MG_HIST_CAP = 12288;      // tiles x segments of one image (the histogram lives in the idle sample buffer)
constexpr int MG_TILE_NB = 2048;        // buckets of the distribution sort of a tile / of an image's samples
constexpr int MG_SPLIT_NB = 8192;
constexpr int MG_BUCKET_MAX = 48;       // elements in one bucket beyond which a workgroup takes the merge-path levels
static_assert(MG_TILE_THREADS * MG_TILE_VT >= MG_CAP && MG_SPLIT_THREADS * MG_SPLIT_VT >= MG_NS_CAP, "outputs per thread");

typedef unsigned long long u64;

// Measurement aid (detbox_debug_merge_stats): [0] tile workgroups, [1] of them sorted by merge-path levels, [2..5] clock
// cycles of a tile workgroup's first wave in its phases (set-up and piece bounds; piece loads; sort; output), summed;
// [6] split workgroups that sampled, [7] their cycles.
__device__ u64 g_merge_stats[8];

__device__ inline u64 run_word(const float2 v) {
    return ((u64)__float_as_uint(v.x) << 32) | (u64)(~__float_as_uint(v.y));
}

// Exclusive prefix sums of one value per thread over the workgroup (THREADS threads, all active); the total is
// returned in `total`.  s_w: one word per wave.
template <int THREADS>
__device__ inline uint32_t block_exclusive_add(const uint32_t v, uint32_t *s_w, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_add(v);
    __syncthreads();                                    // s_w may still be read from an earlier call
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        const uint32_t x = s_w[w];
        base += w < wave ? x : 0u;
        tot += x;
    }
    total = tot;
    return base + inc - v;
}

// Minimum over the wave, left in every lane.
template <int O>
__device__ inline uint32_t wave_min_u32(uint32_t v) {
    v = min(v, lane_xor_u32<O>(v));
    if constexpr (O > 1) return wave_min_u32<O / 2>(v);
    else return v;
}

// Largest r < S with off[r] <= e (off ascending, off[0] <= e): the run that holds element (or row) e.
__device__ inline int run_of(const int *__restrict__ off, const int S, const int e) {
    int r = 0;
#pragma unroll
    for (int step = MG_MAX_S / 2; step > 0; step >>= 1)
        if (r + step < S && off[r + step] <= e) r += step;
    return r;
}

// One merge level over `total` elements that lie in `cur` as the runs [off[r], off[r+1]), r < S: groups of g = 2^lvl
// consecutive runs are sorted (descending) already; neighbouring groups are merged into `nxt`.
template <int VT>
__device__ inline void merge_level(const u64 *__restrict__ cur, u64 *__restrict__ nxt, const int *__restrict__ off,
                                   const int S, const int g, const int total) {
    int o = min((int)threadIdx.x * VT, total);
    const int o_end = min(o + VT, total);
    while (o < o_end) {
        int pr = 0;                        // pair containing output position o: largest pr with off[pr*2g] <= o
#pragma unroll
        for (int step = MG_MAX_S / 2; step > 0; step >>= 1)
            if ((pr + step) * 2 * g < S && off[(pr + step) * 2 * g] <= o) pr += step;
        const int a_lo = off[min(S, pr * 2 * g)], a_hi = off[min(S, pr * 2 * g + g)];
        const int b_hi = off[min(S, (pr + 1) * 2 * g)];
        const int na = a_hi - a_lo, nb = b_hi - a_hi;
        const u64 *A = cur + a_lo, *Bp = cur + a_hi;
        const int d = o - a_lo;            // diagonal inside the pair
        int lo = max(0, d - nb), hi = min(d, na);
        while (lo < hi) {                  // A[mid] goes before B[d-1-mid]  <=>  A[mid] > B[d-1-mid] (the words are distinct)
            const int mid = (lo + hi) >> 1;
            if (A[mid] > Bp[d - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        int ia = lo, ib = d - lo;
        const int stop = min(o_end, b_hi);
        u64 a = ia < na ? A[ia] : 0ull, bb = ib < nb ? Bp[ib] : 0ull;     // 0 is below every real word
        for (; o < stop; ++o) {
            const bool take_a = a > bb;
            nxt[o] = take_a ? a : bb;
            if (take_a) { ++ia; a = ia < na ? A[ia] : 0ull; }
            else { ++ib; bb = ib < nb ? Bp[ib] : 0ull; }
        }
    }
}

// All levels; returns the buffer that holds the merged sequence (a or b).
template <int VT>
__device__ inline u64 *merge_all(u64 *a, u64 *b, const int *off, const int S, const int total) {
    for (int g = 1; g < S; g <<= 1) {
        merge_level<VT>(a, b, off, S, g, total);
        __syncthreads();
        u64 *t = a; a = b; b = t;
    }
    return a;
}

// Distribution sort of the n words in `a` (descending), result in `a`; `b` is scratch, hist holds NB + 2 words.
// Returns false, with `a` untouched, when some bucket would hold more than MG_BUCKET_MAX elements.  Block-uniform.
// Every thread of the workgroup must call it; ends with a barrier.
template <int THREADS, int E, int NB>
__device__ inline bool bucket_sort(u64 *__restrict__ a, u64 *__restrict__ b, const int n, uint32_t *__restrict__ hist) {
    static_assert(NB % THREADS == 0, "buckets per thread");
    constexpr int BPT = NB / THREADS;
    const int tid = threadIdx.x;
    uint32_t *s_min = hist + NB, *s_max = hist + NB + 1, *s_big = hist + NB + 2;
    u64 w[E];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int e = tid + k * THREADS;
        w[k] = e < n ? a[e] : 0ull;
        if (e < n) { kmin = min(kmin, (uint32_t)(w[k] >> 32)); kmax = max(kmax, (uint32_t)(w[k] >> 32)); }
    }
#pragma unroll
    for (int q = 0; q < BPT; ++q) hist[tid + q * THREADS] = 0u;
    if (tid == 0) { *s_min = 0xffffffffu; *s_max = 0u; *s_big = 0u; }
    __syncthreads();
    kmin = wave_min_u32<32>(kmin);
    kmax = ~wave_min_u32<32>(~kmax);
    if ((tid & 63) == 0) { atomicMin(s_min, kmin); atomicMax(s_max, kmax); }
    __syncthreads();
    const uint32_t hi = *s_max, width = hi - *s_min;
    // monotone in the key: int -> float rounding, the product with a positive scale and the truncation all are
    const float scale = width ? (float)(NB - 1) / (float)width : 0.0f;
    int bucket[E];
    uint32_t slot[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int e = tid + k * THREADS;
        bucket[k] = min(NB - 1, (int)((float)(hi - (uint32_t)(w[k] >> 32)) * scale));
        slot[k] = e < n ? atomicAdd(&hist[bucket[k]], 1u) : 0u;
    }
    __syncthreads();
    {   // exclusive prefix sums of the bucket counts, in place; the largest count
        uint32_t c[BPT], sum = 0, big = 0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) { c[q] = hist[tid * BPT + q]; sum += c[q]; big = max(big, c[q]); }
        uint32_t total;
        uint32_t base = block_exclusive_add<THREADS>(sum, hist + NB + 4, total);
        if (big > (uint32_t)MG_BUCKET_MAX) *s_big = 1u;
#pragma unroll
        for (int q = 0; q < BPT; ++q) { hist[tid * BPT + q] = base; base += c[q]; }
    }
    __syncthreads();
    if (*s_big) return false;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int e = tid + k * THREADS;
        if (e < n) b[hist[bucket[k]] + slot[k]] = w[k];
    }
    __syncthreads();
    {   // rank inside the bucket: the E elements of a thread walk their buckets together, so the LDS reads overlap
        int bs[E], len[E], ahead[E], longest = 0;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int e = tid + k * THREADS;
            bs[k] = (int)hist[bucket[k]];
            len[k] = e < n ? (bucket[k] + 1 < NB ? (int)hist[bucket[k] + 1] : n) - bs[k] : 0;
            ahead[k] = 0;
            longest = max(longest, len[k]);
        }
        for (int j = 0; j < longest; ++j) {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const u64 x = b[min(bs[k] + j, n - 1)];
                ahead[k] += (j < len[k] && x > w[k]) ? 1 : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int e = tid + k * THREADS;
            if (e < n) a[bs[k] + ahead[k]] = w[k];
        }
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(MG_SPLIT_THREADS) void nms_split_kernel(
    const float2 *__restrict__ kept, const int32_t *__restrict__ kcount, const int32_t *__restrict__ seg_off,
    const int S, const long long stride, const int SP, const int Q, const int T_max, const int one_tile_rows,
    int32_t *__restrict__ keep_counts, int32_t *__restrict__ csmat, u64 *__restrict__ kmat, int32_t *__restrict__ ntiles,
    const int stats) {
    extern __shared__ u64 s_dyn[];
    const long long c_begin = stats ? (long long)__builtin_readcyclecounter() : 0ll;
    __shared__ int s_soff[MG_MAX_S + 1], s_so[MG_MAX_S + 1];
    __shared__ uint32_t s_w[MG_SPLIT_THREADS / 64];
    __shared__ int s_bad;
    constexpr int THREADS = MG_SPLIT_THREADS;
    constexpr int E = MG_NS_CAP / THREADS;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *kc = kcount + (size_t)b * S;
    const int32_t *so = seg_off + (size_t)b * (S + 1);
    if (tid == 0) s_bad = 0;
    int c = tid < S ? kc[tid] : 0;
    if (tid <= S) s_so[tid] = so[tid];
    __syncthreads();
    if (c < 0) { s_bad = 1; c = 0; }            // a segment the NMS kernel refused: empty, and the image is flagged
    uint32_t n = 0, nsamp = 0;
    block_exclusive_add<THREADS>((uint32_t)c, s_w, n);
    const uint32_t q0 = block_exclusive_add<THREADS>((uint32_t)((c + SP - 1) / SP), s_w, nsamp);
    if (tid < S) s_soff[tid] = (int)q0;
    if (tid == 0) s_soff[S] = (int)nsamp;
    __syncthreads();
    if (tid == 0) keep_counts[b] = s_bad ? -1 : (int)n;
    if (n <= (uint32_t)one_tile_rows) {          // the whole image is one tile (one_tile_rows >= SP * Q)
        if (tid == 0) ntiles[b] = 1;
        return;
    }
    const int T = (int)((nsamp + Q - 1) / Q);    // >= 2: nsamp * SP >= n > SP * Q
    if (tid == 0) ntiles[b] = T;
    u64 *buf_a = s_dyn, *buf_b = s_dyn + MG_NS_CAP;
    const float2 *kb = kept + (size_t)b * stride;
    {
        float2 v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {            // the samples of a run sit SP elements apart
            const int e = tid + k * THREADS;
            const int r = run_of(s_soff, S, e);
            v[k] = make_float2(0.f, 0.f);
            if (e < (int)nsamp) v[k] = kb[s_so[r] + (size_t)(e - s_soff[r]) * SP];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int e = tid + k * THREADS;
            if (e < (int)nsamp) buf_a[e] = run_word(v[k]);
        }
    }
    __syncthreads();
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dyn + 2 * MG_NS_CAP);
    const u64 *sorted = bucket_sort<THREADS, E, MG_SPLIT_NB>(buf_a, buf_b, (int)nsamp, s_hist)
                            ? buf_a : merge_all<MG_SPLIT_VT>(buf_a, buf_b, s_soff, S, (int)nsamp);
    uint32_t *hist = reinterpret_cast<uint32_t *>(sorted == buf_a ? buf_b : buf_a);      // [T][S], the idle buffer
    for (int i = tid; i < T * S; i += THREADS) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; ++k) {                // the run of a sample follows from its row
        const int i = tid + k * THREADS;
        if (i < (int)nsamp) atomicAdd(&hist[(i / Q) * S + run_of(s_so, S, (int)~(uint32_t)sorted[i])], 1u);
    }
    __syncthreads();
    // per run: samples in front of splitter t = the samples of the intervals before t; splitter t = merged sample t * Q
    int32_t *cs = csmat + (size_t)b * T_max * S;
    if (tid < S) {
        uint32_t acc = 0;
        for (int t0 = 0; t0 < T; t0 += 8) {
            uint32_t h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = t0 + u < T ? hist[(t0 + u) * S + tid] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < T) cs[(size_t)(t0 + u) * S + tid] = (int32_t)acc;
                acc += h[u];
            }
        }
    }
    for (int t = 1 + tid; t < T; t += THREADS) kmat[(size_t)b * T_max + t] = sorted[t * Q];
    if (stats && tid == 0) {
        atomicAdd(&g_merge_stats[6], 1ull);
        atomicAdd(&g_merge_stats[7], (u64)((long long)__builtin_readcyclecounter() - c_begin));
    }
}

typedef double mg_v2d __attribute__((ext_vector_type(2), aligned(8)));
typedef long long mg_v2l __attribute__((ext_vector_type(2)));
typedef uint32_t mg_v4u __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t mg_v2u __attribute__((ext_vector_type(2)));

// ROWS: the result rows (and, with WIRE, the message of detbox_pack_wire) leave from here; otherwise only the keep list.
template <bool ROWS, bool WIRE>
__global__ __launch_bounds__(MG_TILE_THREADS) void nms_tile_kernel(
    const float2 *__restrict__ kept, const int32_t *__restrict__ kcount, const int32_t *__restrict__ seg_off,
    const int S, const long long stride, const int SP, const int T_max,
    const int32_t *__restrict__ keep_counts, const int32_t *__restrict__ csmat, const u64 *__restrict__ kmat,
    const int32_t *__restrict__ ntiles, int64_t *__restrict__ keep,
    const float4 *__restrict__ boxes, const float *__restrict__ scores, const int64_t *__restrict__ cls,
    const int64_t *__restrict__ image_ids, const int32_t *__restrict__ image_sizes, const int B,
    int64_t *__restrict__ rows_ids, double *__restrict__ rows_vals, int32_t *__restrict__ row_offsets,
    const int hdr_rows, uint32_t *__restrict__ wire, const int stats) {
    extern __shared__ u64 s_dyn[];
    const long long c_begin = stats ? (long long)__builtin_readcyclecounter() : 0ll;
    __shared__ int s_off[MG_MAX_S + 1], s_src[MG_MAX_S], s_so[MG_MAX_S + 1];
    __shared__ long long s_cls[ROWS ? MG_MAX_S : 1];
    // kept counts, sample counts and piece bounds per run: dead before the sort starts, they share its bucket counters
    static_assert(MG_TILE_NB >= 5 * MG_MAX_S, "scratch of the set-up phase");
    int *s_k = reinterpret_cast<int *>(s_dyn + 2 * MG_CAP);
    int (*s_cs)[MG_MAX_S] = reinterpret_cast<int (*)[MG_MAX_S]>(s_k + MG_MAX_S);
    int (*s_lb)[MG_MAX_S] = reinterpret_cast<int (*)[MG_MAX_S]>(s_k + 3 * MG_MAX_S);
    __shared__ uint32_t s_w[MG_TILE_THREADS / 64];
    constexpr int THREADS = MG_TILE_THREADS;
    constexpr int E = MG_CAP / THREADS;
    const TileOfImage wg = xcd_affine_tile();            // the gathers of one image stay in one L2
    const int t = wg.x, b = wg.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int32_t *so = seg_off + (size_t)b * (S + 1);
    const size_t base = (size_t)b * stride;
    const float2 *kb = kept + base;
    // ---- where this tile's piece of every run starts and ends.  Every load whose address does not depend on loaded
    // data is issued before the first one is waited for (rows t and t+1 of the sample counts exist for every t < T_max).
    const int T = ntiles[b];
    u64 K0 = 0, K1 = 0;
    int k_run = 0, seg0 = 0, cs0 = 0, cs1 = 0;
    if (tid < S) {
        k_run = kcount[(size_t)b * S + tid];
        seg0 = so[tid];
        if (t > 0) cs0 = csmat[((size_t)b * T_max + t) * S + tid];
        if (t + 1 < T_max) cs1 = csmat[((size_t)b * T_max + t + 1) * S + tid];
    }
    if (t > 0) K0 = kmat[(size_t)b * T_max + t];
    if (t + 1 < T_max) K1 = kmat[(size_t)b * T_max + t + 1];
    uint32_t rows_before = 0;
    int n_img = 0;
    double W = 0.0, H = 0.0;
    long long img_id = 0;
    if (ROWS) {
        n_img = max(keep_counts[b], 0);                  // a flagged image has a keep list but no rows
        W = (double)image_sizes[2 * b + 1]; H = (double)image_sizes[2 * b]; img_id = image_ids[b];
    }
    if (t >= T) return;                                  // a workgroup beyond the image's tiles has issued a dozen loads, no more
    if (ROWS)
        for (int i = tid; i < b; i += THREADS) rows_before += (uint32_t)max(keep_counts[i], 0);
    if (tid < S) {
        k_run = max(k_run, 0);
        s_k[tid] = k_run;
        s_so[tid] = seg0;
        s_lb[0][tid] = 0;
        s_lb[1][tid] = k_run;
        s_cs[0][tid] = cs0;
        s_cs[1][tid] = cs1;
        if (ROWS) s_cls[tid] = seg0 < stride ? cls[base + seg0] : 0;      // a segment is one class: its first row names it
    }
    if (tid == 0) s_so[S] = so[S];
    __syncthreads();
    if (T > 1) {
        // probe (u, r, j): is element SP*(cs-1)+j of run r in front of splitter t+u?  SP probes share cs and are counted
        // by one ballot; SP divides 64, so a group never straddles a wave.
        const int per = S * SP, total = 2 * per;
        const int grp = lane & ~(SP - 1);
        const u64 gmask = SP == 64 ? ~0ull : ((1ull << SP) - 1ull);
        for (int i0 = 0; i0 < total; i0 += 4 * THREADS) {
            bool live[4];
            int u_[4], r_[4], cs_[4];
            u64 w_[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * THREADS + tid;
                const int u = i >= per ? 1 : 0;
                const int pr = (i - u * per) / SP, j = i & (SP - 1);
                u_[q] = u; r_[q] = pr;
                live[q] = i < total && (u == 0 ? t > 0 : t + 1 < T);
                cs_[q] = live[q] ? s_cs[u][pr] : 0;
                w_[q] = 0ull;
                if (cs_[q] > 0) {
                    const int pos = SP * (cs_[q] - 1) + j;
                    if (pos < s_k[pr]) w_[q] = run_word(kb[s_so[pr] + pos]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u64 ball = __ballot(w_[q] > (u_[q] ? K1 : K0));     // a missing element (word 0) is never in front
                if (live[q] && (lane & (SP - 1)) == 0)
                    s_lb[u_[q]][r_[q]] = cs_[q] > 0 ? SP * (cs_[q] - 1) + __popcll((ball >> grp) & gmask) : 0;
            }
        }
        __syncthreads();
    }
    int lb0 = 0, len = 0;
    if (tid < S) { lb0 = s_lb[0][tid]; len = s_lb[1][tid] - lb0; }
    uint32_t n_tile = 0, tile_start = 0;
    const uint32_t o0 = block_exclusive_add<THREADS>((uint32_t)len, s_w, n_tile);
    block_exclusive_add<THREADS>((uint32_t)lb0, s_w, tile_start);
    if (tid < S) { s_off[tid] = (int)o0; s_src[tid] = s_so[tid] + lb0; }
    if (tid == 0) s_off[S] = (int)n_tile;
    uint32_t row0 = 0;
    if (ROWS) {
        block_exclusive_add<THREADS>(rows_before, s_w, row0);
        if (t == 0 && tid == 0) {
            row_offsets[b] = (int32_t)row0;
            if (b == B - 1) row_offsets[B] = (int32_t)row0 + n_img;
            if (WIRE) {
                wire[2 + b] = row0;
                const u64 id = (u64)img_id;
                wire[2 + (B + 1) + 2 * b] = (uint32_t)id;
                wire[2 + (B + 1) + 2 * b + 1] = (uint32_t)(id >> 32);
                wire[2 + (B + 1) + 2 * B + 2 * b] = (uint32_t)image_sizes[2 * b];
                wire[2 + (B + 1) + 2 * B + 2 * b + 1] = (uint32_t)image_sizes[2 * b + 1];
                if (b == B - 1) { wire[0] = row0 + (uint32_t)n_img; wire[1] = (uint32_t)B; wire[2 + B] = row0 + (uint32_t)n_img; }
            }
        }
    }
    __syncthreads();
    if (n_tile == 0) return;
    const long long c_setup = stats ? (long long)__builtin_readcyclecounter() : 0ll;
    // ---- the pieces into LDS, run after run
    u64 *buf_a = s_dyn, *buf_b = s_dyn + MG_CAP;
    {
        float2 v[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int e = tid + k * THREADS;
            const int r = run_of(s_off, S, e);
            v[k] = make_float2(0.f, 0.f);
            if (e < (int)n_tile) v[k] = kb[s_src[r] + (e - s_off[r])];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int e = tid + k * THREADS;
            if (e < (int)n_tile) buf_a[e] = run_word(v[k]);
        }
    }
    __syncthreads();
    const long long c_loaded = stats ? (long long)__builtin_readcyclecounter() : 0ll;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dyn + 2 * MG_CAP);
    const bool by_buckets = bucket_sort<THREADS, E, MG_TILE_NB>(buf_a, buf_b, (int)n_tile, s_hist);
    const u64 *fin = by_buckets ? buf_a : merge_all<MG_TILE_VT>(buf_a, buf_b, s_off, S, (int)n_tile);
    const long long c_sorted = stats ? (long long)__builtin_readcyclecounter() : 0ll;
    // ---- this tile's slice of the keep list and of the rows, THREADS rows at a time.  The float64 values (and the message
    // words) of a chunk are built in LDS -- the sort's scratch buffer is idle now -- and leave as contiguous streams: a
    // non-temporal store only pays when whole lines are written.  The gathers of the next chunk are in flight meanwhile.
    double *s_vals = reinterpret_cast<double *>(fin == buf_a ? buf_b : buf_a);       // THREADS x 5 doubles
    uint32_t *s_wire = reinterpret_cast<uint32_t *>(s_vals + 5 * THREADS);            // THREADS x 6 words
    static_assert((size_t)MG_TILE_THREADS * (5 * 8 + 6 * 4) <= (size_t)MG_CAP * 8, "row staging fits the idle buffer");
    const int rows_tile = ROWS ? max(0, min((int)n_tile, n_img - (int)tile_start)) : 0;   // rows of this tile (flagged image: 0)
    u64 w_nx = tid < (int)n_tile ? fin[tid] : 0ull;
    float4 bx_nx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ROWS && tid < rows_tile) bx_nx = boxes[base + (~(uint32_t)w_nx)];
    for (int c0 = 0; c0 < (int)n_tile; c0 += THREADS) {
        const int i = c0 + tid;
        const u64 w = w_nx;
        const float4 bx = bx_nx;
        if (c0 + THREADS < (int)n_tile) {                                            // uniform
            const int i2 = i + THREADS;
            w_nx = i2 < (int)n_tile ? fin[i2] : 0ull;
            if (ROWS && i2 < rows_tile) bx_nx = boxes[base + (~(uint32_t)w_nx)];
        }
        const uint32_t pos = ~(uint32_t)w, key = (uint32_t)(w >> 32);
        const size_t o = (size_t)tile_start + i;                // position in the image's keep list
        if (keep && i < (int)n_tile) __builtin_nontemporal_store((long long)pos, reinterpret_cast<long long *>(keep) + base + o);
        if (ROWS) {
            const int nr = min(THREADS, rows_tile - c0);        // rows of this chunk (uniform)
            if (nr <= 0) continue;
            if (tid < nr) {
                float sc = __uint_as_float(bits_from_okey(key));
                if (key == 0x80000000u) sc = scores[base + pos];             // the key does not tell -0 from +0
                const long long cid = s_cls[run_of(s_so, S, (int)pos)];
                s_vals[5 * tid + 0] = (double)bx.x * W;
                s_vals[5 * tid + 1] = (double)bx.y * H;
                s_vals[5 * tid + 2] = (double)(bx.z - bx.x) * W;
                s_vals[5 * tid + 3] = (double)(bx.w - bx.y) * H;
                s_vals[5 * tid + 4] = (double)sc;
                mg_v2l ids = {img_id, cid};
                __builtin_nontemporal_store(ids, reinterpret_cast<mg_v2l *>(rows_ids + 2 * ((size_t)row0 + o)));
                if (WIRE) {
                    uint32_t *q = s_wire + 6 * tid;
                    q[0] = (uint32_t)cid;
                    q[1] = __float_as_uint(bx.x); q[2] = __float_as_uint(bx.y);
                    q[3] = __float_as_uint(bx.z); q[4] = __float_as_uint(bx.w);
                    q[5] = __float_as_uint(sc);
                }
            }
            __syncthreads();
            const size_t R0 = (size_t)row0 + tile_start + c0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int j = tid + q * THREADS;
                if (j < 5 * nr) store_nt(rows_vals + 5 * R0 + j, s_vals[j]);
            }
            if (WIRE) {
                uint32_t *out = wire + 6 * ((size_t)hdr_rows + R0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int j = tid + q * THREADS;
                    if (j < 6 * nr) store_nt(out + j, s_wire[j]);
                }
            }
            __syncthreads();
        }
    }
    if (stats && tid == 0) {
        const long long c_end = (long long)__builtin_readcyclecounter();
        atomicAdd(&g_merge_stats[0], 1ull);
        if (!by_buckets) atomicAdd(&g_merge_stats[1], 1ull);
        atomicAdd(&g_merge_stats[2], (u64)(c_setup - c_begin));
        atomicAdd(&g_merge_stats[3], (u64)(c_loaded - c_setup));
        atomicAdd(&g_merge_stats[4], (u64)(c_sorted - c_loaded));
        atomicAdd(&g_merge_stats[5], (u64)(c_end - c_sorted));
    }
}

// two element buffers, the bucket counts, min / max / flag and one word per wave for the prefix sums
constexpr size_t split_lds_bytes() { return (size_t)MG_NS_CAP * 2 * 8 + (size_t)(MG_SPLIT_NB + 4 + MG_SPLIT_THREADS / 64) * 4; }
constexpr size_t tile_lds_bytes() { return (size_t)MG_CAP * 2 * 8 + (size_t)(MG_TILE_NB + 4 + MG_TILE_THREADS / 64) * 4; }

struct MergeWs { int32_t *csmat; u64 *kmat; int32_t *ntiles; size_t total; };

MergeWs merge_ws(const MergePlan &p, int64_t B, int64_t S, void *workspace) {
    Carver cv(workspace);
    MergeWs w;
    w.csmat = cv.take<int32_t>((size_t)B * p.T_max * S);
    w.kmat = cv.take<u64>((size_t)B * p.T_max);
    w.ntiles = cv.take<int32_t>((size_t)B);
    w.total = cv.off;
    return w;
}

}  // namespace

namespace {
bool g_merge_stats_on = false;
}

extern "C" int detbox_debug_merge_stats(uint64_t *out8, int enable, int reset) {
    DETBOX_CHECK_ARG(out8, "null pointer");
    g_merge_stats_on = enable != 0;
    unsigned long long host[8];
    DETBOX_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_merge_stats), sizeof(host)));
    for (int i = 0; i < 8; ++i) out8[i] = host[i];
    if (reset) {
        for (int i = 0; i < 8; ++i) host[i] = 0ull;
        DETBOX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_merge_stats), host, sizeof(host)));
    }
    return DETBOX_OK;
}

namespace detbox {

MergePlan merge_plan(int64_t B, int64_t stride, int64_t S) {
    MergePlan p;
    p.ok = false; p.SP = 64; p.Q = 1; p.T_max = 1; p.one_tile_rows = MG_CAP;
    if (S < 1 || S > MG_MAX_S || stride < 1 || B < 1) return p;
    int SP = 8;
    while (stride / SP + S > MG_NS_CAP && SP < 64) SP *= 2;
    if (stride / SP + S > MG_NS_CAP) return p;
    const int64_t err = (int64_t)(SP - 1) * S;
    if (MG_CAP - err < 8 * SP) return p;
    // Rows per tile: as many as a workgroup holds when there are rows enough to fill the chip that way; with fewer rows
    // (few images, short keep lists) smaller tiles, so that about two workgroups per CU share the work, down to 512 rows.
    int64_t target = B * stride / 480;
    if (target < 512) target = 512;
    if (target > MG_CAP - err) target = MG_CAP - err;
    const int Q = (int)(target / SP) > 0 ? (int)(target / SP) : 1;
    // An image is cut only if that buys parallelism: with about two images per CU or more, one that fits a workgroup
    // stays whole.
    const int one = B >= 480 ? MG_CAP : SP * Q;
    const int64_t T_max = stride <= one ? 1 : (stride / SP + S + Q - 1) / Q;
    if (T_max * S > MG_HIST_CAP) return p;
    p.ok = true; p.SP = SP; p.Q = Q; p.T_max = (int)T_max; p.one_tile_rows = one;
    return p;
}

size_t merge_workspace_bytes(const MergePlan &p, int64_t B, int64_t S) { return merge_ws(p, B, S, nullptr).total; }

int merge_rows_launch(const MergePlan &p, const float2 *kept, const int32_t *kcount, const int32_t *seg_offsets,
                      int64_t B, int64_t stride, int64_t S, int64_t *keep, int32_t *keep_counts, const RowsOut *rows,
                      void *workspace, hipStream_t st) {
    const MergeWs w = merge_ws(p, B, S, workspace);
    {
        ProfScope ps("nms_split", st);
        const size_t lds = split_lds_bytes();
        DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_split_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        nms_split_kernel<<<dim3((unsigned)B), dim3(MG_SPLIT_THREADS), lds, st>>>(kept, kcount, seg_offsets, (int)S, stride, p.SP, p.Q,
                                                                          p.T_max, p.one_tile_rows, keep_counts, w.csmat, w.kmat, w.ntiles, g_merge_stats_on ? 1 : 0);
        DETBOX_LAUNCH_CHECK();
    }
    {
        ProfScope ps("nms_tile", st);
        const size_t lds = tile_lds_bytes();
        const dim3 grid((unsigned)p.T_max, (unsigned)B);
#define DETBOX_TILE(R, W)                                                                                                   \
    {                                                                                                                       \
        DETBOX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_tile_kernel<R, W>),                               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                              \
        nms_tile_kernel<R, W><<<grid, dim3(MG_TILE_THREADS), lds, st>>>(                                                      \
            kept, kcount, seg_offsets, (int)S, stride, p.SP, p.T_max, keep_counts, w.csmat, w.kmat, w.ntiles, keep,         \
            rows ? rows->boxes : nullptr, rows ? rows->scores : nullptr, rows ? rows->class_ids : nullptr,                  \
            rows ? rows->image_ids : nullptr, rows ? rows->image_sizes : nullptr, (int)B,                                   \
            rows ? rows->rows_ids : nullptr, rows ? rows->rows_vals : nullptr, rows ? rows->row_offsets : nullptr,          \
            rows ? rows->hdr_rows : 0, rows ? rows->wire : nullptr, g_merge_stats_on ? 1 : 0);                                                    \
    }
        if (!rows) DETBOX_TILE(false, false)
        else if (rows->wire) DETBOX_TILE(true, true)
        else DETBOX_TILE(true, false)
#undef DETBOX_TILE
        DETBOX_LAUNCH_CHECK();
    }
    return DETBOX_OK;
}

}  // namespace detbox
