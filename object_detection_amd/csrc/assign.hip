// Training-time target assignment for a whole batch.
//
// assign_anchor_kernel fuses, per (image, anchor): box conversion, the IoU row
// against every GT of the image, the row arg-max (matched GT), the label
// thresholds and the offset encoding, and at the same time reduces the IoU
// COLUMNS (best anchor per GT) with a wavefront max + ballot and one 64-bit
// atomicMax per (wave, GT).  The P x G IoU matrix never leaves registers.
// Replaces src/models/ssd300.py:245-266, src/models/retinanet.py:176-198,
// src/models/yolov3.py:161-186; assign_fcos_kernel replaces src/models/fcos.py:211-243.
//
// Layout: one thread owns one anchor (coalesced float4 loads/stores); the GT
// boxes of the current image sit in LDS as xyxy + area; a block walks several
// images so the anchor conversion is paid once.
//
// Bit-exactness: every fp32 expression keeps the operation order of the
// reference (no FMA contraction: the file is built with -ffp-contract=off;
// division is IEEE-correct).  Ties resolve to the lowest index like torch.max /
// torch.min on CPU: rows by a strict compare in ascending GT order, columns by
// packing (iou bits, ~anchor index) into one 64-bit key.
#include "detbox_internal.h"
#include <cstdlib>
#include <mutex>

using namespace detbox;

namespace {

constexpr int ASSIGN_BLOCK = 256;
constexpr int GCHUNK = 128;       // GT boxes staged in LDS at a time

// Maximum over the 64 lanes of a wavefront with data-parallel-primitive moves (no LDS traffic):
// quad swaps, half-row and row mirrors, then the row broadcasts leave the result in lane 63.
// Every lane of the wave must be active.
__device__ inline float wave_max(float v) {
#define DETBOX_DPP_MAX(ctrl, rows)                                                                                  \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rows, 0xf, false)))
    DETBOX_DPP_MAX(0xB1, 0xf);      // quad_perm [1,0,3,2]
    DETBOX_DPP_MAX(0x4E, 0xf);      // quad_perm [2,3,0,1]
    DETBOX_DPP_MAX(0x141, 0xf);     // row_half_mirror
    DETBOX_DPP_MAX(0x140, 0xf);     // row_mirror
    DETBOX_DPP_MAX(0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    DETBOX_DPP_MAX(0x143, 0xc);     // row_bcast:31 into rows 2 and 3
#undef DETBOX_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// APT anchors per thread: with several, the staging of
// an image's GT boxes, the barriers and the per-GT atomics are shared by APT * 256 anchors.
template <int MODE, int APT>
__global__ __launch_bounds__(ASSIGN_BLOCK) void assign_anchor_kernel(
    const float *__restrict__ anchors, int P,
    const float4 *__restrict__ gt_boxes, const int64_t *__restrict__ gt_labels,
    const int32_t *__restrict__ gt_off, int B, int imgs_per_block, float thr_lo, float thr_hi,
    float4 *__restrict__ locs, int64_t *__restrict__ labels, int64_t *__restrict__ matched,
    unsigned long long *__restrict__ colbest, unsigned long long *__restrict__ colpart) {
    // Per-GT arg-max over the anchors: every workgroup has its own best (IoU, anchor) word per GT box of its images.  With
    // colpart != null it stores that word, zero included, into its own column of colpart [total_gt, gridDim.x] and the
    // relabelling kernel takes the maximum of a row: plain stores, nothing to clear beforehand.  Otherwise (callers that
    // bring one slot per GT box only) the workgroups meet in colbest [total_gt] with one global atomicMax each.
    __shared__ float4 s_xyxy[GCHUNK];
    __shared__ float s_area[GCHUNK];
    __shared__ float2 s_xy[GCHUNK];                 // YOLO: raw GT top-left
    __shared__ unsigned long long s_col[GCHUNK];
    __shared__ float4 s_raw[GCHUNK];                // the GT boxes as given, for the encoding
    __shared__ int64_t s_lbl[GCHUNK];

    const int tid = threadIdx.x;
    // A wave owns APT * 64 consecutive anchors (anchor m of a lane is 64 after anchor m - 1): every access is a
    // contiguous kilobyte, and the bounds of the wave's anchors -- what lets it step over GT boxes -- stay tight.
    constexpr int MSTRIDE = 64;
    const int p0 = blockIdx.x * (ASSIGN_BLOCK * APT) + (tid >> 6) * (64 * APT) + (tid & 63);

    // anchors in registers: raw (for encoding), xyxy + area (for IoU)
    float a0[APT], a1[APT], a2[APT], a3[APT], grid[APT];
    float ax1[APT], ay1[APT], ax2[APT], ay2[APT], gx_hi[APT], gy_hi[APT], area_a[APT];
    bool valid[APT];
    const float inf = __builtin_inff();
    float bx1 = inf, by1 = inf, bx2 = -inf, by2 = -inf;
#pragma unroll
    for (int m = 0; m < APT; ++m) {
        const int p = p0 + m * MSTRIDE;
        valid[m] = p < P;
        a0[m] = 0.f; a1[m] = 0.f; a2[m] = 1.f; a3[m] = 1.f; grid[m] = 1.f;
        if (valid[m]) {
            if (MODE == DETBOX_ASSIGN_CXCYWH) {
                const float4 a = reinterpret_cast<const float4 *>(anchors)[p];
                a0[m] = a.x; a1[m] = a.y; a2[m] = a.z; a3[m] = a.w;
            } else {
                const float *a = anchors + (size_t)p * 5;
                a0[m] = a[0]; a1[m] = a[1]; a2[m] = a[2]; a3[m] = a[3]; grid[m] = a[4];
            }
        }
        gx_hi[m] = 0.f; gy_hi[m] = 0.f;
        if (MODE == DETBOX_ASSIGN_CXCYWH) {
            ax1[m] = a0[m] - 0.5f * a2[m]; ay1[m] = a1[m] - 0.5f * a3[m];
            ax2[m] = a0[m] + 0.5f * a2[m]; ay2[m] = a1[m] + 0.5f * a3[m];
        } else {
            ax1[m] = a0[m]; ay1[m] = a1[m]; ax2[m] = a0[m] + a2[m]; ay2[m] = a1[m] + a3[m];
            gx_hi[m] = a0[m] + grid[m]; gy_hi[m] = a1[m] + grid[m];
        }
        area_a[m] = (ax2[m] - ax1[m]) * (ay2[m] - ay1[m]);
        if (valid[m]) {
            bx1 = fminf(bx1, ax1[m]); by1 = fminf(by1, ay1[m]);
            bx2 = fmaxf(bx2, ax2[m]); by2 = fmaxf(by2, ay2[m]);
        }
    }
    // Bounds of the anchors of this wave: a GT box that lies outside them has IoU exactly 0 with every one of
    // them, so the wave steps over it (an IoU of 0 never enters the per-GT arg-max).
    const float wx1 = -wave_max(-bx1), wy1 = -wave_max(-by1);
    const float wx2 = wave_max(bx2), wy2 = wave_max(by2);

    const int b_begin = blockIdx.y * imgs_per_block;
    const int b_end = min(B, b_begin + imgs_per_block);
    for (int b = b_begin; b < b_end; ++b) {
        const int g0 = gt_off[b];
        const int G = gt_off[b + 1] - g0;
        const size_t o0 = (size_t)b * P + p0;
        if (G <= 0) {
#pragma unroll
            for (int m = 0; m < APT; ++m)
                if (valid[m]) {
                    const size_t o = o0 + m * MSTRIDE;
                    locs[o] = make_float4(0.f, 0.f, 0.f, 0.f); labels[o] = 0; matched[o] = -1;
                }
            continue;
        }
        float best[APT];
        int bidx[APT];
#pragma unroll
        for (int m = 0; m < APT; ++m) { best[m] = -1.0f; bidx[m] = 0; }
        for (int c0 = 0; c0 < G; c0 += GCHUNK) {
            const int gc = min(GCHUNK, G - c0);
            __syncthreads();
            if (tid < gc) {
                float4 g = gt_boxes[g0 + c0 + tid];
                float4 x;
                if (MODE == DETBOX_ASSIGN_CXCYWH)
                    x = make_float4(g.x - 0.5f * g.z, g.y - 0.5f * g.w, g.x + 0.5f * g.z, g.y + 0.5f * g.w);
                else
                    x = make_float4(g.x, g.y, g.x + g.z, g.y + g.w);
                s_xyxy[tid] = x;
                s_area[tid] = (x.z - x.x) * (x.w - x.y);
                s_xy[tid] = make_float2(g.x, g.y);
                s_col[tid] = 0ull;
                s_raw[tid] = g;
                s_lbl[tid] = gt_labels[g0 + c0 + tid];
            }
            __syncthreads();
            for (int j = 0; j < gc; ++j) {
                const float4 g = s_xyxy[j];
                // the same value in every lane: the width or the height of every intersection clamps to 0
                if (__ballot(g.x >= wx2 || g.z <= wx1 || g.y >= wy2 || g.w <= wy1) != 0ull) {
#pragma unroll
                    for (int m = 0; m < APT; ++m)
                        if (0.0f > best[m]) { best[m] = 0.0f; bidx[m] = c0 + j; }       // what an IoU of 0 does below
                    continue;
                }
                const float ga = s_area[j];
                float vmax = 0.0f;                      // largest IoU of this thread's anchors with GT j, first anchor on ties
                int mmax = 0;
#pragma unroll
                for (int m = 0; m < APT; ++m) {
                    const float w = fmaxf(fminf(ax2[m], g.z) - fmaxf(ax1[m], g.x), 0.0f);
                    const float h = fmaxf(fminf(ay2[m], g.w) - fmaxf(ay1[m], g.y), 0.0f);
                    const float inter = w * h;
                    float iou = 0.0f;
                    // inter == 0 gives exactly +0 (the union is positive), so the division is skipped
                    if (inter > 0.0f) iou = inter / ((area_a[m] + ga) - inter);
                    if (!valid[m]) iou = 0.0f;
                    if (iou > best[m]) { best[m] = iou; bidx[m] = c0 + j; }
                    float v = iou;
                    if (MODE == DETBOX_ASSIGN_YOLO) {
                        const float2 q = s_xy[j];
                        const bool in_grid = (a0[m] <= q.x) && (q.x < gx_hi[m]) && (a1[m] <= q.y) && (q.y < gy_hi[m]);
                        v = in_grid ? iou : 0.0f;
                    }
                    if (v > vmax) { vmax = v; mmax = m; }
                }
                if (__ballot(vmax > 0.0f) != 0ull) {           // wave-uniform
                    const float wm = wave_max(vmax);
                    if (vmax == wm) {                          // several lanes only on exact ties: the lowest anchor wins the max
                        const unsigned p = (unsigned)(p0 + mmax * MSTRIDE);
                        atomicMax(&s_col[j], ((unsigned long long)__float_as_uint(wm) << 32) | (unsigned long long)(0xffffffffu - p));
                    }
                }
            }
            __syncthreads();
            if (tid < gc) {
                const unsigned long long k = s_col[tid];
                if (colpart) colpart[(size_t)(g0 + c0 + tid) * gridDim.x + blockIdx.x] = k;
                else if (k != 0ull) atomicMax(&colbest[g0 + c0 + tid], k);
            }
        }
        // the last chunk is still in LDS (the only one unless an image has more than GCHUNK boxes)
        const int c_last = ((G - 1) / GCHUNK) * GCHUNK;
#pragma unroll
        for (int m = 0; m < APT; ++m) {
            if (!valid[m]) continue;
            const size_t o = o0 + m * MSTRIDE;
            const bool in_lds = bidx[m] >= c_last;
            const float4 g = in_lds ? s_raw[bidx[m] - c_last] : gt_boxes[g0 + bidx[m]];
            const int64_t lbl = in_lds ? s_lbl[bidx[m] - c_last] : gt_labels[g0 + bidx[m]];
            float4 d;
            if (MODE == DETBOX_ASSIGN_CXCYWH) {
                d.x = (10.0f * (g.x - a0[m])) / a2[m];
                d.y = (10.0f * (g.y - a1[m])) / a3[m];
            } else {
                d.x = (10.0f * (g.x - a0[m])) / grid[m];
                d.y = (10.0f * (g.y - a1[m])) / grid[m];
            }
            d.z = 5.0f * logf(g.z / a2[m]);
            d.w = 5.0f * logf(g.w / a3[m]);
            store_nt(&locs[o], d);
            store_nt(&matched[o], (int64_t)bidx[m]);                // the relabelling kernel reads back one row per GT box
            store_nt(&labels[o], (best[m] < thr_lo) ? (int64_t)0 : ((best[m] < thr_hi) ? (int64_t)-1 : lbl));
        }
    }
}

// The best anchor of every GT is labelled as if its max IoU were thr_hi
// (`max_ious[best_ids] = thr`, src/models/ssd300.py:258).  It keeps the label
// of its own arg-max GT: the reference's "assign GT j" loop writes into a
// temporary (ssd300.py:256-257).
__global__ __launch_bounds__(256) void assign_fixup_kernel(
    const unsigned long long *__restrict__ colbest, const unsigned long long *__restrict__ colpart, int nwg,
    const int32_t *__restrict__ gt_off, int B, int total_gt, int P,
    const int64_t *__restrict__ gt_labels, const int64_t *__restrict__ matched, float thr_lo, float thr_hi,
    int64_t *__restrict__ labels, int64_t *__restrict__ best_anchor) {
    // with the per-workgroup table: one wave per GT box (the row's words are spread over the lanes); else one thread
    const int t = colpart ? (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= total_gt) return;
    int lo = 0, hi = B;                      // largest b with gt_off[b] <= t
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (gt_off[mid] <= t) lo = mid; else hi = mid;
    }
    const int b = lo;
    unsigned long long k = 0ull;
    if (colpart) {                           // the best word over the assignment's workgroups (a row of colpart)
        const unsigned long long *row = colpart + (size_t)t * nwg;
        for (int w = (int)(threadIdx.x & 63); w < nwg; w += 64) { const unsigned long long v = row[w]; k = v > k ? v : k; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long v = ((unsigned long long)(unsigned)__shfl_xor((int)(k >> 32), o) << 32) | (unsigned)__shfl_xor((int)k, o);
            k = v > k ? v : k;
        }
        if ((threadIdx.x & 63) != 0) return;
    } else {
        k = colbest[t];
    }
    const unsigned pbest = (k == 0ull) ? 0u : (0xffffffffu - (unsigned)(k & 0xffffffffull));
    if (best_anchor) best_anchor[t] = (int64_t)pbest;
    const size_t o = (size_t)b * P + pbest;
    const int64_t m = matched[o];
    labels[o] = (thr_hi < thr_lo) ? 0 : gt_labels[gt_off[b] + m];
}

__global__ __launch_bounds__(ASSIGN_BLOCK) void assign_fcos_kernel(
    const float2 *__restrict__ points, const float2 *__restrict__ ranges, int P,
    const float4 *__restrict__ gt_boxes, const int64_t *__restrict__ gt_labels,
    const int32_t *__restrict__ gt_off, int B, int imgs_per_block,
    float4 *__restrict__ locs, float *__restrict__ cents, int64_t *__restrict__ labels,
    int64_t *__restrict__ matched) {
    __shared__ float4 s_xyxy[GCHUNK];
    __shared__ float s_area[GCHUNK];
    const float INF = 1e8f;
    const int tid = threadIdx.x;
    const int p = blockIdx.x * ASSIGN_BLOCK + tid;
    const bool valid = p < P;
    float2 pt = make_float2(0.f, 0.f), rr = make_float2(0.f, 0.f);
    if (valid) { pt = points[p]; rr = ranges[p]; }

    const int b_begin = blockIdx.y * imgs_per_block;
    const int b_end = min(B, b_begin + imgs_per_block);
    for (int b = b_begin; b < b_end; ++b) {
        const int g0 = gt_off[b];
        const int G = gt_off[b + 1] - g0;
        const size_t o = (size_t)b * P + p;
        if (G <= 0) {
            if (valid) { locs[o] = make_float4(0.f, 0.f, 0.f, 0.f); cents[o] = 0.f; labels[o] = 0; matched[o] = -1; }
            continue;
        }
        float best = __builtin_inff();
        int bidx = 0;
        for (int c0 = 0; c0 < G; c0 += GCHUNK) {
            const int gc = min(GCHUNK, G - c0);
            __syncthreads();
            if (tid < gc) {
                float4 g = gt_boxes[g0 + c0 + tid];
                float4 x = make_float4(g.x - 0.5f * g.z, g.y - 0.5f * g.w, g.x + 0.5f * g.z, g.y + 0.5f * g.w);
                s_xyxy[tid] = x;
                s_area[tid] = (x.z - x.x) * (x.w - x.y);
            }
            __syncthreads();
            for (int j = 0; j < gc; ++j) {
                const float4 g = s_xyxy[j];
                const float l = pt.x - g.x, r = g.z - pt.x, t = pt.y - g.y, bt = g.w - pt.y;
                const float mn = fminf(fminf(l, r), fminf(t, bt));
                const float mx = fmaxf(fmaxf(l, r), fmaxf(t, bt));
                const bool ok = (mn > 0.0f) && (rr.x <= mx) && (mx <= rr.y);
                const float area = ok ? s_area[j] : INF;
                if (area < best) { best = area; bidx = c0 + j; }
            }
        }
        if (valid) {
            const float4 g = gt_boxes[g0 + bidx];
            const float x1 = g.x - 0.5f * g.z, y1 = g.y - 0.5f * g.w, x2 = g.x + 0.5f * g.z, y2 = g.y + 0.5f * g.w;
            const float l = pt.x - x1, r = x2 - pt.x, t = pt.y - y1, bt = y2 - pt.y;
            locs[o] = make_float4(l, r, t, bt);
            // a / b * c / d evaluated left to right (src/models/fcos.py:235-237)
            cents[o] = sqrtf(((fminf(l, r) / fmaxf(l, r)) * fminf(t, bt)) / fmaxf(t, bt));
            matched[o] = bidx;
            labels[o] = (best == INF) ? 0 : gt_labels[g0 + bidx];
        }
    }
}

int pick_imgs_per_block(int64_t tiles, int64_t B) {
    // One image per workgroup measured fastest at every batch size (41 us at 1, 46 at 4, 80 at 16 images per
    // workgroup on 16 x 120087 anchors): the per-image work is a latency chain that only more resident
    // workgroups hide.  More than one only where the grid's y limit forces it.
    (void)tiles;
    int64_t ipb = 1;
    while ((B + ipb - 1) / ipb > 65535) ++ipb;
    return (int)ipb;
}

}  // namespace

extern "C" size_t detbox_assign_anchor_workspace_bytes(int64_t total_gt) {
    if (total_gt < 0) total_gt = 0;
    return align_up((size_t)total_gt * sizeof(unsigned long long) + 256, 256);
}

namespace {
// four anchors per thread once that still leaves about four workgroups per CU (33 against 44 us on 16 x 120087 anchors)
int assign_apt(int64_t P, int64_t B) { return (P * B >= (int64_t)4 * ASSIGN_BLOCK * 1024) ? 4 : 1; }
int64_t assign_tiles(int64_t P, int64_t B) { const int apt = assign_apt(P, B); return (P + ASSIGN_BLOCK * apt - 1) / (ASSIGN_BLOCK * apt); }
// bytes of the [total_gt, workgroups] table of per-workgroup arg-max words
size_t assign_part_bytes(int64_t total_gt, int64_t P, int64_t B) {
    return align_up((size_t)(total_gt < 0 ? 0 : total_gt) * (size_t)assign_tiles(P, B) * sizeof(unsigned long long) + 256, 256);
}
constexpr size_t ASSIGN_PART_MAX_BYTES = (size_t)256 << 20;       // beyond this the one-slot-per-GT form is asked for
}  // namespace

extern "C" size_t detbox_assign_anchor_workspace_bytes_for(int64_t total_gt, int64_t P, int64_t B) {
    const size_t one = detbox_assign_anchor_workspace_bytes(total_gt);
    if (P <= 0 || B <= 0) return one;
    const size_t part = assign_part_bytes(total_gt, P, B);
    return part <= ASSIGN_PART_MAX_BYTES && part > one ? part : one;
}

namespace {

// Shared by the two entry points.  Either `colpart` holds one 64-bit word per (GT box, workgroup of the assignment) --
// written whole by the assignment, nothing to clear -- or `colbest` holds one slot per GT box (cleared here, on the stream).
int assign_anchor_run(int mode, const float *anchors, int64_t P,
                      const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                      int64_t B, int64_t total_gt, float thr_lo, float thr_hi,
                      float *target_locs, int64_t *target_labels, int64_t *matched,
                      int64_t *best_anchor, unsigned long long *colbest, unsigned long long *colpart, hipStream_t s) {
    if (total_gt > 0 && !colpart) DETBOX_HIP(hipMemsetAsync(colbest, 0, (size_t)total_gt * sizeof(unsigned long long), s));
    const int apt = assign_apt(P, B);
    const int64_t tiles = assign_tiles(P, B);
    const int ipb = pick_imgs_per_block(tiles, B);
    dim3 grid((unsigned)tiles, (unsigned)((B + ipb - 1) / ipb));
    {
        ProfScope ps("assign_anchor", s);
#define DETBOX_ASSIGN(MODE, APT)                                                                                   \
    assign_anchor_kernel<MODE, APT><<<grid, dim3(ASSIGN_BLOCK), 0, s>>>(                                           \
        anchors, (int)P, reinterpret_cast<const float4 *>(gt_boxes), gt_labels, gt_offsets, (int)B, ipb, thr_lo,   \
        thr_hi, reinterpret_cast<float4 *>(target_locs), target_labels, matched, colbest, colpart)
        if (mode == DETBOX_ASSIGN_CXCYWH) { if (apt == 4) DETBOX_ASSIGN(DETBOX_ASSIGN_CXCYWH, 4); else DETBOX_ASSIGN(DETBOX_ASSIGN_CXCYWH, 1); }
        else { if (apt == 4) DETBOX_ASSIGN(DETBOX_ASSIGN_YOLO, 4); else DETBOX_ASSIGN(DETBOX_ASSIGN_YOLO, 1); }
#undef DETBOX_ASSIGN
        DETBOX_LAUNCH_CHECK();
    }
    if (total_gt > 0) {
        ProfScope ps("assign_fixup", s);
        // (table form: a wave per GT box, four to a workgroup)
        assign_fixup_kernel<<<dim3((unsigned)(colpart ? (total_gt + 3) / 4 : (total_gt + 255) / 256)), dim3(256), 0, s>>>(
            colbest, colpart, (int)tiles, gt_offsets, (int)B, (int)total_gt, (int)P, gt_labels, matched, thr_lo, thr_hi,
            target_labels, best_anchor);
        DETBOX_LAUNCH_CHECK();
    }
    return DETBOX_OK;
}

int assign_anchor_check(int mode, const float *anchors, int64_t P, const float *gt_boxes, const int64_t *gt_labels,
                        const int32_t *gt_offsets, int64_t B, int64_t total_gt, const float *target_locs,
                        const int64_t *target_labels, const int64_t *matched) {
    DETBOX_CHECK_ARG(mode == DETBOX_ASSIGN_CXCYWH || mode == DETBOX_ASSIGN_YOLO, "unknown mode");
    DETBOX_CHECK_ARG(P >= 0 && B >= 0 && total_gt >= 0, "negative size");
    DETBOX_CHECK_ARG(P < (1ll << 31) && total_gt < (1ll << 31), "P and total_gt must fit 31 bits");
    if (P == 0 || B == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(anchors && gt_offsets && target_locs && target_labels && matched, "null pointer");
    DETBOX_CHECK_ARG(total_gt == 0 || (gt_boxes && gt_labels), "null GT pointer");
    DETBOX_CHECK_ARG((uintptr_t)target_locs % 16 == 0 && (uintptr_t)gt_boxes % 16 == 0, "float4 arrays must be 16-byte aligned");
    DETBOX_CHECK_ARG(mode != DETBOX_ASSIGN_CXCYWH || (uintptr_t)anchors % 16 == 0, "anchors must be 16-byte aligned");
    return DETBOX_OK;
}

}  // namespace

extern "C" int detbox_assign_anchor(int mode, const float *anchors, int64_t P,
                                    const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                                    int64_t B, int64_t total_gt, float thr_lo, float thr_hi,
                                    float *target_locs, int64_t *target_labels, int64_t *matched,
                                    int64_t *best_anchor, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = assign_anchor_check(mode, anchors, P, gt_boxes, gt_labels, gt_offsets, B, total_gt, target_locs,
                                     target_labels, matched)) return rc;
    if (P == 0 || B == 0) return DETBOX_OK;
    if (workspace_bytes < detbox_assign_anchor_workspace_bytes(total_gt) || !workspace || (uintptr_t)workspace % 256) {
        set_error("detbox_assign_anchor: workspace missing, misaligned or smaller than %zu bytes",
                  detbox_assign_anchor_workspace_bytes(total_gt));
        return DETBOX_ERR_WORKSPACE;
    }
    // a workspace of detbox_assign_anchor_workspace_bytes_for(...) bytes holds the per-workgroup table; a smaller one the slots
    unsigned long long *ws = static_cast<unsigned long long *>(workspace);
    const bool table = workspace_bytes >= assign_part_bytes(total_gt, P, B) && assign_part_bytes(total_gt, P, B) <= ASSIGN_PART_MAX_BYTES;
    return assign_anchor_run(mode, anchors, P, gt_boxes, gt_labels, gt_offsets, B, total_gt, thr_lo, thr_hi, target_locs,
                             target_labels, matched, best_anchor, table ? nullptr : ws, table ? ws : nullptr,
                             static_cast<hipStream_t>(stream));
}

// ---- the same assignment for callers that cannot allocate -------------------------------------------------
// The per-GT arg-max slots live in a device array of this library instead of a caller workspace.  Every GPU the
// library runs on holds its own copy at its own address; the address is looked up once per device (with that
// device current) and kept.  Host code only ever uses that looked-up pointer.
static __device__ __align__(256) unsigned long long scratch[DETBOX_ASSIGN_NOALLOC_MAX_GT];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
unsigned long long *g_scratch_ptr[MAX_DEVICES];       // per device: where `scratch` lives there

int scratch_of_current_device(unsigned long long **out) {
    int dev = -1;
    DETBOX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) {
        set_error("detbox_assign_anchor_noalloc: device ordinal %d beyond the %d this entry point keeps track of", dev, MAX_DEVICES);
        return DETBOX_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        DETBOX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { set_error("detbox_assign_anchor_noalloc: no address for the scratch array on device %d", dev); return DETBOX_ERR_HIP; }
        g_scratch_ptr[dev] = static_cast<unsigned long long *>(p);
    }
    *out = g_scratch_ptr[dev];
    return DETBOX_OK;
}
}  // namespace

extern "C" int detbox_assign_anchor_noalloc(int mode, const float *anchors, int64_t P,
                                            const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                                            int64_t B, int64_t total_gt, float thr_lo, float thr_hi,
                                            float *target_locs, int64_t *target_labels, int64_t *matched,
                                            int64_t *best_anchor, void *stream) {
    if (int rc = assign_anchor_check(mode, anchors, P, gt_boxes, gt_labels, gt_offsets, B, total_gt, target_locs,
                                     target_labels, matched)) return rc;
    if (total_gt > DETBOX_ASSIGN_NOALLOC_MAX_GT) {
        set_error("detbox_assign_anchor_noalloc: %lld ground-truth boxes in the batch, the built-in scratch holds %d; "
                  "use detbox_assign_anchor with a workspace", (long long)total_gt, DETBOX_ASSIGN_NOALLOC_MAX_GT);
        return DETBOX_ERR_UNSUPPORTED;
    }
    if (P == 0 || B == 0) return DETBOX_OK;
    unsigned long long *colbest = nullptr;
    if (int rc = scratch_of_current_device(&colbest)) return rc;
    // assign_anchor_run clears the slots it uses on the stream before the kernels: the array keeps the last call's contents
    return assign_anchor_run(mode, anchors, P, gt_boxes, gt_labels, gt_offsets, B, total_gt, thr_lo, thr_hi, target_locs,
                             target_labels, matched, best_anchor, colbest, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int detbox_assign_fcos(const float *points, const float *regress_ranges, int64_t P,
                                  const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                                  int64_t B, int64_t total_gt,
                                  float *target_locs, float *target_cents, int64_t *target_labels, int64_t *matched,
                                  void *stream) {
    DETBOX_CHECK_ARG(P >= 0 && B >= 0 && total_gt >= 0, "negative size");
    DETBOX_CHECK_ARG(P < (1ll << 31) && total_gt < (1ll << 31), "P and total_gt must fit 31 bits");
    if (P == 0 || B == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(points && regress_ranges && gt_offsets && target_locs && target_cents && target_labels && matched,
                     "null pointer");
    DETBOX_CHECK_ARG(total_gt == 0 || (gt_boxes && gt_labels), "null GT pointer");
    DETBOX_CHECK_ARG((uintptr_t)target_locs % 16 == 0 && (uintptr_t)gt_boxes % 16 == 0 &&
                     (uintptr_t)points % 8 == 0 && (uintptr_t)regress_ranges % 8 == 0, "misaligned array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t tiles = (P + ASSIGN_BLOCK - 1) / ASSIGN_BLOCK;
    const int ipb = pick_imgs_per_block(tiles, B);
    dim3 grid((unsigned)tiles, (unsigned)((B + ipb - 1) / ipb));
    ProfScope ps("assign_fcos", s);
    assign_fcos_kernel<<<grid, dim3(ASSIGN_BLOCK), 0, s>>>(
        reinterpret_cast<const float2 *>(points), reinterpret_cast<const float2 *>(regress_ranges), (int)P,
        reinterpret_cast<const float4 *>(gt_boxes), gt_labels, gt_offsets, (int)B, ipb,
        reinterpret_cast<float4 *>(target_locs), target_cents, target_labels, matched);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
