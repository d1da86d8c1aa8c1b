/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for greedy non-maximum suppression.
 *
 * This file is a plain-C restatement of the greedy CPU NMS that the reference
 * reaches through torchvision.ops.batched_nms (call site:
 * src/utilities/predictor.py:55).  torchvision is an un-vendored, un-pinned
 * third-party dependency of the reference (setup.cfg:14-15 pins only
 * albumentations), so the algorithm below follows torchvision's published CPU
 * kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp):
 *
 *   areas = (x2-x1)*(y2-y1)                      fp32
 *   order = stable argsort(scores, descending)   ties: lower index first
 *   for i in order (not suppressed): keep i; for every later j not suppressed:
 *       w = max(0, min(x2i,x2j) - max(x1i,x1j)); h likewise      fp32
 *       inter = w*h; ovr = inter / (area_i + area_j - inter)      fp32
 *       suppress j iff (double)ovr > iou_threshold                (NaN -> keep)
 *   return kept indices in descending-score order.
 *
 * Nothing in the shipped package links or loads this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: no FMA contraction, so
 * every fp32 operation rounds exactly as the scalar C++ kernel does).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float score;
    int64_t idx;
} oracle_item_t;

/* Stable descending merge sort on (score) with original order kept for ties. */
static void oracle_merge_sort(oracle_item_t *a, oracle_item_t *tmp, int64_t n) {
    for (int64_t width = 1; width < n; width *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * width) {
            int64_t mid = lo + width < n ? lo + width : n;
            int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take from the right run only when strictly greater: stable */
                if (a[j].score > a[i].score) tmp[k++] = a[j++];
                else tmp[k++] = a[i++];
            }
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(oracle_item_t));
    }
}

/*
 * boxes  : [n,4] xyxy fp32, row-major
 * scores : [n]
 * keep   : [n] out, kept indices in descending-score order
 * returns the number of kept boxes (or -1 when out of memory).
 */
int64_t oracle_nms(const float *boxes, const float *scores, int64_t n,
                   double iou_threshold, int64_t *keep) {
    if (n <= 0) return 0;
    oracle_item_t *items = (oracle_item_t *)malloc((size_t)n * sizeof(oracle_item_t));
    oracle_item_t *tmp = (oracle_item_t *)malloc((size_t)n * sizeof(oracle_item_t));
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    uint8_t *suppressed = (uint8_t *)calloc((size_t)n, 1);
    if (!items || !tmp || !areas || !suppressed) {
        free(items); free(tmp); free(areas); free(suppressed);
        return -1;
    }
    for (int64_t i = 0; i < n; ++i) {
        items[i].score = scores[i];
        items[i].idx = i;
        areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    oracle_merge_sort(items, tmp, n);

    int64_t num_keep = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = items[_i].idx;
        if (suppressed[i]) continue;
        keep[num_keep++] = i;
        float ix1 = boxes[4 * i + 0], iy1 = boxes[4 * i + 1];
        float ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = items[_j].idx;
            if (suppressed[j]) continue;
            /* std::max(a,b) == (a<b)?b:a and std::min(a,b) == (b<a)?b:a */
            float xx1 = ix1 < boxes[4 * j + 0] ? boxes[4 * j + 0] : ix1;
            float yy1 = iy1 < boxes[4 * j + 1] ? boxes[4 * j + 1] : iy1;
            float xx2 = boxes[4 * j + 2] < ix2 ? boxes[4 * j + 2] : ix2;
            float yy2 = boxes[4 * j + 3] < iy2 ? boxes[4 * j + 3] : iy2;
            float w = xx2 - xx1; w = 0.0f < w ? w : 0.0f;
            float h = yy2 - yy1; h = 0.0f < h ? h : 0.0f;
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr > iou_threshold) suppressed[j] = 1;
        }
    }
    free(items); free(tmp); free(areas); free(suppressed);
    return num_keep;
}
