/*
 * detbox -- C ABI of the MI355X (gfx950) detection box-ops library.
 *
 * The reference (rs1004/object_detection) has no FFI layer: its box-ops path is
 * plain Python on torch + torchvision.ops.  Each entry point below names the
 * reference code it replaces (paths relative to the reference root) and is what
 * a binding for that call site would bind.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller, contiguous, row-major;
 *     fp32 boxes/scores, int64 labels/class ids/indices, int32 offsets/counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued asynchronously on it, nothing synchronises the device;
 *   - functions return DETBOX_OK or a negative DETBOX_ERR_* code and never throw;
 *     detbox_last_error() returns a thread-local message for the last failure;
 *   - functions taking `workspace` need `*_workspace_bytes(...)` bytes of
 *     256-byte-aligned scratch; its contents are undefined afterwards, and two
 *     calls sharing one workspace must be ordered on one stream;
 *   - no entry point allocates or frees device memory.
 */
#ifndef DETBOX_H
#define DETBOX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DETBOX_OK 0
#define DETBOX_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, unknown mode) */
#define DETBOX_ERR_WORKSPACE (-2)   /* workspace missing, misaligned or too small */
#define DETBOX_ERR_HIP (-3)         /* a HIP runtime call failed */
#define DETBOX_ERR_UNSUPPORTED (-4) /* size beyond a documented limit */

/* box formats of torchvision.ops.box_convert */
#define DETBOX_FMT_XYXY 0
#define DETBOX_FMT_XYWH 1
#define DETBOX_FMT_CXCYWH 2

/* anchor-assignment modes */
#define DETBOX_ASSIGN_CXCYWH 0 /* SSD / RetinaNet: anchors [P,4] cxcywh, GT cxcywh */
#define DETBOX_ASSIGN_YOLO 1   /* YOLOv3: priors [P,5] = x,y,w,h,grid_length; GT xywh */

/* decode modes */
#define DETBOX_DECODE_CXCYWH 0 /* anchors [P,4] cxcywh  (SSD / RetinaNet) */
#define DETBOX_DECODE_YOLO 1   /* priors  [P,5] x,y,w,h,grid  (YOLOv3)    */
#define DETBOX_DECODE_FCOS 2   /* points  [P,2] pixels  (FCOS)            */

#define DETBOX_MAX_TOP_K 4096     /* largest top_k of detbox_select_decode */
#define DETBOX_NMS_MAX_SORT 4096  /* largest unsorted segment of detbox_batched_nms_segments */

int detbox_version(void);
const char *detbox_last_error(void);

/* ---- primitives (torchvision.ops; third-party to the reference) -------------------------- */

/* torchvision.ops.box_convert(boxes[n,4], in_fmt, out_fmt) -> out[n,4].
 * Reference call sites: src/models/ssd300.py:249-250,374; src/models/retinanet.py:180-181,302;
 * src/models/yolov3.py:168-169,321; src/models/fcos.py:215; src/datasets/dataset.py:59. */
int detbox_box_convert(const float *boxes, float *out, int64_t n, int in_fmt, int out_fmt, void *stream);

/* torchvision.ops.box_iou(boxes1[n,4] xyxy, boxes2[m,4] xyxy) -> iou[n,m]  (no epsilon).
 * Reference call sites: src/models/ssd300.py:251; src/models/retinanet.py:182; src/models/yolov3.py:170. */
int detbox_box_iou(const float *boxes1, int64_t n, const float *boxes2, int64_t m, float *iou, void *stream);

/* ---- training-time target assignment ("Step 1" of every loss()) -------------------------- */

/* Replaces src/models/ssd300.py:245-266, src/models/retinanet.py:176-198 and
 * src/models/yolov3.py:161-186 for a whole batch in one call.
 *   anchors      [P,4] (CXCYWH mode) or [P,5] (YOLO mode)
 *   gt_boxes     [sum G,4], gt_labels [sum G], gt_offsets [B+1] (image i owns rows
 *                gt_offsets[i] .. gt_offsets[i+1]-1 of both)
 *   thr_lo/hi    labels: max_iou < thr_lo -> 0 (background); else max_iou < thr_hi -> -1 (ignore);
 *                else the label of the arg-max GT.  SSD: lo = hi = 0.5; RetinaNet: 0.4, 0.5;
 *                YOLOv3: 0.5, 1.0.  The best anchor of every GT has its max_iou replaced by thr_hi
 *                before labelling (and keeps the label of its OWN arg-max GT, as the reference does).
 *   target_locs  [B,P,4] encoded offsets of the arg-max GT (ssd300.py:299-318 / yolov3.py:236-257)
 *   target_labels[B,P], matched [B,P] (arg-max GT index inside the image; lowest index on ties)
 *   best_anchor  [sum G] or NULL: per GT the anchor of maximal IoU (YOLO: IoU masked by the
 *                in-grid test), lowest index on ties
 * An image with no GT gets labels 0, matched -1 and zero offsets (the reference raises). */
size_t detbox_assign_anchor_workspace_bytes(int64_t total_gt);
/* The larger workspace with which detbox_assign_anchor does without global atomics: one word per (GT box, workgroup of
 * the assignment) that the workgroups store and the relabelling step reduces (nothing to clear beforehand: one launch less per call).  Never smaller than detbox_assign_anchor_workspace_bytes(total_gt), which detbox_assign_anchor
 * accepts as well (it then meets in one slot per GT box with an atomic per workgroup).  No counterpart in the reference. */
size_t detbox_assign_anchor_workspace_bytes_for(int64_t total_gt, int64_t P, int64_t B);
int detbox_assign_anchor(int mode, const float *anchors, int64_t P,
                         const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                         int64_t B, int64_t total_gt, float thr_lo, float thr_hi,
                         float *target_locs, int64_t *target_labels, int64_t *matched,
                         int64_t *best_anchor, void *workspace, size_t workspace_bytes, void *stream);

/* detbox_assign_anchor for callers that cannot allocate device memory or bring a workspace: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same
 * arguments, results and kernels as detbox_assign_anchor, minus the workspace: the per-GT arg-max slots live in an
 * array inside the library (8 * DETBOX_ASSIGN_NOALLOC_MAX_GT bytes = 512 KiB on every GPU the library runs on,
 * held until the process ends).  total_gt beyond DETBOX_ASSIGN_NOALLOC_MAX_GT returns DETBOX_ERR_UNSUPPORTED.
 * The call makes no hipMalloc / hipFree; the device that `stream` belongs to must be current.  All calls on one
 * GPU share that GPU's array: calls that overlap there (two streams, two threads) race; order them on one
 * stream or use detbox_assign_anchor with separate workspaces.
 * Graph capture: the first call on a device looks the array's address up (hipGetSymbolAddress), which is set-up work,
 * not a stream operation.  Make one ordinary call per device (any small input) before capturing; from then on the
 * entry point issues a memset and kernel launches on `stream` and nothing else, and can be captured and replayed. */
#define DETBOX_ASSIGN_NOALLOC_MAX_GT 65536
int detbox_assign_anchor_noalloc(int mode, const float *anchors, int64_t P,
                                 const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                                 int64_t B, int64_t total_gt, float thr_lo, float thr_hi,
                                 float *target_locs, int64_t *target_labels, int64_t *matched,
                                 int64_t *best_anchor, void *stream);

/* Replaces src/models/fcos.py:211-243.  points [P,2], regress_ranges [P,2] (inclusive both ends),
 * GT cxcywh in pixels.  target_locs [B,P,4] = rays (l,r,t,b) of the matched box, target_cents [B,P],
 * target_labels [B,P], matched [B,P].  Rows with label 0 hold the rays of GT 0 and a centerness
 * that may be NaN, exactly like the reference; images with no GT get zeros and matched -1. */
int detbox_assign_fcos(const float *points, const float *regress_ranges, int64_t P,
                       const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_offsets,
                       int64_t B, int64_t total_gt,
                       float *target_locs, float *target_cents, int64_t *target_labels, int64_t *matched,
                       void *stream);

/* Hard-negative mining rank mask.  Replaces `loss_neg.argsort(descending=True).argsort() < 3 * num_pos`
 * of src/models/ssd300.py:277-278 (ssd512 inherits it): out_mask[b,p] = 1 iff the descending rank of
 * values[b,p] inside row b is below k[b]; equal values rank by ascending index (the reference's sort
 * leaves that order unspecified); NaN ranks first, -0 equals +0.  values [B,P], k [B], out_mask [B,P]
 * bytes (0 / 1).  k[b] <= 0 selects nothing, k[b] >= P everything. */
int detbox_topk_mask(const float *values, const int64_t *k, int64_t B, int64_t P, uint8_t *out_mask, void *stream);

/* Focal confidence loss, summed over the rows that take part (label >= 0; the ignore label -1 and any
 * label >= C are skipped).  Replaces the boolean-mask gather and focal_loss(..., reduction='sum') of
 * src/models/losses.py:5-9 as called at src/models/retinanet.py:215 and src/models/fcos.py:271:
 *   loss_row = -((alpha * (1 - pt)^gamma) * log_pt),  pt = softmax(logits[row])[label],
 *   log_pt = log_softmax(logits[row])[label].
 * logits [rows, C] (C <= 256), labels [rows]; *loss_sum (device, float64) receives the sum.
 * The backward call writes grad_logits [rows, C] = *grad_scale * d(sum)/d(logits) (zero rows where the
 * label does not take part); grad_scale is a device scalar (the upstream gradient, e.g. 1/N).
 * Floating point: the sum is accumulated in float64, so it differs from the reference's fp32 sum only by
 * the reference's own rounding (tested to 1e-5 relative). */
int detbox_focal_loss_forward(const float *logits, const int64_t *labels, int64_t rows, int64_t C,
                              float alpha, float gamma, double *loss_sum, void *stream);
int detbox_focal_loss_backward(const float *logits, const int64_t *labels, int64_t rows, int64_t C,
                               float alpha, float gamma, const float *grad_scale, float *grad_logits, void *stream);

/* Localisation loss of the SSD and RetinaNet heads with its mask, summed:
 *   F.smooth_l1_loss(out_locs[pos_mask], target_locs[pos_mask], reduction='sum'),  pos_mask = labels > 0
 * (src/models/ssd300.py:283, src/models/retinanet.py:211; beta = 1).  input, target [rows, 4], labels [rows];
 * *loss_sum (device, float64) receives the sum.  The backward call writes grad_input [rows, 4] =
 * *grad_scale * d(sum)/d(input) (zero rows where labels <= 0); grad_scale is a device scalar.
 * Floating point: every term is the reference's fp32 value, the sum is accumulated in float64 (tested to
 * 1e-5 relative against the reference's fp32 sum). */
int detbox_smooth_l1_masked_forward(const float *input, const float *target, const int64_t *labels,
                                    int64_t rows, double *loss_sum, void *stream);
int detbox_smooth_l1_masked_backward(const float *input, const float *target, const int64_t *labels,
                                     int64_t rows, const float *grad_scale, float *grad_input, void *stream);

/* ---- inference: activation + threshold + exact top-k + decode ---------------------------- */

/* activations applied to the head output before the threshold */
#define DETBOX_ACT_NONE 0            /* input already holds the scores                                  */
#define DETBOX_ACT_SOFTMAX 1         /* softmax over C                (ssd300.py:354, retinanet.py:282) */
#define DETBOX_ACT_SOFTMAX_SIGMOID 2 /* softmax over C * sigmoid(extra[b,p])             (fcos.py:314)  */
#define DETBOX_ACT_SIGMOID_SIGMOID 3 /* sigmoid * sigmoid(extra[b,p])                (yolov3.py:298-300) */

/* Writes the activated scores out[B,P,C] of in[B,P,C] (extra [B,P] for the two activations that use
 * it).  The arithmetic is the one detbox_select_decode applies internally, bit for bit, so a caller
 * (or a test) can see exactly the scores the selection ranked.  Limit: C <= 256. */
int detbox_activate_scores(int activation, const float *in, const float *extra, int64_t B, int64_t P, int64_t C,
                           float *out, void *stream);

/* Replaces the activation and the per-image, per-class loop of every pre_predict
 * (src/models/ssd300.py:354-382, src/models/retinanet.py:282-310, src/models/fcos.py:314-341,
 * src/models/yolov3.py:298-329): for every image b and class c in [class_begin, C) keep the anchors with
 *     score[b,p,c] > conf_thresh  and  descending rank of score[b,:,c] < top_k
 * (ties at the boundary: lower anchor index wins), decode their boxes to xyxy and emit them
 * class-major, anchor-index ascending.
 *   scores        [B,P,C] head output (logits, or scores when activation == DETBOX_ACT_NONE);
 *                 extra [B,P] or NULL; locs [B,P,4]; anchors per decode_mode
 *   score_hi_hint upper end of the expected score range (1.0 for probabilities).  Speed only.
 *   out_*         image b owns rows b*stride .. b*stride+out_counts[b]-1, stride = (C-class_begin)*top_k;
 *                 out_boxes [B*stride,4], out_scores [B*stride], out_class_ids [B*stride]
 *                 (= c + class_id_offset)
 *   out_seg_offsets [B, (C-class_begin)+1]: class segment boundaries inside image b's rows
 *   out_counts    [B]
 * Limits: 1 <= top_k <= DETBOX_MAX_TOP_K, C <= 256, C - class_begin <= 128.  The workspace holds
 * the staging lists, the samples and the histograms; no copy of the activated scores is kept (the exact path, taken
 * when the sampled guess fails or P < 4096, recomputes them from the head output), so its size does not depend on
 * `activation`. */
size_t detbox_select_decode_workspace_bytes(int64_t B, int64_t P, int64_t C, int class_begin, int top_k,
                                            int activation);
int detbox_select_decode(int decode_mode, int activation, const float *scores, const float *extra,
                         const float *locs, const float *anchors,
                         int64_t B, int64_t P, int64_t C, int class_begin, int class_id_offset,
                         float conf_thresh, int top_k, float score_hi_hint,
                         float *out_boxes, float *out_scores, int64_t *out_class_ids,
                         int32_t *out_seg_offsets, int32_t *out_counts,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- inference: per-class NMS and result rows -------------------------------------------- */

/* Replaces torchvision.ops.batched_nms as called at src/utilities/predictor.py:55, for a batch whose
 * boxes are already grouped by class: image b owns rows b*stride.., segment s of image b is rows
 * seg_offsets[b,s] .. seg_offsets[b,s+1]-1 (relative to the image), S segments per image.
 * Greedy NMS inside each segment in stable descending score order, suppressing when
 * (double)iou > iou_threshold; keep[b, 0..keep_counts[b]) are the kept row indices (relative to the
 * image) of all segments in stable descending score order.
 *   max_segment   : an upper bound on the segment lengths (e.g. top_k); <= 0 means "unknown" (= stride).
 *                   It sizes the on-chip buffers; an image holding a longer segment gets keep_counts -1.
 *   presorted != 0: every segment is already in descending score order (up to 262144 boxes each).
 *   presorted == 0: segments are sorted here; a segment longer than DETBOX_NMS_MAX_SORT makes
 *                   keep_counts[b] = -1 for its image. */
size_t detbox_batched_nms_segments_workspace_bytes(int64_t B, int64_t stride, int64_t S);
int detbox_batched_nms_segments(const float *boxes, const float *scores, const int32_t *seg_offsets,
                                int64_t B, int64_t stride, int64_t S, int64_t max_segment,
                                double iou_threshold, int presorted,
                                int64_t *keep, int32_t *keep_counts,
                                void *workspace, size_t workspace_bytes, void *stream);

/* Replaces the per-detection Python loop of src/utilities/predictor.py:56-73: gathers the kept rows
 * of every image and writes result rows, images in order, detections in keep order:
 *   rows_ids  [N,2] = image_id, category_id;  rows_vals [N,5] (float64) =
 *   x1*W, y1*H, (x2-x1)*W, (y2-y1)*H, score   (fp32 differences, float64 products, as in Python)
 * N = sum keep_counts; row_offsets [B+1] receives the per-image row ranges.  image_sizes [B,2] = H, W. */
int detbox_pack_detections(const float *boxes, const float *scores, const int64_t *class_ids,
                           const int64_t *keep, const int32_t *keep_counts,
                           const int64_t *image_ids, const int32_t *image_sizes,
                           int64_t B, int64_t stride,
                           int64_t *rows_ids, double *rows_vals, int32_t *row_offsets, void *stream);

/* detbox_batched_nms_segments and detbox_pack_detections (and, with wire != NULL, detbox_pack_wire) as one call:
 * replaces src/utilities/predictor.py:52-73 for a batch -- batched_nms at :55, the gathers at :56-58 and the
 * per-detection loop at :60-73.  The order step of the NMS writes the result rows itself, so the keep lists are
 * not read back and the kept boxes are gathered once.
 *   class_ids     [B*stride]; every row of a segment carries the same class id (a segment IS a class)
 *   keep          [B*stride] (required), keep_counts [B]: as detbox_batched_nms_segments
 *   rows_ids, rows_vals, row_offsets: as detbox_pack_detections; an image whose keep_counts is -1 gets no rows
 *   images_cap, wire: as detbox_pack_wire, or wire == NULL for no message
 * Workspace: detbox_nms_rows_workspace_bytes (equal to detbox_batched_nms_segments_workspace_bytes). */
size_t detbox_nms_rows_workspace_bytes(int64_t B, int64_t stride, int64_t S);
int detbox_nms_rows(const float *boxes, const float *scores, const int64_t *class_ids,
                    const int32_t *seg_offsets, int64_t B, int64_t stride, int64_t S, int64_t max_segment,
                    double iou_threshold, int presorted,
                    const int64_t *image_ids, const int32_t *image_sizes,
                    int64_t *keep, int32_t *keep_counts,
                    int64_t *rows_ids, double *rows_vals, int32_t *row_offsets,
                    int64_t images_cap, uint32_t *wire,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---- multi-GPU exchange of detections ---------------------------------------------------- */

/* Compact, self-describing message with one rank's kept detections, for the all-gather that precedes
 * evaluation (the reference is single-process: src/test.py:98-99 appends Predictor.run's rows; here every
 * rank ships its images' rows).  No counterpart in the reference.
 * wire: 32-bit words, 6 per row; detbox_wire_header_rows(images_cap) header rows, then up to B*stride
 * detection rows (class id, x1, y1, x2, y2, score as fp32 bits), images in order, detections in keep
 * order.  Header words: [0] rows, [1] B, [2..2+B] row offsets per image, then B image ids (low, high
 * word), then B image sizes (H, W).  images_cap >= B must be the same on every rank (it fixes where the
 * rows start).  The rows of detbox_pack_detections follow from these words bit for bit:
 * x1*W, y1*H, (x2-x1)*W, (y2-y1)*H (fp32 differences, float64 products), score. */
int64_t detbox_wire_header_rows(int64_t images_cap);
int detbox_pack_wire(const float *boxes, const float *scores, const int64_t *class_ids,
                     const int64_t *keep, const int32_t *keep_counts,
                     const int64_t *image_ids, const int32_t *image_sizes,
                     int64_t B, int64_t stride, int64_t images_cap, uint32_t *wire, void *stream);
/* detbox_pack_detections and detbox_pack_wire in one pass over the kept rows (one gather feeds the result
 * rows and the message).  No counterpart in the reference. */
int detbox_pack_detections_wire(const float *boxes, const float *scores, const int64_t *class_ids,
                                const int64_t *keep, const int32_t *keep_counts,
                                const int64_t *image_ids, const int32_t *image_sizes,
                                int64_t B, int64_t stride,
                                int64_t *rows_ids, double *rows_vals, int32_t *row_offsets,
                                int64_t images_cap, uint32_t *wire, void *stream);

/* ---- per-kernel timing (HIP events on the launch stream) --------------------------------- */

/* While enabled, every kernel launch is bracketed by hipEvents recorded on its stream.
 * detbox_profile_read synchronises those events and returns the accumulated time and launch
 * count of the kernel called `name`; reset drops all samples.  Kernel names: "box_convert",
 * "box_iou", "assign_anchor", "assign_fixup", "assign_fcos", "activate_scores", "select_sample",
 * "select_plan", "select_scan", "select_check", "score_hist", "select_kth",
 * "select_refine", "select_emit", "select_finalize", "select_direct_stats", "select_direct", "nms_segments", "nms_scan", "nms_merge",
 * "nms_split", "nms_tile", "pack_detections", "pack_wire", "topk_mask", "focal_forward", "focal_backward",
 * "smooth_l1_forward", "smooth_l1_backward". */
int detbox_profile_enable(int on);
/* Restricts the timing to the one kernel called `name` (NULL or "" = every kernel again), so that a
 * timed run pays the event overhead for that kernel only. */
int detbox_profile_only(const char *name);
int detbox_profile_reset(void);
int detbox_profile_read(const char *name, double *total_ms, int64_t *launches);

/* Measurement aid, not part of the box-ops path: reads in[0..n) once with 16-byte loads from `blocks`
 * workgroups (kernel name "stream_read").  Its achieved GB/s is the read-bandwidth yardstick the
 * streaming kernels are compared with. */
int detbox_debug_stream_read(const float *in, int64_t n, float *out, int blocks, void *stream);

/* Measurement aid, not part of the box-ops path: counters of the pair-list NMS kernel summed over all launches
 * on the current device since the last reset, collected while switched on (detbox_debug_nms_stats_enable, or the
 * environment variable DETBOX_NMS_STATS at load time; a few atomics per wave).  Synchronises the device. out8[0] segments, [1] segments redone by
 * the chunked algorithm because a pair queue overflowed, [2] pairs that passed the centre-distance filter,
 * [3] pairs with iou above the threshold, [4] queue drains; [5..7] clock cycles of each segment's first wave in the
 * three phases (sort and records; all pairs; resolve and output), summed over segments. */
int detbox_debug_nms_stats_enable(int on);
/* Measurement and test aid: on != 0 makes detbox_batched_nms_segments / detbox_nms_rows order the kept boxes with the
 * level-by-level merge (and pack the rows in a pass of their own) for every shape, as they do for shapes the tile merge
 * does not hold (more than 256 segments per image).  Also set by the environment variable DETBOX_NMS_LEVEL_MERGE at
 * load time.  The workspace sizes follow the switch: query them again after changing it. */
int detbox_debug_nms_level_merge(int on);
int detbox_debug_nms_stats(uint64_t *out8, int reset);

/* Measurement aid, not part of the box-ops path: counters of the order step (nms_split / nms_tile kernels), summed over
 * all launches on the current device since the last reset while switched on (enable != 0 switches the collection on for
 * the launches that follow, 0 off).  Synchronises the device.  out8[0] tile workgroups, [1] of them ordered by the
 * merge-path levels instead of the distribution sort, [2..5] clock cycles of a tile workgroup's first wave in its four
 * phases (set-up and piece bounds; piece loads; sort; output), [6] split workgroups that sampled, [7] their cycles. */
int detbox_debug_merge_stats(uint64_t *out8, int enable, int reset);

#ifdef __cplusplus
}
#endif
#endif /* DETBOX_H */
