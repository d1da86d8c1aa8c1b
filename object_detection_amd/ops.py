"""Operator layer over the detbox C ABI.

Level 1 -- torchvision.ops-compatible signatures, the three symbols the reference
imports (src/models/ssd300.py:5, src/utilities/predictor.py:3, src/datasets/dataset.py:2):
    box_convert(boxes, in_fmt, out_fmt), box_iou(boxes1, boxes2),
    batched_nms(boxes, scores, idxs, iou_threshold)
Level 1b -- the fused batch operators the model classes in models.py are built on:
    assign_anchor, assign_fcos, select_decode, nms_candidates, pack_detections, nms_rows

Every operator runs on the GPU through libdetbox.so.  CPU tensors are accepted for
the Level-1 functions (they are copied to the current GPU and the result is copied
back, because the reference calls them on CPU tensors); nothing is ever computed on
the host, and a missing library or GPU raises.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

FMT = {'xyxy': 0, 'xywh': 1, 'cxcywh': 2}
ASSIGN_CXCYWH, ASSIGN_YOLO = 0, 1
DECODE_CXCYWH, DECODE_YOLO, DECODE_FCOS = 0, 1, 2
ACT_NONE, ACT_SOFTMAX, ACT_SOFTMAX_SIGMOID, ACT_SIGMOID_SIGMOID = 0, 1, 2, 3
MAX_TOP_K = 4096
NMS_MAX_SORT = 4096

_workspaces = {}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.DetboxError('no GPU visible: the detbox operators only run on a ROCm device')
    return torch.device('cuda', torch.cuda.current_device())


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# DETBOX_POISON_WORKSPACE=1 (tests): every call finds its workspace filled with garbage
_POISON = bool(os.environ.get('DETBOX_POISON_WORKSPACE'))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for a launch; nothing to do (and nothing spent) when the device is already current."""
    return _NO_GUARD if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _to(t: torch.Tensor, device) -> torch.Tensor:
    return t if t.device == device else t.to(device)


_ws_sizes = {}


def _ws_bytes(device, fn, *args) -> int:
    """Workspace size of an operator for one set of sizes on one device, asked from the library once (the library may
    size a workspace by the device's CU count, so the answer of one device is not reused for another)."""
    key = (device.index, fn.__name__, args)
    n = _ws_sizes.get(key)
    if n is None:
        if len(_ws_sizes) > 4096:
            _ws_sizes.clear()
        n = _ws_sizes[key] = int(fn(*args))
    return n


def _workspace(device, nbytes: int, tag: str, stream: Optional[int] = None) -> torch.Tensor:
    """Grow-only scratch buffer per (device, operator, stream): calls that share one are ordered on that stream."""
    key = (device.index, tag, _stream(device) if stream is None else stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    if _POISON:
        ws.fill_(0xA5)       # the operators must not rely on what a workspace holds on entry
    return ws


def release_workspaces(device=None) -> int:
    """Drops the cached scratch buffers (all of them, or those of one device) and returns how many bytes they held.
    The buffers are grow-only per (device, operator, stream): after a sweep over very large batches this hands the memory
    (staging lists and kept runs that grow with the batch) back to the caching allocator.  Only call it when no operator is in flight on the streams concerned.  A captured HIP graph
    holds the ADDRESSES of the buffers its launches were given: releasing (or outgrowing) a buffer a graph was captured
    with leaves that graph pointing at freed memory, so destroy such graphs first and capture them again afterwards."""
    index = None if device is None else torch.device(device).index
    freed = 0
    for key in [k for k in _workspaces if index is None or k[0] == index]:
        freed += _workspaces.pop(key).numel()
    return freed


def _dense(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[torch.Tensor]:
    """The dtype, dense layout and 16-byte alignment the kernels assume; a no-op for a freshly allocated
    tensor of the right type, a copy for a strided view or a slice that starts off a 16-byte boundary."""
    if t is None:
        return None
    if t.dtype is dtype and t.is_contiguous() and not (t.data_ptr() & 15):
        return t
    if t.dtype is not dtype:
        t = t.to(dtype)
    if not t.is_contiguous():
        t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return _dense(t.to(device=device))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()


# ------------------------------------------------------------------ Level 1

def box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    if in_fmt not in FMT or out_fmt not in FMT:
        raise ValueError('Unsupported Bounding Box Conversions for given in_fmt and out_fmt')
    src_device = boxes.device
    dev = boxes.device if boxes.is_cuda else _device()
    b = _f32(boxes, dev).reshape(-1, 4)
    out = torch.empty_like(b)
    with _on(dev):
        _lib.check(_lib.lib().detbox_box_convert(_ptr(b), _ptr(out), b.shape[0], FMT[in_fmt], FMT[out_fmt], _stream(dev)),
                   'detbox_box_convert')
    return out.reshape(boxes.shape).to(src_device)


def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    src_device = boxes1.device
    dev = boxes1.device if boxes1.is_cuda else _device()
    b1, b2 = _f32(boxes1, dev), _f32(boxes2, dev)
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().detbox_box_iou(_ptr(b1), b1.shape[0], _ptr(b2), b2.shape[0], _ptr(out), _stream(dev)),
                   'detbox_box_iou')
    return out.to(src_device)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.batched_nms: NMS inside every class, kept indices in descending score order
    (ties: lower index first)."""
    src_device = boxes.device
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=src_device)
    dev = boxes.device if boxes.is_cuda else _device()
    with _on(dev):
        b, s = _f32(boxes, dev), _f32(scores, dev)
        cls = idxs.to(dev)
        # group by class (stable: original order inside a class); long classes are pre-sorted by score
        order = torch.argsort(cls, stable=True)
        _, cnt = torch.unique_consecutive(cls[order], return_counts=True)
        presorted = int(cnt.max().item()) > NMS_MAX_SORT
        if presorted:
            by_score = torch.argsort(s, descending=True, stable=True)
            order = by_score[torch.argsort(cls[by_score], stable=True)]
        S = cnt.numel()
        seg = torch.zeros((1, S + 1), dtype=torch.int32, device=dev)
        seg[0, 1:] = torch.cumsum(cnt, 0).to(torch.int32)
        keep, kc = nms_segments(b[order].unsqueeze(0), s[order].unsqueeze(0), seg, iou_threshold,
                                max_segment=int(cnt.max().item()), presorted=presorted)
        k = int(kc[0].item())
        if k < 0:
            raise _lib.DetboxError('detbox_batched_nms_segments reported an over-long segment')
        kept = order[keep[0, :k]]
        # score ties between different classes: torchvision orders them by original index
        kept = kept.sort().values
        kept = kept[torch.argsort(s[kept], descending=True, stable=True)]
        return kept.to(src_device)


# ------------------------------------------------------------------ fused batch operators

def pack_gt(gt_bboxes: Sequence[torch.Tensor], gt_labels: Sequence[torch.Tensor], device):
    """List of per-image [G_i,4] / [G_i] tensors (the reference's collate contract,
    src/datasets/dataset.py:88-100) -> packed [sum G,4] f32, [sum G] i64, offsets [B+1] i32 on `device`
    with one host-to-device copy each."""
    B = len(gt_bboxes)
    counts = [int(b.shape[0]) for b in gt_bboxes]
    off = torch.zeros(B + 1, dtype=torch.int32)
    if B:
        off[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0).to(torch.int32)
    total = int(off[-1])
    if total:
        boxes = torch.cat([b.reshape(-1, 4).to(torch.float32) for b in gt_bboxes]).contiguous()
        labels = torch.cat([l.reshape(-1).to(torch.int64) for l in gt_labels]).contiguous()
    else:
        boxes, labels = torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64)
    return boxes.to(device, non_blocking=True), labels.to(device, non_blocking=True), off.to(device, non_blocking=True), total


def assign_anchor(anchors: torch.Tensor, gt_boxes: torch.Tensor, gt_labels: torch.Tensor, gt_offsets: torch.Tensor,
                  total_gt: int, mode: int, thr_lo: float, thr_hi: float, want_best: bool = False):
    """-> target_locs [B,P,4] f32, target_labels [B,P] i64, matched [B,P] i64 (, best_anchor [sum G] i64)."""
    dev = anchors.device
    anchors, gt_boxes = _dense(anchors), _dense(gt_boxes)
    gt_labels, gt_offsets = _dense(gt_labels, torch.int64), _dense(gt_offsets, torch.int32)
    B, P = gt_offsets.numel() - 1, anchors.shape[0]
    if gt_boxes.shape[0] < total_gt or gt_labels.numel() < total_gt:
        raise ValueError('assign_anchor: fewer ground-truth rows than total_gt')
    locs = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
    labels = torch.empty((B, P), dtype=torch.int64, device=dev)
    matched = torch.empty((B, P), dtype=torch.int64, device=dev)
    best = torch.empty((max(total_gt, 1),), dtype=torch.int64, device=dev) if want_best else None
    L = _lib.lib()
    with _on(dev):
        st = _stream(dev)
        ws = _workspace(dev, _ws_bytes(dev, L.detbox_assign_anchor_workspace_bytes_for, total_gt, P, B), 'assign', st)
        _lib.check(L.detbox_assign_anchor(mode, _ptr(anchors), P, _ptr(gt_boxes), _ptr(gt_labels), _ptr(gt_offsets),
                                          B, total_gt, thr_lo, thr_hi, _ptr(locs), _ptr(labels), _ptr(matched),
                                          _ptr(best), ws.data_ptr(), ws.numel(), st), 'detbox_assign_anchor')
    if want_best:
        return locs, labels, matched, best[:total_gt]
    return locs, labels, matched


def assign_fcos(points: torch.Tensor, ranges: torch.Tensor, gt_boxes, gt_labels, gt_offsets, total_gt: int):
    """-> target_locs [B,P,4] (l,r,t,b), target_cents [B,P], target_labels [B,P], matched [B,P]."""
    dev = points.device
    points, ranges, gt_boxes = _dense(points), _dense(ranges), _dense(gt_boxes)
    gt_labels, gt_offsets = _dense(gt_labels, torch.int64), _dense(gt_offsets, torch.int32)
    B, P = gt_offsets.numel() - 1, points.shape[0]
    if gt_boxes.shape[0] < total_gt or gt_labels.numel() < total_gt or ranges.shape[0] != P:
        raise ValueError('assign_fcos: table or ground-truth sizes do not fit')
    locs = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
    cents = torch.empty((B, P), dtype=torch.float32, device=dev)
    labels = torch.empty((B, P), dtype=torch.int64, device=dev)
    matched = torch.empty((B, P), dtype=torch.int64, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().detbox_assign_fcos(_ptr(points), _ptr(ranges), P, _ptr(gt_boxes), _ptr(gt_labels),
                                                 _ptr(gt_offsets), B, total_gt, _ptr(locs), _ptr(cents), _ptr(labels),
                                                 _ptr(matched), _stream(dev)), 'detbox_assign_fcos')
    return locs, cents, labels, matched


@dataclass
class Candidates:
    """Output of select_decode: image b owns rows [b, :counts[b]] of the [B, stride] arrays, grouped by
    class (seg_offsets[b, s] .. seg_offsets[b, s+1]) and ordered by anchor index inside a class."""
    boxes: torch.Tensor        # [B, stride, 4] xyxy
    scores: torch.Tensor       # [B, stride]
    class_ids: torch.Tensor    # [B, stride] int64
    seg_offsets: torch.Tensor  # [B, S+1] int32
    counts: torch.Tensor       # [B] int32
    top_k: int


def activate_scores(head: torch.Tensor, extra: Optional[torch.Tensor], activation: int) -> torch.Tensor:
    """The activated scores [B,P,C] exactly as select_decode computes them internally."""
    dev = head.device
    head, extra = _dense(head), _dense(extra)
    B, P, C = head.shape
    out = torch.empty_like(head)
    with _on(dev):
        _lib.check(_lib.lib().detbox_activate_scores(activation, _ptr(head), _ptr(extra), B, P, C, _ptr(out), _stream(dev)),
                   'detbox_activate_scores')
    return out


def select_decode(scores: torch.Tensor, locs: torch.Tensor, anchors: torch.Tensor, decode_mode: int,
                  class_begin: int, class_id_offset: int, conf_thresh: float, top_k: int,
                  score_hi_hint: float = 1.0, activation: int = ACT_NONE,
                  extra: Optional[torch.Tensor] = None) -> Candidates:
    """scores: [B,P,C] activated scores (activation == ACT_NONE) or raw head output to be activated in
    the kernel; extra: [B,P] centerness / objectness logits for the activations that use them."""
    dev = scores.device
    if not scores.is_cuda:
        raise _lib.DetboxError('select_decode runs on a ROCm device only')
    scores, locs, anchors, extra = _dense(scores), _dense(_to(locs, dev)), _dense(_to(anchors, dev)), _dense(None if extra is None else _to(extra, dev))
    B, P, C = scores.shape
    if locs.shape != (B, P, 4) or anchors.shape[0] != P or (extra is not None and extra.shape != (B, P)):
        raise ValueError(f'select_decode: shapes {tuple(scores.shape)}, {tuple(locs.shape)}, {tuple(anchors.shape)} do not fit')
    S = C - class_begin
    stride = S * top_k
    out_boxes = torch.empty((B, stride, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, stride), dtype=torch.float32, device=dev)
    out_cls = torch.empty((B, stride), dtype=torch.int64, device=dev)
    seg = torch.empty((B, S + 1), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    with _on(dev):
        st = _stream(dev)
        ws = _workspace(dev, _ws_bytes(dev, L.detbox_select_decode_workspace_bytes, B, P, C, class_begin, top_k, activation), 'select', st)
        _lib.check(L.detbox_select_decode(decode_mode, activation, _ptr(scores), _ptr(extra), _ptr(locs), _ptr(anchors),
                                          B, P, C, class_begin, class_id_offset, conf_thresh, top_k, score_hi_hint,
                                          out_boxes.data_ptr(), out_scores.data_ptr(), out_cls.data_ptr(),
                                          seg.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), st),
                   'detbox_select_decode')
    return Candidates(out_boxes, out_scores, out_cls, seg, counts, top_k)


def nms_segments(boxes: torch.Tensor, scores: torch.Tensor, seg_offsets: torch.Tensor, iou_threshold: float,
                 max_segment: int = 0, presorted: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """boxes [B,stride,4], scores [B,stride], seg_offsets [B,S+1] -> keep [B,stride] i64, keep_counts [B] i32."""
    dev = boxes.device
    boxes, scores, seg_offsets = _dense(boxes), _dense(scores), _dense(seg_offsets, torch.int32)
    B, stride = scores.shape
    if boxes.shape != (B, stride, 4) or seg_offsets.shape[0] != B:
        raise ValueError(f'nms_segments: shapes {tuple(boxes.shape)}, {tuple(scores.shape)}, {tuple(seg_offsets.shape)} do not fit')
    S = seg_offsets.shape[1] - 1
    keep = torch.empty((B, stride), dtype=torch.int64, device=dev)
    kc = torch.empty((B,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    with _on(dev):
        st = _stream(dev)
        ws = _workspace(dev, _ws_bytes(dev, L.detbox_batched_nms_segments_workspace_bytes, B, stride, S), 'nms', st)
        _lib.check(L.detbox_batched_nms_segments(_ptr(boxes), _ptr(scores), _ptr(seg_offsets), B, stride, S,
                                                 int(max_segment), float(iou_threshold), int(presorted),
                                                 _ptr(keep), kc.data_ptr(), ws.data_ptr(), ws.numel(), st),
                   'detbox_batched_nms_segments')
    return keep, kc


def nms_candidates(c: Candidates, iou_threshold: float) -> Tuple[torch.Tensor, torch.Tensor]:
    return nms_segments(c.boxes, c.scores, c.seg_offsets, iou_threshold, max_segment=c.top_k, presorted=False)


def nms_rows(c: Candidates, iou_threshold: float, image_ids: torch.Tensor, image_sizes: torch.Tensor,
             wire_images_cap: Optional[int] = None, presorted: bool = False, max_segment: Optional[int] = None):
    """nms_candidates + pack_detections as one operator (src/utilities/predictor.py:52-73 for a batch): the order step
    of the NMS writes the result rows itself.
    -> keep [B,stride] i64, keep_counts [B] i32, rows_ids [B*stride,2] i64, rows_vals [B*stride,5] f64,
       row_offsets [B+1] i32 (, wire [header_rows + B*stride, 6] i32 with wire_images_cap)."""
    dev = c.boxes.device
    boxes, scores, cls = _dense(c.boxes), _dense(c.scores), _dense(c.class_ids, torch.int64)
    seg = _dense(c.seg_offsets, torch.int32)
    image_sizes = _dense(_to(image_sizes, dev), torch.int32)
    image_ids = _dense(_to(image_ids, dev), torch.int64)
    B, stride = scores.shape
    S = seg.shape[1] - 1
    if boxes.shape != (B, stride, 4) or cls.shape != (B, stride) or seg.shape[0] != B or image_ids.numel() != B \
            or image_sizes.shape != (B, 2):
        raise ValueError('nms_rows: shapes do not fit')
    if wire_images_cap is not None and B > wire_images_cap:
        raise ValueError('nms_rows: more images than wire_images_cap')
    keep = torch.empty((B, stride), dtype=torch.int64, device=dev)
    kc = torch.empty((B,), dtype=torch.int32, device=dev)
    rows_ids = torch.empty((B * stride, 2), dtype=torch.int64, device=dev)
    rows_vals = torch.empty((B * stride, 5), dtype=torch.float64, device=dev)
    row_off = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    wire = None
    if wire_images_cap is not None:
        wire = torch.empty((wire_header_rows(wire_images_cap) + B * stride, 6), dtype=torch.int32, device=dev)
    L = _lib.lib()
    with _on(dev):
        st = _stream(dev)
        ws = _workspace(dev, _ws_bytes(dev, L.detbox_nms_rows_workspace_bytes, B, stride, S), 'nms', st)
        _lib.check(L.detbox_nms_rows(_ptr(boxes), _ptr(scores), _ptr(cls), _ptr(seg), B, stride, S,
                                     int(c.top_k if max_segment is None else max_segment), float(iou_threshold), int(presorted),
                                     image_ids.data_ptr(), image_sizes.data_ptr(), _ptr(keep), kc.data_ptr(),
                                     _ptr(rows_ids), _ptr(rows_vals), row_off.data_ptr(),
                                     int(wire_images_cap or 0), None if wire is None else wire.data_ptr(),
                                     ws.data_ptr(), ws.numel(), st), 'detbox_nms_rows')
    if wire is None:
        return keep, kc, rows_ids, rows_vals, row_off
    return keep, kc, rows_ids, rows_vals, row_off, wire


def nms_level_merge(on: bool):
    """Test / measurement switch: order the kept boxes with the level-by-level merge for every shape."""
    _lib.check(_lib.lib().detbox_debug_nms_level_merge(int(on)), 'detbox_debug_nms_level_merge')
    _ws_sizes.clear()


def pack_detections(boxes, scores, class_ids, keep, keep_counts, image_ids: torch.Tensor, image_sizes: torch.Tensor,
                    wire_images_cap: Optional[int] = None):
    """-> rows_ids [B*stride,2] i64, rows_vals [B*stride,5] f64, row_offsets [B+1] i32 (rows beyond
    row_offsets[B] are unused).  With wire_images_cap the same pass also writes the message of pack_wire,
    returned as a fourth value."""
    dev = boxes.device
    boxes, scores, class_ids, keep = _dense(boxes), _dense(scores), _dense(class_ids, torch.int64), _dense(keep, torch.int64)
    keep_counts, image_sizes = _dense(keep_counts, torch.int32), _dense(_to(image_sizes, dev), torch.int32)
    image_ids = _dense(_to(image_ids, dev), torch.int64)
    B, stride = scores.shape
    if boxes.shape != (B, stride, 4) or keep.shape != (B, stride) or image_ids.numel() != B or image_sizes.shape != (B, 2):
        raise ValueError('pack_detections: shapes do not fit')
    rows_ids = torch.empty((B * stride, 2), dtype=torch.int64, device=dev)
    rows_vals = torch.empty((B * stride, 5), dtype=torch.float64, device=dev)
    row_off = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    with _on(dev):
        if wire_images_cap is None:
            _lib.check(_lib.lib().detbox_pack_detections(_ptr(boxes), _ptr(scores), _ptr(class_ids), _ptr(keep),
                                                         keep_counts.data_ptr(), image_ids.data_ptr(), image_sizes.data_ptr(),
                                                         B, stride, rows_ids.data_ptr(), rows_vals.data_ptr(),
                                                         row_off.data_ptr(), _stream(dev)), 'detbox_pack_detections')
            return rows_ids, rows_vals, row_off
        if B > wire_images_cap:
            raise ValueError('pack_detections: more images than wire_images_cap')
        wire = torch.empty((wire_header_rows(wire_images_cap) + B * stride, 6), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().detbox_pack_detections_wire(_ptr(boxes), _ptr(scores), _ptr(class_ids), _ptr(keep),
                                                          keep_counts.data_ptr(), image_ids.data_ptr(), image_sizes.data_ptr(),
                                                          B, stride, rows_ids.data_ptr(), rows_vals.data_ptr(),
                                                          row_off.data_ptr(), int(wire_images_cap), wire.data_ptr(),
                                                          _stream(dev)), 'detbox_pack_detections_wire')
    return rows_ids, rows_vals, row_off, wire


class _FocalLossSum(torch.autograd.Function):
    """sum over rows with label >= 0 of the focal loss (src/models/losses.py:5-9), fused: no masked gather,
    no [rows, C] temporaries; forward and backward each read the logits once."""

    @staticmethod
    def forward(ctx, logits, labels, alpha, gamma):
        dev = logits.device
        C = logits.shape[-1]
        x = _dense(logits.detach().reshape(-1, C))
        t = labels.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty((), dtype=torch.float64, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().detbox_focal_loss_forward(_ptr(x), _ptr(t), x.shape[0], C, float(alpha), float(gamma),
                                                            out.data_ptr(), _stream(dev)), 'detbox_focal_loss_forward')
        ctx.save_for_backward(x, t)
        ctx.alpha, ctx.gamma, ctx.shape, ctx.dtype = float(alpha), float(gamma), logits.shape, logits.dtype
        return out.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        x, t = ctx.saved_tensors
        dev = x.device
        gs = grad_out.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(x)
        with _on(dev):
            _lib.check(_lib.lib().detbox_focal_loss_backward(_ptr(x), _ptr(t), x.shape[0], x.shape[1], ctx.alpha, ctx.gamma,
                                                             gs.data_ptr(), _ptr(grad), _stream(dev)),
                       'detbox_focal_loss_backward')
        return grad.reshape(ctx.shape).to(ctx.dtype), None, None, None


def focal_loss_sum(logits: torch.Tensor, labels: torch.Tensor, alpha: float = 0.25, gamma: float = 2.0) -> torch.Tensor:
    """focal_loss(logits[labels >= 0], labels[labels >= 0], reduction='sum') of the reference
    (src/models/retinanet.py:215, src/models/fcos.py:271), differentiable in `logits`.
    logits [..., C] on the GPU, labels [...] (negative = ignored)."""
    if not logits.is_cuda:
        raise _lib.DetboxError('focal_loss_sum runs on a ROCm device only')
    return _FocalLossSum.apply(logits, labels, alpha, gamma)


class _SmoothL1MaskedSum(torch.autograd.Function):
    """sum over rows with label > 0 of smooth_l1(input, target) (beta = 1), fused with the mask: no nonzero(),
    no [N_pos, 4] gathers, no host synchronisation."""

    @staticmethod
    def forward(ctx, input, target, labels):
        dev = input.device
        x = _dense(input.detach().reshape(-1, 4))
        y = _dense(target.detach().to(dev).reshape(-1, 4))
        t = _dense(labels.to(dev).reshape(-1), torch.int64)
        if y.shape != x.shape or t.numel() != x.shape[0]:
            raise ValueError('smooth_l1_masked_sum: input, target [..., 4] and labels [...] do not fit')
        out = torch.empty((), dtype=torch.float64, device=dev)
        with _on(dev):
            _lib.check(_lib.lib().detbox_smooth_l1_masked_forward(_ptr(x), _ptr(y), _ptr(t), x.shape[0], out.data_ptr(),
                                                                  _stream(dev)), 'detbox_smooth_l1_masked_forward')
        ctx.save_for_backward(x, y, t)
        ctx.shape, ctx.dtype = input.shape, input.dtype
        return out.to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        x, y, t = ctx.saved_tensors
        dev = x.device
        gs = grad_out.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(x)
        with _on(dev):
            _lib.check(_lib.lib().detbox_smooth_l1_masked_backward(_ptr(x), _ptr(y), _ptr(t), x.shape[0], gs.data_ptr(),
                                                                   _ptr(grad), _stream(dev)), 'detbox_smooth_l1_masked_backward')
        return grad.reshape(ctx.shape).to(ctx.dtype), None, None


def smooth_l1_masked_sum(input: torch.Tensor, target: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """F.smooth_l1_loss(input[labels > 0], target[labels > 0], reduction='sum') of the reference
    (src/models/ssd300.py:283, src/models/retinanet.py:211), differentiable in `input`.
    input, target [..., 4] on the GPU, labels [...]."""
    if not input.is_cuda:
        raise _lib.DetboxError('smooth_l1_masked_sum runs on a ROCm device only')
    return _SmoothL1MaskedSum.apply(input, target, labels)


def wire_header_rows(images_cap: int) -> int:
    return int(_lib.lib().detbox_wire_header_rows(int(images_cap)))


def pack_wire(boxes, scores, class_ids, keep, keep_counts, image_ids: torch.Tensor, image_sizes: torch.Tensor,
              images_cap: int) -> torch.Tensor:
    """-> wire [header_rows + B*stride, 6] int32: the kept detections of this rank as a compact message
    (see detbox_pack_wire); `dist.unpack_wire` turns gathered messages back into result rows."""
    dev = boxes.device
    boxes, scores, class_ids, keep = _dense(boxes), _dense(scores), _dense(class_ids, torch.int64), _dense(keep, torch.int64)
    keep_counts, image_sizes = _dense(keep_counts, torch.int32), _dense(image_sizes.to(dev), torch.int32)
    image_ids = _dense(image_ids.to(dev), torch.int64)
    B, stride = scores.shape
    if boxes.shape != (B, stride, 4) or keep.shape != (B, stride) or image_ids.numel() != B or image_sizes.shape != (B, 2):
        raise ValueError('pack_wire: shapes do not fit')
    if B > images_cap:
        raise ValueError('pack_wire: more images than images_cap')
    wire = torch.empty((wire_header_rows(images_cap) + B * stride, 6), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().detbox_pack_wire(_ptr(boxes), _ptr(scores), _ptr(class_ids), _ptr(keep),
                                               keep_counts.data_ptr(), image_ids.data_ptr(), image_sizes.data_ptr(),
                                               B, stride, int(images_cap), wire.data_ptr(), _stream(dev)),
                   'detbox_pack_wire')
    return wire


# ------------------------------------------------------------------ per-kernel timing

def topk_mask(values: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """mask[b, p] = descending rank of values[b, p] in row b < k[b], ties by ascending index: the
    `values.argsort(descending=True).argsort() < k[:, None]` of the reference's hard-negative mining
    (src/models/ssd300.py:277-278) without the two sorts.  values [B, P] float, k [B] integer -> bool [B, P]."""
    dev = values.device if values.is_cuda else _device()
    if values.dim() != 2:
        raise ValueError('values must be [B, P]')
    B, P = values.shape
    v = _f32(values.detach(), dev)
    kk = k.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    if kk.numel() != B:
        raise ValueError('k must hold one count per row')
    out = torch.empty((B, P), dtype=torch.bool, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().detbox_topk_mask(_ptr(v), _ptr(kk), B, P, _ptr(out), _stream(dev)), 'detbox_topk_mask')
    return out if values.is_cuda else out.to(values.device)


def profile_enable(on: bool = True):
    _lib.check(_lib.lib().detbox_profile_enable(int(on)), 'detbox_profile_enable')


def profile_only(name: str = ''):
    """Time only the kernel called `name` ('' = all kernels)."""
    _lib.check(_lib.lib().detbox_profile_only(name.encode()), 'detbox_profile_only')


def profile_reset():
    _lib.check(_lib.lib().detbox_profile_reset(), 'detbox_profile_reset')


def profile_read(name: str) -> Tuple[float, int]:
    import ctypes
    ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    _lib.check(_lib.lib().detbox_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n)), 'detbox_profile_read')
    return ms.value, n.value
