"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the detection box-ops path.

A CPU restatement (PyTorch CPU tensors, fp32) of the box-ops algorithms of
rs1004/object_detection, written to follow the reference's own loop structure
so that it can serve both as the parity checker for the HIP kernels and as the
"reference CPU path" timing baseline.  Every function cites the reference
file:line it follows (paths relative to the reference root).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the shipped package (object_detection_amd/) never does.

Pinning: tests/golden/generate_golden.py imports the reference's model files
(with this module's box_iou / box_convert / batched_nms standing in for the
missing torchvision) and records the reference's own outputs; the tests check
this oracle against those fixtures.  Two behaviours have no runnable reference
and are therefore "parity unpinned":
  * the three torchvision primitives themselves (torchvision is not installed
    and not vendored; they follow its documented/published semantics), and
  * YoloV3.pre_predict decode, which raises TypeError in the reference
    (src/models/yolov3.py:320); the evident intent (yolov3.py:259-280) is
    implemented.

Deliberate, documented resolutions of unspecified reference behaviour:
  * top-k boundary ties: the reference ranks with a non-stable
    argsort(descending=True) (e.g. src/models/ssd300.py:370) whose tie order is
    unspecified; this oracle (and the HIP path) define "lower anchor index
    wins" (= stable descending sort).
  * images with zero ground-truth boxes crash the reference (IndexError from
    max() over an empty dim); here they yield all-background targets with
    matched index -1 and zero offsets.
"""
from __future__ import annotations

import ctypes
import math
import os
from itertools import product
from typing import List, Sequence, Tuple

import numpy as np
import torch

INF = 1e8  # src/models/fcos.py:9

_HERE = os.path.dirname(os.path.abspath(__file__))
_NMS_LIB = None


# --------------------------------------------------------------------------
# torchvision.ops primitives (third-party; restated from published semantics)
# --------------------------------------------------------------------------

def box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    """torchvision.ops.box_convert; call sites src/models/ssd300.py:249-250,374,
    src/models/yolov3.py:168-169,321, src/models/fcos.py:215,
    src/datasets/dataset.py:59."""
    allowed = ('xyxy', 'xywh', 'cxcywh')
    if in_fmt not in allowed or out_fmt not in allowed:
        raise ValueError('Unsupported Bounding Box Conversions for given in_fmt and out_fmt')
    if in_fmt == out_fmt:
        return boxes.clone()
    if in_fmt != 'xyxy' and out_fmt != 'xyxy':
        boxes = box_convert(boxes, in_fmt, 'xyxy')
        in_fmt = 'xyxy'
    if in_fmt == 'xyxy':
        x1, y1, x2, y2 = boxes.unbind(-1)
        if out_fmt == 'xywh':
            return torch.stack((x1, y1, x2 - x1, y2 - y1), dim=-1)
        cx = (x1 + x2) / 2
        cy = (y1 + y2) / 2
        return torch.stack((cx, cy, x2 - x1, y2 - y1), dim=-1)
    if in_fmt == 'xywh':
        x, y, w, h = boxes.unbind(-1)
        return torch.stack((x, y, x + w, y + h), dim=-1)
    cx, cy, w, h = boxes.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


def box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """torchvision.ops.box_iou ([N,4] x [M,4] xyxy -> [N,M]); call sites
    src/models/ssd300.py:251, src/models/retinanet.py:182, src/models/yolov3.py:170."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union


def _nms_lib():
    global _NMS_LIB
    if _NMS_LIB is None:
        path = os.path.join(_HERE, '_build', 'liboracle_nms.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle` (or __graft_entry__.build())')
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int64
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_double, ctypes.c_void_p]
        _NMS_LIB = lib
    return _NMS_LIB


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.nms CPU kernel (greedy, stable descending order); C body in
    oracle/nms_oracle.c."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    keep = np.empty((n,), dtype=np.int64)
    k = _nms_lib().oracle_nms(b.ctypes.data, s.ctypes.data, n, float(iou_threshold), keep.ctypes.data)
    if k < 0:
        raise MemoryError('oracle_nms')
    return torch.from_numpy(keep[:k].copy())


def nms_python(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Slow pure-Python twin of nms() used to cross-check the C body on small inputs."""
    n = boxes.shape[0]
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    order = sorted(range(n), key=lambda i: -float(s[i]))  # sorted() is stable
    suppressed = [False] * n
    keep = []
    f32 = np.float32
    for a in range(n):
        i = order[a]
        if suppressed[i]:
            continue
        keep.append(i)
        for c in range(a + 1, n):
            j = order[c]
            if suppressed[j]:
                continue
            xx1 = max(b[i, 0], b[j, 0]); yy1 = max(b[i, 1], b[j, 1])
            xx2 = min(b[i, 2], b[j, 2]); yy2 = min(b[i, 3], b[j, 3])
            w = max(f32(0), f32(xx2 - xx1)); h = max(f32(0), f32(yy2 - yy1))
            inter = f32(w * h)
            with np.errstate(invalid='ignore', divide='ignore'):
                ovr = f32(inter / f32(f32(areas[i] + areas[j]) - inter))
            if float(ovr) > iou_threshold:
                suppressed[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor,
                iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.batched_nms, per-class ("vanilla") form: NMS inside every
    class on the un-offset boxes, kept indices returned in stable descending
    score order.  Call site src/utilities/predictor.py:55."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr = torch.where(idxs == class_id)[0]
        curr_keep = nms(boxes[curr], scores[curr], iou_threshold)
        keep_mask[curr[curr_keep]] = True
    keep = torch.where(keep_mask)[0]
    return keep[scores[keep].sort(descending=True, stable=True)[1]]


def batched_nms_coordinate_trick(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor,
                                 iou_threshold: float) -> torch.Tensor:
    """The other published torchvision strategy (one NMS on class-offset boxes).
    Used only by the golden generator to reject fixtures on which the two
    strategies disagree through fp32 rounding of the offset coordinates."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


# --------------------------------------------------------------------------
# anchor / point tables (rows a1-a4)
# --------------------------------------------------------------------------

def _ssd_boxes(cfg, m, s_min, s_max) -> torch.Tensor:
    def s_(k):
        return s_min + (s_max - s_min) * (k - 1) / (m - 1)
    dboxes = []
    for k, (f_k, num_aspects) in enumerate(cfg, start=1):
        aspects = [1, 'add', 2, 1 / 2] if num_aspects == 4 else [1, 'add', 2, 1 / 2, 3, 1 / 3]
        for i, j in product(range(f_k), repeat=2):
            cx = (j + 0.5) / f_k
            cy = (i + 0.5) / f_k
            for a in aspects:
                if a == 'add':
                    w = h = pow(s_(k) * s_(k + 1), 0.5)
                else:
                    w = s_(k) * pow(a, 0.5)
                    h = s_(k) * pow(1 / a, 0.5)
                dboxes.append([cx, cy, w, h])
    return torch.tensor(dboxes).clamp(min=0, max=1)


def ssd300_default_boxes() -> torch.Tensor:
    """src/models/ssd300.py:182-208 -> [8732,4] cxcywh."""
    return _ssd_boxes([[38, 4], [19, 6], [10, 6], [5, 6], [3, 4], [1, 4]], 6, 0.1, 0.88)


def ssd512_default_boxes() -> torch.Tensor:
    """src/models/ssd512.py:51-77 -> [24564,4] cxcywh."""
    return _ssd_boxes([[64, 4], [32, 6], [16, 6], [8, 6], [4, 6], [2, 4], [1, 4]], 7, 0.07, 0.88)


def retinanet_prior_boxes(f_k_list: Sequence[int]) -> torch.Tensor:
    """src/models/retinanet.py:120-139 -> [9*sum(f_k^2),4] cxcywh."""
    pboxes = []
    for f_k in f_k_list:
        for i, j in product(range(f_k), repeat=2):
            cx = (j + 0.5) / f_k
            cy = (i + 0.5) / f_k
            for aspect in [0.5, 1, 2]:
                for scale in [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]:
                    w = 4 * scale / f_k * pow(1 / aspect, 0.5)
                    h = 4 * scale / f_k * pow(aspect, 0.5)
                    pboxes.append([cx, cy, w, h])
    return torch.tensor(pboxes).clamp(min=0, max=1)


def fcos_points(f_k_list: Sequence[int], image_size: float = 512) -> torch.Tensor:
    """src/models/fcos.py:142-155 -> [sum(f_k^2),2] pixels.  The reference
    hard-codes 512 (fcos.py:152); image_size generalises it by formula."""
    points = []
    for f_k in f_k_list:
        for y, x in product(range(f_k), repeat=2):
            points.append([(x + 0.5) / f_k * image_size, (y + 0.5) / f_k * image_size])
    return torch.tensor(points)


def fcos_regress_ranges(f_k_list: Sequence[int]) -> torch.Tensor:
    """src/models/fcos.py:157-173 -> [sum(f_k^2),2]."""
    regress_range_list = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]
    rr = []
    for i, f_k in enumerate(f_k_list):
        low, upp = regress_range_list[i]
        for _ in product(range(f_k), repeat=2):
            rr.append([low, upp])
    return torch.tensor(rr)


def yolov3_prior_boxes() -> torch.Tensor:
    """src/models/yolov3.py:107-125 -> [10647,5] (x, y, w, h, grid_length)."""
    S = 416
    pboxes = []
    for f_k, aspects in [
        [13, [[116, 90], [156, 198], [373, 326]]],
        [26, [[30, 61], [62, 45], [59, 119]]],
        [52, [[10, 13], [16, 30], [33, 23]]],
    ]:
        for i, j in product(range(f_k), repeat=2):
            x = j / f_k
            y = i / f_k
            grid_length = 1 / f_k
            for w, h in aspects:
                pboxes.append([x, y, w / S, h / S, grid_length])
    return torch.tensor(pboxes)


# --------------------------------------------------------------------------
# encode / decode (rows a9, a10)
# --------------------------------------------------------------------------

def calc_delta(bboxes, dboxes, std=(0.1, 0.2)):
    """src/models/ssd300.py:299-318 == src/models/retinanet.py:227-246."""
    db_cx = (1 / std[0]) * (bboxes[:, 0] - dboxes[:, 0]) / dboxes[:, 2]
    db_cy = (1 / std[0]) * (bboxes[:, 1] - dboxes[:, 1]) / dboxes[:, 3]
    db_w = (1 / std[1]) * (bboxes[:, 2] / dboxes[:, 2]).log()
    db_h = (1 / std[1]) * (bboxes[:, 3] / dboxes[:, 3]).log()
    return torch.stack([db_cx, db_cy, db_w, db_h], dim=1)


def calc_delta_yolo(bboxes, pboxes, grid_length, std=(0.1, 0.2)):
    """src/models/yolov3.py:236-257."""
    g = grid_length.reshape(-1)
    db_x = (1 / std[0]) * (bboxes[:, 0] - pboxes[:, 0]) / g
    db_y = (1 / std[0]) * (bboxes[:, 1] - pboxes[:, 1]) / g
    db_w = (1 / std[1]) * (bboxes[:, 2] / pboxes[:, 2]).log()
    db_h = (1 / std[1]) * (bboxes[:, 3] / pboxes[:, 3]).log()
    return torch.stack([db_x, db_y, db_w, db_h], dim=1)


def calc_coord(locs, dboxes, std=(0.1, 0.2)):
    """src/models/ssd300.py:320-337 == src/models/retinanet.py:248-265 -> cxcywh."""
    b_cx = dboxes[:, 0] + std[0] * locs[:, 0] * dboxes[:, 2]
    b_cy = dboxes[:, 1] + std[0] * locs[:, 1] * dboxes[:, 3]
    b_w = dboxes[:, 2] * (std[1] * locs[:, 2]).exp()
    b_h = dboxes[:, 3] * (std[1] * locs[:, 3]).exp()
    return torch.stack([b_cx, b_cy, b_w, b_h], dim=1)


def calc_coord_yolo(locs, pboxes, grid_length, std=(0.1, 0.2)):
    """src/models/yolov3.py:259-280 -> xywh."""
    g = grid_length.reshape(-1)
    b_x = pboxes[:, 0] + std[0] * locs[:, 0] * g
    b_y = pboxes[:, 1] + std[0] * locs[:, 1] * g
    b_w = pboxes[:, 2] * (std[1] * locs[:, 2]).exp()
    b_h = pboxes[:, 3] * (std[1] * locs[:, 3]).exp()
    return torch.stack([b_x, b_y, b_w, b_h], dim=1)


def calc_coord_fcos(locs, points):
    """src/models/fcos.py:281-297 -> xyxy (ray order read as l,t,r,b here)."""
    xmin = points[:, 0] - locs[:, 0].exp()
    ymin = points[:, 1] - locs[:, 1].exp()
    xmax = points[:, 0] + locs[:, 2].exp()
    ymax = points[:, 1] + locs[:, 3].exp()
    return torch.stack([xmin, ymin, xmax, ymax], dim=1)


# --------------------------------------------------------------------------
# training-time target assignment, "Step 1" of every loss() (rows a5-a8)
# --------------------------------------------------------------------------

def _empty_targets(P):
    return (torch.zeros(P, 4), torch.zeros(P, dtype=torch.long),
            torch.full((P,), -1, dtype=torch.long))


def assign_ssd(dboxes, gt_bboxes: List[torch.Tensor], gt_labels: List[torch.Tensor],
               iou_thresh: float = 0.5):
    """src/models/ssd300.py:240-266.  Returns (target_locs [B,P,4] f32,
    target_labels [B,P] i64, matched_bbox_ids [B,P] i64)."""
    B, P = len(gt_bboxes), dboxes.shape[0]
    target_locs = torch.zeros(B, P, 4)
    target_labels = torch.zeros(B, P, dtype=torch.long)
    matched_all = torch.zeros(B, P, dtype=torch.long)
    for i in range(B):
        bboxes, labels = gt_bboxes[i], gt_labels[i]
        if bboxes.shape[0] == 0:
            target_locs[i], target_labels[i], matched_all[i] = _empty_targets(P)
            continue
        bboxes_xyxy = box_convert(bboxes, in_fmt='cxcywh', out_fmt='xyxy')
        dboxes_xyxy = box_convert(dboxes, in_fmt='cxcywh', out_fmt='xyxy')
        ious = box_iou(dboxes_xyxy, bboxes_xyxy)
        best_ious, best_dbox_ids = ious.max(dim=0)
        max_ious, matched_bbox_ids = ious.max(dim=1)
        # ssd300.py:256-257 writes into a temporary produced by advanced indexing:
        # matched_bbox_ids is unchanged, so the loop is intentionally absent here.
        max_ious[best_dbox_ids] = iou_thresh
        locs = calc_delta(bboxes[matched_bbox_ids], dboxes)
        labels = labels[matched_bbox_ids]
        labels[max_ious.less(iou_thresh)] = 0
        target_locs[i], target_labels[i], matched_all[i] = locs, labels, matched_bbox_ids
    return target_locs, target_labels, matched_all


def assign_retinanet(pboxes, gt_bboxes, gt_labels, iou_threshs=(0.4, 0.5)):
    """src/models/retinanet.py:171-198 (labels: -1 ignore, 0 background)."""
    B, P = len(gt_bboxes), pboxes.shape[0]
    target_locs = torch.zeros(B, P, 4)
    target_labels = torch.zeros(B, P, dtype=torch.long)
    matched_all = torch.zeros(B, P, dtype=torch.long)
    for i in range(B):
        bboxes, labels = gt_bboxes[i], gt_labels[i]
        if bboxes.shape[0] == 0:
            target_locs[i], target_labels[i], matched_all[i] = _empty_targets(P)
            continue
        bboxes_xyxy = box_convert(bboxes, in_fmt='cxcywh', out_fmt='xyxy')
        pboxes_xyxy = box_convert(pboxes, in_fmt='cxcywh', out_fmt='xyxy')
        ious = box_iou(pboxes_xyxy, bboxes_xyxy)
        best_ious, best_pbox_ids = ious.max(dim=0)
        max_ious, matched_bbox_ids = ious.max(dim=1)
        max_ious[best_pbox_ids] = iou_threshs[1]  # retinanet.py:187-189 (loop is a no-op)
        locs = calc_delta(bboxes[matched_bbox_ids], pboxes)
        labels = labels[matched_bbox_ids]
        labels[max_ious.less(iou_threshs[1])] = -1
        labels[max_ious.less(iou_threshs[0])] = 0
        target_locs[i], target_labels[i], matched_all[i] = locs, labels, matched_bbox_ids
    return target_locs, target_labels, matched_all


def assign_yolov3(pboxes5, gt_bboxes, gt_labels, iou_thresh=0.5):
    """src/models/yolov3.py:157-186 (GT format xywh, normalised)."""
    B, P = len(gt_bboxes), pboxes5.shape[0]
    target_locs = torch.zeros(B, P, 4)
    target_labels = torch.zeros(B, P, dtype=torch.long)
    matched_all = torch.zeros(B, P, dtype=torch.long)
    pboxes, grid_length = pboxes5.split(4, dim=1)
    for i in range(B):
        bboxes, labels = gt_bboxes[i], gt_labels[i]
        if bboxes.shape[0] == 0:
            target_locs[i], target_labels[i], matched_all[i] = _empty_targets(P)
            continue
        is_in_grid = (pboxes[:, [0]] <= bboxes[:, 0]) * (bboxes[:, 0] < pboxes[:, [0]] + grid_length) * \
            (pboxes[:, [1]] <= bboxes[:, 1]) * (bboxes[:, 1] < pboxes[:, [1]] + grid_length)
        bboxes_xyxy = box_convert(bboxes, in_fmt='xywh', out_fmt='xyxy')
        pboxes_xyxy = box_convert(pboxes, in_fmt='xywh', out_fmt='xyxy')
        ious = box_iou(pboxes_xyxy, bboxes_xyxy)
        best_ious, best_pbox_ids = (ious * is_in_grid).max(dim=0)
        max_ious, matched_bbox_ids = ious.max(dim=1)
        max_ious[best_pbox_ids] = 1.  # yolov3.py:175-177 (loop is a no-op)
        locs = calc_delta_yolo(bboxes[matched_bbox_ids], pboxes, grid_length)
        labels = labels[matched_bbox_ids]
        labels[max_ious < 1.] = -1
        labels[max_ious.less(iou_thresh)] = 0
        target_locs[i], target_labels[i], matched_all[i] = locs, labels, matched_bbox_ids
    return target_locs, target_labels, matched_all


def assign_fcos(points, regress_ranges, gt_bboxes, gt_labels):
    """src/models/fcos.py:204-247 (GT cxcywh in pixels).  Returns (target_locs
    [B,P,4] = rays l,r,t,b of the matched box, target_cents [B,P],
    target_labels [B,P], matched [B,P]).  Rows whose label is 0 carry the rays
    of GT 0 and a possibly-NaN centerness exactly like the reference."""
    B, P = len(gt_bboxes), points.shape[0]
    target_locs = torch.zeros(B, P, 4)
    target_cents = torch.zeros(B, P)
    target_labels = torch.zeros(B, P, dtype=torch.long)
    matched_all = torch.zeros(B, P, dtype=torch.long)
    for i in range(B):
        bboxes, labels = gt_bboxes[i], gt_labels[i]
        if bboxes.shape[0] == 0:
            target_locs[i], target_labels[i], matched_all[i] = _empty_targets(P)
            continue
        bboxes_xyxy = box_convert(bboxes, in_fmt='cxcywh', out_fmt='xyxy')
        areas = (bboxes_xyxy[:, 2] - bboxes_xyxy[:, 0]) * (bboxes_xyxy[:, 3] - bboxes_xyxy[:, 1]).repeat(len(points), 1)
        left = points[:, [0]] - bboxes_xyxy[:, 0]
        right = bboxes_xyxy[:, 2] - points[:, [0]]
        top = points[:, [1]] - bboxes_xyxy[:, 1]
        bottom = bboxes_xyxy[:, 3] - points[:, [1]]
        rays = torch.stack([left, right, top, bottom], dim=-1)
        inside_bbox = rays.min(dim=-1).values > 0
        areas[~inside_bbox] = INF
        max_ray = rays.max(dim=-1).values
        inside_regress_range = (regress_ranges[:, [0]] <= max_ray) * (max_ray <= regress_ranges[:, [1]])
        areas[~inside_regress_range] = INF
        min_areas, matched_bbox_ids = areas.min(dim=1)
        locs = rays[range(len(points)), matched_bbox_ids]
        cents = (
            locs[:, 0:2].min(dim=1).values / locs[:, 0:2].max(dim=1).values
            * locs[:, 2:4].min(dim=1).values / locs[:, 2:4].max(dim=1).values
        ).sqrt()
        labels = labels[matched_bbox_ids]
        labels[min_areas == INF] = 0
        target_locs[i], target_cents[i], target_labels[i], matched_all[i] = locs, cents, labels, matched_bbox_ids
    return target_locs, target_cents, target_labels, matched_all


# --------------------------------------------------------------------------
# inference: threshold + top-k + decode (rows a11-a13)
# --------------------------------------------------------------------------

def _rank_mask(col: torch.Tensor, conf_thresh: float, top_k: int) -> torch.Tensor:
    """(score > thr) * (descending rank < top_k), src/models/ssd300.py:370;
    stable sort = "lower anchor index wins" at a boundary tie."""
    rank = col.argsort(descending=True, stable=True).argsort()
    return (col > conf_thresh) * (rank < top_k)


def select_decode(locs, confs, anchors, conf_thresh, top_k, mode, class_begin, class_id_offset):
    """Per-image body shared by every pre_predict (src/models/ssd300.py:364-382,
    retinanet.py:292-310, fcos.py:324-341, yolov3.py:311-329).  `confs` are the
    activated scores [P,C].  mode: 'cxcywh' | 'yolo' | 'fcos'."""
    bboxes, scores, class_ids = [], [], []
    for class_id in range(class_begin, confs.size(1)):
        pos_mask = _rank_mask(confs[:, class_id], conf_thresh, top_k)
        scores_ = confs[pos_mask, class_id]
        class_ids_ = torch.full_like(scores_, class_id + class_id_offset, dtype=torch.long)
        if mode == 'cxcywh':
            bboxes_ = box_convert(calc_coord(locs[pos_mask], anchors[pos_mask]), in_fmt='cxcywh', out_fmt='xyxy')
        elif mode == 'yolo':
            a = anchors[pos_mask]
            bboxes_ = box_convert(calc_coord_yolo(locs[pos_mask], a[:, :4], a[:, 4:]), in_fmt='xywh', out_fmt='xyxy')
        elif mode == 'fcos':
            bboxes_ = calc_coord_fcos(locs[pos_mask], anchors[pos_mask])
        else:
            raise ValueError(mode)
        bboxes.append(bboxes_); scores.append(scores_); class_ids.append(class_ids_)
    return torch.cat(bboxes), torch.cat(scores), torch.cat(class_ids)


def select_decode_fast(locs, confs, anchors, conf_thresh, top_k, mode, class_begin, class_id_offset):
    """Vectorised twin of select_decode (one stable sort over all class columns
    at once); identical outputs, used at sizes where the class loop takes too long."""
    P, C = confs.shape
    sub = confs[:, class_begin:]
    order = sub.argsort(dim=0, descending=True, stable=True)
    rank = torch.empty_like(order)
    ar = torch.arange(P).unsqueeze(1).expand_as(order)
    rank.scatter_(0, order, ar)
    mask = (sub > conf_thresh) & (rank < top_k)          # [P, Cs]
    cls_idx, anc_idx = torch.nonzero(mask.t(), as_tuple=True)  # class-major, anchor ascending
    scores = sub[anc_idx, cls_idx]
    class_ids = cls_idx + class_begin + class_id_offset
    l, a = locs[anc_idx], anchors[anc_idx]
    if mode == 'cxcywh':
        bboxes = box_convert(calc_coord(l, a), in_fmt='cxcywh', out_fmt='xyxy')
    elif mode == 'yolo':
        bboxes = box_convert(calc_coord_yolo(l, a[:, :4], a[:, 4:]), in_fmt='xywh', out_fmt='xyxy')
    elif mode == 'fcos':
        bboxes = calc_coord_fcos(l, a)
    else:
        raise ValueError(mode)
    return bboxes, scores, class_ids.to(torch.long)


def _per_image(fn, out_locs, out_confs, anchors, conf_thresh, top_k, mode, class_begin, class_id_offset):
    pb, ps, pc = [], [], []
    for locs, confs in zip(out_locs, out_confs):
        b, s, c = fn(locs, confs, anchors, conf_thresh, top_k, mode, class_begin, class_id_offset)
        pb.append(b); ps.append(s); pc.append(c)
    return pb, ps, pc


def pre_predict_softmax(out_locs, out_confs, anchors, conf_thresh=0.01, top_k=200, fast=False):
    """src/models/ssd300.py:339-384 == src/models/retinanet.py:267-312."""
    confs = torch.softmax(out_confs, dim=-1)
    fn = select_decode_fast if fast else select_decode
    return _per_image(fn, out_locs, confs, anchors, conf_thresh, top_k, 'cxcywh', 1, 0)


def pre_predict_fcos(out_locs, out_confs, out_cents, points, conf_thresh=0.01, top_k=200, fast=False):
    """src/models/fcos.py:299-343."""
    confs = torch.softmax(out_confs, dim=-1) * out_cents.sigmoid()[..., None]
    fn = select_decode_fast if fast else select_decode
    return _per_image(fn, out_locs, confs, points, conf_thresh, top_k, 'fcos', 1, 0)


def pre_predict_yolov3(out_locs, out_objs, out_confs, pboxes5, conf_thresh=0.01, top_k=200, fast=False):
    """src/models/yolov3.py:282-330 with the evident intent at :320
    (`_calc_coord(locs, pboxes[:, :4], pboxes[:, 4:])`); like the reference it
    overwrites out_locs[..., :2] with their sigmoid in place."""
    out_locs[..., :2] = out_locs[..., :2].sigmoid()
    confs = out_confs.sigmoid() * out_objs.sigmoid()[..., None]
    fn = select_decode_fast if fast else select_decode
    return _per_image(fn, out_locs, confs, pboxes5, conf_thresh, top_k, 'yolo', 0, 1)


# --------------------------------------------------------------------------
# Predictor.run (row a14)
# --------------------------------------------------------------------------

def predictor_run(image_metas, pred_bboxes, pred_scores, pred_class_ids, iou_thresh=0.45,
                  return_keeps=False):
    """src/utilities/predictor.py:51-73."""
    result, keeps = [], []
    for image_meta, bboxes, scores, class_ids in zip(image_metas, pred_bboxes, pred_scores, pred_class_ids):
        keep = batched_nms(bboxes, scores, class_ids, iou_threshold=iou_thresh)
        keeps.append(keep)
        bboxes, scores, class_ids = bboxes[keep], scores[keep], class_ids[keep]
        H, W = image_meta['height'], image_meta['width']
        for (xmin, ymin, xmax, ymax), score, class_id in zip(bboxes, scores, class_ids):
            result.append({
                'image_id': image_meta['image_id'],
                'category_id': class_id.item(),
                'bbox': [xmin.item() * W, ymin.item() * H, (xmax - xmin).item() * W, (ymax - ymin).item() * H],
                'score': score.item(),
            })
    return (result, keeps) if return_keeps else result


def predictor_run_rows(image_metas, pred_bboxes, pred_scores, pred_class_ids, iou_thresh=0.45):
    """Array form of predictor_run: (ids int64 [N,2] = image_id, category_id;
    vals float64 [N,5] = x, y, w, h, score) with the same arithmetic
    (fp32 differences, float64 products) but no per-detection Python objects."""
    ids, vals = [], []
    for image_meta, bboxes, scores, class_ids in zip(image_metas, pred_bboxes, pred_scores, pred_class_ids):
        keep = batched_nms(bboxes, scores, class_ids, iou_threshold=iou_thresh)
        b, s, c = bboxes[keep], scores[keep], class_ids[keep]
        H, W = image_meta['height'], image_meta['width']
        v = torch.stack([b[:, 0].double() * W, b[:, 1].double() * H,
                         (b[:, 2] - b[:, 0]).double() * W, (b[:, 3] - b[:, 1]).double() * H,
                         s.double()], dim=1)
        ids.append(torch.stack([torch.full_like(c, image_meta['image_id']), c], dim=1))
        vals.append(v)
    if not ids:
        return torch.zeros(0, 2, dtype=torch.long), torch.zeros(0, 5, dtype=torch.double)
    return torch.cat(ids), torch.cat(vals)


# --------------------------------------------------------------------------
# synthetic workload generator shared by tests and bench (host-side, seeded)
# --------------------------------------------------------------------------

def synth_gt(i: int, num_fg_classes: int, coco_like: bool, fmt: str = 'cxcywh', scale: float = 1.0):
    """GT boxes for image i: G = 1 + (i mod 16) (VOC-like) or 1 + (i mod 64)
    (COCO-like); centres U(0.15,0.85), sizes U(0.04,0.5); labels in 1..C_fg."""
    g = torch.Generator().manual_seed(1000 + i)
    G = 1 + (i % (64 if coco_like else 16))
    c = 0.15 + 0.70 * torch.rand(G, 2, generator=g)
    wh = 0.04 + 0.46 * torch.rand(G, 2, generator=g)
    boxes = torch.cat([c, wh], dim=1)
    labels = torch.randint(1, num_fg_classes + 1, (G,), generator=g)
    if fmt == 'xywh':
        boxes = torch.cat([c - wh / 2, wh], dim=1)
    return boxes * scale, labels


def hard_negative_mask(loss_neg: torch.Tensor, num_pos: torch.Tensor) -> torch.Tensor:
    """src/models/ssd300.py:277-278: `loss_neg.argsort(descending=True).argsort() < 3 * num_pos`.
    The reference's sort is not stable, so its choice among equal losses at the boundary is unspecified;
    here (and in the build) the lower index wins."""
    rank = loss_neg.argsort(dim=1, descending=True, stable=True).argsort(dim=1)
    return rank < (3 * num_pos).reshape(-1, 1)


def focal_loss_sum(logits: torch.Tensor, labels: torch.Tensor, alpha: float = 0.25, gamma: float = 2) -> torch.Tensor:
    """src/models/losses.py:5-9 with reduction='sum' over the rows the callers select
    (`out_confs[pos_mask + neg_mask]`, src/models/retinanet.py:215, src/models/fcos.py:271): labels >= 0."""
    import torch.nn.functional as F
    C = logits.shape[-1]
    x, t = logits.reshape(-1, C), labels.reshape(-1)
    sel = t >= 0
    x, t = x[sel], t[sel]
    pt = F.softmax(x, dim=-1)
    log_pt = F.log_softmax(x, dim=-1)
    return F.nll_loss(alpha * (1 - pt).pow(gamma) * log_pt, t, reduction='sum')


def smooth_l1_masked_sum(input: torch.Tensor, target: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """src/models/ssd300.py:283 and src/models/retinanet.py:211:
    F.smooth_l1_loss(out_locs[pos_mask], target_locs[pos_mask], reduction='sum') with pos_mask = labels > 0."""
    import torch.nn.functional as F
    pos = labels > 0
    return F.smooth_l1_loss(input[pos], target[pos], reduction='sum')
