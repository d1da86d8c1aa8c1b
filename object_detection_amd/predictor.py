"""Predictor: per-class NMS + COCO-style result dicts, mirroring
src/utilities/predictor.py:8-83 (same constructor arguments, same `run` signature and
result format) with the NMS and the row arithmetic on the GPU.

`run` accepts what the reference's `run` accepts (lists of per-image boxes / scores /
class ids, on any device).  When every image's class ids are non-decreasing -- which is
how every pre_predict emits them -- the whole batch goes through one call that does the
segmented NMS and writes the result rows; otherwise each image goes through the generic
batched_nms and one packing call follows.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import ops
from ._lib import DetboxError


class CandidateList(list):
    """One of the three lists pre_predict returns (boxes, scores or class ids per image).  It is an ordinary list
    of CPU tensors that also keeps the device-resident candidates it was copied from, so that Predictor.run,
    handed the three lists of one pre_predict call back unmodified (the flow of src/test.py:93-99), continues on
    the device instead of uploading and re-segmenting what just came down.  The link lives and dies with the
    lists: nothing is kept in the module or on the model."""

    def __init__(self, items, resident=None, role=0):
        super().__init__(items)
        self._resident, self._role = resident, role


class _Resident:
    """The device candidates of one pre_predict call and a fingerprint of every tensor that was handed out."""

    def __init__(self, candidates, lists):
        self.candidates = candidates
        self.prints = [[_fingerprint(t) for t in lst] for lst in lists]
        self.refs = [[t for t in lst] for lst in lists]           # identity: the very tensor objects


def _fingerprint(t: torch.Tensor):
    # in-place operators bump the tensor's version counter; a replaced element, a slice or a copy is another object
    return (t.data_ptr(), tuple(t.shape), t.dtype, t._version)


def remember(lists, candidates):
    """Wraps the three lists of a pre_predict call so that Predictor.run can find the device copy again."""
    r = _Resident(candidates, lists)
    return tuple(CandidateList(lst, r, role) for role, lst in enumerate(lists))


def recall(pb, ps, pc):
    """The device candidates behind (pb, ps, pc) if these are the three lists of one pre_predict call, in their
    roles, with every element still the tensor that was handed out and untouched by in-place operators; else None
    (the generic path then uploads the lists, which is always correct).  Writes through views that bypass the
    version counter (`.numpy()`, `.data`) are not seen: pass plain lists (`list(pb)`) after such edits."""
    r = getattr(pb, '_resident', None)
    if r is None or getattr(ps, '_resident', None) is not r or getattr(pc, '_resident', None) is not r:
        return None
    if (getattr(pb, '_role', -1), getattr(ps, '_role', -1), getattr(pc, '_role', -1)) != (0, 1, 2):
        return None
    for lst, refs, prints in zip((pb, ps, pc), r.refs, r.prints):
        if len(lst) != len(refs):
            return None
        for t, ref, fp in zip(lst, refs, prints):
            if t is not ref or _fingerprint(t) != fp:
                return None
    return r.candidates


def rows_to_dicts(rows_ids: torch.Tensor, rows_vals: torch.Tensor) -> List[dict]:
    """[N,2] i64 + [N,5] f64 -> the reference's list of dicts (src/utilities/predictor.py:62-72)."""
    ids, vals = rows_ids.tolist(), rows_vals.tolist()
    return [{'image_id': i[0], 'category_id': i[1], 'bbox': v[:4], 'score': v[4]} for i, v in zip(ids, vals)]


class Predictor:
    def __init__(self, iou_thresh: float = 0.45, classes: dict = None, out_dir: str = None):
        self.iou_thresh = iou_thresh
        self.classes = classes
        self.out_dir = out_dir      # drawing is not part of the box-ops path

    def run_rows(self, image_metas: Sequence[dict], pred_bboxes, pred_scores, pred_class_ids):
        """-> (rows_ids [N,2] i64, rows_vals [N,5] f64) on the GPU."""
        B = len(image_metas)
        dev = pred_bboxes[0].device if B and pred_bboxes[0].is_cuda else ops._device()
        if B == 0:
            return (torch.zeros((0, 2), dtype=torch.int64, device=dev), torch.zeros((0, 5), dtype=torch.float64, device=dev))
        c = recall(pred_bboxes, pred_scores, pred_class_ids)
        if c is not None and c.boxes.shape[0] == B:
            dev = c.boxes.device
            with torch.cuda.device(dev):
                ids = torch.tensor([m['image_id'] for m in image_metas], dtype=torch.int64).to(dev, non_blocking=True)
                hw = torch.tensor([[m['height'], m['width']] for m in image_metas], dtype=torch.int32).to(dev, non_blocking=True)
                keep, kc, rows_ids, rows_vals, row_off = ops.nms_rows(c, self.iou_thresh, ids, hw)
                n = int(row_off[-1].item())
                if bool((kc < 0).any().item()):
                    raise DetboxError('NMS reported an over-long segment')
                return rows_ids[:n], rows_vals[:n]
        counts = [int(s.shape[0]) for s in pred_scores]
        n_max = max(max(counts), 1)
        with torch.cuda.device(dev):
            boxes = torch.zeros((B, n_max, 4), dtype=torch.float32, device=dev)
            scores = torch.zeros((B, n_max), dtype=torch.float32, device=dev)
            cls = torch.full((B, n_max), -1, dtype=torch.int64, device=dev)
            for b in range(B):
                n = counts[b]
                if n:
                    boxes[b, :n] = pred_bboxes[b].to(dev, torch.float32)
                    scores[b, :n] = pred_scores[b].to(dev, torch.float32)
                    cls[b, :n] = pred_class_ids[b].to(dev, torch.int64)
            valid = torch.arange(n_max, device=dev)[None, :] < torch.tensor(counts, device=dev)[:, None]
            ids = torch.tensor([m['image_id'] for m in image_metas], dtype=torch.int64, device=dev)
            hw = torch.tensor([[m['height'], m['width']] for m in image_metas], dtype=torch.int32, device=dev)
            # fast path conditions (one sync): ids grouped ascending, small non-negative, segments sortable on chip
            nondecr = ((cls[:, 1:] >= cls[:, :-1]) | ~valid[:, 1:]).all()
            cmin = torch.where(valid, cls, torch.zeros_like(cls)).min()
            cmax = cls.max()
            ok, cmin, cmax = bool(nondecr.item()), int(cmin.item()), int(cmax.item())
            fast = ok and cmin >= 0 and cmax < 4096
            if fast:
                S = cmax + 1
                onehot_counts = torch.zeros((B, S), dtype=torch.int32, device=dev)
                onehot_counts.scatter_add_(1, cls.clamp(min=0), valid.to(torch.int32))
                fast = int(onehot_counts.max().item()) <= ops.NMS_MAX_SORT
            if fast:
                seg = torch.zeros((B, S + 1), dtype=torch.int32, device=dev)
                seg[:, 1:] = torch.cumsum(onehot_counts, 1).to(torch.int32)
                cand = ops.Candidates(boxes, scores, cls, seg, seg[:, -1].contiguous(), min(n_max, ops.NMS_MAX_SORT))
                keep, kc, rows_ids, rows_vals, row_off = ops.nms_rows(cand, self.iou_thresh, ids, hw)
            else:
                keep = torch.zeros((B, n_max), dtype=torch.int64, device=dev)
                kcl = []
                for b in range(B):
                    n = counts[b]
                    k = ops.batched_nms(boxes[b, :n], scores[b, :n], cls[b, :n], self.iou_thresh)
                    keep[b, :k.numel()] = k
                    kcl.append(k.numel())
                kc = torch.tensor(kcl, dtype=torch.int32, device=dev)
                rows_ids, rows_vals, row_off = ops.pack_detections(boxes, scores, cls, keep, kc, ids, hw)
            offs = row_off.tolist()
            if any(o < 0 for o in offs) or bool((kc < 0).any().item()):
                raise DetboxError('NMS reported an over-long segment')
            return rows_ids[:offs[-1]], rows_vals[:offs[-1]]

    def run(self, image_metas: list, pred_bboxes, pred_scores, pred_class_ids) -> list:
        rows_ids, rows_vals = self.run_rows(image_metas, pred_bboxes, pred_scores, pred_class_ids)
        return rows_to_dicts(rows_ids, rows_vals)
