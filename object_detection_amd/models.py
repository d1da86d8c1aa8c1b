"""Box-ops halves of the reference's detection models, on the HIP operators.

Each class mirrors the interface of its namesake in the reference
(`DetectionNet`, src/models/base.py:45-51): `loss(outputs, gt_bboxes, gt_labels, ...)`
and `pre_predict(outputs, conf_thresh=0.01, top_k=200)` with the same argument
meaning and return types, so `train.py:149,155-156` and `test.py:98-99` work
unchanged when a model's two methods are bound to these.  The conv network is not
part of this package: a class here holds only the anchor/point table.

Inside `loss`, Step 1 (target assignment) is one batched HIP call; Steps 2-5 are
the reference's differentiable loss arithmetic in plain PyTorch.  `pre_predict`
hands the raw head outputs to the HIP path, which fuses the activation (soft-max,
soft-max x sigmoid(centerness), sigmoid x sigmoid(objectness)) into the
threshold/top-k/decode pass.  `detect` chains pre_predict -> NMS -> result rows
without leaving the device.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import anchors as A
from . import ops


def iou_loss_with_distance(input, target, reduction='none'):
    """src/models/losses.py:12-29 (rays ordered l, r, t, b)."""
    def area(t):
        return (t[:, 1] + t[:, 0]) * (t[:, 3] + t[:, 2])
    inter = area(torch.minimum(input, target))
    union = area(input) + area(target) - inter
    loss = -(inter / union.clamp(min=1e-8)).log()
    return loss.sum() if reduction == 'sum' else (loss.mean() if reduction == 'mean' else loss)


class _BoxOps:
    """Shared plumbing: device-resident anchor table, candidate selection, fused detection."""
    decode_mode = ops.DECODE_CXCYWH
    class_begin = 1          # column 0 is background for the softmax heads
    class_id_offset = 0
    return_cpu = True        # pre_predict returns CPU tensors like the reference (ssd300.py:357-358)

    def __init__(self):
        self._dev_tables = {}

    def _table(self, t: torch.Tensor, device) -> torch.Tensor:
        device = torch.device(device)
        key = (id(t), device.type, device.index)
        if key not in self._dev_tables:
            self._dev_tables[key] = t.to(device=device, dtype=torch.float32).contiguous()
        return self._dev_tables[key]

    activation = ops.ACT_SOFTMAX

    # -- subclasses provide: _anchors_for_decode(), _head(outputs) -> (locs, class head output, extra)
    def _activate(self, outputs: tuple):
        """(locs, activated scores [B,P,C]) with the scores exactly as the fused kernels compute them."""
        locs, head, extra = self._head(outputs)
        return locs, ops.activate_scores(head.contiguous(), None if extra is None else extra.contiguous(), self.activation)

    # True: the activation runs inside the selection pass (the head output is read once).  False: the activation is
    # torch's own, exactly the reference's lines (ssd300.py:354, fcos.py:314, yolov3.py:298-300), on the device the
    # head output lives on, and the selection ranks those very bits -- for callers who need the candidate sets of a
    # particular torch build bit for bit, at the price of one more pass over [B,P,C].
    fused_activation = True

    def _torch_scores(self, head, extra):
        if self.activation == ops.ACT_SOFTMAX:
            return F.softmax(head, dim=-1)
        if self.activation == ops.ACT_SOFTMAX_SIGMOID:
            return F.softmax(head, dim=-1) * extra.sigmoid()[..., None]
        if self.activation == ops.ACT_SIGMOID_SIGMOID:
            return head.sigmoid() * extra.sigmoid()[..., None]
        return head

    def candidates(self, outputs: tuple, conf_thresh: float = 0.01, top_k: int = 200) -> ops.Candidates:
        locs, head, extra = self._head(outputs)
        anchors = self._table(self._anchors_for_decode(), head.device)
        if not self.fused_activation:
            return ops.select_decode(self._torch_scores(head, extra).contiguous(), locs.contiguous(), anchors, self.decode_mode,
                                     self.class_begin, self.class_id_offset, conf_thresh, top_k, activation=ops.ACT_NONE)
        return ops.select_decode(head.contiguous(), locs.contiguous(), anchors, self.decode_mode,
                                 self.class_begin, self.class_id_offset, conf_thresh, top_k,
                                 activation=self.activation, extra=None if extra is None else extra.contiguous())

    def pre_predict(self, outputs: tuple, conf_thresh: float = 0.01, top_k: int = 200) -> tuple:
        c = self.candidates(outputs, conf_thresh, top_k)
        counts = c.counts.tolist()                      # the one device->host sync
        boxes, scores, cls = c.boxes, c.scores, c.class_ids
        if self.return_cpu:
            if 2 * sum(counts) >= scores.numel():       # mostly full: three large copies instead of 3 * B small ones
                boxes, scores, cls = boxes.cpu(), scores.cpu(), cls.cpu()
                pb = [boxes[b, :n] for b, n in enumerate(counts)]
                ps = [scores[b, :n] for b, n in enumerate(counts)]
                pc = [cls[b, :n] for b, n in enumerate(counts)]
            else:
                pb = [boxes[b, :n].cpu() for b, n in enumerate(counts)]
                ps = [scores[b, :n].cpu() for b, n in enumerate(counts)]
                pc = [cls[b, :n].cpu() for b, n in enumerate(counts)]
        else:
            pb = [boxes[b, :n] for b, n in enumerate(counts)]
            ps = [scores[b, :n] for b, n in enumerate(counts)]
            pc = [cls[b, :n] for b, n in enumerate(counts)]
        from . import predictor
        return predictor.remember((pb, ps, pc), c)       # Predictor.run continues on the device copy if handed these back

    def detect(self, outputs: tuple, image_metas: Sequence[dict], conf_thresh: float = 0.01, top_k: int = 200,
               iou_thresh: float = 0.45, trim: bool = True):
        """pre_predict + Predictor.run on the device.  Returns (rows_ids [N,2] i64, rows_vals [N,5] f64)
        device tensors; nothing is synchronised except the final row count.  trim=False skips that one
        synchronisation too and returns (rows_ids, rows_vals, row_offsets) with the buffers at full length
        (rows beyond row_offsets[-1] unused) -- the form for batches in flight on several streams."""
        c = self.candidates(outputs, conf_thresh, top_k)
        dev = c.boxes.device
        ids = torch.tensor([m['image_id'] for m in image_metas], dtype=torch.int64).to(dev, non_blocking=True)
        hw = torch.tensor([[m['height'], m['width']] for m in image_metas], dtype=torch.int32).to(dev, non_blocking=True)
        _, _, rows_ids, rows_vals, row_off = ops.nms_rows(c, iou_thresh, ids, hw)
        if not trim:
            return rows_ids, rows_vals, row_off
        n = int(row_off[-1].item())
        return rows_ids[:n], rows_vals[:n]


class SSD300(_BoxOps):
    """Box-ops of src/models/ssd300.py (default boxes :182-208, loss :210-297, pre_predict :339-384)."""

    def __init__(self, num_classes: int):
        super().__init__()
        self.nc = num_classes + 1
        self.dboxes = self._get_default_boxes()

    def _get_default_boxes(self):
        return A.ssd300_default_boxes()

    def _anchors_for_decode(self):
        return self.dboxes

    def _head(self, outputs):
        out_locs, out_confs = outputs
        return out_locs.detach(), out_confs.detach(), None

    def assign(self, gt_bboxes: list, gt_labels: list, device, iou_thresh: float = 0.5, want_best=False):
        gb, gl, off, total = ops.pack_gt(gt_bboxes, gt_labels, device)
        return ops.assign_anchor(self._table(self.dboxes, device), gb, gl, off, total, ops.ASSIGN_CXCYWH,
                                 iou_thresh, iou_thresh, want_best)

    def loss(self, outputs: tuple, gt_bboxes: list, gt_labels: list, iou_thresh: float = 0.5, alpha: float = 1.0) -> dict:
        out_locs, out_confs = outputs
        B, P, C = out_confs.size()
        target_locs, target_labels, _ = self.assign(gt_bboxes, gt_labels, out_locs.device, iou_thresh)
        # Steps 2-5: src/models/ssd300.py:273-297
        pos_mask = target_labels > 0
        loss_neg = F.cross_entropy(out_confs.view(-1, C), target_labels.view(-1), reduction='none').view(B, -1)
        loss_neg[pos_mask] = 0
        neg_mask = ops.topk_mask(loss_neg, 3 * pos_mask.sum(dim=1))      # hard-negative mining, ssd300.py:277-278
        N = pos_mask.sum()
        loss_loc = ops.smooth_l1_masked_sum(out_locs, target_locs, target_labels) / N      # ssd300.py:283, fused with the mask
        # Step 4 (ssd300.py:288): cross entropy over positives and mined negatives = the fused focal sum with
        # alpha = 1, gamma = 0; every other row carries the ignore label
        sel_labels = torch.where(pos_mask | neg_mask, target_labels, torch.full_like(target_labels, -1))
        loss_conf = ops.focal_loss_sum(out_confs, sel_labels, 1.0, 0.0) / N
        return {'loss': loss_conf + alpha * loss_loc, 'loss_loc': loss_loc, 'loss_conf': loss_conf}


class SSD512(SSD300):
    """Box-ops of src/models/ssd512.py (24564 default boxes; loss / pre_predict inherited)."""

    def _get_default_boxes(self):
        return A.ssd512_default_boxes()


class RetinaNet(_BoxOps):
    """Box-ops of src/models/retinanet.py (prior boxes :120-139, loss :141-225, pre_predict :267-312).
    The reference builds its priors lazily from the feature-map sizes of the first forward
    (:116-117); here they are given up front (512 px: [64,32,16,8,4]; 800 px: [100,50,25,13,7])."""

    def __init__(self, num_classes: int, f_k_list: Sequence[int] = (64, 32, 16, 8, 4)):
        super().__init__()
        self.nc = num_classes + 1
        self.pboxes = self._get_prior_boxes(list(f_k_list))

    def _get_prior_boxes(self, f_k_list):
        return A.retinanet_prior_boxes(f_k_list)

    def _anchors_for_decode(self):
        return self.pboxes

    def _head(self, outputs):
        out_locs, out_confs = outputs
        return out_locs.detach(), out_confs.detach(), None

    def assign(self, gt_bboxes, gt_labels, device, iou_threshs: tuple = (0.4, 0.5), want_best=False):
        gb, gl, off, total = ops.pack_gt(gt_bboxes, gt_labels, device)
        return ops.assign_anchor(self._table(self.pboxes, device), gb, gl, off, total, ops.ASSIGN_CXCYWH,
                                 iou_threshs[0], iou_threshs[1], want_best)

    def loss(self, outputs: tuple, gt_bboxes: list, gt_labels: list, iou_threshs: tuple = (0.4, 0.5)) -> dict:
        out_locs, out_confs = outputs
        target_locs, target_labels, _ = self.assign(gt_bboxes, gt_labels, out_locs.device, iou_threshs)
        # Steps 2-5: src/models/retinanet.py:205-225
        pos_mask = target_labels > 0
        neg_mask = target_labels == 0
        N = pos_mask.sum()
        loss_loc = ops.smooth_l1_masked_sum(out_locs, target_locs, target_labels) / N      # retinanet.py:211, fused with the mask
        # Step 4 fused: rows with label -1 (ignored) are skipped inside the kernel (retinanet.py:215)
        loss_conf = ops.focal_loss_sum(out_confs, target_labels) / N
        return {'loss': loss_conf + loss_loc, 'loss_loc': loss_loc, 'loss_conf': loss_conf}


class FCOS(_BoxOps):
    """Box-ops of src/models/fcos.py (points/ranges :142-173, loss :175-279, pre_predict :299-343).
    image_size is 512 in the reference (hard-coded at :152)."""
    decode_mode = ops.DECODE_FCOS
    activation = ops.ACT_SOFTMAX_SIGMOID

    def __init__(self, num_classes: int, f_k_list: Sequence[int] = (64, 32, 16, 8, 4), image_size: float = 512):
        super().__init__()
        self.nc = num_classes + 1
        self.points = A.fcos_points(list(f_k_list), image_size)
        self.regress_ranges = A.fcos_regress_ranges(list(f_k_list))

    def _anchors_for_decode(self):
        return self.points

    def _head(self, outputs):
        out_locs, out_confs, out_cents = outputs
        return out_locs.detach(), out_confs.detach(), out_cents.detach()

    def assign(self, gt_bboxes, gt_labels, device):
        gb, gl, off, total = ops.pack_gt(gt_bboxes, gt_labels, device)
        return ops.assign_fcos(self._table(self.points, device), self._table(self.regress_ranges, device),
                               gb, gl, off, total)

    def loss(self, outputs: tuple, gt_bboxes: list, gt_labels: list) -> dict:
        out_locs, out_confs, out_cents = outputs
        target_locs, target_cents, target_labels, _ = self.assign(gt_bboxes, gt_labels, out_locs.device)
        # Steps 2-5: src/models/fcos.py:254-279
        pos_mask = target_labels > 0
        neg_mask = target_labels == 0
        N = pos_mask.sum()
        loss_loc = iou_loss_with_distance(out_locs[pos_mask].exp(), target_locs[pos_mask], reduction='sum') / target_cents[pos_mask].sum()
        loss_cent = F.binary_cross_entropy_with_logits(out_cents[pos_mask], target_cents[pos_mask], reduction='sum') / N
        loss_conf = ops.focal_loss_sum(out_confs, target_labels) / N         # fcos.py:271, fused
        return {'loss': loss_conf + loss_cent + loss_loc, 'loss_loc': loss_loc, 'loss_cent': loss_cent, 'loss_conf': loss_conf}


class YoloV3(_BoxOps):
    """Box-ops of src/models/yolov3.py (priors :107-125, loss :127-234, pre_predict :282-330).
    pre_predict decodes with `_calc_coord(locs, pboxes[:, :4], pboxes[:, 4:])`, the evident intent
    of :320 (which raises TypeError in the reference), and like the reference overwrites
    out_locs[..., :2] with their sigmoid in place (:297)."""
    decode_mode = ops.DECODE_YOLO
    class_begin = 0
    class_id_offset = 1      # emitted class id = column + 1 (yolov3.py:319)
    activation = ops.ACT_SIGMOID_SIGMOID

    def __init__(self, num_classes: int):
        super().__init__()
        self.nc = num_classes
        self.pboxes = A.yolov3_prior_boxes()

    def _anchors_for_decode(self):
        return self.pboxes

    def _head(self, outputs):
        out_locs, out_objs, out_confs = outputs
        out_locs = out_locs.detach()
        out_locs[..., :2] = out_locs[..., :2].sigmoid()        # in place, like yolov3.py:297
        return out_locs, out_confs.detach(), out_objs.detach()

    def assign(self, gt_bboxes, gt_labels, device, iou_thresh: float = 0.5, want_best=False):
        gb, gl, off, total = ops.pack_gt(gt_bboxes, gt_labels, device)
        return ops.assign_anchor(self._table(self.pboxes, device), gb, gl, off, total, ops.ASSIGN_YOLO,
                                 iou_thresh, 1.0, want_best)

    def loss(self, outputs: tuple, gt_bboxes: list, gt_labels: list, iou_thresh: float = 0.5) -> dict:
        out_locs, out_objs, out_confs = outputs
        target_locs, target_labels, _ = self.assign(gt_bboxes, gt_labels, out_locs.device, iou_thresh)
        # Steps 2-5: src/models/yolov3.py:192-234
        pos_mask = target_labels > 0
        neg_mask = target_labels == 0
        N = pos_mask.sum()
        loss_loc = (F.binary_cross_entropy_with_logits(out_locs[pos_mask][..., :2], target_locs[pos_mask][..., :2], reduction='sum')
                    + F.mse_loss(out_locs[pos_mask][..., 2:], target_locs[pos_mask][..., 2:], reduction='sum')) / N
        loss_conf = F.binary_cross_entropy_with_logits(
            out_confs[pos_mask], F.one_hot(target_labels[pos_mask] - 1, num_classes=self.nc).float(), reduction='sum') / N
        sel = pos_mask + neg_mask
        loss_obj = F.binary_cross_entropy_with_logits(out_objs[sel], pos_mask[sel].float(), reduction='sum') / N
        return {'loss': loss_loc + loss_obj + loss_conf, 'loss_loc': loss_loc, 'loss_conf': loss_conf, 'loss_obj': loss_obj}


MODELS = {'ssd300': SSD300, 'ssd512': SSD512, 'yolov3': YoloV3, 'retinanet': RetinaNet, 'fcos': FCOS}
