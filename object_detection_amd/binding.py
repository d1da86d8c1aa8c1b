"""Maintainer-side binding of the box-ops classes into the reference's model objects (Seam 2).

The reference's loops call `model.loss(outputs, gt_bboxes, gt_labels)` (src/train.py:149),
`model.pre_predict(outputs)` (src/train.py:155, src/test.py:98) and `predictor.run(...)` (src/train.py:156,
src/test.py:99).  `bind_boxops` routes the first two of a reference model object to the HIP path and leaves the
network part (forward, backbone, heads) alone; `Predictor` of this package replaces the reference's class.
"""
from __future__ import annotations

from .models import MODELS


def bind_boxops(model, name: str, **ctor):
    """model: an instance of the reference's SSD300 / SSD512 / RetinaNet / FCOS / YoloV3 (anything with `.nc`).
    name: 'ssd300' | 'ssd512' | 'retinanet' | 'fcos' | 'yolov3'.  ctor: what the box-ops class needs beyond the
    class count -- `f_k_list` (feature-map sizes) for RetinaNet and FCOS, whose tables the reference builds from
    the first forward (src/models/retinanet.py:116-117).
    The reference stores `nc = num_classes + 1` for the soft-max heads (src/models/ssd300.py:57) and
    `nc = num_classes` for YOLOv3; the box-ops constructors take num_classes."""
    num_classes = model.nc if name == 'yolov3' else model.nc - 1
    impl = MODELS[name](num_classes, **ctor)
    model.boxops = impl
    model.loss = impl.loss
    model.pre_predict = impl.pre_predict
    model.detect = impl.detect
    return model
