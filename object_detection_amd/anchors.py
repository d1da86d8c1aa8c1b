"""Anchor / point tables of the four head families, built once on the host.

Same numbers as the reference's generators (src/models/ssd300.py:182-208,
src/models/ssd512.py:51-77, src/models/retinanet.py:120-139, src/models/fcos.py:142-173,
src/models/yolov3.py:107-125): every value is computed in float64 exactly as the
reference's Python scalar arithmetic does, then rounded once to fp32.  Cell centres are
vectorised with numpy; the handful of per-level sizes use Python floats so that
`pow(x, 0.5)` stays the libm call the reference makes.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

INF = 1e8


def _grid(f_k: int):
    """Row-major (i = y outer, j = x inner) cell indices of an f_k x f_k map."""
    i, j = np.meshgrid(np.arange(f_k, dtype=np.float64), np.arange(f_k, dtype=np.float64), indexing='ij')
    return i.reshape(-1), j.reshape(-1)


def _ssd(cfg, m, s_min, s_max) -> torch.Tensor:
    def s_(k):
        return s_min + (s_max - s_min) * (k - 1) / (m - 1)

    levels = []
    for k, (f_k, num_aspects) in enumerate(cfg, start=1):
        aspects = [1, 'add', 2, 1 / 2] if num_aspects == 4 else [1, 'add', 2, 1 / 2, 3, 1 / 3]
        wh = []
        for a in aspects:
            if a == 'add':
                w = h = pow(s_(k) * s_(k + 1), 0.5)
            else:
                w, h = s_(k) * pow(a, 0.5), s_(k) * pow(1 / a, 0.5)
            wh.append((w, h))
        wh = np.array(wh, dtype=np.float64)                      # [A,2]
        i, j = _grid(f_k)
        cxy = np.stack([(j + 0.5) / f_k, (i + 0.5) / f_k], axis=1)  # [f_k^2,2]
        lvl = np.concatenate([np.repeat(cxy[:, None, :], len(wh), axis=1),
                              np.repeat(wh[None, :, :], cxy.shape[0], axis=0)], axis=2)
        levels.append(lvl.reshape(-1, 4))
    t = torch.from_numpy(np.concatenate(levels).astype(np.float32))
    return t.clamp(min=0, max=1)


def ssd300_default_boxes() -> torch.Tensor:
    """[8732,4] cxcywh, clamped to [0,1]."""
    return _ssd([[38, 4], [19, 6], [10, 6], [5, 6], [3, 4], [1, 4]], 6, 0.1, 0.88)


def ssd512_default_boxes() -> torch.Tensor:
    """[24564,4] cxcywh, clamped to [0,1]."""
    return _ssd([[64, 4], [32, 6], [16, 6], [8, 6], [4, 6], [2, 4], [1, 4]], 7, 0.07, 0.88)


def retinanet_prior_boxes(f_k_list: Sequence[int]) -> torch.Tensor:
    """[9*sum f_k^2, 4] cxcywh: aspect (0.5,1,2) outer x scale (2^0,2^(1/3),2^(2/3)) inner."""
    levels = []
    for f_k in f_k_list:
        wh = []
        for aspect in [0.5, 1, 2]:
            for scale in [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]:
                wh.append((4 * scale / f_k * pow(1 / aspect, 0.5), 4 * scale / f_k * pow(aspect, 0.5)))
        wh = np.array(wh, dtype=np.float64)
        i, j = _grid(f_k)
        cxy = np.stack([(j + 0.5) / f_k, (i + 0.5) / f_k], axis=1)
        lvl = np.concatenate([np.repeat(cxy[:, None, :], 9, axis=1), np.repeat(wh[None, :, :], cxy.shape[0], axis=0)], axis=2)
        levels.append(lvl.reshape(-1, 4))
    t = torch.from_numpy(np.concatenate(levels).astype(np.float32))
    return t.clamp(min=0, max=1)


def fcos_points(f_k_list: Sequence[int], image_size: float = 512) -> torch.Tensor:
    """[sum f_k^2, 2] pixel centres.  The reference hard-codes 512 px (src/models/fcos.py:152)."""
    levels = []
    for f_k in f_k_list:
        y, x = _grid(f_k)
        levels.append(np.stack([(x + 0.5) / f_k * image_size, (y + 0.5) / f_k * image_size], axis=1))
    return torch.from_numpy(np.concatenate(levels).astype(np.float32))


def fcos_regress_ranges(f_k_list: Sequence[int]) -> torch.Tensor:
    rr = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]
    levels = [np.repeat(np.array([rr[i]], dtype=np.float64), f_k * f_k, axis=0) for i, f_k in enumerate(f_k_list)]
    return torch.from_numpy(np.concatenate(levels).astype(np.float32))


def yolov3_prior_boxes() -> torch.Tensor:
    """[10647,5] = x, y (cell top-left), w, h, grid_length; coarse map first."""
    S = 416
    levels = []
    for f_k, aspects in [[13, [[116, 90], [156, 198], [373, 326]]],
                         [26, [[30, 61], [62, 45], [59, 119]]],
                         [52, [[10, 13], [16, 30], [33, 23]]]]:
        i, j = _grid(f_k)
        xy = np.stack([j / f_k, i / f_k], axis=1)
        whg = np.array([[w / S, h / S, 1 / f_k] for w, h in aspects], dtype=np.float64)
        lvl = np.concatenate([np.repeat(xy[:, None, :], 3, axis=1), np.repeat(whg[None, :, :], xy.shape[0], axis=0)], axis=2)
        levels.append(lvl.reshape(-1, 5))
    return torch.from_numpy(np.concatenate(levels).astype(np.float32))
