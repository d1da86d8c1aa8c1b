#!/usr/bin/env python3
"""Golden values of the whole loss() of every family, produced by the REAL reference models on CPU.
Inputs are seeded (torch.Generator), so only the seeds and the returned scalars are stored.
Uses the import scaffold of generate_golden.py (run in the build container only)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_golden as G  # noqa: E402
from oracle import boxops_oracle as O  # noqa: E402


def outputs_for(family, seed, B, P, C):
    g = torch.Generator().manual_seed(seed)
    locs = torch.randn(B, P, 4, generator=g)
    confs = 1.5 * torch.randn(B, P, C, generator=g)
    if family == 'fcos':
        return (locs, confs, torch.randn(B, P, generator=g))
    if family == 'yolo':
        return (locs, torch.randn(B, P, generator=g), confs)
    return (locs, confs)


def main():
    out = {}
    fk = [8, 4, 2, 1, 1]
    ssd = G.bare(G.SSD300, 21); ssd.dboxes = ssd._get_default_boxes()
    ret = G.bare(G.RetinaNet, 21); ret.pboxes = ret._get_prior_boxes(fk)
    fcos = G.bare(G.FCOS, 21); fcos.points = fcos._get_points(fk); fcos.regress_ranges = fcos._get_regress_ranges(fk)
    yolo = G.bare(G.YoloV3, 20); yolo.pboxes = yolo._get_prior_boxes()
    cases = [('ssd300', 'ssd', ssd, ssd.dboxes.shape[0], 21, dict(fmt='cxcywh', scale=1.0)),
             ('retinanet_small', 'retina', ret, ret.pboxes.shape[0], 21, dict(fmt='cxcywh', scale=1.0)),
             ('fcos_small', 'fcos', fcos, fcos.points.shape[0], 21, dict(fmt='cxcywh', scale=512.0)),
             ('yolov3', 'yolo', yolo, yolo.pboxes.shape[0], 20, dict(fmt='xywh', scale=1.0))]
    for name, family, model, P, C, gtkw in cases:
        gts = [O.synth_gt(i, 20, False, **gtkw) for i in (3, 8)]
        gb, gl = [g[0] for g in gts], [g[1] for g in gts]
        seed = 4000 + len(out)
        res = model.loss(outputs_for(family, seed, 2, P, C), [b.clone() for b in gb], [l.clone() for l in gl])
        out[name] = dict(family=family, seed=seed, P=int(P), C=int(C), gt_images=[3, 8], gt_kw=gtkw,
                         loss={k: float(v) for k, v in res.items()})
        print(name, out[name]['loss'])
    with open(os.path.join(HERE, 'loss_values.json'), 'w') as f:
        json.dump(dict(torch=torch.__version__, cases=out), f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
