#!/usr/bin/env python3
"""Golden fixture for the focal confidence loss (src/models/losses.py:5-9 as called at
src/models/retinanet.py:215), produced by the REAL reference RetinaNet.loss(): its locals `target_labels`,
`loss_conf` and `pos_mask` are captured when loss() returns.  Uses the import scaffold of generate_golden.py
(run in the build container only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_golden as G  # noqa: E402  (stubs torchvision, imports the reference models)
from oracle import boxops_oracle as O  # noqa: E402


def main():
    m = G.bare(G.RetinaNet, 21)
    m.pboxes = m._get_prior_boxes([8, 4, 2, 1, 1])
    P = m.pboxes.shape[0]
    gts = [O.synth_gt(i, 20, False) for i in (2, 9)]
    gb, gl = [g[0] for g in gts], [g[1] for g in gts]
    gen = torch.Generator().manual_seed(321)
    out_locs = torch.randn(len(gb), P, 4, generator=gen)
    out_confs = 1.5 * torch.randn(len(gb), P, 21, generator=gen)
    got = G.capture_loss_locals(m, (out_locs, out_confs), gb, gl, ['target_labels', 'loss_conf', 'pos_mask'])
    assert '_exception' not in got, got.get('_exception')
    labels = got['target_labels']
    n_pos = int(got['pos_mask'].sum())
    assert n_pos > 0 and int((labels < 0).sum()) > 0, 'want positives and ignored anchors in the fixture'
    np.savez_compressed(os.path.join(HERE, 'focal_retinanet_small.npz'), out_confs=out_confs.numpy(),
                        target_labels=labels.numpy().astype(np.int64), loss_conf=np.float64(float(got['loss_conf'])),
                        num_pos=np.int64(n_pos))
    print('focal_retinanet_small: P =', P, 'positives =', n_pos, 'ignored =', int((labels < 0).sum()),
          'loss_conf =', float(got['loss_conf']))


if __name__ == '__main__':
    main()
