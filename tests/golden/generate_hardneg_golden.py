#!/usr/bin/env python3
"""Golden fixture for the hard-negative-mining rank mask (src/models/ssd300.py:275-278), produced by the
REAL reference loss(): its local variables `loss_neg`, `pos_mask` and `neg_mask` are captured when loss()
returns.  Uses the import scaffold of generate_golden.py (run in the build container only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_golden as G  # noqa: E402  (stubs torchvision, imports the reference models)
from oracle import boxops_oracle as O  # noqa: E402


def main():
    ssd = G.bare(G.SSD300, 21)
    ssd.dboxes = ssd._get_default_boxes()
    P = ssd.dboxes.shape[0]
    gts = [O.synth_gt(i, 20, False) for i in (0, 5, 15, 9)]
    gb, gl = [g[0] for g in gts], [g[1] for g in gts]
    gen = torch.Generator().manual_seed(123)
    outs = (torch.randn(len(gb), P, 4, generator=gen), 2.0 * torch.randn(len(gb), P, 21, generator=gen))
    got = G.capture_loss_locals(ssd, outs, gb, gl, ['loss_neg', 'neg_mask', 'pos_mask'])
    assert '_exception' not in got, got.get('_exception')
    loss_neg, neg_mask, pos_mask = got['loss_neg'].detach(), got['neg_mask'], got['pos_mask']
    k = 3 * pos_mask.sum(dim=1)
    for b in range(loss_neg.shape[0]):            # the reference's argsort is not stable: no tie at the boundary
        v = loss_neg[b].sort(descending=True).values
        kb = int(k[b])
        assert 0 < kb < P and float(v[kb - 1]) != float(v[kb]), 'boundary tie: pick another seed'
        assert int(neg_mask[b].sum()) == kb
    np.savez_compressed(os.path.join(HERE, 'hardneg_ssd300.npz'), loss_neg=loss_neg.numpy().astype(np.float32),
                        k=k.numpy().astype(np.int64), neg_mask=np.packbits(neg_mask.numpy().astype(np.uint8), axis=1),
                        num_pos=pos_mask.sum(dim=1).numpy().astype(np.int64))
    print('hardneg_ssd300: k =', k.tolist())


if __name__ == '__main__':
    main()
