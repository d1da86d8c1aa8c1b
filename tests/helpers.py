"""Shared helpers for the tests: golden-fixture loading and input regeneration."""
import hashlib
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def manifest():
    with open(os.path.join(GOLDEN, 'manifest.json')) as f:
        return json.load(f)


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def unpack_gt(z):
    off = z['gt_offsets']
    gb, gl = torch.from_numpy(z['gt_boxes']), torch.from_numpy(z['gt_labels'])
    return ([gb[off[i]:off[i + 1]].clone() for i in range(len(off) - 1)],
            [gl[off[i]:off[i + 1]].clone() for i in range(len(off) - 1)])


def head_outputs(seed, B, P, C, sparse_images=(), extra=False, bg_bias=8.0, obj_bias=-6.0, yolo=False, loc_bias=0.0):
    """Seeded synthetic head outputs; the same recipe the golden generator used."""
    g = torch.Generator().manual_seed(seed)
    locs = torch.randn(B, P, 4, generator=g) + loc_bias
    confs = torch.randn(B, P, C, generator=g)
    ext = torch.randn(B, P, generator=g) if extra else None
    for b in sparse_images:
        if yolo:
            ext[b] += obj_bias
        else:
            confs[b, :, 0] += bg_bias
    return locs, confs, ext


def predict_inputs(z, P):
    """Rebuild (out_locs, out_confs, extra) of a predict_* fixture, checking the
    recorded input hashes when the inputs are seed-generated."""
    spec = json.loads(str(z['spec']))
    if 'out_locs' in z.files:
        return spec, torch.from_numpy(z['out_locs']), torch.from_numpy(z['out_confs']), None
    locs, confs, ext = head_outputs(spec['seed'], spec['B'], P, spec['C'], tuple(spec['sparse_images']),
                                    extra=spec['family'] in ('fcos',), bg_bias=spec['bg_bias'],
                                    loc_bias=spec.get('loc_bias', 0.0))
    hashes = [sha(locs), sha(confs)] + ([sha(ext)] if ext is not None else [])
    assert hashes == spec['inputs_sha256'], 'seeded inputs differ from the ones the fixture was generated with'
    return spec, locs, confs, ext


def split(t, counts):
    out, o = [], 0
    for c in counts:
        out.append(t[o:o + int(c)])
        o += int(c)
    return out


def rel_close(a: torch.Tensor, b: torch.Tensor, rtol=1e-5, atol=1e-6):
    """|a-b| <= atol + rtol*|b| elementwise, NaN == NaN."""
    a, b = a.double(), b.double()
    both_nan = a.isnan() & b.isnan()
    ok = ((a - b).abs() <= atol + rtol * b.abs()) | both_nan | (a == b)
    return bool(ok.all())


def box_close(a: torch.Tensor, b: torch.Tensor, rtol=1e-5):
    """Decoded boxes: |a-b| <= rtol * (largest |coordinate| of that box).  A coordinate such as
    `px - exp(l)` (FCOS) cancels, so its error is relative to the operands' magnitude -- which the
    box's largest coordinate bounds within a factor 2 -- not to the (possibly tiny) difference."""
    a, b = a.double(), b.double()
    scale = b.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
    ok = ((a - b).abs() <= rtol * scale) | (a == b) | (a.isnan() & b.isnan())
    return bool(ok.all())


def wire_from_rows(per_image, image_ids, image_sizes, header_rows):
    """Torch restatement of the wire message (include/detbox.h, detbox_pack_wire) for CPU tests:
    per_image = list of (class_ids [n] i64, boxes [n,4] f32, scores [n] f32) in keep order."""
    import torch
    B = len(per_image)
    counts = [int(p[0].numel()) for p in per_image]
    n = sum(counts)
    words = torch.zeros((header_rows + n) * 6, dtype=torch.int32)
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    words[0], words[1] = n, B
    words[2:2 + B + 1] = torch.tensor(off, dtype=torch.int32)
    for b in range(B):
        iid = int(image_ids[b])
        lo, hi = iid & 0xffffffff, (iid >> 32) & 0xffffffff
        words[2 + B + 1 + 2 * b] = lo - (1 << 32) if lo >= (1 << 31) else lo
        words[2 + B + 1 + 2 * b + 1] = hi - (1 << 32) if hi >= (1 << 31) else hi
        words[2 + 3 * B + 1 + 2 * b] = int(image_sizes[b][0])
        words[2 + 3 * B + 1 + 2 * b + 1] = int(image_sizes[b][1])
    wire = words.reshape(-1, 6)
    r = header_rows
    for cls, boxes, scores in per_image:
        k = cls.numel()
        wire[r:r + k, 0] = cls.to(torch.int32)
        wire[r:r + k, 1:5] = boxes.to(torch.float32).contiguous().view(torch.int32)
        wire[r:r + k, 5] = scores.to(torch.float32).contiguous().view(torch.int32)
        r += k
    return wire


def rows_from_detections(per_image, image_ids, image_sizes):
    """The result rows (ids i64 [n,2], vals f64 [n,5]) as src/utilities/predictor.py:60-73 builds them."""
    import torch
    ids, vals = [], []
    for b, (cls, boxes, scores) in enumerate(per_image):
        H, W = float(image_sizes[b][0]), float(image_sizes[b][1])
        bx = boxes.to(torch.float32)
        ids.append(torch.stack([torch.full_like(cls, int(image_ids[b])), cls], dim=1))
        vals.append(torch.stack([bx[:, 0].double() * W, bx[:, 1].double() * H, (bx[:, 2] - bx[:, 0]).double() * W,
                                 (bx[:, 3] - bx[:, 1]).double() * H, scores.to(torch.float32).double()], dim=1))
    return torch.cat(ids), torch.cat(vals)
