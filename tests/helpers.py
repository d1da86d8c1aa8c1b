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
                                    extra=spec['family'] in ('fcos', 'yolov3'), bg_bias=spec['bg_bias'],
                                    obj_bias=spec.get('obj_bias', -6.0), yolo=spec['family'] == 'yolov3',
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


def big_predict_reference(O, z, anchors, strict=False):
    """Full-size predict_* fixtures hold hashes, a subsample and the keep lists as one bit per candidate.  The
    reference's full arrays are rebuilt here by the CPU oracle from the seeded inputs.  On the machine that made
    the fixtures (strict=True, the CPU suite) every hash must be reproduced bit for bit, which pins the oracle at
    full size.  On another CPU torch's exp / soft-max may round differently (another vector ISA), so there the
    integer results must match exactly and the float arrays agree with the stored subsample to 1e-6.  Returns
    (spec, inputs (locs, confs, ext), lists (boxes, scores, class ids per image), keeps per image, rows_ids, rows_vals)."""
    import hashlib
    spec, locs, confs, ext = predict_inputs(z, anchors.shape[0])
    thr, k = spec['conf_thresh'], spec['top_k']
    if spec['family'] == 'fcos':
        pb, ps, pc = O.pre_predict_fcos(locs, confs, ext, anchors, thr, k, fast=True)
    else:
        pb, ps, pc = O.pre_predict_softmax(locs, confs, anchors, thr, k, fast=True)
    counts = z['counts'].tolist()
    assert [t.shape[0] for t in pb] == counts
    ab, asc, ac = torch.cat(pb), torch.cat(ps), torch.cat(pc)
    assert sha(ac) == str(z['class_ids_sha256'])
    exact = sha(ab) == str(z['bboxes_sha256']) and sha(asc) == str(z['scores_sha256'])
    assert exact or not strict
    assert rel_close(asc[::16], torch.from_numpy(z['sub_scores']), 1e-6, atol=0.0)
    assert box_close(ab[::16], torch.from_numpy(z['sub_bboxes']), 1e-6)
    bits = torch.from_numpy(np.unpackbits(z['keep_bits'])[:sum(counts)].astype(bool))
    keeps = []
    for m, s_ in zip(split(bits, counts), ps):
        members = torch.nonzero(m).flatten()
        keeps.append(members[torch.argsort(s_[members], descending=True, stable=True)])
    assert [k_.numel() for k_ in keeps] == z['keep_counts'].tolist()
    metas = json.loads(str(z['metas']))
    ids, vals = [], []
    for meta, b, s_, c, kp in zip(metas, pb, ps, pc, keeps):       # the rows of src/utilities/predictor.py:56-73 for these keep lists
        bb, ss, cc = b[kp], s_[kp], c[kp]
        H, W = meta['height'], meta['width']
        ids.append(torch.stack([torch.full_like(cc, meta['image_id']), cc], dim=1))
        vals.append(torch.stack([bb[:, 0].double() * W, bb[:, 1].double() * H, (bb[:, 2] - bb[:, 0]).double() * W,
                                 (bb[:, 3] - bb[:, 1]).double() * H, ss.double()], dim=1))
    rid, rv = torch.cat(ids), torch.cat(vals)
    if exact:
        assert hashlib.sha256(rid.numpy().tobytes()).hexdigest() == str(z['rows_ids_sha256'])
        assert hashlib.sha256(rv.numpy().tobytes()).hexdigest() == str(z['rows_vals_sha256'])
        assert np.array_equal(rid[::16].numpy(), z['sub_rows_ids']) and np.array_equal(rv[::16].numpy(), z['sub_rows_vals'])
    return spec, (locs, confs, ext), (pb, ps, pc), keeps, rid, rv
