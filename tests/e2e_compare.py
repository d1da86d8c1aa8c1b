"""How far the shipped inference path may differ from the reference when both start from the same raw logits.

The HIP path fuses the activation (its own exp) and decodes with the GPU's exp; the reference's golden rows
were produced by torch's CPU soft-max and exp.  Both are correct fp32 evaluations that round differently, so a
score can land on the other side of `conf_thresh` or of the k-th score, two near-equal scores can swap places,
and an IoU can land on the other side of the NMS threshold.  Everything else must be identical.  This module
finds every difference between the two results and checks that each one sits on such a boundary:

  selection   a (class, anchor) selected by one side only must have a reference score within `tol` (relative)
              of conf_thresh or of the k-th largest reference score of its class;
  keep lists  compared as lists of (class, anchor).  At the first rank where they differ the two entries must
              be explained by (a) a selection difference of that very candidate, (b) a near-tie of the two
              scores, or (c) an entry kept by one side only whose IoU against a box that the OTHER side kept
              earlier in its class is within `iou_tol` of the threshold, whose score nearly ties with such a box
              that it overlaps above the threshold, or which overlaps, above the threshold, a candidate of its class
              that only one side selected.  What follows the first divergence of an image may cascade and is only
              counted.

Test infrastructure only (CPU, torch); nothing in the package imports it.
"""
import torch


def select_mask(scores, class_begin, conf_thresh, top_k):
    """[P,C] activated scores -> bool [P,Cs]: score > thr and descending rank < top_k, lower anchor first on ties
    (src/models/ssd300.py:370)."""
    sub = scores[:, class_begin:]
    P = sub.shape[0]
    order = sub.argsort(dim=0, descending=True, stable=True)
    rank = torch.empty_like(order)
    rank.scatter_(0, order, torch.arange(P).unsqueeze(1).expand_as(order))
    return (sub > conf_thresh) & (rank < top_k)


def keys_of(mask):
    """Candidate order of pre_predict (class ascending, anchor ascending) as keys class * P + anchor."""
    P = mask.shape[0]
    c, p = torch.nonzero(mask.t(), as_tuple=True)
    return c * P + p


def iou_one_to_many(box, boxes):
    """torchvision's box_iou arithmetic for one box against [n,4]."""
    area = (box[2] - box[0]) * (box[3] - box[1])
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    w = (torch.minimum(box[2], boxes[:, 2]) - torch.maximum(box[0], boxes[:, 0])).clamp(min=0)
    h = (torch.minimum(box[3], boxes[:, 3]) - torch.maximum(box[1], boxes[:, 1])).clamp(min=0)
    inter = w * h
    return inter / (area + areas - inter)


def compare_image(ref_scores, hip_scores, ref, hip, class_begin, conf_thresh, top_k, iou_thresh,
                  tol=1e-5, iou_tol=1e-4):
    """One image.  ref_scores / hip_scores: [P,C] activated scores of the two sides.
    ref / hip: dicts with 'boxes' [n,4], 'scores' [n], 'keep' [m] (indices into the candidate list, descending
    score) of each side.  Returns a report dict; report['unexplained'] lists what no boundary accounts for."""
    P = ref_scores.shape[0]
    mr = select_mask(ref_scores, class_begin, conf_thresh, top_k)
    mh = select_mask(hip_scores, class_begin, conf_thresh, top_k)
    kr, kh = keys_of(mr), keys_of(mh)
    rep = dict(candidates_ref=int(kr.numel()), candidates_hip=int(kh.numel()), selection_diff=0, keep_ref=int(ref['keep'].numel()),
               keep_hip=int(hip['keep'].numel()), keep_differing=0, first_divergence=None, unexplained=[])
    assert kr.numel() == ref['scores'].numel(), 'reference candidates are not what the reference scores select'
    assert kh.numel() == hip['scores'].numel(), 'HIP candidates are not what the HIP scores select'

    # ---- selection
    sub_r = ref_scores[:, class_begin:]
    diff_p, diff_c = torch.nonzero(mr != mh, as_tuple=True)
    rep['selection_diff'] = int(diff_p.numel())
    only_one_side = set()
    if diff_p.numel():
        kth = sub_r.topk(min(top_k, P), dim=0).values[-1]           # k-th largest reference score per class
        for p, c in zip(diff_p.tolist(), diff_c.tolist()):
            s = float(sub_r[p, c])
            near_thr = abs(s - conf_thresh) <= tol * abs(conf_thresh)
            near_kth = abs(s - float(kth[c])) <= tol * abs(float(kth[c]))
            only_one_side.add(c * P + p)
            if not (near_thr or near_kth):
                rep['unexplained'].append(dict(kind='selection', anchor=p, cls=c + class_begin, ref_score=s,
                                               hip_score=float(hip_scores[p, c + class_begin]), kth=float(kth[c])))

    # ---- keep lists, as (class, anchor) keys
    keep_r, keep_h = kr[ref['keep']], kh[hip['keep']]
    n = min(keep_r.numel(), keep_h.numel())
    d = torch.nonzero(keep_r[:n] != keep_h[:n]).flatten()
    rep['keep_differing'] = int(d.numel()) + abs(keep_r.numel() - keep_h.numel())
    if d.numel() == 0 and keep_r.numel() == keep_h.numel():
        return rep
    r = int(d[0]) if d.numel() else n
    a = int(keep_h[r]) if r < keep_h.numel() else None
    b = int(keep_r[r]) if r < keep_r.numel() else None
    rep['first_divergence'] = dict(rank=r, hip=a, ref=b)

    def info(key):          # box and reference-side score of a candidate, from whichever side has it
        i = torch.nonzero(kr == key).flatten()
        if i.numel():
            return ref['boxes'][int(i[0])], float(ref['scores'][int(i[0])])
        i = torch.nonzero(kh == key).flatten()
        return hip['boxes'][int(i[0])], float(hip['scores'][int(i[0])])

    set_r, set_h = set(keep_r.tolist()), set(keep_h.tolist())

    def explained(key, other):
        if key is None:
            return 'end of list'
        if key in only_one_side:
            return 'selection difference'
        box, s = info(key)
        if other is not None and key in set_r and key in set_h and other in set_r and other in set_h:
            so = info(other)[1]
            if abs(s - so) <= tol * max(abs(s), abs(so)):
                return 'near-tie of two kept scores'
        # Kept by one side only.  On the side that dropped it, the decisive IoU is the one against a box of its class that
        # this side kept EARLIER (score at least as high): that IoU must sit on the threshold, or the two scores must
        # nearly tie (then the other side processed them in the other order) with the boxes overlapping above it.
        c = key // P
        for keys, kept, side in ((kr, set_r, ref), (kh, set_h, hip)):
            if key in kept:
                continue
            same = torch.nonzero((keys // P) == c).flatten()
            earlier = [int(i) for i in same.tolist() if int(keys[i]) in kept and int(keys[i]) != key]
            if not earlier:
                continue
            earlier = torch.tensor(earlier)
            sc = side['scores'][earlier]
            iou = iou_one_to_many(box, side['boxes'][earlier])
            near = (sc - s).abs() <= tol * max(abs(s), 1e-30)
            if bool((((iou - iou_thresh).abs() <= iou_tol) & ((sc >= s) | near)).any()):
                return 'IoU within %g of the threshold against a box kept earlier' % iou_tol
            if bool((near & (iou > iou_thresh - iou_tol)).any()):
                return 'near-tie with an overlapping kept box'
        # a candidate that only one side selected, in this class and overlapping this box above the threshold
        for k in only_one_side:
            if k // P == c and k != key:
                if float(iou_one_to_many(box, info(k)[0][None, :])[0]) > iou_thresh - iou_tol:
                    return 'selection difference of an overlapping box'
        return None

    ea, eb = explained(a, b), explained(b, a)
    rep['first_divergence'].update(hip_why=ea, ref_why=eb)
    # one explanation accounts for the divergence: the other entry is simply what comes next in that list
    if ea is None and eb is None:
        rep['unexplained'].append(dict(kind='keep', rank=r, hip=a, ref=b))
    return rep


def summarise(reports):
    tot = dict(images=len(reports), candidates=0, selection_diff=0, kept=0, images_with_keep_diff=0, keep_differing=0, unexplained=0)
    for r in reports:
        tot['candidates'] += r['candidates_ref']
        tot['selection_diff'] += r['selection_diff']
        tot['kept'] += r['keep_ref']
        tot['images_with_keep_diff'] += 1 if r['keep_differing'] else 0
        tot['keep_differing'] += r['keep_differing']
        tot['unexplained'] += len(r['unexplained'])
    return tot
