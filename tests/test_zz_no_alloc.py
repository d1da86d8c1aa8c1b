"""detbox_assign_anchor_noalloc (the entry point that allocates nothing: its per-GT slots live in an array inside
the library) against detbox_assign_anchor, through the C ABI: two calls in a row, the second on a smaller batch,
so that whatever the first call left in the shared array must not leak into the second."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gt(n_images, seed, num_fg=20):
    g = torch.Generator().manual_seed(seed)
    gb, gl = [], []
    for i in range(n_images):
        G = 1 + (i * 7 + seed) % 23
        c = 0.15 + 0.70 * torch.rand(G, 2, generator=g)
        wh = 0.04 + 0.46 * torch.rand(G, 2, generator=g)
        gb.append(torch.cat([c, wh], 1))
        gl.append(torch.randint(1, num_fg + 1, (G,), generator=g))
    return gb, gl


def _both(D, dev, anchors, gb, gl, mode, lo, hi, first=False):
    ops, L = D.ops, D.ops._lib.lib()
    gbp, glp, off, total = ops.pack_gt(gb, gl, dev)
    want = ops.assign_anchor(anchors, gbp, glp, off, total, mode, lo, hi, want_best=True)
    B, P = len(gb), anchors.shape[0]
    locs = torch.full((B, P, 4), float('nan'), device=dev)
    labels = torch.full((B, P), -7, dtype=torch.int64, device=dev)
    matched = torch.full((B, P), -7, dtype=torch.int64, device=dev)
    best = torch.full((max(total, 1),), -7, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(dev)['allocation.all.allocated']
    free_before = torch.cuda.mem_get_info(dev)[0]
    rc = L.detbox_assign_anchor_noalloc(mode, anchors.data_ptr(), P, gbp.data_ptr(), glp.data_ptr(), off.data_ptr(), B, total,
                                        lo, hi, locs.data_ptr(), labels.data_ptr(), matched.data_ptr(), best.data_ptr(),
                                        torch.cuda.current_stream(dev).cuda_stream)
    D.ops._lib.check(rc, 'detbox_assign_anchor_noalloc')
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(dev)['allocation.all.allocated'] == before
    if not first:
        assert torch.cuda.mem_get_info(dev)[0] == free_before       # no device memory was taken or given back
    wl, wlab, wm, wbest = want
    assert torch.equal(labels, wlab) and torch.equal(matched, wm) and torch.equal(best[:total], wbest)
    assert torch.equal(locs, wl)                 # the same kernels on the same inputs: identical bits
    # the workspace entry point with the small workspace (one slot per GT box: the atomic form of the per-GT arg-max; `want`
    # above came from the larger one, where every workgroup stores its own word and nothing is cleared)
    small = L.detbox_assign_anchor_workspace_bytes(total)
    assert small <= L.detbox_assign_anchor_workspace_bytes_for(total, P, B)
    ws = torch.full((small,), 0xA5, dtype=torch.uint8, device=dev)
    locs2, labels2, matched2, best2 = torch.empty_like(locs), torch.empty_like(labels), torch.empty_like(matched), torch.empty_like(best)
    rc = L.detbox_assign_anchor(mode, anchors.data_ptr(), P, gbp.data_ptr(), glp.data_ptr(), off.data_ptr(), B, total, lo, hi,
                                locs2.data_ptr(), labels2.data_ptr(), matched2.data_ptr(), best2.data_ptr(), ws.data_ptr(), small,
                                torch.cuda.current_stream(dev).cuda_stream)
    D.ops._lib.check(rc, 'detbox_assign_anchor')
    torch.cuda.synchronize()
    assert torch.equal(labels2, wlab) and torch.equal(matched2, wm) and torch.equal(best2[:total], wbest) and torch.equal(locs2, wl)
    return total


def test_noalloc_assign_matches_workspace_entry_point():
    import object_detection_amd as D
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    L = D.ops._lib.lib()
    a300 = D.anchors.ssd300_default_boxes().to(dev)
    gb, gl = _gt(12, 3)
    n1 = _both(D, dev, a300, gb, gl, D.ops.ASSIGN_CXCYWH, 0.5, 0.5, first=True)
    gb2, gl2 = _gt(3, 11)                          # smaller batch, fewer GT slots: stale slots beyond it stay untouched
    n2 = _both(D, dev, a300, gb2, gl2, D.ops.ASSIGN_CXCYWH, 0.4, 0.5)
    assert n2 < n1
    yolo = D.anchors.yolov3_prior_boxes().to(dev)
    gby = [torch.cat([b[:, :2] - b[:, 2:] / 2, b[:, 2:]], 1) for b in gb2]
    _both(D, dev, yolo, gby, gl2, D.ops.ASSIGN_YOLO, 0.5, 1.0)
    # an image without ground truth and an empty batch
    _both(D, dev, a300, [gb[0], torch.zeros(0, 4), gb[1]], [gl[0], torch.zeros(0, dtype=torch.int64), gl[1]], D.ops.ASSIGN_CXCYWH, 0.5, 0.5)
    # beyond the built-in array: a clear error, nothing launched
    gbp, glp, off, total = D.ops.pack_gt(gb, gl, dev)
    out = torch.empty((12, 8732, 4), device=dev)
    lab = torch.empty((12, 8732), dtype=torch.int64, device=dev)
    rc = L.detbox_assign_anchor_noalloc(0, a300.data_ptr(), 8732, gbp.data_ptr(), glp.data_ptr(), off.data_ptr(), 12, 65537, 0.5, 0.5,
                                        out.data_ptr(), lab.data_ptr(), lab.data_ptr(), None, None)
    assert rc == -4 and b'65537' in L.detbox_last_error()


def test_noalloc_assign_replays_from_a_captured_graph():
    """The callers this entry point is for: a HIP graph.  The first call on a device looks the scratch array's address
    up (hipGetSymbolAddress), which is set-up work and not a stream operation, so one eager call per device comes
    first; after it the entry point is made of a memset and kernel launches only and can be captured and replayed."""
    import object_detection_amd as D
    dev = torch.device('cuda', 0)
    ops, L = D.ops, D.ops._lib.lib()
    a300 = D.anchors.ssd300_default_boxes().to(dev)
    gb, gl = _gt(5, 21)
    _both(D, dev, a300, gb, gl, ops.ASSIGN_CXCYWH, 0.5, 0.5, first=True)        # the eager warm-up call (and a check)
    gbp, glp, off, total = ops.pack_gt(gb, gl, dev)
    B, P = len(gb), a300.shape[0]
    locs = torch.full((B, P, 4), float('nan'), device=dev)
    labels = torch.full((B, P), -7, dtype=torch.int64, device=dev)
    matched = torch.full((B, P), -7, dtype=torch.int64, device=dev)
    best = torch.full((total,), -7, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        rc = L.detbox_assign_anchor_noalloc(ops.ASSIGN_CXCYWH, a300.data_ptr(), P, gbp.data_ptr(), glp.data_ptr(), off.data_ptr(),
                                            B, total, 0.5, 0.5, locs.data_ptr(), labels.data_ptr(), matched.data_ptr(),
                                            best.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    D.ops._lib.check(rc, 'detbox_assign_anchor_noalloc under capture')
    assert torch.isnan(locs).all() and (labels == -7).all()        # captured, not run
    for rep in range(2):
        if rep:                                   # new ground truth in the same buffers: the replay must follow it
            gb2, gl2 = _gt(5, 21 + rep)
            g2, l2, o2, t2 = ops.pack_gt(gb2, gl2, dev)
            if t2 != total:
                continue
            gbp.copy_(g2); glp.copy_(l2); off.copy_(o2)
        graph.replay()
        torch.cuda.synchronize()
        wl, wlab, wm, wbest = ops.assign_anchor(a300, gbp, glp, off, total, ops.ASSIGN_CXCYWH, 0.5, 0.5, want_best=True)
        assert torch.equal(labels, wlab) and torch.equal(matched, wm) and torch.equal(best, wbest) and torch.equal(locs, wl)
