#!/usr/bin/env python3
"""Benchmark of the detection box-ops hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--regime dense|sparse|trained]

With --gpus N > 1 and no RANK / WORLD_SIZE in the environment the script starts its N ranks itself (one process
per GPU, rendezvous on 127.0.0.1) and relays rank 0's line; under torch.distributed.run it is simply one rank.

One "step" = one pass of the whole path over one batch of synthetic head outputs that already sit
in HBM: target assignment (train side) + activation + threshold/top-k/decode + per-class NMS +
result rows (inference side); with N > 1 ranks also the all-gather of the result rows.  Each rank
owns its own images (weak scaling, image-sharded); `value` = images of all ranks / wall time.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel of the step with HIP events
recorded on its launch stream during the timed steps; `cpu_baseline` times the CPU oracle (the
loop-faithful restatement of the reference path) on a small sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import object_detection_amd as D  # noqa: E402
from object_detection_amd import ops  # noqa: E402
from object_detection_amd.dist import RowExchange  # noqa: E402
from object_detection_amd.pipeline import StreamRing  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, spec

WORKLOADS = {
    # name: (family, ctor, B, C_stored, top_k, coco_like, gt_fmt, gt_scale)
    'retinanet800': dict(family='retina', B=16, C=81, top_k=1000, coco=True, fmt='cxcywh', scale=1.0,
                         make=lambda: D.RetinaNet(80, [100, 50, 25, 13, 7])),
    'retinanet800_b512': dict(family='retina', B=512, C=81, top_k=1000, coco=True, fmt='cxcywh', scale=1.0,
                              make=lambda: D.RetinaNet(80, [100, 50, 25, 13, 7])),
    'retinanet800_b128': dict(family='retina', B=128, C=81, top_k=1000, coco=True, fmt='cxcywh', scale=1.0,
                              make=lambda: D.RetinaNet(80, [100, 50, 25, 13, 7])),
    'ssd300': dict(family='ssd', B=8, C=21, top_k=200, coco=False, fmt='cxcywh', scale=1.0, make=lambda: D.SSD300(20)),
    'ssd300_b512': dict(family='ssd', B=512, C=21, top_k=200, coco=False, fmt='cxcywh', scale=1.0, make=lambda: D.SSD300(20)),
    'fcos800': dict(family='fcos', B=16, C=81, top_k=200, coco=True, fmt='cxcywh', scale=512.0,
                    make=lambda: D.FCOS(80, [100, 50, 25, 13, 7])),
    'yolov3': dict(family='yolo', B=64, C=20, top_k=200, coco=False, fmt='xywh', scale=1.0, make=lambda: D.YoloV3(20)),
}


def synth_gt(i, num_fg, coco_like, fmt, scale):
    """Same recipe as the oracle's synth_gt (GT seed 1000+i); duplicated here so that the timed
    product path imports nothing from oracle/."""
    g = torch.Generator().manual_seed(1000 + i)
    G = 1 + (i % (64 if coco_like else 16))
    c = 0.15 + 0.70 * torch.rand(G, 2, generator=g)
    wh = 0.04 + 0.46 * torch.rand(G, 2, generator=g)
    boxes = torch.cat([c, wh], dim=1) if fmt != 'xywh' else torch.cat([c - wh / 2, wh], dim=1)
    labels = torch.randint(1, num_fg + 1, (G,), generator=g)
    return boxes * scale, labels


def synth_head(i, P, C, family, regime):
    """Head outputs of image i (seed 2000+i): locs ~ N(0,1), logits ~ N(0,1) (dense), with the background
    column +8 / objectness -6 (sparse), or with a background column that dominates (trained)."""
    g = torch.Generator().manual_seed(2000 + i)
    locs = torch.randn(P, 4, generator=g)
    confs = torch.randn(P, C, generator=g)
    ext = torch.randn(P, generator=g) if family in ('fcos', 'yolo') else None
    if regime == 'sparse':
        if family == 'yolo':
            ext -= 6.0
        else:
            confs[:, 0] += 8.0
    elif regime == 'trained':
        # what a trained model looks like to this path: nearly every anchor is confidently background, a few
        # hundred scores per image pass the threshold, so the threshold binds (not top-k) and NMS has little to do
        if family == 'yolo':
            ext -= 9.0
        else:
            confs[:, 0] += (10.5 if P > 50000 else 9.0)
    return locs, confs, ext


class Workload:
    def __init__(self, name, regime, rank, dev, exact=False):
        w = WORKLOADS[name]
        self.name, self.regime, self.w, self.dev = name, regime, w, dev
        self.exchange = None
        self.exchange_root = 0          # the evaluator is one process (rank 0), as in the reference's test loop
        self.exchange_payload = 'rows'
        self.model = w['make']()
        m = self.model
        # exact: torch's own activation on the device, then the selection on those very bits (the mode whose candidate
        # sets are bit-exact against a torch build); default: the activation fused into the selection pass
        self.exact = exact
        m.fused_activation = not exact
        self.anchors_cpu = m._anchors_for_decode()
        self.P = self.anchors_cpu.shape[0]
        self.B, self.C, self.top_k = w['B'], w['C'], w['top_k']
        self.images = list(range(rank * self.B, (rank + 1) * self.B))
        num_fg = self.C - 1 if w['family'] != 'yolo' else self.C
        gts = [synth_gt(i, num_fg, w['coco'], w['fmt'], w['scale']) for i in self.images]
        self.gt_b, self.gt_l = [g[0] for g in gts], [g[1] for g in gts]
        self.gtb, self.gtl, self.goff, self.total_gt = ops.pack_gt(self.gt_b, self.gt_l, dev)
        locs, confs, ext = [], [], []
        for i in self.images:
            l, c, e = synth_head(i, self.P, self.C, w['family'], regime)
            locs.append(l); confs.append(c); ext.append(e)
        self.locs_cpu, self.confs_cpu = torch.stack(locs), torch.stack(confs)
        self.ext_cpu = torch.stack(ext) if ext[0] is not None else None
        self.locs, self.confs = self.locs_cpu.to(dev), self.confs_cpu.to(dev)
        self.ext = self.ext_cpu.to(dev) if self.ext_cpu is not None else None
        self.metas = [{'image_id': i, 'height': 480, 'width': 640} for i in self.images]
        self.ids = torch.tensor([m_['image_id'] for m_ in self.metas], dtype=torch.int64, device=dev)
        self.hw = torch.tensor([[m_['height'], m_['width']] for m_ in self.metas], dtype=torch.int32, device=dev)
        self.anchors = m._table(self.anchors_cpu, dev)
        if w['family'] == 'fcos':
            self.ranges = m._table(m.regress_ranges, dev)

    def outputs(self):
        f = self.w['family']
        if f == 'fcos':
            return (self.locs, self.confs, self.ext)
        if f == 'yolo':
            return (self.locs.clone(), self.ext, self.confs)     # pre_predict overwrites locs[..., :2]
        return (self.locs, self.confs)

    def assign(self):
        f = self.w['family']
        if f == 'ssd':
            return ops.assign_anchor(self.anchors, self.gtb, self.gtl, self.goff, self.total_gt, ops.ASSIGN_CXCYWH, 0.5, 0.5)
        if f == 'retina':
            return ops.assign_anchor(self.anchors, self.gtb, self.gtl, self.goff, self.total_gt, ops.ASSIGN_CXCYWH, 0.4, 0.5)
        if f == 'yolo':
            return ops.assign_anchor(self.anchors, self.gtb, self.gtl, self.goff, self.total_gt, ops.ASSIGN_YOLO, 0.5, 1.0)
        return ops.assign_fcos(self.anchors, self.ranges, self.gtb, self.gtl, self.goff, self.total_gt)

    def step(self, world):
        """The whole path once.  Returns device tensors; no host synchronisation for world == 1."""
        extra = os.environ.get('DETBOX_BENCH_TWICE', '').split(',')     # measurement aid: marginal cost of one stage
        targets = self.assign()
        if 'assign' in extra:
            targets = self.assign()
        m = self.model
        c = m.candidates(self.outputs(), 0.01, self.top_k)      # activation fused into the selection pass
        if 'select' in extra:
            c = m.candidates(self.outputs(), 0.01, self.top_k)
        # per-class NMS + order step + result rows: one operator (the order step writes the rows itself)
        if world == 1:
            keep, kc, rows_ids, rows_vals, row_off = ops.nms_rows(c, 0.45, self.ids, self.hw)
            if 'nms' in extra:
                keep, kc, rows_ids, rows_vals, row_off = ops.nms_rows(c, 0.45, self.ids, self.hw)
        else:
            # the exchange step: this rank's detections as a compact message; the gather of the previous
            # batch's messages starts now, on a side stream, and overlaps this batch's kernels
            if self.exchange is None:
                self.exchange = RowExchange(keep_results=False, root=self.exchange_root, payload=self.exchange_payload,
                                            timing=True)
                self.hdr_rows = ops.wire_header_rows(self.B)
            keep, kc, rows_ids, rows_vals, row_off, wire = ops.nms_rows(c, 0.45, self.ids, self.hw, wire_images_cap=self.B)
            self.exchange.submit(wire, self.B, self.hdr_rows)
        return targets, c, kc, rows_ids, rows_vals, row_off


def cpu_baseline(wl, n_img):
    """The CPU oracle (loop-faithful restatement of the reference: per-image Python loop, per-class
    double argsort, greedy CPU NMS, per-detection dict loop) on the first n_img images of rank 0."""
    from oracle import boxops_oracle as O
    f = wl.w['family']
    gb, gl = wl.gt_b[:n_img], wl.gt_l[:n_img]
    locs, confs = wl.locs_cpu[:n_img].clone(), wl.confs_cpu[:n_img].clone()
    ext = wl.ext_cpu[:n_img].clone() if wl.ext_cpu is not None else None
    a = wl.anchors_cpu
    t0 = time.perf_counter()
    if f == 'ssd':
        O.assign_ssd(a, gb, gl)
        pb, ps, pc = O.pre_predict_softmax(locs, confs, a, 0.01, wl.top_k)
    elif f == 'retina':
        O.assign_retinanet(a, gb, gl)
        pb, ps, pc = O.pre_predict_softmax(locs, confs, a, 0.01, wl.top_k)
    elif f == 'fcos':
        O.assign_fcos(a, wl.model.regress_ranges, gb, gl)
        pb, ps, pc = O.pre_predict_fcos(locs, confs, ext, a, 0.01, wl.top_k)
    else:
        O.assign_yolov3(a, gb, gl)
        pb, ps, pc = O.pre_predict_yolov3(locs, ext, confs, a, 0.01, wl.top_k)
    res = O.predictor_run(wl.metas[:n_img], pb, ps, pc, 0.45)
    dt = time.perf_counter() - t0
    return n_img / dt, dt, len(res)


def counters_current(entry_sha):
    """True when a committed counter pass was taken on the kernel sources this run was built from."""
    from object_detection_amd.build import source_hash
    return entry_sha is not None and entry_sha == source_hash()


def measured_traffic(workload, regime, kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes (profiles/traffic.json, written
    by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command), or None -- also when the
    pass was taken on other kernel sources than the ones built now (stale counters never reach the line)."""
    try:
        with open(os.path.join(REPO, 'profiles', 'traffic.json')) as f:
            t = json.load(f)[f'{workload}/{regime}']
        if not counters_current(t.get('_source_sha256')):
            return None
        return t[kernel]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


KERNELS = ['assign_anchor', 'assign_fixup', 'assign_fcos', 'select_sample', 'select_plan', 'select_scan',
           'select_check', 'select_fallback', 'score_hist', 'select_kth',
           'select_refine', 'select_emit', 'select_finalize', 'select_direct_stats', 'select_direct', 'nms_segments', 'nms_scan', 'nms_merge',
           'nms_split', 'nms_tile', 'pack_detections']
PRICED = ('select_scan', 'score_hist', 'select_emit', 'assign_anchor', 'assign_fcos', 'nms_segments', 'select_finalize',
          'select_direct_stats', 'select_direct')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--blocks', type=int, default=7,
                    help='the timed region of exactly --steps steps is repeated this many times; value is the median block')
    ap.add_argument('--workload', default='retinanet800', choices=sorted(WORKLOADS))
    ap.add_argument('--regime', default='dense', choices=['dense', 'sparse', 'trained'])
    ap.add_argument('--cpu-images', type=int, default=-1, help='images in the CPU baseline sample (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle timing (profiling runs)')
    ap.add_argument('--no-extra', action='store_true', help='skip the `workloads` object (the other headline configurations)')
    ap.add_argument('--graph', action='store_true',
                    help='N=1 only: capture one step per stream in a HIP graph and replay it in the timed region (launch-bound workloads)')
    ap.add_argument('--streams', type=int, default=3,
                    help='steps are issued round-robin on this many HIP streams (independent batches in flight)')
    ap.add_argument('--exchange-payload', default='rows', choices=['rows', 'header'],
                    help='N>1: what every rank ships: its detections, or only the per-image row counts (a consumer that needs no boxes)')
    ap.add_argument('--exchange', default='all', choices=['all', 'root'],
                    help='N>1: every rank receives the detections of all ranks (all-gather), or only rank 0 does (gather)')
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """--gpus N without a launcher: this process starts the N ranks (fresh interpreters of this script, one per
    GPU), relays rank 0's JSON line and fails if any rank fails.  Nothing here touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f'[bench] ranks failed (rank, exit code): {bad}', file=sys.stderr)
        return 1
    return 0


def measure(wl, args, world, dist, steps, warmup, blocks, streams_n, use_graph):
    """Warm-up, an untimed survey pass with every kernel bracketed by events, then `blocks` timed regions of
    exactly `steps` steps, each bracketed by barrier + synchronize, with events on the dominant kernel only."""
    dev, B = wl.dev, wl.B

    def barrier():
        if world > 1:
            if wl.exchange is not None:
                wl.exchange.finish()            # every batch's gather is complete before the clock is read
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(max(warmup, 1)):
        out = wl.step(world)
    torch.cuda.synchronize()
    _, c, kc, _, _, _ = out
    n_cand = float(c.counts.sum().item()) / B
    n_keep = float(kc.clamp(min=0).sum().item()) / B
    if bool((kc < 0).any().item()):
        raise RuntimeError('NMS reported an error flag')

    ops.profile_reset()
    ops.profile_only('')
    ops.profile_enable(True)
    for _ in range(5):
        wl.step(world)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    kern = {}
    for nm in KERNELS:
        ms, n = ops.profile_read(nm)
        if n:
            kern[nm] = dict(total_ms=ms, launches=n, avg_us=1e3 * ms / n)
    dom = max((k for k in kern if k in PRICED), key=lambda k: kern[k]['total_ms'])

    ops.profile_reset()
    ops.profile_only(dom)
    ops.profile_enable(True)
    graph = None
    streams = StreamRing(dev, streams_n) if streams_n > 1 else None
    if use_graph and world == 1:
        # every launch of a step goes into one HIP graph per stream (buffers come from the capture's private
        # pool, workspaces are per stream); a replay runs exactly the same kernels on the same inputs
        ops.profile_enable(False)
        graph = []
        for st in (streams.streams if streams is not None else [torch.cuda.Stream(dev)]):
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                wl.step(world)
            st.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=st):
                graph_out = wl.step(world)
            graph.append((g_, st, graph_out))
        for g_, st, _ in graph:
            with torch.cuda.stream(st):
                g_.replay()
        torch.cuda.synchronize()
    elif streams is not None:
        for _ in range(len(streams)):                 # workspaces and allocator pools of every stream exist before the clock starts
            with streams.next():
                wl.step(world)
        torch.cuda.synchronize()
    times = []
    for _ in range(max(blocks, 1)):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if graph is not None:
                g_, st, _ = graph[i % len(graph)]
                with torch.cuda.stream(st):
                    g_.replay()
            elif streams is not None:
                with streams.next():
                    wl.step(world)
            else:
                wl.step(world)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        times.append(elapsed)
    ops.profile_enable(False)
    ops.profile_only('')
    ms, n = ops.profile_read(dom)
    if n == 0:                                   # graph replay: no events inside; the survey pass stands in
        ms, n = kern[dom]['total_ms'], kern[dom]['launches']
    launch = (f'hip-graph replay on {len(graph)} stream(s)' if graph is not None
              else ('stream' if streams is None else f'{len(streams)} streams, steps round-robin'))
    return dict(times=times, kern=kern, dom=dom, dom_us=1e3 * ms / max(n, 1), dom_launches=n, n_cand=n_cand, n_keep=n_keep,
                launch=launch, streams=(1 if streams is None else len(streams)))


def price(wl, m, world, steps):
    """Throughput and roofline figures of one measured workload."""
    B, P, C = wl.B, wl.P, wl.C
    n_cand, n_keep, kern, dom = m['n_cand'], m['n_keep'], m['kern'], m['dom']
    G_mean = wl.total_gt / B
    score_bytes = 4.0 * B * P * C
    alg_bytes = {      # algorithmic bytes per launch (whole batch of this rank)
        'select_scan': score_bytes, 'score_hist': score_bytes, 'select_emit': score_bytes,
        'assign_anchor': B * (48.0 * P + 24.0 * G_mean), 'assign_fcos': B * (52.0 * P + 24.0 * G_mean),
        'nms_segments': B * (28.0 * n_cand + 8.0 * n_keep),
        'select_finalize': B * (28.0 * n_cand + 32.0 * n_cand),
        'select_direct_stats': score_bytes, 'select_direct': score_bytes + B * (28.0 * n_cand + 32.0 * n_cand),
    }
    achieved = alg_bytes[dom] / (m['dom_us'] * 1e-6) / 1e9
    roofline = dict(kernel=dom, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(wl.name, wl.regime, dom), avg_us=m['dom_us'],
                    launches=m['dom_launches'], algorithmic_bytes=alg_bytes[dom],
                    avg_us_one_batch_in_flight=kern[dom]['avg_us'],
                    note='avg_us is measured in the timed region, where the kernel shares the CUs with the other '
                         'batch in flight; avg_us_one_batch_in_flight comes from the single-stream survey pass')
    if dom == 'nms_segments':
        # greedy NMS is O(n * kept) pair tests on data that stays on chip: VALU issue, not HBM, is what it can
        # saturate.  valu_frac = VALU wave-instructions per second (committed counter pass of this command, profiles/,
        # over the duration measured here) / the chip's measured issue peak (tools/bw/exp_valu.hip).
        v = measured_valu(wl.name, wl.regime)
        if v is not None:
            roofline.update(valu_fields(v, m['dom_us']))
    streaming = None
    if 'select_scan' in kern:
        a = score_bytes / (kern['select_scan']['avg_us'] * 1e-6) / 1e9
        streaming = dict(kernel='select_scan', bound='hbm', achieved=a, peak=HBM_PEAK_GBS, unit='GB/s',
                         frac=a / HBM_PEAK_GBS, traffic=measured_traffic(wl.name, wl.regime, 'select_scan'),
                         avg_us=kern['select_scan']['avg_us'], algorithmic_bytes=score_bytes,
                         note='survey pass: every kernel bracketed by events')
    # the greedy NMS next to it, priced the same way whichever of the two kernels happened to be the longer one in
    # the survey pass (they are within a few percent of each other on the dense workload)
    pairs = None
    if 'nms_segments' in kern:
        us = kern['nms_segments']['avg_us']
        a = alg_bytes['nms_segments'] / (us * 1e-6) / 1e9
        pairs = dict(kernel='nms_segments', bound='valu', achieved=a, peak=HBM_PEAK_GBS, unit='GB/s', frac=a / HBM_PEAK_GBS,
                     traffic=measured_traffic(wl.name, wl.regime, 'nms_segments'), avg_us=us,
                     algorithmic_bytes=alg_bytes['nms_segments'],
                     note='survey pass; O(n * kept) pair tests on data that stays on chip: HBM frac is reported for '
                          'completeness, valu_frac (VALU wave-instructions per second / measured issue peak of the chip, '
                          'committed counter pass and microbenchmark) is the yardstick')
        v = measured_valu(wl.name, wl.regime)
        if v is not None:
            pairs.update(valu_fields(v, us))
    anchor_bytes = {'fcos': 8, 'yolo': 20}.get(wl.w['family'], 16)
    bytes_img = ((52.0 if wl.w['family'] == 'fcos' else 48.0) * P + 24.0 * G_mean
                 + 4.0 * P * C + 16.0 * P + anchor_bytes * P + (4.0 * P if wl.ext is not None else 0.0) + 28.0 * n_cand
                 + 28.0 * n_cand + 8.0 * n_keep)
    ts = sorted(m['times'])
    med = ts[len(ts) // 2]
    ips = world * B * steps / med
    return dict(ips=ips, ms_per_step=1e3 * med / steps, roofline=roofline, streaming=streaming, pairs=pairs, bytes_img=bytes_img,
                hbm_frac=bytes_img * ips / world / (HBM_PEAK_GBS * 1e9),
                blocks=dict(n=len(ts), steps_each=steps, ms_per_step_min=1e3 * ts[0] / steps, ms_per_step_median=1e3 * med / steps,
                            ms_per_step_max=1e3 * ts[-1] / steps),
                sub30_us_total=sum(v['avg_us'] for v in kern.values() if v['avg_us'] < 30.0),
                launches_per_step=sum(v['launches'] for v in kern.values()) / 5.0)


def measured_valu(workload, regime):
    """The committed SQ counter pass of the NMS kernel (profiles/valu.json), or None when it is missing or was taken on
    other kernel sources than the ones built now."""
    try:
        with open(os.path.join(REPO, 'profiles', 'valu.json')) as f:
            v = json.load(f)[f'{workload}/{regime}']['nms_segments']
        return v if counters_current(v.get('source_sha256')) else None
    except (OSError, KeyError, ValueError):
        return None


def valu_peaks():
    """Measured VALU issue peaks of the chip in wave-instructions per second (profiles/valu_peak.json, from
    tools/bw/exp_valu.hip): plain 32-bit instructions, and the half-rate ones the NMS pair filter is made of."""
    try:
        with open(os.path.join(REPO, 'profiles', 'valu_peak.json')) as f:
            p = json.load(f)
        return float(p['plain_wave_insts_per_s']), float(p['half_rate_wave_insts_per_s'])
    except (OSError, KeyError, ValueError):
        return None


def valu_fields(v, us):
    """VALU issue share of a kernel that ran `us` microseconds: its wave-instructions per second against the measured
    peak for plain instructions (a lower bound of the share: every instruction priced as a full-rate one) and against
    the peak for half-rate ones (an upper bound)."""
    out = {'valu_insts_per_launch': v['valu_insts_per_launch']}
    pk = valu_peaks()
    if pk is not None:
        rate = v['valu_insts_per_launch'] / (us * 1e-6)
        out['valu_frac'] = rate / pk[0]
        out['valu_frac_if_all_half_rate'] = rate / pk[1]
        out['valu_peak_wave_insts_per_s'] = {'plain': pk[0], 'half_rate': pk[1]}
    return out


def main():
    if os.environ.get('DETBOX_BENCH_DEBUG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['DETBOX_BENCH_DEBUG']), exit=True)
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))      # before anything initialises the GPU in this process

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE', file=sys.stderr)
    # DETBOX_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get('DETBOX_BENCH_BACKEND', 'nccl')
    local_dev = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = Workload(args.workload, args.regime, rank, dev)
    wl.exchange_root = 0 if args.exchange == 'root' else None
    wl.exchange_payload = args.exchange_payload
    B, P, C = wl.B, wl.P, wl.C
    m = measure(wl, args, world, dist, args.steps, args.warmup, args.blocks, args.streams, args.graph)
    host_waits = wl.exchange.host_waits if wl.exchange is not None else None
    gt_mean = wl.total_gt / B
    exchange = None
    if wl.exchange is not None:
        # what the exchange asks of the links, next to what the kernels ask of the chip: bytes INTO this rank per step,
        # the time its gathers took on the side stream (events; device collectives only), and the rate that would hide
        # them behind a step.  A GPU's seven xGMI links carry 7 x 76.8 GB/s in one direction at their peak.
        ex = wl.exchange
        n_g = max(ex.gathers, 1)
        bytes_in = ex.gather_bytes_in / n_g
        ms = ex.gather_ms / n_g if ex.gather_ms > 0 else None
        peak = 7 * 76.8
        exchange = dict(collective='gather to rank 0' if args.exchange == 'root' else 'all-gather', payload=args.exchange_payload,
                        bytes_in_per_step_per_rank=bytes_in, gather_ms_per_step=ms, batches_measured=ex.gathers,
                        roofline=dict(bound='xgmi', peak=peak, unit='GB/s',
                                      achieved=(bytes_in / (ms * 1e-3) / 1e9 if ms else None),
                                      frac=(bytes_in / (ms * 1e-3) / 1e9 / peak if ms else None),
                                      note='peak = 7 links x 76.8 GB/s into one GPU; achieved = bytes received by rank 0 per batch '
                                           '/ time of its gather on the side stream'))

    if rank == 0:
        pr = price(wl, m, world, args.steps)
        n_cpu = args.cpu_images if args.cpu_images >= 0 else (8 if P > 50000 else wl.B)
        cpu = None
        if world == 1 and n_cpu > 0 and not args.no_cpu_baseline:
            n_cpu = min(n_cpu, B)
            v, dt, nres = cpu_baseline(wl, n_cpu)
            cpu = dict(value=v, unit='images/s', cores=torch.get_num_threads(), kind='port',
                       sample=f'{n_cpu} image(s) of the same workload through the CPU oracle '
                              f'(assign + loop-faithful pre_predict + greedy NMS + dict loop), {dt:.1f} s, {nres} detections')
        # the other configurations the metric is quoted on, and the regime the HBM roofline was chosen for
        extra = None
        if world == 1 and not args.no_extra:
            extra = {}
            del wl
            torch.cuda.empty_cache()
            # graph replay where the host's launch rate is what binds (issuing a stage twice costs more than the
            # stage's kernels take: tools/twice.sh): the small workload and the trained-like regime
            # sparse: a head whose background column is raised by 8 (objectness lowered by 6): a few thousand candidates per image.  exact: the same dense
            # workload with torch's own activation instead of the fused one -- the mode whose keep lists are bit-exact.
            for name, regime, steps, graph, exact in (
                    ('ssd300', 'dense', 200, True, False), ('ssd300_b512', 'dense', 20, False, False),
                    ('retinanet800', 'sparse', 20, False, False), ('ssd300_b512', 'sparse', 20, False, False),
                    ('retinanet800', 'trained', 50, True, False), ('ssd300_b512', 'trained', 50, True, False),
                    ('retinanet800', 'dense', 20, False, True), ('ssd300_b512', 'dense', 20, False, True),
                    ('fcos800', 'dense', 100, True, False), ('yolov3', 'dense', 100, True, False)):
                if name == args.workload and regime == args.regime and not exact:
                    continue
                w2 = Workload(name, regime, 0, dev, exact=exact)
                m2 = measure(w2, args, 1, None, steps, 3, 5, args.streams, graph)
                p2 = price(w2, m2, 1, steps)
                extra[f'{name}/{regime}' + ('/exact' if exact else '')] = {
                    'images_per_s': p2['ips'], 'ms_per_step': p2['ms_per_step'], 'images_per_gpu': w2.B,
                    'activation': 'torch on the device, then the selection (bit-exact candidate sets)' if exact else 'fused into the selection pass',
                    'hbm_frac_whole_step': p2['hbm_frac'], 'algorithmic_bytes_per_image': p2['bytes_img'],
                    'candidates_per_image': m2['n_cand'], 'kept_per_image': m2['n_keep'], 'launch': m2['launch'],
                    'dominant_kernel': m2['dom'], 'dominant_avg_us': round(m2['kern'][m2['dom']]['avg_us'], 2),
                    'dominant_hbm_frac': p2['roofline']['frac'], 'blocks': p2['blocks'],
                    'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in m2['kern'].items()}}
                del w2
                torch.cuda.empty_cache()
        line = {
            'metric': 'images/sec assign+decode+NMS', 'value': pr['ips'], 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': pr['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'launch': m['launch'],
            'config': {'workload': f'{args.workload} {args.regime}: B={B}/GPU P={P} C={C} top_k={WORKLOADS[args.workload]["top_k"]} conf>0.01 nms=0.45',
                       'images_per_gpu': B, 'global_batch': world * B, 'candidates_per_image': m['n_cand'],
                       'kept_per_image': m['n_keep'], 'gt_per_image': gt_mean, 'parallelism': f'image-sharded x{world}',
                       'streams': m['streams'],
                       'exchange': ('none' if world == 1 else ('gather to rank 0' if args.exchange == 'root' else 'all-gather')),
                       'exchange_host_waits': host_waits},
            'exchange': (dict(exchange, rate_to_hide_behind_a_step_GBps=exchange['bytes_in_per_step_per_rank'] / (pr['ms_per_step'] * 1e-3) / 1e9)
                         if exchange is not None else None),
            'blocks': pr['blocks'],
            'roofline': pr['roofline'],
            'roofline_streaming': pr['streaming'],
            'roofline_nms': pr['pairs'],
            'cpu_baseline': cpu,
            'pipeline': {'algorithmic_bytes_per_image': pr['bytes_img'], 'hbm_frac_whole_step': pr['hbm_frac'],
                         'launches_per_step': pr['launches_per_step'], 'sub30_us_kernels_total_us': round(pr['sub30_us_total'], 1)},
            'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in m['kern'].items()},
            'workloads': extra,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
