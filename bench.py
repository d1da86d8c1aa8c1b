#!/usr/bin/env python3
"""Benchmark of the detection box-ops hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--regime dense|sparse]

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
    """Head outputs of image i (seed 2000+i): locs ~ N(0,1), logits ~ N(0,1) (dense) or with the
    background column +8 / objectness -6 (sparse)."""
    g = torch.Generator().manual_seed(2000 + i)
    locs = torch.randn(P, 4, generator=g)
    confs = torch.randn(P, C, generator=g)
    ext = torch.randn(P, generator=g) if family in ('fcos', 'yolo') else None
    if regime == 'sparse':
        if family == 'yolo':
            ext -= 6.0
        else:
            confs[:, 0] += 8.0
    return locs, confs, ext


class Workload:
    def __init__(self, name, regime, rank, dev):
        w = WORKLOADS[name]
        self.name, self.regime, self.w, self.dev = name, regime, w, dev
        self.exchange = None
        self.exchange_root = 0          # the evaluator is one process (rank 0), as in the reference's test loop
        self.model = w['make']()
        m = self.model
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
        targets = self.assign()
        m = self.model
        c = m.candidates(self.outputs(), 0.01, self.top_k)      # activation fused into the selection pass
        keep, kc = ops.nms_candidates(c, 0.45)
        if world == 1:
            rows_ids, rows_vals, row_off = ops.pack_detections(c.boxes, c.scores, c.class_ids, keep, kc, self.ids, self.hw)
        else:
            # the exchange step: this rank's detections as a compact message; the gather of the previous
            # batch's messages starts now, on a side stream, and overlaps this batch's kernels
            if self.exchange is None:
                self.exchange = RowExchange(keep_results=False, root=self.exchange_root)
                self.hdr_rows = ops.wire_header_rows(self.B)
            rows_ids, rows_vals, row_off, wire = ops.pack_detections(c.boxes, c.scores, c.class_ids, keep, kc, self.ids,
                                                                     self.hw, wire_images_cap=self.B)
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


def measured_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes (profiles/traffic.json, written
    by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command), or None."""
    try:
        with open(os.path.join(REPO, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        return t[f'{args.workload}/{args.regime}'][kernel]['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def main():
    if os.environ.get('DETBOX_BENCH_DEBUG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['DETBOX_BENCH_DEBUG']), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='retinanet800', choices=sorted(WORKLOADS))
    ap.add_argument('--regime', default='dense', choices=['dense', 'sparse'])
    ap.add_argument('--cpu-images', type=int, default=-1, help='images in the CPU baseline sample (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle timing (profiling runs)')
    ap.add_argument('--graph', action='store_true',
                    help='N=1 only: capture one step per stream in a HIP graph and replay it in the timed region (launch-bound workloads)')
    ap.add_argument('--streams', type=int, default=2,
                    help='steps are issued round-robin on this many HIP streams (independent batches in flight)')
    ap.add_argument('--exchange', default='root', choices=['root', 'all'],
                    help='N>1: detections are gathered to rank 0 (one evaluator process) or to every rank')
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = Workload(args.workload, args.regime, rank, dev)
    wl.exchange_root = 0 if args.exchange == 'root' else None
    B, P, C = wl.B, wl.P, wl.C

    def barrier():
        if world > 1:
            if wl.exchange is not None:
                wl.exchange.finish()            # every batch's all-gather is complete before the clock is read
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(max(args.warmup, 1)):
        out = wl.step(world)
    torch.cuda.synchronize()
    _, c, kc, _, _, row_off = out
    n_cand = float(c.counts.sum().item()) / B
    n_keep = float(kc.clamp(min=0).sum().item()) / B
    if bool((kc < 0).any().item()):
        raise RuntimeError('NMS reported an error flag')

    names = ['assign_anchor', 'assign_fixup', 'assign_fcos', 'select_sample', 'select_plan', 'select_scan',
             'select_bucket', 'select_check', 'select_materialize', 'score_hist', 'select_kth', 'select_refine', 'select_emit',
             'select_finalize', 'nms_segments', 'nms_scan', 'nms_merge', 'pack_detections']
    priced = ('select_scan', 'score_hist', 'select_emit', 'assign_anchor', 'assign_fcos', 'nms_segments', 'select_finalize')

    # Untimed survey pass: every kernel bracketed by events -> the per-kernel table and the dominant kernel.
    ops.profile_reset()
    ops.profile_only('')
    ops.profile_enable(True)
    for _ in range(5):
        wl.step(world)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    kern = {}
    for nm in names:
        ms, n = ops.profile_read(nm)
        if n:
            kern[nm] = dict(total_ms=ms, launches=n, avg_us=1e3 * ms / n)
    dom = max((k for k in kern if k in priced), key=lambda k: kern[k]['total_ms'])

    # Timed region: only the dominant kernel carries events, on the stream it is launched on.
    ops.profile_reset()
    ops.profile_only(dom)
    ops.profile_enable(True)
    graph = None
    streams = StreamRing(dev, args.streams) if args.streams > 1 else None
    if args.graph and world == 1:
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
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
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
    ops.profile_enable(False)
    ops.profile_only('')
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms, n = ops.profile_read(dom)
        if n == 0:                                   # graph replay: no events inside; fall back to the survey pass
            ms, n = kern[dom]['total_ms'], kern[dom]['launches']
        dom_us = 1e3 * ms / max(n, 1)
        G_mean = wl.total_gt / B
        score_bytes = 4.0 * B * P * C
        alg_bytes = {      # algorithmic bytes per launch (whole batch of this rank)
            'select_scan': score_bytes, 'score_hist': score_bytes, 'select_emit': score_bytes,
            'assign_anchor': B * (48.0 * P + 24.0 * G_mean), 'assign_fcos': B * (52.0 * P + 24.0 * G_mean),
            'nms_segments': B * (28.0 * n_cand + 8.0 * n_keep),
            'select_finalize': B * (28.0 * n_cand + 32.0 * n_cand),
        }
        achieved = alg_bytes[dom] / (dom_us * 1e-6) / 1e9
        roofline = dict(kernel=dom, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(args, dom), avg_us=dom_us,
                        launches=n, algorithmic_bytes=alg_bytes[dom],
                        avg_us_one_batch_in_flight=kern[dom]['avg_us'],
                        note='avg_us is measured in the timed region, where the kernel shares the CUs with the other '
                             'batch in flight; avg_us_one_batch_in_flight comes from the single-stream survey pass')
        # the streaming kernel of the path (reads the logits once), from the survey pass
        streaming = None
        if 'select_scan' in kern:
            a = score_bytes / (kern['select_scan']['avg_us'] * 1e-6) / 1e9
            streaming = dict(kernel='select_scan', bound='hbm', achieved=a, peak=HBM_PEAK_GBS, unit='GB/s',
                             frac=a / HBM_PEAK_GBS, traffic=measured_traffic(args, 'select_scan'),
                             avg_us=kern['select_scan']['avg_us'], algorithmic_bytes=score_bytes,
                             note='survey pass: every kernel bracketed by events')
        anchor_bytes = {'fcos': 8, 'yolo': 20}.get(wl.w['family'], 16)
        bytes_img = ((52.0 if wl.w['family'] == 'fcos' else 48.0) * P + 24.0 * G_mean
                     + 4.0 * P * C + 16.0 * P + anchor_bytes * P + (4.0 * P if wl.ext is not None else 0.0) + 28.0 * n_cand
                     + 28.0 * n_cand + 8.0 * n_keep)
        ips = world * B * args.steps / elapsed
        n_cpu = args.cpu_images if args.cpu_images >= 0 else (8 if P > 50000 else wl.B)
        cpu = None
        if world == 1 and n_cpu > 0 and not args.no_cpu_baseline:
            n_cpu = min(n_cpu, B)
            v, dt, nres = cpu_baseline(wl, n_cpu)
            cpu = dict(value=v, unit='images/s', cores=torch.get_num_threads(), kind='port',
                       sample=f'{n_cpu} image(s) of the same workload through the CPU oracle '
                              f'(assign + loop-faithful pre_predict + greedy NMS + dict loop), {dt:.1f} s, {nres} detections')
        line = {
            'metric': 'images/sec assign+decode+NMS', 'value': ips, 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'launch': (f'hip-graph replay on {len(graph)} stream(s)' if graph is not None else ('stream' if streams is None else f'{len(streams)} streams, steps round-robin')),
            'config': {'workload': f'{args.workload} {args.regime}: B={B}/GPU P={P} C={C} top_k={wl.top_k} conf>0.01 nms=0.45',
                       'images_per_gpu': B, 'global_batch': world * B, 'candidates_per_image': n_cand,
                       'kept_per_image': n_keep, 'gt_per_image': G_mean, 'parallelism': f'image-sharded x{world}', 'streams': (1 if streams is None else len(streams)), 'exchange': ('none' if world == 1 else ('gather to rank 0' if args.exchange == 'root' else 'all-gather'))},
            'roofline': roofline,
            'roofline_streaming': streaming,
            'cpu_baseline': cpu,
            'pipeline': {'algorithmic_bytes_per_image': bytes_img,
                         'hbm_frac_whole_step': bytes_img * ips / world / (HBM_PEAK_GBS * 1e9)},
            'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in kern.items()},
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
