"""GPU-side measurement: host time to issue each operator of a step (no synchronisation inside the loop; the
trained-like regime keeps the GPU ahead of the host, so the wall time per call is the host's own cost)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detection_amd import ops

name, regime = (sys.argv[1:] + ['retinanet800', 'trained'])[:2]
dev = torch.device('cuda', 0)
wl = bench.Workload(name, regime, 0, dev)
N = 300
for _ in range(5):
    out = wl.step(1)
torch.cuda.synchronize()
targets, c, kc, rows_ids, rows_vals, row_off = out
keep, kc = ops.nms_candidates(c, 0.45)


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{label:12s} issue {1e6 * (t1 - t0) / N:7.1f} us/call   drained after {1e6 * (t2 - t0) / N:7.1f} us/call')


timed('assign', wl.assign)
timed('candidates', lambda: wl.model.candidates(wl.outputs(), 0.01, wl.top_k))
timed('nms', lambda: ops.nms_candidates(c, 0.45))
timed('pack', lambda: ops.pack_detections(c.boxes, c.scores, c.class_ids, keep, kc, wl.ids, wl.hw))
timed('step', lambda: wl.step(1))
timed('empty x3', lambda: (torch.empty(1000, device=dev), torch.empty(1000, device=dev), torch.empty(1000, device=dev)))
