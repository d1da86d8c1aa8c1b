"""GPU-side experiment: the NMS call of a bench workload alone, with the pair-list kernel's counters."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detection_amd import ops, _lib

name, regime = (sys.argv[1:] + ['retinanet800', 'dense'])[:2]
dev = torch.device('cuda', 0)
wl = bench.Workload(name, regime, 0, dev)
c = wl.model.candidates(wl.outputs(), 0.01, wl.top_k)
L = _lib.lib()
out = (ctypes.c_uint64 * 8)()
for _ in range(3):
    keep, kc = ops.nms_candidates(c, 0.45)
torch.cuda.synchronize()
L.detbox_debug_nms_stats(out, 1)
ops.profile_reset(); ops.profile_only(''); ops.profile_enable(True)
N = 20
for _ in range(N):
    keep, kc = ops.nms_candidates(c, 0.45)
torch.cuda.synchronize()
ops.profile_enable(False)
L.detbox_debug_nms_stats(out, 1)
st = [int(x) / N for x in out]
ms, n = ops.profile_read('nms_segments')
print(f'{name}/{regime}: nms_segments {1e3 * ms / n:.1f} us; per launch: segments {st[0]:.0f}, fallbacks {st[1]:.0f}, '
      f'passed {st[2]:.0f}, edges {st[3]:.0f}, drains {st[4]:.0f}; cycles per segment: sort {st[5] / max(st[0], 1):.0f}, pairs {st[6] / max(st[0], 1):.0f}, resolve {st[7] / max(st[0], 1):.0f}; candidates {int(c.counts.sum())}, kept {int(kc.sum())}')
