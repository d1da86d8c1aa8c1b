"""Phase clocks of the order step (nms_split / nms_tile) on a bench workload:  python tools/exp_merge.py [workload] [regime]"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from object_detection_amd import ops, _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'retinanet800'
regime = sys.argv[2] if len(sys.argv) > 2 else 'dense'
dev = torch.device('cuda', 0)
wl = bench.Workload(name, regime, 0, dev)
L = _lib.lib()
out = (ctypes.c_uint64 * 8)()
for _ in range(3):
    wl.step(1)
torch.cuda.synchronize()
L.detbox_debug_merge_stats(out, 1, 1)
n = 10
for _ in range(n):
    wl.step(1)
torch.cuda.synchronize()
L.detbox_debug_merge_stats(out, 0, 1)
st = [int(x) for x in out]
tiles = max(st[0], 1)
print(f'{name}/{regime}: {st[0] / n:.0f} tile workgroups per step, {st[1] / n:.0f} by merge-path levels')
for nm, v in zip(('set-up', 'piece loads', 'sort', 'output'), st[2:6]):
    print(f'  {nm:12s} {v / tiles:9.0f} cycles per workgroup')
print(f'  split: {st[6] / n:.0f} workgroups per step, {st[7] / max(st[6], 1):.0f} cycles each')
