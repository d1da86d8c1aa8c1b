"""Runs ONE select/nms configuration a few times (for rocprofv3 counter passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import object_detection_amd as D
from object_detection_amd import ops
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', 16)); C = 81
act = int(os.environ.get('ACT', 1)); thr = float(os.environ.get('THR', 0.01)); k = int(os.environ.get('K', 1000))
what = os.environ.get('WHAT', 'select')
m = D.RetinaNet(80, [100, 50, 25, 13, 7]); P = m.pboxes.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
locs = torch.randn(B, P, 4, device=dev, generator=g)
confs = torch.randn(B, P, C, device=dev, generator=g)
if act == 0: confs = torch.softmax(confs, -1)
anch = m.pboxes.to(dev)
c = None
for _ in range(3):
    c = ops.select_decode(confs, locs, anch, 0, 1, 0, thr, k, activation=act)
    if what == 'nms':
        ops.nms_candidates(c, 0.45)
torch.cuda.synchronize()
