"""select_scan time on the RetinaNet-800 B=16 logits, dense / sparse thresholds (development aid; DETBOX_LIB picks the build)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import object_detection_amd as D
from object_detection_amd import ops

dev = torch.device('cuda', 0)
B, C = 16, 81
m = D.RetinaNet(80, [100, 50, 25, 13, 7]); P = m.pboxes.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
locs = torch.randn(B, P, 4, device=dev, generator=g)
confs = torch.randn(B, P, C, device=dev, generator=g)
anch = m.pboxes.to(dev)
res = []
for thr in (0.01, 2.0):
    f = lambda: ops.select_decode(confs, locs, anch, 0, 1, 0, thr, 1000, activation=1)
    for _ in range(3): f()
    torch.cuda.synchronize(); ops.profile_reset(); ops.profile_enable(True)
    for _ in range(10): c = f()
    torch.cuda.synchronize(); ops.profile_enable(False)
    ms, k = ops.profile_read('select_scan'); ms2, k2 = ops.profile_read('select_finalize')
    res.append(f'thr {thr}: scan {1e3 * ms / k:.1f} us, finalize {1e3 * ms2 / max(k2, 1):.1f} us, candidates {int(c.counts.sum())}')
print(os.environ.get('DETBOX_LIB', 'default'), '|', ' | '.join(res))
