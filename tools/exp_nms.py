"""Ad-hoc timing of the NMS stage on RetinaNet-800 dense candidates (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import object_detection_amd as D
from object_detection_amd import ops
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', 16)); C = 81
m = D.RetinaNet(80, [100, 50, 25, 13, 7]); P = m.pboxes.shape[0]
g = torch.Generator(device=dev).manual_seed(0)
locs = torch.randn(B, P, 4, device=dev, generator=g)
confs = torch.randn(B, P, C, device=dev, generator=g)
anch = m.pboxes.to(dev)
c = ops.select_decode(confs, locs, anch, 0, 1, 0, 0.01, int(os.environ.get('K', 1000)), activation=1)
for _ in range(3): keep, kc = ops.nms_candidates(c, 0.45)
torch.cuda.synchronize(); ops.profile_reset(); ops.profile_enable(True)
for _ in range(5): ops.nms_candidates(c, 0.45)
torch.cuda.synchronize(); ops.profile_enable(False)
for nm in ['nms_segments', 'nms_scan', 'nms_merge']:
    ms, k = ops.profile_read(nm); print(nm, round(1e3 * ms / max(k, 1), 1), 'us x', k)
print('kept/img', kc.float().mean().item())
