"""Ad-hoc timing of the anchor assignment on RetinaNet-800 shapes (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detection_amd import ops
dev = torch.device('cuda', 0)
wl = bench.Workload('retinanet800', 'dense', 0, dev)
for _ in range(3): wl.assign()
torch.cuda.synchronize(); ops.profile_reset(); ops.profile_enable(True)
for _ in range(10): wl.assign()
torch.cuda.synchronize(); ops.profile_enable(False)
for nm in ['assign_anchor', 'assign_fixup']:
    ms, k = ops.profile_read(nm); print(os.environ.get('DETBOX_ASSIGN_IPB'), nm, round(1e3 * ms / max(k, 1), 1), 'us')
