"""Ad-hoc timing of select_decode / nms variants on RetinaNet-800 shapes (development aid)."""
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
names = ['select_sample', 'select_plan', 'select_scan', 'select_check', 'select_finalize', 'nms_segments', 'nms_merge', 'activate_scores']

def run(tag, fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ops.profile_reset(); ops.profile_enable(True)
    for _ in range(n): fn()
    torch.cuda.synchronize(); ops.profile_enable(False)
    out = {}
    for nm in names:
        ms, k = ops.profile_read(nm)
        if k: out[nm] = round(1e3 * ms / k, 1)
    print(tag, out, flush=True)

run('softmax thr=0.01 k=1000', lambda: ops.select_decode(confs, locs, anch, 0, 1, 0, 0.01, 1000, activation=1))
run('softmax thr=2.0 (no emission)', lambda: ops.select_decode(confs, locs, anch, 0, 1, 0, 2.0, 1000, activation=1))
sc = torch.softmax(confs, -1)
run('none thr=0.01 k=1000', lambda: ops.select_decode(sc, locs, anch, 0, 1, 0, 0.01, 1000, activation=0))
run('none thr=2.0 (no emission)', lambda: ops.select_decode(sc, locs, anch, 0, 1, 0, 2.0, 1000, activation=0))
run('softmax thr=0.01 k=200', lambda: ops.select_decode(confs, locs, anch, 0, 1, 0, 0.01, 200, activation=1))
run('activate only', lambda: ops.activate_scores(confs, None, 1))
c = ops.select_decode(confs, locs, anch, 0, 1, 0, 0.01, 1000, activation=1)
run('nms', lambda: ops.nms_candidates(c, 0.45))
