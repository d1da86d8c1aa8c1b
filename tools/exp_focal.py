"""Ad-hoc timing of the fused focal loss on RetinaNet-800 shapes (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import ops
dev = torch.device('cuda', 0)
B, P, C = 16, 120087, 81
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, P, C, device=dev, generator=g, requires_grad=True)
t = torch.randint(-1, 3, (B, P), device=dev, generator=g).clamp(max=80)
for _ in range(2):
    l = ops.focal_loss_sum(x, t); l.backward(); x.grad = None
torch.cuda.synchronize(); ops.profile_reset(); ops.profile_enable(True)
for _ in range(5):
    l = ops.focal_loss_sum(x, t); l.backward(); x.grad = None
torch.cuda.synchronize(); ops.profile_enable(False)
for nm in ['focal_forward', 'focal_backward']:
    ms, k = ops.profile_read(nm); us = 1e3 * ms / max(k, 1)
    byt = B * P * C * 4 * (1 if nm == 'focal_forward' else 2)
    print(nm, round(us, 1), 'us', round(byt / us / 1e6, 2), 'TB/s')
# the unfused reference formulation on the same device, for scale
import torch.nn.functional as F, time
def ref():
    sel = t >= 0
    xs, ts = x[sel], t[sel]
    pt = F.softmax(xs, -1); lp = F.log_softmax(xs, -1)
    return F.nll_loss(0.25 * (1 - pt).pow(2) * lp, ts, reduction='sum')
for _ in range(2): ref().backward(); x.grad = None
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ref().backward(); x.grad = None
torch.cuda.synchronize(); print('unfused torch fwd+bwd', round((time.perf_counter() - t0) / 5 * 1e6, 1), 'us')
