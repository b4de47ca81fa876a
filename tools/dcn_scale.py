"""DCN 256->256 @14x14, 501 RoIs: time against the spread of the offsets (LDS bank conflicts of the gather)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
n = 501
w = torch.randn(256, 256, 3, 3, device=dev) / 48
wq = ops.pack_conv_weight(w)
x = torch.randn(n, 256, 14, 14, device=dev)
base = torch.randn(n, 36, 14, 14, device=dev)
for sc in (0.0, 0.1, 0.5, 1.5, 4.0):
    off = base * sc
    print(f'offset std {sc:4.1f}: dcn {t(lambda: ops.deform_conv(x, off, wq, 256, 2, relu=True), iters=20, warmup=3):.3f} ms')
print(f'conv3x3: {t(lambda: ops.conv2d(x, wq, None, 256, 3, relu=True), iters=20, warmup=3):.3f} ms')
