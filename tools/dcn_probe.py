"""A few launches of conv3x3 and DCN 256->256 at 14x14 (501 RoIs = 3 full rounds) for PMC runs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 501
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
w = torch.randn(256, 256, 3, 3, device=dev) / 48
wq = ops.pack_conv_weight(w)
x = torch.randn(n, 256, 14, 14, device=dev)
off = torch.randn(n, 36, 14, 14, device=dev) * scale
for _ in range(4):
    ops.conv2d(x, wq, None, 256, 3, relu=True)
    ops.deform_conv(x, off, wq, 256, 2, relu=True)
torch.cuda.synchronize()
