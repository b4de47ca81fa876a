"""A few launches of conv3x3 and DCN 256->256 at 14x14 (501 RoIs = 3 full rounds) for PMC runs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 501
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
C = int(sys.argv[3]) if len(sys.argv) > 3 else 256
S = int(sys.argv[4]) if len(sys.argv) > 4 else 14
w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
wq = ops.pack_conv_weight(w)
x = torch.randn(n, C, S, S, device=dev)
off = torch.randn(n, 36, S, S, device=dev) * scale
for _ in range(4):
    ops.conv2d(x, wq, None, C, 3, relu=True)
    ops.deform_conv(x, off, wq, C, 2, relu=True)
torch.cuda.synchronize()
