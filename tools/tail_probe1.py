"""Rounds of workgroups in the 1x1 convolutions (128 couts x 128 px tiles, three workgroups per CU = 768 slots):
time per launch around the RoI counts of the training step (256 RoIs, or 128 per half) for the shapes it runs.
DM_CONV_TAIL=0 disables the small-tile tail launch (read once per process).
usage: python tools/tail_probe1.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
for (cin, cout, S, ns) in ((256, 256, 14, (120, 125, 128, 192, 250, 256, 262)), (128, 128, 28, (120, 122, 128, 244, 256)),
                           (256, 126, 14, (128, 256)), (128, 256, 56, (64, 128)),
                           (576, 64, 56, (83, 84, 96, 120, 125, 126, 128)), (64, 64, 56, (244, 250, 256))):
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    b = torch.randn(cout, device=dev)
    wq = ops.pack_conv_weight(w)
    xs = torch.randn(max(ns), cin, S, S, device=dev)
    ref = ops.conv2d(xs[:8], wq, b, cout, 1, relu=True)
    for n in ns:
        x = xs[:n]
        ms = t(lambda: ops.conv2d(x, wq, b, cout, 1, relu=True), iters=30, warmup=5)
        y = ops.conv2d(x, wq, b, cout, 1, relu=True)
        same = torch.equal(y[:8], ref) and torch.equal(y[n - 4:], ops.conv2d(x[n - 4:].contiguous(), wq, b, cout, 1, relu=True))
        mt = (cout + 127) // 128
        wgs = mt * ((n * S * S + 127) // 128)
        tf = 2.0 * n * S * S * cin * cout / ms / 1e9
        print(f'{cin:4d}->{cout:4d} @{S:3d}^2 N={n:4d} wgs={wgs:5d} ({wgs / 768:5.2f} rounds)  {ms:6.3f} ms  {tf:6.1f} TF/s  bits {"same" if same else "DIFFER"}', flush=True)
