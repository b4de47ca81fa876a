import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
for C, S in ((256, 14), (128, 28), (64, 56)):
    wq = ops.pack_conv_weight(torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5)
    row = f'C={C} S={S}:'
    for n in (8, 16, 32, 64, 100, 128):
        x = torch.randn(n, C, S, S, device=dev); off = torch.randn(n, 36, S, S, device=dev)
        row += f' N={n} {t(lambda: ops.deform_conv(x, off, wq, C, 2, relu=True), iters=20, warmup=3):.3f}'
    print(row)
