#!/usr/bin/env python
"""Launch the three roofline kernels (conv3x3, DCN 3x3, RoIAlign 14x14) at the bench shapes a few times (for rocprofv3
--kernel-trace --stats and the --pmc passes: FETCH_SIZE and WRITE_SIZE need separate runs on gfx950)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth  # noqa: E402

dev = torch.device('cuda')
N = 512
x = torch.randn(N, 256, 14, 14, device=dev)
w = torch.randn(256, 256, 3, 3, device=dev) / 48
b = torch.randn(256, device=dev)
wq = ops.pack_conv_weight(w)
off = torch.randn(N, 36, 14, 14, device=dev) * 0.5             # non-zero offsets: the deformable gather is exercised
wd = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, device=dev) / 48)
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, N, 800, 1333, seed=1).to(dev)
for _ in range(int(os.environ.get("PROBE_ITERS", 4))):
    ops.conv2d(x, wq, b, 256, 3, relu=True)
    ops.deform_conv(x, off, wd, 256, 2, relu=True)
    ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
torch.cuda.synchronize()
print('done')
