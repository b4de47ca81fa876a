"""Does a convolution give the same bits when other kernels share the GPU?  (An intra-kernel race -- a missing barrier
or wait between LDS staging and use -- is closed by lock-step timing when the kernel runs alone and opens when its waves
are delayed by neighbours.)  For split in 0 / 3 / 6 (exact fp32, bf16-split products) and three layer shapes: the
output alone vs 30 runs beside a second stream that keeps the chip busy with memory-bound and MFMA kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynamask_amd import ops, streams  # noqa: E402

dev = torch.device('cuda', 0)
g = torch.Generator(device='cpu').manual_seed(3)
side = streams.side(dev, 0)
noise_x = torch.randn(256, 64, 56, 56, device=dev)
noise_w = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
noise_g = torch.randn(256, 64, 56, 56, device=dev)
shapes = [(256, 256, 256, 14, 3), (256, 514, 256, 14, 1), (128, 64, 64, 56, 3), (256, 128, 128, 28, 1), (256, 256, 2, 14, 1), (256, 128, 2, 28, 1),
          (256, 64, 2, 56, 1), (256, 36, 256, 14, 3), (256, 126, 256, 14, 1)]
modes = [0] + ([3, 6] if hasattr(ops, 'split_packing') else [])
for split in modes:
    for (n, cin, cout, s, ks) in shapes:
        x = torch.randn(n, cin, s, s, generator=g).to(dev)
        w = (torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        wp = ops.pack_conv_weight(w, split=split) if split else ops.pack_conv_weight(w)
        torch.cuda.synchronize()
        ref = ops.conv2d(x, wp, b, cout, ks, relu=True).clone()
        torch.cuda.synchronize()
        alone = sum(0 if torch.equal(ops.conv2d(x, wp, b, cout, ks, relu=True), ref) else 1 for _ in range(10))
        torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for rep in range(30):
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2 + rep % 3):
                    ops.relu_backward_(noise_g, noise_x)
                    ops.conv2d(noise_x, noise_w, None, 64, 3)
            if rep % 2:
                torch.cuda._sleep(20000 * (rep % 7))
            y = ops.conv2d(x, wp, b, cout, ks, relu=True)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
                worst = max(worst, (y - ref).abs().max().item())
        print(f'split={split} conv{ks}x{ks} {cin}->{cout} @{s}x{s} N={n}: alone {alone}/10 differ, beside other work {bad}/30 differ'
              f' (max |diff| {worst:.3e})', flush=True)
