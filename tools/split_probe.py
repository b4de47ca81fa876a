"""bf16-split MFMA mode of the implicit-GEMM convolutions (DM_MFMA_SPLIT=3, opt-in): error against the exact fp32
kernels and against float64, and time, per layer shape of the path.  usage: python tools/split_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
import torch.nn.functional as F
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)


def t_ms(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(name, N, srcs_c, cout, S, ks, relu=True, ref64=True):
    srcs = [torch.randn(N, c, S, S, generator=g).to(dev) for c in srcs_c]
    cin = sum(srcs_c)
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    wq = ops.pack_conv_weight(w, src_channels=srcs_c, split=False)
    ws = ops.pack_conv_weight(w, src_channels=srcs_c, split=True)
    exact = ops.conv2d(srcs, wq, b, cout, ks, relu=relu)
    split = ops.conv2d(srcs, ws, b, cout, ks, relu=relu)
    scale = float(exact.abs().max())
    msg = f'{name:44s} split - fp32 kernel: max {float((split - exact).abs().max()):.3g} (scale {scale:.3g})'
    if ref64:
        n64 = min(N, 8)
        r = F.conv2d(torch.cat([s[:n64] for s in srcs], 1).double(), w.double(), b.double(), padding=ks // 2)
        r = r.relu() if relu else r
        msg += f'; vs f64: fp32 kernel {float((exact[:n64].double() - r).abs().max()):.3g}, split {float((split[:n64].double() - r).abs().max()):.3g}'
    flops = 2.0 * N * S * S * cin * cout * ks * ks
    te = t_ms(lambda: ops.conv2d(srcs, wq, b, cout, ks, relu=relu))
    ts = t_ms(lambda: ops.conv2d(srcs, ws, b, cout, ks, relu=relu))
    print(msg + f'; time fp32 {te:.3f} ms ({flops / te / 1e9:.0f} TF/s) split {ts:.3f} ms ({flops / ts / 1e9:.0f} TF/s fp32-equivalent)', flush=True)


case('conv3x3 256->256 @14, 512 RoIs', 512, [256], 256, 14, 3)
case('conv3x3 256->256 @14, 100 RoIs', 100, [256], 256, 14, 3)
case('conv3x3 256->36 @14 (DCN offsets), 512', 512, [256], 36, 14, 3, relu=False)
case('conv3x3 128->36 @28, 512', 512, [128], 36, 28, 3, relu=False)
case('conv3x3 64->36 @56, 256', 256, [64], 36, 56, 3, relu=False)
case('fuse 1x1 [256,256,2]->256 @14, 512', 512, [256, 256, 2], 256, 14, 1)
case('fuse 1x1 [128,128,2]->128 @28, 512', 512, [128, 128, 2], 128, 28, 1)
case('fuse 1x1 [64,64,2]->64 @56, 256', 256, [64, 64, 2], 64, 56, 1)
case('out 1x1 256->126 @14, 512', 512, [256], 126, 14, 1)
case('out 1x1 64->30 @56, 256', 256, [64], 30, 56, 1)
case('semantic 1x1 256->256 on P4 [1,256,50,84]', 1, [256], 256, 50, 1, ref64=False)
case('colgrad 1x1 64->576 @56, 256', 256, [64], 576, 56, 1, relu=False)
case('col GEMM 1x1 576->64 @56, 128', 128, [576], 64, 56, 1)
