"""bf16-split MFMA modes of the implicit-GEMM convolutions (DM_MFMA_SPLIT=3 | 6, opt-in): error against the exact fp32
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
    ws = {k: ops.pack_conv_weight(w, src_channels=srcs_c, split=k) for k in (3, 6)}
    exact = ops.conv2d(srcs, wq, b, cout, ks, relu=relu)
    split = {k: ops.conv2d(srcs, ws[k], b, cout, ks, relu=relu) for k in (3, 6)}
    scale = float(exact.abs().max())
    msg = f'{name:44s} scale {scale:.3g}; split3 / split6 - fp32 kernel: {float((split[3] - exact).abs().max()):.3g} / {float((split[6] - exact).abs().max()):.3g}'
    if ref64:
        n64 = min(N, 8)
        r = F.conv2d(torch.cat([s[:n64] for s in srcs], 1).double(), w.double(), b.double(), padding=ks // 2)
        r = r.relu() if relu else r
        msg += (f'; vs f64: fp32 kernel {float((exact[:n64].double() - r).abs().max()):.3g}, split3 {float((split[3][:n64].double() - r).abs().max()):.3g}, '
                f'split6 {float((split[6][:n64].double() - r).abs().max()):.3g}')
    flops = 2.0 * N * S * S * cin * cout * ks * ks
    te = t_ms(lambda: ops.conv2d(srcs, wq, b, cout, ks, relu=relu))
    ts = {k: t_ms(lambda: ops.conv2d(srcs, ws[k], b, cout, ks, relu=relu)) for k in (3, 6)}
    print(msg + f'; ms fp32 {te:.3f} ({flops / te / 1e9:.0f} TF/s), split3 {ts[3]:.3f} ({flops / ts[3] / 1e9:.0f}), split6 {ts[6]:.3f} ({flops / ts[6] / 1e9:.0f} fp32-equivalent TF/s)', flush=True)


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

# ---- the data-gradient forms: transposed + rotated packs, channel windows, masked / accumulating epilogues
def bwd_case(name, N, cin, cout, S, ks, lo=None, hi=None, srcs=None, mask=False, accumulate=False):
    w = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).to(dev)
    dy = torch.randn(N, cout, S, S, generator=g).to(dev) * 1e-3
    lo_, hi_ = (0, cin) if lo is None else (lo, hi)
    wwin = w[:, lo_:hi_].contiguous()
    outs = []
    for split in (0, 3, 6):
        wq = ops.pack_conv_weight(wwin, transpose_flip=True, split=split)
        out = torch.full((N, hi_ - lo_, S, S), 0.5e-3, device=dev) if accumulate else None
        m = (torch.randn(N, hi_ - lo_, S, S, generator=torch.Generator().manual_seed(5)).to(dev)) if mask else None
        outs.append(ops.conv2d(dy, wq, None, hi_ - lo_, ks, out=out, accumulate=accumulate, mask=m))
    r = F.conv_transpose2d(dy[:4].double(), w.double(), padding=ks // 2)[:, lo_:hi_]
    if accumulate:
        r = r + 0.5e-3
    e = [float((o[:4].double() - r).abs().max()) for o in outs] if not mask else [float('nan')] * 3
    print(f'{name:44s} scale {float(outs[0].abs().max()):.3g}; split3 / split6 - fp32 kernel: {float((outs[1] - outs[0]).abs().max()):.3g} / {float((outs[2] - outs[0]).abs().max()):.3g}; vs f64: fp32 {e[0]:.3g} split3 {e[1]:.3g} split6 {e[2]:.3g}', flush=True)


bwd_case('dgrad 3x3 256->256 @14', 64, 256, 256, 14, 3)
bwd_case('dgrad 3x3 256->256 @14 masked', 64, 256, 256, 14, 3, mask=True)
bwd_case('dgrad 3x3 36->256 @14 accumulate', 64, 256, 36, 14, 3, accumulate=True)
bwd_case('dgrad 1x1 window 256:512 of 514 -> @14', 64, 514, 256, 14, 1, lo=256, hi=512)
bwd_case('dgrad 1x1 window 512:514 of 514 -> @14', 64, 514, 256, 14, 1, lo=512, hi=514)
bwd_case('dgrad 1x1 126->256 @14', 64, 256, 126, 14, 1)
bwd_case('dgrad 1x1 30->64 @56', 32, 64, 30, 56, 1)
bwd_case('dgrad 1x1 80->256 (class logits shape) @14', 64, 256, 80, 14, 1)
