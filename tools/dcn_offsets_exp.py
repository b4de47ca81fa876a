"""How much of the fused DCN forward's time is the bilinear gather's LDS bank conflicts?  Same shapes, offsets
drawn at several scales (0 = the regular grid, every lane pair reads neighbouring words)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')


def t(fn, iters=8, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for C, S in ((256, 14), (128, 28), (64, 56)):
    x = torch.randn(N, C, S, S, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
    wq = ops.pack_conv_weight(w)
    t(lambda: ops.deform_conv(x, torch.zeros(N, 36, S, S, device=dev), wq, C, 2, relu=True), iters=10)
    row = f'dcn {C} @{S} x{N}:'
    for sc in (0.0, 0.25, 0.5, 1.0, 3.0):
        off = torch.randn(N, 36, S, S, device=dev) * sc
        ms = t(lambda: ops.deform_conv(x, off, wq, C, 2, relu=True))
        row += f'  sigma {sc:4.2f}: {ms:.3f} ms {2.0 * N * S * S * C * C * 9 / ms / 1e9:6.1f} TF/s'
    print(row, flush=True)
