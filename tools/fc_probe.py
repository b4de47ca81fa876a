import os, sys, torch
sys.path.insert(0, '/root/repo')
from dynamask_amd import ops
dev = torch.device('cuda')
def t(fn, iters=10, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for N in (100, 512, 1000):
    x = torch.randn(N, 12544, 1, 1, device=dev)
    w = torch.randn(1024, 12544, 1, 1, device=dev) / 112
    b = torch.randn(1024, device=dev)
    wq = ops.pack_conv_weight(w)
    ms = t(lambda: ops.conv2d(x, wq, b, 1024, 1, relu=True))
    y = ops.conv2d(x, wq, b, 1024, 1, relu=True).flatten(1)
    ref = torch.relu(torch.addmm(b, x.flatten(1), w.flatten(1).t()))
    ms2 = t(lambda: torch.relu(torch.addmm(b, x.flatten(1), w.flatten(1).t())))
    print(f'N={N}: conv-as-fc {ms:.3f} ms, torch.addmm {ms2:.3f} ms, max err {(y-ref).abs().max().item():.2e}')
