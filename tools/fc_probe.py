"""FC layer: dm_fc_fwd vs the conv kernel used as an FC vs the library GEMM (torch.addmm)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
for N, K, M in ((100, 12544, 1024), (1000, 12544, 1024), (1000, 1024, 1024), (1000, 1024, 320), (1000, 1024, 81), (512, 3136, 512)):
    x = torch.randn(N, K, device=dev)
    w = torch.randn(M, K, device=dev) / K ** 0.5
    b = torch.randn(M, device=dev)
    ms = t(lambda: ops.fc(x, w, b, relu=True))
    ref = torch.relu(torch.addmm(b, x, w.t()))
    ms2 = t(lambda: torch.relu(torch.addmm(b, x, w.t())))
    err = (ops.fc(x, w, b, relu=True) - ref).abs().max().item()
    print(f'N={N} K={K} M={M}: dm_fc_fwd {ms:.3f} ms ({2*N*K*M/ms/1e9:.1f} TF/s), torch.addmm {ms2:.3f} ms, max diff {err:.2e}')
