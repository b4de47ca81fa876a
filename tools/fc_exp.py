"""dm_fc_fwd at the bbox head's layers against torch's addmm (the vendor GEMM); DM_FC_SEG is read once per process.
(Round 3: double-buffered operand fragments at three workgroups per CU 0.298 ms, 64-deep chunks 0.328 ms, against 0.298 for
the kernel as it is and 0.221 for the vendor GEMM at 1000 x 12544 -> 1024: neither is the lever.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
for N, K, M in ((1000, 12544, 1024), (1000, 1024, 1024), (256, 3136, 512)):
    x = torch.randn(N, K, device=dev); w = torch.randn(M, K, device=dev) / K ** 0.5; b = torch.randn(M, device=dev)

    def t(fn, it=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    ms = t(lambda: ops.fc(x, w, b, relu=True))
    ms2 = t(lambda: torch.relu(torch.addmm(b, x, w.t())))
    err = (ops.fc(x, w, b, relu=True) - torch.relu(torch.addmm(b, x, w.t()))).abs().max().item()
    print(f'fc {N} x {K} -> {M}  SEG={os.environ.get("DM_FC_SEG", "-")}: dm_fc_fwd {ms:.3f} ms {2*N*K*M/ms/1e9:.1f} TF/s | torch addmm + relu {ms2:.3f} ms | max diff {err:.2e}', flush=True)
