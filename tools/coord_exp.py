"""DCN coordinate gradient alone, the three stage shapes (256 RoIs); DM_COORD_V1=1 selects the first-generation kernel
(read once per process).  Prints a checksum so that two runs can be compared."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
N = 256
for C, S, sc in ((64, 56, 1.0), (128, 28, 1.0), (256, 14, 1.0), (64, 56, 4.0)):
    torch.manual_seed(0)
    x = torch.randn(N, C, S, S, device=dev)
    cg = torch.randn(N, 9 * C, S, S, device=dev)
    off = torch.randn(N, 36, S, S, device=dev) * sc
    out = torch.empty_like(off)
    for _ in range(3): ops.deform_coord_grad(cg, x, off, 2, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.deform_coord_grad(cg, x, off, 2, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f'coord grad {C} ch @{S}x{S} x{N}, offsets x{sc}: {e0.elapsed_time(e1) / 10:.3f} ms  sum {out.double().sum().item():.4f} abs {out.double().abs().sum().item():.4f}', flush=True)
