"""dcn col2im at 56x56 (64 channels, 256 RoIs), timed alone.  (Round 3 swept channels per workgroup x threads per workgroup
with a build-time knob: 2 planes x 512 threads 1.02 ms, 2 x 256 1.09, 2 x 1024 1.21, 4 x 512 1.17, 4 x 256 1.85, 1 x 256 / 512 / 1024 1.29-1.32.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops
dev = torch.device('cuda')
N, C, S = 256, 64, 56
torch.manual_seed(0)
cg = torch.randn(N, 9 * C, S, S, device=dev)
off = torch.randn(N, 18, S, S, device=dev)
gx = torch.empty(N, C, S, S, device=dev)
for _ in range(5): ops.deform_col2im(cg, off, (N, C, S, S), 1, out=gx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.deform_col2im(cg, off, (N, C, S, S), 1, out=gx)
e1.record(); torch.cuda.synchronize()
print(f'col2im 64 ch @56x56 x256: {e0.elapsed_time(e1) / 10:.3f} ms  checksum {gx.double().sum().item():.6f} {gx.double().abs().sum().item():.6f}', flush=True)
# the offsets of the benchmark's training step (a zero-initialised offset conv: every interior pixel links with its neighbour)
for sigma in (0.0, 0.3):
    off0 = torch.randn(N, 18, S, S, device=dev) * sigma
    for _ in range(3): ops.deform_col2im(cg, off0, (N, C, S, S), 1, out=gx)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): ops.deform_col2im(cg, off0, (N, C, S, S), 1, out=gx)
    e1.record(); torch.cuda.synchronize()
    print(f'   offsets sigma {sigma}: {e0.elapsed_time(e1) / 10:.3f} ms  checksum {gx.double().sum().item():.6f} {gx.double().abs().sum().item():.6f}', flush=True)
for (C2, S2) in ((128, 28), (256, 14)):
    cg2 = torch.randn(N, 9 * C2, S2, S2, device=dev)
    off2 = torch.zeros(N, 36, S2, S2, device=dev)
    for _ in range(3): ops.deform_col2im(cg2, off2, (N, C2, S2, S2), 2)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): ops.deform_col2im(cg2, off2, (N, C2, S2, S2), 2)
    e1.record(); torch.cuda.synchronize()
    print(f'col2im {C2} ch @{S2}x{S2} x256, zero offsets: {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
