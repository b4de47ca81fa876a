import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/dynamask_amd') else os.environ.get('GRAFT_REPO_ROOT', '.'))
from dynamask_amd import ops
def t(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for C, S in ((128, 14), (64, 28)):
    go = torch.randn(256, C, 2 * S, 2 * S, device='cuda')
    y = torch.randn(256, C, 2 * S, 2 * S, device='cuda')
    ms = t(lambda: ops.upsample2x_backward(go, y, (256, C, S, S), False))
    print(f'upsample bwd {C}@{S}->{2*S}: {ms:.3f} ms  ({(2 * go.numel() + go.numel() // 4) * 4 / ms / 1e9:.2f} TB/s)')
