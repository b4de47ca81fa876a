"""MaskPre's BatchNorm + ReLU + max-pool backward alone at the training shape (256 x 128 x 56 x 56) and the second layer's
(256 x 16 x 28 x 28).  DM_BN_BWD_V1=1 (read once per process) = the three-kernel backward, DM_BN_POOL_V1=1 = the per-output forward.
usage: python tools/bn_bwd_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(0)
for N, C, S in ((256, 128, 56), (256, 16, 28)):
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    mean = x.mean((0, 2, 3)); var = x.var((0, 2, 3), unbiased=False)
    gamma = torch.rand(C, device=dev, generator=g) + 0.5; beta = torch.randn(C, device=dev, generator=g) * 0.1
    go = torch.randn(N, C, S // 2, S // 2, device=dev, generator=g)
    mf = t(lambda: ops.bn_relu_maxpool(x, mean, var, gamma, beta), iters=30, warmup=5)
    y = ops.bn_relu_maxpool(x, mean, var, gamma, beta)
    print(f'bn_relu_maxpool {N}x{C}x{S}x{S}: {mf:.3f} ms  (checksum {float(y.double().sum()):.9e})')
    ms = t(lambda: ops.bn_relu_maxpool_backward(x, mean, var, gamma, beta, go), iters=30, warmup=5)
    gx, gg, gb = ops.bn_relu_maxpool_backward(x, mean, var, gamma, beta, go)
    print(f'bn_relu_maxpool_backward {N}x{C}x{S}x{S}: {ms:.3f} ms  (checksums gx {float(gx.double().abs().sum()):.6e} gg {float(gg.double().sum()):.6e} gb {float(gb.double().sum()):.6e})')
