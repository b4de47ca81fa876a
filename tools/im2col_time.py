"""deform_im2col alone at the training step's shapes (DM_IM2COL_V1=1 = the per-element kernel).
usage: python tools/im2col_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynamask_amd import ops
from kbench import t
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(1)
for N, C, S in ((128, 64, 56), (256, 128, 28), (256, 256, 14)):
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    off = torch.randn(N, 36, S, S, device=dev, generator=g) * 0.7
    out = torch.empty(N, 9 * C, S, S, device=dev)
    ms = t(lambda: ops.deform_im2col(x, off, 2, out=out), iters=20, warmup=5)
    print(f'deform_im2col {N}x{C}x{S}x{S}: {ms:.3f} ms = {out.numel() * 4 / ms / 1e9:.2f} TB/s written  (checksum {float(out.double().sum()):.9e})')
