"""Times of the class-gathered logit kernels alone (the fused exit `class_logits_up2x`, `class_logits`) and of
`point_sample` at the bench / training shapes.  DYNAMASK_HIP_LIB=<other .so> times another build of the library.
usage: python tools/up2x_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynamask_amd import ops, synth
from kbench import t
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(0)
for N, C, S in ((512, 128, 14), (256, 64, 28), (100, 128, 14)):
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    wi = torch.randn(80, C, device=dev, generator=g); wd = torch.randn(80, C, device=dev, generator=g)
    bi = torch.randn(80, device=dev, generator=g); bd = torch.randn(80, device=dev, generator=g)
    lab = torch.randint(0, 80, (N,), device=dev, generator=g)
    ms = t(lambda: ops.class_logits_up2x(x, wi, bi, wd, bd, lab), iters=50, warmup=10)
    print(f'class_logits_up2x N={N} C={C} {S}x{S}: {ms*1e3:.1f} us')
for N, C, S in ((512, 256, 14), (256, 256, 14), (256, 128, 28), (256, 64, 56), (100, 256, 14)):
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    wi = torch.randn(80, C, device=dev, generator=g); wd = torch.randn(80, C, device=dev, generator=g)
    bi = torch.randn(80, device=dev, generator=g); bd = torch.randn(80, device=dev, generator=g)
    lab = torch.randint(0, 80, (N,), device=dev, generator=g)
    sig = torch.empty(N, 2, S, S, device=dev)
    ms = t(lambda: ops.class_logits(x, wi, bi, wd, bd, lab, sig_out=sig), iters=50, warmup=10)
    print(f'class_logits N={N} C={C} {S}x{S}: {ms*1e3:.1f} us = {x.numel() * 4 / ms / 1e9:.2f} TB/s read')
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
for N, lvl, C, S in ((512, 2, 256, 14), (256, 1, 128, 28), (256, 0, 64, 56)):
    sem = torch.randn(1, C, feats[lvl].shape[2], feats[lvl].shape[3], device=dev, generator=g)
    rois = synth.make_rois(1, N, 800, 1333, seed=1).to(dev)
    ms = t(lambda: ops.point_sample(sem, rois, S, 0.25), iters=50, warmup=10)
    print(f'point_sample N={N} C={C} {S}x{S}: {ms*1e3:.1f} us = {N * C * S * S * 4 / ms / 1e9:.2f} TB/s written')
for N, C, S in ((256, 256, 14), (256, 128, 28), (256, 64, 56)):
    x = torch.randn(N, C, S, S, device=dev, generator=g)
    wi = torch.randn(80, C, device=dev, generator=g); wd = torch.randn(80, C, device=dev, generator=g)
    lab = torch.randint(0, 80, (N,), device=dev, generator=g)
    gi = torch.randn(N, 1, S, S, device=dev, generator=g); gd = torch.randn(N, 1, S, S, device=dev, generator=g)
    gx = torch.randn(N, C, S, S, device=dev, generator=g)
    gwi = torch.zeros(80, C, device=dev); gwd = torch.zeros(80, C, device=dev); gbi = torch.zeros(80, device=dev); gbd = torch.zeros(80, device=dev)
    ms = t(lambda: ops.class_logits_backward(x, wi, wd, lab, gi, gd, gx, True, gwi, gbi, gwd, gbd), iters=50, warmup=10)
    print(f'class_logits_backward N={N} C={C} {S}x{S} (accumulating): {ms*1e3:.1f} us = {3 * x.numel() * 4 / ms / 1e9:.2f} TB/s (x, gx read; gx written)')
    lab0 = torch.full((N,), 3, device=dev, dtype=torch.int64)
    ms = t(lambda: ops.class_logits_backward(x, wi, wd, lab0, gi, gd, gx, True, gwi, gbi, gwd, gbd), iters=50, warmup=10)
    print(f'class_logits_backward N={N} C={C} {S}x{S}, every RoI the same class: {ms*1e3:.1f} us')
