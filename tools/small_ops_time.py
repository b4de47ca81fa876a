"""Back-to-back times of the small memory-bound launches of a training step (the synchronised figures of
tools/step_shapes.py carry ~25 us of host time each): which of them are far from what their bytes cost?
usage: python tools/small_ops_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynamask_amd import ops, synth
from kbench import t
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def line(name, ms, nbytes):
    print(f'{name:64s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s of {nbytes / 1e6:7.1f} MB', flush=True)


go = R(256, 1, 112, 112)
line('upsample2x_backward 256x1x56x56 <- 112x112 (align_corners)', t(lambda: ops.upsample2x_backward(go, None, (256, 1, 56, 56), True), iters=50, warmup=10), go.numel() * 5)
for C, S in ((64, 28), (128, 14)):
    go = R(256, C, 2 * S, 2 * S); fo = torch.relu(R(256, C, 2 * S, 2 * S))
    line(f'upsample2x_backward 256x{C}x{S}x{S} (+ReLU mask)', t(lambda: ops.upsample2x_backward(go, fo, (256, C, S, S), False), iters=50, warmup=10), go.numel() * 9)
    x = R(128, C, S, S)
    line(f'upsample2x 128x{C}x{S}x{S} (+ReLU)', t(lambda: ops.upsample2x(x, align_corners=False, relu=True), iters=50, warmup=10), x.numel() * 20)
x = R(128, 1, 56, 56)
line('upsample2x 128x1x56x56 (align_corners)', t(lambda: ops.upsample2x(x, align_corners=True), iters=50, warmup=10), x.numel() * 20)
x = R(256, 128, 56, 56)
line('bn_stats 256x128x56x56', t(lambda: ops.bn_stats(x), iters=30, warmup=5), x.numel() * 8)
rois = synth.make_rois(2, 128, 800, 1333, seed=1).to(dev)
for C, S, (H, W) in ((64, 56, (200, 336)), (128, 28, (100, 168)), (256, 14, (50, 84))):
    go = R(256, C, S, S)
    line(f'point_sample_backward 256x{C}x{S}x{S} -> 2x{C}x{H}x{W}', t(lambda: ops.point_sample_backward(go, (2, C, H, W), rois, 0.25), iters=30, warmup=5), go.numel() * 4 + 2 * C * H * W * 4)
a = R(256, 32, 56, 56); b = R(256, 32, 56, 56)
line('relu_backward_ 256x32x56x56', t(lambda: ops.relu_backward_(a, b), iters=50, warmup=10), a.numel() * 12)
