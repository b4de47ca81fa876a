#!/usr/bin/env python
"""Per-kernel timings at the benchmark shapes (512 RoIs, 1333x800 FPN): one
process, event-timed, prints a table with achieved TFLOP/s or GB/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth  # noqa: E402


def t(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    N = int(os.environ.get('KB_N', 512))
    dev = torch.device('cuda')
    rows = []

    def conv(name, srcs_c, cout, S, ks, nb=N, hw=None):
        H, W = (S, S) if hw is None else hw
        xs = [torch.randn(nb, c, H, W, device=dev) for c in srcs_c]
        cin = sum(srcs_c)
        w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
        b = torch.randn(cout, device=dev)
        wq = ops.pack_conv_weight(w, src_channels=list(srcs_c))
        ms = t(lambda: ops.conv2d(xs, wq, b, cout, ks, relu=True))
        fl = 2.0 * nb * H * W * cin * cout * ks * ks
        rows.append((name, ms, fl / ms / 1e9, 'TF/s'))

    conv('conv3x3 256->256 @14', [256], 256, 14, 3)
    conv('offconv3x3 256->36 @14', [256], 36, 14, 3)
    conv('offconv3x3 128->36 @28', [128], 36, 28, 3)
    conv('offconv3x3 64->36 @56', [64], 36, 56, 3)
    conv('fuse1x1 514->256 @14', [256, 256, 2], 256, 14, 1)
    conv('out1x1 256->126 @14', [256], 126, 14, 1)
    conv('fuse1x1 258->128 @28', [128, 128, 2], 128, 28, 1)
    conv('out1x1 128->62 @28', [128], 62, 28, 1)
    conv('fuse1x1 130->64 @56', [64, 64, 2], 64, 56, 1)
    conv('out1x1 64->30 @56', [64], 30, 56, 1)
    conv('colgrad1x1 64->576 @56 (256 rois)', [64], 576, 56, 1, nb=256)
    conv('colgrad1x1 128->1152 @28 (256)', [128], 1152, 28, 1, nb=256)
    conv('colgrad1x1 256->2304 @14 (256)', [256], 2304, 14, 1, nb=256)
    conv('sem1x1 256->256 P4', [256], 256, 0, 1, nb=1, hw=(50, 84))
    conv('sem1x1 256->128 P3', [256], 128, 0, 1, nb=1, hw=(100, 168))
    conv('sem1x1 256->64 P2', [256], 64, 0, 1, nb=1, hw=(200, 336))

    def dcn(name, C, S):
        x = torch.randn(N, C, S, S, device=dev)
        off = torch.randn(N, 36, S, S, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5
        wq = ops.pack_conv_weight(w)
        ms = t(lambda: ops.deform_conv(x, off, wq, C, 2, relu=True), iters=5)
        rows.append((name, ms, 2.0 * N * S * S * C * C * 9 / ms / 1e9, 'TF/s'))

    dcn('dcn 256 @14', 256, 14)
    dcn('dcn 128 @28', 128, 28)
    dcn('dcn 64 @56', 64, 56)

    feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
    rois = synth.make_rois(1, N, 800, 1333, seed=1).to(dev)
    ms = t(lambda: ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32]))
    rows.append(('roialign14 P2-5', ms, (N * 256 * 196 * 4 * 2) / ms / 1e6, 'GB/s(2x out)'))
    ms = t(lambda: ops.roi_align([feats[0]], rois[:128], 56, [1 / 4]), iters=3)
    rows.append(('roialign56 P2 (128 rois)', ms, (128 * 256 * 3136 * 4) / ms / 1e6, 'GB/s(out)'))
    sem = torch.randn(1, 256, 50, 84, device=dev)
    ms = t(lambda: ops.point_sample(sem, rois, 14, 0.25))
    rows.append(('pointsample 256 @14', ms, (N * 256 * 196 * 4) / ms / 1e6, 'GB/s(out)'))
    sem2 = torch.randn(1, 64, 200, 336, device=dev)
    ms = t(lambda: ops.point_sample(sem2, rois, 56, 0.25))
    rows.append(('pointsample 64 @56', ms, (N * 64 * 3136 * 4) / ms / 1e6, 'GB/s(out)'))
    for C, S in ((256, 14), (128, 28), (64, 56), (32, 56)):
        x = torch.randn(N, C, S, S, device=dev)
        nc = 80 if C > 32 else 1
        wi = torch.randn(nc, C, device=dev)
        bi = torch.randn(nc, device=dev)
        lab = torch.randint(0, nc, (N,), device=dev)
        ms = t(lambda: ops.class_logits(x, wi, bi, wi, bi, lab))
        rows.append((f'logits {C} @{S}', ms, (N * C * S * S * 4) / ms / 1e6, 'GB/s(in)'))
    for C, S in ((128, 14), (64, 28)):
        x = torch.randn(N, C, S, S, device=dev)
        ms = t(lambda: ops.upsample2x(x, relu=True))
        rows.append((f'upsample {C} @{S}', ms, (N * C * S * S * 4 * 5) / ms / 1e6, 'GB/s(in+out)'))
    x = torch.randn(N, 256, 14, 14, device=dev)
    enc = torch.randn(N, 100, 14, 14, device=dev)
    ms = t(lambda: ops.carafe(x, enc))
    rows.append(('carafe 256 @14->28', ms, (N * 256 * 196 * 4 * 5 + N * 100 * 196 * 4) / ms / 1e6, 'GB/s(in+out)'))
    if os.environ.get('KB_WGRAD', '1') == '1':
        NT = 256          # training RoI batch (2 images x 128 positives)
        def wgrad(name, cin, cout, S, ks):
            x = torch.randn(NT, cin, S, S, device=dev)
            dy = torch.randn(NT, cout, S, S, device=dev)
            dw = torch.zeros(cout, cin, ks, ks, device=dev)
            ms = t(lambda: ops.conv2d_wgrad(dy, x, ks, dw=dw))
            rows.append((name, ms, 2.0 * NT * S * S * cin * cout * ks * ks / ms / 1e9, 'TF/s'))
        wgrad('wgrad3x3 256->256 @14', 256, 256, 14, 3)
        wgrad('wgrad3x3 256->36 @14', 256, 36, 14, 3)
        wgrad('wgrad3x3 64->36 @56', 64, 36, 56, 3)
        wgrad('wgrad1x1 2304->256 @14 (dcn)', 2304, 256, 14, 1)
        wgrad('wgrad1x1 1152->128 @28 (dcn)', 1152, 128, 28, 1)
        wgrad('wgrad1x1 576->64 @56 (dcn)', 576, 64, 56, 1)
        wgrad('wgrad1x1 256->256 @14', 256, 256, 14, 1)
        wgrad('wgrad1x1 128->128 @28', 128, 128, 28, 1)
        wgrad('wgrad1x1 64->64 @56', 64, 64, 56, 1)
        wgrad('wgrad1x1 64->30 @56', 64, 30, 56, 1)
    print(f'{"kernel":32s} {"ms":>9s} {"rate":>10s}')
    for name, ms, rate, unit in rows:
        print(f'{name:32s} {ms:9.3f} {rate:10.1f} {unit}')


if __name__ == '__main__':
    main()
