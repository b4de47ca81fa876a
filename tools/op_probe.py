"""Event-timed single ops at the training step's shapes (256 RoIs on the 800x1333 pyramid).
  python tools/op_probe.py [case ...]      cases: psb (point-sample adjoint), wgrad (narrow weight gradients)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import synth, ops

dev = torch.device('cuda')
B, per = 2, 128
rois = synth.make_rois(B, per, bench.IMG_H, bench.IMG_W, seed=11).to(dev)
N = B * per


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


cases = sys.argv[1:] or ['psb', 'wgrad']
g = torch.Generator().manual_seed(5)
if 'psb' in cases:
    for C, S, (H, W), sc in ((64, 56, (200, 336), 0.25), (128, 28, (100, 168), 0.125), (256, 14, (50, 84), 0.0625)):
        go = torch.randn(N, C, S, S, generator=g).to(dev)
        gf = torch.zeros(B, C, H, W, device=dev)
        ms = timed(lambda: ops.point_sample_backward(go, (B, C, H, W), rois, sc, grad_feat=gf))
        print(f'point_sample_backward {N}x{C}x{S}x{S} -> {B}x{C}x{H}x{W}: {ms:.3f} ms  ({go.numel() * 4 / ms / 1e9:.2f} TB/s of gradient read)', flush=True)
if 'psb_split' in cases:       # by RoI size: which share of the 56 x 56 adjoint goes to the LDS-tile path / the global-atomic path
    C, S, (H, W), sc = 64, 56, (200, 336), 0.25
    side = ((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])).sqrt()
    for lo, hi in ((0, 100), (100, 180), (180, 300), (300, 450), (450, 2000)):
        sel = rois[(side >= lo) & (side < hi)].contiguous()
        if sel.shape[0] == 0:
            continue
        go = torch.randn(sel.shape[0], C, S, S, generator=g).to(dev)
        gf = torch.zeros(B, C, H, W, device=dev)
        ms = timed(lambda: ops.point_sample_backward(go, (B, C, H, W), sel, sc, grad_feat=gf))
        print(f'point_sample_backward sqrt(wh) in [{lo},{hi}): {sel.shape[0]} RoIs {ms:.3f} ms = {1e3 * ms / sel.shape[0]:.2f} us/RoI', flush=True)
if 'dcnfwd' in cases:       # fused DCN forward vs column matrix + 1x1 GEMM (the column matrix is what the backward needs anyway)
    for C, S in ((64, 56), (128, 28), (256, 14)):
        x = torch.randn(N, C, S, S, generator=g).to(dev)
        off = torch.randn(N, 36, S, S, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev)
        pk = ops.pack_conv_weight(w)
        wt = ops.dcn_weight_permute(w, C, C, True)
        pk_cm = ops.pack_conv_weight(wt, transpose_flip=True)
        t_f = timed(lambda: ops.deform_conv(x, off, pk, C, 2, relu=True))
        t_i = timed(lambda: ops.deform_im2col(x, off, 2), reps=5)
        col = ops.deform_im2col(x, off, 2)
        t_g = timed(lambda: ops.conv2d([col], pk_cm, None, C, 1, relu=True))
        a, b = ops.deform_conv(x, off, pk, C, 2, relu=True), ops.conv2d([col], pk_cm, None, C, 1, relu=True)
        print(f'DCN fwd C={C} @{S}: fused {t_f:.3f} ms; im2col {t_i:.3f} + GEMM {t_g:.3f} ms; max diff {float((a - b).abs().max()):.2e}', flush=True)
        del col
if 'dcnchunk' in cases:     # inference DCN at 512 RoIs: fused kernel vs RoI chunks of (im2col -> 1x1 GEMM) small enough for the
    from dynamask_amd import streams      # column matrix to stay in the Infinity Cache, the two kernels of successive chunks on two streams
    NI = 512
    for C, S in ((64, 56), (128, 28)):
        x = torch.randn(NI, C, S, S, generator=g).to(dev)
        off = torch.randn(NI, 36, S, S, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev)
        pk = ops.pack_conv_weight(w)
        pk_cm = ops.pack_conv_weight(ops.dcn_weight_permute(w, C, C, True), transpose_flip=True)
        t_f = timed(lambda: ops.deform_conv(x, off, pk, C, 2, relu=True), reps=5)
        print(f'DCN fwd C={C} @{S} x{NI}: fused {t_f:.3f} ms', flush=True)
        out = torch.empty(NI, C, S, S, device=dev)
        s1 = streams.side(dev, 0)
        for chunk in (16, 32, 64, 128):
            cols = [torch.empty(chunk, 9 * C, S, S, device=dev) for _ in range(2)]

            def run(two):
                main = torch.cuda.current_stream()
                evs = []
                for k, lo in enumerate(range(0, NI, chunk)):
                    col = cols[k & 1]
                    if two and len(evs) >= 2:
                        main.wait_event(evs[-2])          # the GEMM that last read this buffer
                    lib_call = ops.lib().dm_deform_im2col
                    ops.check(lib_call(ops._p(x[lo:lo + chunk]), ops._p(off[lo:lo + chunk]), chunk, C, S, S, 2, ops._p(col), ops._stream()), 'im2col')
                    if two:
                        e = torch.cuda.Event(); e.record(main); s1.wait_event(e)
                        with torch.cuda.stream(s1):
                            ops.conv2d([col], pk_cm, None, C, 1, relu=True, out=out[lo:lo + chunk])
                            e2 = torch.cuda.Event(); e2.record(s1); evs.append(e2)
                    else:
                        ops.conv2d([col], pk_cm, None, C, 1, relu=True, out=out[lo:lo + chunk])
                if two:
                    main.wait_stream(s1)
            for two in (False, True):
                t = timed(lambda: run(two), reps=3)
                print(f'   chunks of {chunk:3d} RoIs ({chunk * 9 * C * S * S * 4 / 2**20:.0f} MiB), {"two streams" if two else "one stream"}: {t:.3f} ms', flush=True)
        ref = ops.deform_conv(x, off, pk, C, 2, relu=True)
        print('   max diff vs fused', float((ref - out).abs().max()), flush=True)
if 'wgrad' in cases:
    for cout, cin, S, ks in ((36, 64, 56, 3), (36, 128, 28, 3), (36, 256, 14, 3), (16, 128, 28, 3), (30, 64, 56, 1), (62, 128, 28, 1),
                             (126, 256, 14, 1), (256, 256, 14, 3), (64, 576, 56, 1)):
        dy = torch.randn(N, cout, S, S, generator=g).to(dev)
        x = torch.randn(N, cin, S, S, generator=g).to(dev)
        dw = torch.zeros(cout, cin, ks, ks, device=dev)
        ms = timed(lambda: ops.conv2d_wgrad(dy, x, ks, dw=dw))
        fl = 2.0 * N * S * S * cout * cin * ks * ks
        print(f'conv2d_wgrad {cout}x{cin}x{ks}x{ks} @{S}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({(dy.numel() + x.numel()) * 4 / ms / 1e9:.2f} TB/s of operands)', flush=True)
if 'wgradcat' in cases:      # the concat weight gradients (feat, semantic, 2 coordinate planes) per source, and the slab reduce's share
    for cout, cs, S in ((128, (128, 128, 2), 28), (64, (64, 64, 2), 56), (256, (256, 256, 2), 14)):
        dy = torch.randn(N, cout, S, S, generator=g).to(dev)
        srcs = [torch.randn(N, c, S, S, generator=g).to(dev) for c in cs]
        dw = torch.zeros(cout, sum(cs), 1, 1, device=dev)
        fl = 2.0 * N * S * S * cout * sum(cs)
        ms = timed(lambda: ops.conv2d_wgrad(dy, srcs, 1, dw=dw))
        print(f'concat wgrad {cout}x{sum(cs)} @{S}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s', flush=True)
        for s in srcs:
            d1 = torch.zeros(cout, s.shape[1], 1, 1, device=dev)
            ms = timed(lambda: ops.conv2d_wgrad(dy, s, 1, dw=d1))
            print(f'     source of {s.shape[1]:3d} channels alone: {ms:.3f} ms', flush=True)
        for slab in (False,):
            ops.WGRAD_SLAB[0] = slab
            ms = timed(lambda: ops.conv2d_wgrad(dy, srcs, 1, dw=dw))
            print(f'     atomics instead of slabs: {ms:.3f} ms', flush=True)
            ops.WGRAD_SLAB[0] = True
