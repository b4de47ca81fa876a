"""The one-kernel DCN data gradient (dm_dcn_bwd_data_fused) against the three-kernel path: differences and times.
  python tools/dcn_fused_probe.py [check] [time]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops

dev = torch.device('cuda')
g = torch.Generator().manual_seed(3)
what = sys.argv[1:] or ['check', 'time']


def both(N, C, Cout, S, dg, sigma, H=None):
    H = H or S
    x = torch.randn(N, C, H, S, generator=g).to(dev)
    off = (torch.randn(N, 18 * dg, H, S, generator=g) * sigma).to(dev)
    go = torch.randn(N, Cout, H, S, generator=g).to(dev)
    w = (torch.randn(Cout, C, 3, 3, generator=g) * 0.05).to(dev)
    return x, off, go, w


def timed(fn, reps=10):
    for _ in range(2):
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


if 'check' in what:
    for (N, C, Cout, S, dg, sigma, H) in ((3, 64, 64, 56, 2, 0.0, None), (3, 64, 64, 56, 2, 0.5, None), (3, 64, 64, 56, 2, 3.0, None),
                                          (2, 128, 128, 28, 2, 0.7, None), (2, 128, 128, 28, 2, 4.0, None), (2, 32, 64, 12, 2, 1.0, 10),
                                          (1, 16, 64, 8, 1, 1.0, 6), (2, 64, 128, 20, 4, 1.5, 14), (16, 64, 64, 56, 2, 0.3, None)):
        x, off, go, w = both(N, C, Cout, S, dg, sigma, H)
        ops.DCN_BWD_FUSED[0] = True               # (opt-in by default: DM_DCN_FUSED=1)
        assert ops.dcn_bwd_fused_ok(x.shape, Cout, dg), (x.shape, Cout, dg)
        ops.DCN_BWD_FUSED[0] = False
        gx0, goff0 = ops.deform_conv_backward_data(x, off, w, go, dg)
        ops.DCN_BWD_FUSED[0] = True
        gx1, goff1 = ops.deform_conv_backward_data(x, off, w, go, dg)
        gx2, goff2 = ops.deform_conv_backward_data(x, off, w, go, dg)
        torch.cuda.synchronize()
        print(f'N={N} C={C} Cout={Cout} {H or S}x{S} dg={dg} sigma={sigma}: grad_x max|diff| {float((gx1 - gx0).abs().max()):.3g} '
              f'(scale {float(gx0.abs().max()):.3g}), grad_offset {float((goff1 - goff0).abs().max()):.3g} (scale {float(goff0.abs().max()):.3g}); '
              f'run-to-run equal: {bool(torch.equal(gx1, gx2))} {bool(torch.equal(goff1, goff2))}', flush=True)
if 'time' in what:
    for (N, C, S, dg, sigma) in ((256, 64, 56, 2, 0.0), (256, 64, 56, 2, 0.5), (256, 64, 56, 2, 2.0), (256, 128, 28, 2, 0.0), (256, 128, 28, 2, 0.5)):
        x, off, go, w = both(N, C, C, S, dg, sigma)
        wf = ops.pack_dcn_bwd_weight(w, dg)
        wc = ops.pack_dcn_colgrad_weight(w)
        ops.DCN_BWD_FUSED[0] = False
        t0 = timed(lambda: ops.deform_conv_backward_data(x, off, w, go, dg, w_colgrad=wc))
        ops.DCN_BWD_FUSED[0] = True
        t1 = timed(lambda: ops.deform_conv_backward_data(x, off, w, go, dg, w_fused=wf))
        print(f'{N}x{C}x{S}x{S} sigma={sigma}: three kernels (one stream) {t0:.3f} ms, fused {t1:.3f} ms', flush=True)
