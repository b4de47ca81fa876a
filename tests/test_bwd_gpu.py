"""Parity of the backward kernels (C ABI) against autograd of the CPU oracle at the north star's gate,
atol = rtol = 1e-4.  One kind of output needs a second look: a weight gradient is a sum of 10^3..10^5
products, and where those cancel to almost nothing an fp32 sum -- in ANY order, the oracle's included --
is only good to a few ulps of the sum of the products' magnitudes, not of the result.  Such outputs are
checked against an fp64 oracle: within the gate, or within 8 ulps of that magnitude sum (`_close_sum`)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi
from oracle import ref_ops

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def _g(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def _dev(t):
    return t.detach().cuda().contiguous()


def _close(a, b, **kw):
    kw = kw or TOL
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), **kw)


def _close_sum(got, ref64, mag64, what=''):
    """got: fp32 sums from the kernel; ref64: the same sums in fp64; mag64: sum of |products| per output."""
    err = (got.detach().cpu().double() - ref64).abs()
    gate = 1e-4 + 1e-4 * ref64.abs()
    cond = 8 * 2.0 ** -24 * mag64
    bad = (err > torch.maximum(gate, cond))
    assert not bool(bad.any()), (what, int(bad.sum()), float(err.max()))
    return int((err > gate).sum())          # how many outputs needed the conditioning bound


@pytest.fixture(scope='module')
def ops():
    from dynamask_amd import ops as o
    return o


@pytest.mark.parametrize('N,Cs,Cout,S,ks', [
    (7, [256], 256, 14, 3), (5, [48], 36, 28, 3), (3, [64, 64, 2], 64, 56, 1), (6, [128], 62, 28, 1), (9, [24], 40, 7, 3),
])
def test_conv_wgrad_bias_and_data_grad(ops, N, Cs, Cout, S, ks):
    xs = [torch.randn(N, c, S, S, generator=_g(10 + i)) for i, c in enumerate(Cs)]
    cin = sum(Cs)
    w = (torch.randn(Cout, cin, ks, ks, generator=_g(20)) / (cin * ks * ks) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=_g(21)).requires_grad_(True)
    xcat = torch.cat(xs, 1).requires_grad_(True)
    y = F.relu(F.conv2d(xcat, w, b, padding=ks // 2))
    go = torch.randn(y.shape, generator=_g(22))
    y.backward(go)
    gy = _dev(go)
    ops.relu_backward_(gy, _dev(y))
    # weight / bias gradients against fp64 sums of the same fp32 inputs
    gyd, xd = gy.cpu().double(), xcat.detach().double()
    ref64 = torch.nn.grad.conv2d_weight(xd, tuple(w.shape), gyd, padding=ks // 2)
    mag64 = torch.nn.grad.conv2d_weight(xd.abs(), tuple(w.shape), gyd.abs(), padding=ks // 2)
    n_cond = _close_sum(ops.conv2d_wgrad(gy, [_dev(t) for t in xs], ks), ref64, mag64, 'wgrad')
    n_cond += _close_sum(ops.channel_sum(gy), gyd.sum((0, 2, 3)), gyd.abs().sum((0, 2, 3)), 'bias')
    print(f'wgrad {N}x{Cs}->{Cout}@{S} k{ks}: {n_cond} of {w.numel() + Cout} sums outside 1e-4 but within 8 ulp of their magnitude sum')
    assert n_cond <= 0.001 * w.numel()
    # data gradient: forward kernel with transposed / rotated weights, sources = [dy]
    gx = ops.conv2d(gy, ops.pack_conv_weight(_dev(w), transpose_flip=True), None, cin, ks)
    _close(gx, xcat.grad)


@pytest.mark.parametrize('shape', [(5, 6, 14, 14), (3, 1, 56, 56), (2, 3, 9, 13), (2, 2, 2, 2), (1, 2, 72, 72), (2, 1, 1, 5)])
def test_upsample_backward(ops, shape):
    """LDS-staged gathers (half-pixel and align_corners), the plain gather, and the scatter fallback for planes
    beyond 64 KB / one-pixel inputs; the output buffer is handed over uninitialised."""
    for ac, relu in ((False, True), (True, False), (True, True)):
        x = torch.randn(*shape, generator=_g(30), requires_grad=True)
        y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=ac)
        if relu:
            y = F.relu(y)
        go = torch.randn(y.shape, generator=_g(31))
        y.backward(go)
        gin = ops.upsample2x_backward(_dev(go), _dev(y) if relu else None, tuple(x.shape), align_corners=ac)
        _close(gin, x.grad)


def test_point_sample_backward(ops):
    hi = gi.head_inputs()
    feat = hi['feats'][2][:, :24].contiguous().requires_grad_(True)
    out = ref_ops.simple_roi_align(feat, hi['rois'], 14, 0.25)
    go = torch.randn(out.shape, generator=_g(40))
    out.backward(go)
    gf = ops.point_sample_backward(_dev(go), tuple(feat.shape), _dev(hi['rois']), 0.25)
    _close(gf, feat.grad)


@pytest.mark.parametrize('S', [14, 28, 56])
def test_point_sample_backward_corner_case_rois(ops, S):
    """The gather form's index runs (a cell's samples are two runs of consecutive lattice indices per axis) against
    autograd through the oracle: RoIs hanging over every border, larger than the map, one pixel wide, of zero width
    (every sample at one coordinate), inverted (coordinates decreasing in the index), off the map, a bad batch index;
    22 channels (a short last channel group)."""
    B, C, H, W = 2, 22, 25, 42
    feat = torch.randn(B, C, H, W, generator=_g(41)).requires_grad_(True)
    rois = torch.tensor([
        [0, 10.0, 8.0, 90.0, 70.0],
        [1, -30.0, -20.0, 40.0, 35.0],            # over the top-left corner
        [0, 140.0, 80.0, 200.0, 130.0],           # over the bottom-right corner
        [1, -50.0, -50.0, 300.0, 250.0],          # larger than the map
        [0, 60.0, 40.0, 61.0, 41.0],              # one image pixel
        [1, 77.3, 20.0, 77.3, 60.0],              # zero width: all samples of a row at one x
        [0, 30.0, 55.5, 90.0, 55.5],              # zero height
        [1, 120.0, 70.0, 40.0, 10.0],             # inverted
        [0, 400.0, 300.0, 500.0, 380.0],          # off the map
        [0, 3.999, 3.999, 4.001, 4.001],          # around a cell boundary: rounding decides the floor
        [1, 0.0, 0.0, 167.0, 99.0],               # the whole map
    ], dtype=torch.float32)
    rois = rois[torch.argsort(rois[:, 0], stable=True)]      # (the reference's SimpleRoIAlign returns rows grouped by image: bbox2roi order)
    out = ref_ops.simple_roi_align(feat, rois, S, 0.25)
    go = torch.randn(out.shape, generator=_g(42))
    out.backward(go)
    gf = ops.point_sample_backward(_dev(go), tuple(feat.shape), _dev(rois), 0.25)
    # a cell of this gradient sums up to ~S * S samples of every RoI over it, in another order than autograd's: the fp64
    # triangle (tests/tolerances.py) instead of round 3's flat atol = 2e-4
    from tolerances import assert_close_via_f64
    feat64 = feat.detach().double().requires_grad_(True)
    ref_ops.simple_roi_align(feat64, rois, S, 0.25).backward(go.double())
    assert_close_via_f64(gf, feat.grad, feat64.grad, f'point-sample adjoint S={S}')


def test_class_logits_backward(ops):
    N, C, S, nc = 7, 64, 28, 80
    x = torch.randn(N, C, S, S, generator=_g(50), requires_grad=True)
    wi = (torch.randn(nc, C, 1, 1, generator=_g(51)) / 8).requires_grad_(True)
    wd = (torch.randn(nc, C, 1, 1, generator=_g(52)) / 8).requires_grad_(True)
    bi = torch.randn(nc, generator=_g(53)).requires_grad_(True)
    bd = torch.randn(nc, generator=_g(54)).requires_grad_(True)
    labels = torch.tensor([3, 3, 79, 0, 42, 3, 60])
    ar = torch.arange(N)
    ip = F.conv2d(x, wi, bi)[ar, labels][:, None]
    dp = F.conv2d(x, wd, bd)[ar, labels][:, None]
    g1 = torch.randn(ip.shape, generator=_g(55))
    g2 = torch.randn(dp.shape, generator=_g(56))
    (ip * g1).sum().backward(retain_graph=True)
    (dp * g2).sum().backward()
    gx = torch.full((N, C, S, S), 1.0).cuda()
    gwi, gwd = torch.zeros(nc, C).cuda(), torch.zeros(nc, C).cuda()
    gbi, gbd = torch.zeros(nc).cuda(), torch.zeros(nc).cuda()
    ops.class_logits_backward(_dev(x), _dev(wi.view(nc, C)), _dev(wd.view(nc, C)), _dev(labels), _dev(g1), _dev(g2), gx,
                              True, gwi, gbi, gwd, gbd)
    _close(gx - 1.0, x.grad)
    _close(gwi, wi.grad.view(nc, C), atol=1e-4, rtol=1e-4)
    _close(gwd, wd.grad.view(nc, C), atol=1e-4, rtol=1e-4)
    _close(gbi, bi.grad, atol=1e-4, rtol=1e-4)
    _close(gbd, bd.grad, atol=1e-4, rtol=1e-4)


def test_class_logits_backward_slab_matches_atomics_with_shared_classes(ops):
    """The slab form of the parameter gradients (per-RoI sums, added per class in RoI order) against the atomic form: 1100
    RoIs -- more than one 1024-RoI tile of the reduce kernel -- over three classes, one of them empty in the second tile;
    both give the same sums (fp32 order differs: relative gate), and the slab form twice gives the same bits."""
    N, C, S, nc = 1100, 40, 14, 80
    g = _g(57)
    x = torch.randn(N, C, S, S, generator=g).cuda()
    wi, wd = torch.randn(nc, C, generator=g).cuda(), torch.randn(nc, C, generator=g).cuda()
    labels = torch.tensor([5, 17, 79])[torch.randint(0, 3, (N,), generator=g)]
    labels[1024:] = 17
    labels = labels.cuda()
    g1, g2 = torch.randn(N, 1, S, S, generator=g).cuda(), torch.randn(N, 1, S, S, generator=g).cuda()

    def run(slab):
        old = ops.CLB_SLAB[0]
        ops.CLB_SLAB[0] = slab
        try:
            gx = torch.empty(N, C, S, S, device='cuda')
            outs = [torch.zeros(nc, C, device='cuda'), torch.zeros(nc, device='cuda'), torch.zeros(nc, C, device='cuda'), torch.zeros(nc, device='cuda')]
            ops.class_logits_backward(x, wi, wd, labels, g1, g2, gx, False, *outs)
            torch.cuda.synchronize()
            return [gx] + outs
        finally:
            ops.CLB_SLAB[0] = old

    a, b, c = run(True), run(False), run(True)
    assert torch.equal(a[0], b[0])
    for t_slab, t_atom, t_again in zip(a[1:], b[1:], c[1:]):
        assert torch.equal(t_slab, t_again)
        scale = float(t_atom.abs().max())
        assert scale > 0 and float((t_slab - t_atom).abs().max()) <= 2e-5 * scale
        assert float(t_slab[[0, 1, 2, 78]].abs().max()) == 0.0        # classes without RoIs stay untouched


def test_sigmoid_backward(ops):
    logit = torch.randn(4, 1, 9, 9, generator=_g(60), requires_grad=True)
    s = logit.sigmoid()
    ga = torch.randn(s.shape, generator=_g(61))
    gb = torch.randn(s.shape, generator=_g(62))
    (s * (ga + gb)).sum().backward()
    wide = torch.zeros(4, 5, 9, 9).cuda()
    wide[:, 3:4] = _dev(s)
    gw = torch.zeros(4, 3, 9, 9).cuda()
    gw[:, 1:2] = _dev(ga)
    out = ops.sigmoid_backward(wide[:, 3:4], gw[:, 1:2], _dev(gb))
    _close(out, logit.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('N,C,S', [(5, 64, 14), (3, 32, 28), (2, 16, 9)])
def test_deform_conv_backward(ops, N, C, S):
    x = torch.randn(N, C, S, S, generator=_g(70), requires_grad=True)
    w = (torch.randn(C, C, 3, 3, generator=_g(71)) / (9 * C) ** 0.5).requires_grad_(True)
    off = (torch.randn(N, 36, S, S, generator=_g(72)) * 1.2)
    off[0, :, 0, 0] = 30.0
    off = off.requires_grad_(True)
    y = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2)
    go = torch.randn(y.shape, generator=_g(73))
    y.backward(go)
    gx, goff, gw = ops.deform_conv_backward(_dev(x), _dev(off), _dev(w), _dev(go), 2)
    _close(gx, x.grad, atol=1e-4, rtol=1e-4)
    _close(goff, off.grad, atol=1e-4, rtol=1e-4)
    _close(gw, w.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('N,cins,cout,H,W,ks', [(9, [64], 64, 14, 14, 1), (5, [64, 64, 2], 64, 28, 28, 1), (3, [48], 200, 10, 6, 3),
                                                (4, [64], 36, 28, 28, 3), (3, [40], 16, 14, 14, 3), (6, [128], 1152, 7, 8, 1),
                                                (2, [30], 64, 56, 56, 1)])
def test_weight_gradient_by_slab_reduce_is_reproducible_and_equals_the_atomic_sum(ops, N, cins, cout, H, W, ks):
    """dm_conv2d_wgrad_slab: split-K partial tiles written to slabs and added in index order.  Two runs give the same
    bits (the reference's weight gradient is a deterministic addmm_, deform_conv_cuda.cpp:460-465); the values are the
    atomic path's (dm_conv2d_wgrad) and torch's up to summation order; accumulation into an existing dw / db is kept."""
    dy = torch.randn(N, cout, H, W, generator=_g(300)).cuda()
    xs = [torch.randn(N, c, H, W, generator=_g(301 + i)).cuda() for i, c in enumerate(cins)]
    assert ops.WGRAD_SLAB[0]
    dw1, db1 = ops.conv2d_wgrad(dy, xs, ks, want_bias=True)
    dw2, db2 = ops.conv2d_wgrad(dy, xs, ks, want_bias=True)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    xc = torch.cat(xs, 1).double().cpu().requires_grad_(True)
    w = torch.zeros(cout, sum(cins), ks, ks, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xc, w, b, padding=ks // 2).backward(dy.double().cpu())
    from tolerances import assert_grad_close
    assert_grad_close(dw1, w.grad.float(), 'dw', rel=1e-4)
    assert_grad_close(db1, b.grad.float(), 'db', rel=1e-4)
    ops.WGRAD_SLAB[0] = False
    try:
        dwa, dba = ops.conv2d_wgrad(dy, xs, ks, want_bias=True)
    finally:
        ops.WGRAD_SLAB[0] = True
    sw, sb = float(dwa.abs().max()), float(dba.abs().max())           # same sums in another order: a few ulp of the largest value
    assert float((dw1 - dwa).abs().max()) <= 2e-6 * sw and float((db1 - dba).abs().max()) <= 2e-6 * sb
    # accumulate into existing tensors
    dw3, db3 = dw1.clone(), db1.clone()
    ops.conv2d_wgrad(dy, xs, ks, dw=dw3, db=db3)
    assert float((dw3 - 2 * dw1).abs().max()) <= 1e-6 * sw and float((db3 - 2 * db1).abs().max()) <= 1e-6 * sb


@pytest.mark.parametrize('N,C,S,sigma', [(2, 16, 56, 6.0), (3, 32, 28, 8.0), (2, 16, 14, 5.0), (1, 16, 56, 0.0)])
def test_deform_coord_grad_with_samples_far_from_their_pixel(ops, N, C, S, sigma):
    """The coordinate gradient stages a band of rows around a workgroup's pixels (DCN_COORD_HALO = 5 rows either side);
    offsets of many rows leave the band and take the global-memory path, samples at the image border read the band's
    zero rows / columns.  Checked against autograd of the oracle (deform_conv_cuda_kernel.cu:145-188, 338-408)."""
    x = torch.randn(N, C, S, S, generator=_g(170), requires_grad=True)
    w = torch.randn(C, C, 3, 3, generator=_g(171)) / (9 * C) ** 0.5
    off = (torch.randn(N, 36, S, S, generator=_g(172)) * sigma).requires_grad_(True)
    y = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2)
    go = torch.randn(y.shape, generator=_g(173))
    y.backward(go)
    gx, goff = ops.deform_conv_backward_data(_dev(x.detach()), _dev(off.detach()), _dev(w), _dev(go), 2)
    _close(goff, off.grad, atol=1e-4, rtol=1e-4)
    _close(gx, x.grad, atol=1e-4, rtol=1e-4)


def test_fixed_point_scatter_accumulators_propagate_non_finite_gradients(ops):
    """The LDS accumulators of the DCN col2im and the point-sample adjoint are 64-bit fixed point; a NaN or
    Inf gradient has no fixed-point value and must show up as NaN in the plane it belongs to (a diverged step
    has to be visible in the FPN / feature gradients), without touching the other planes."""
    g = _g(90)
    N, C, S = 2, 16, 14
    x = torch.randn(N, C, S, S, generator=g)
    off = torch.randn(N, 36, S, S, generator=g) * 0.5
    colgrad = torch.randn(N, 9 * C, S, S, generator=g)
    gx0, _ = ops.deform_col2im_coord(_dev(colgrad), _dev(x), _dev(off), 2)
    bad = colgrad.clone()
    bad[1, 4 * C + 5, 3, 3] = float('nan')          # tap 4, channel 5 of image 1
    bad[0, 2 * C + 9, 0, 0] = float('inf')
    gx1, _ = ops.deform_col2im_coord(_dev(bad), _dev(x), _dev(off), 2)
    assert torch.isnan(gx1[1, 5]).all() and torch.isnan(gx1[0, 9]).all()
    keep = torch.ones(N, C, dtype=torch.bool)
    keep[1, 5] = keep[0, 9] = False
    assert torch.equal(gx1.cpu()[keep], gx0.cpu()[keep])
    hi = gi.head_inputs()
    feat_shape = (2, 8, 64, 80)
    go = torch.randn(hi['rois'].shape[0], 8, 14, 14, generator=g)
    go[2, 3, 7, 7] = float('nan')
    gf = ops.point_sample_backward(_dev(go), feat_shape, _dev(hi['rois']), 0.25)
    b = int(hi['rois'][2, 0])
    assert torch.isnan(gf[b, 3]).any()
    others = [c for c in range(8) if c != 3]
    assert torch.isfinite(gf[:, others]).all()


@pytest.mark.parametrize('N,Cin,Cout,S,ks,acc', [(6, 64, 64, 56, 1, False), (5, 36, 128, 28, 3, True), (9, 256, 256, 14, 3, False),
                                                 (3, 30, 64, 56, 1, False), (40, 256, 256, 14, 3, False)])
def test_conv_with_fused_relu_adjoint_mask(ops, N, Cin, Cout, S, ks, acc):
    """dm_conv2d_fwd_masked = dm_conv2d_fwd followed by the ReLU mask pass, same bits (every tile shape, with and
    without accumulation, incl. the split last-round launch at 40 RoIs)."""
    x = torch.randn(N, Cin, S, S, generator=_g(95)).cuda()
    w = (torch.randn(Cout, Cin, ks, ks, generator=_g(96)) / (Cin * ks * ks) ** 0.5).cuda()
    y = torch.randn(N, Cout, S, S, generator=_g(97)).clamp(min=0).cuda()          # a ReLU output: half zeros
    base = torch.randn(N, Cout, S, S, generator=_g(98)).cuda()
    wq = ops.pack_conv_weight(w)
    ref = ops.conv2d(x, wq, None, Cout, ks, out=base.clone() if acc else None, accumulate=acc)
    ops.relu_backward_(ref, y)
    got = ops.conv2d(x, wq, None, Cout, ks, out=base.clone() if acc else None, accumulate=acc,
                     mask=y if acc else y)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('N,Cs,Cout,H,W', [(1, 32, 36, 8, 8), (3, 40, 16, 20, 12), (2, 64, 33, 9, 56), (5, 8, 5, 30, 64), (4, 96, 32, 14, 14)])
def test_narrow_wgrad_kernel_odd_shapes(ops, N, Cs, Cout, H, W):
    """conv_wgrad3_narrow_kernel away from the training shapes: one image, channel counts that are not multiples
    of 32 (a partly empty channel group), couts at 5 / 16 / 32 / 33 / 36 (with and without the 4-row tail), maps
    whose last row band is short, the narrowest and widest maps it takes."""
    x = torch.randn(N, Cs, H, W, generator=_g(85))
    gy = torch.randn(N, Cout, H, W, generator=_g(86))
    xd, gyd = x.double(), gy.double()
    ref64 = torch.nn.grad.conv2d_weight(xd, (Cout, Cs, 3, 3), gyd, padding=1)
    mag64 = torch.nn.grad.conv2d_weight(xd.abs(), (Cout, Cs, 3, 3), gyd.abs(), padding=1)
    n_cond = _close_sum(ops.conv2d_wgrad(_dev(gy), [_dev(x)], 3), ref64, mag64, 'narrow wgrad')
    assert n_cond <= 0.001 * ref64.numel() + 1
    # accumulation into an existing buffer at a column offset (concatenated sources)
    x2 = torch.randn(N, 8, H, W, generator=_g(87))
    dw = torch.zeros(Cout, Cs + 8, 3, 3).cuda()
    ops.conv2d_wgrad(_dev(gy), [_dev(x), _dev(x2)], 3, dw=dw)
    ref2 = torch.nn.grad.conv2d_weight(torch.cat([x, x2], 1).double(), (Cout, Cs + 8, 3, 3), gyd, padding=1)
    mag2 = torch.nn.grad.conv2d_weight(torch.cat([x, x2], 1).double().abs(), (Cout, Cs + 8, 3, 3), gyd.abs(), padding=1)
    _close_sum(dw, ref2, mag2, 'narrow wgrad, two sources')


@pytest.mark.parametrize('N,C,H,W', [(3, 5, 10, 6), (2, 7, 7, 9), (4, 16, 28, 28), (1, 3, 12, 20), (2, 2, 56, 56),
                                     (16, 4, 12, 20), (19, 3, 28, 28), (16, 3, 7, 9),
                                     (16, 64, 12, 20), (17, 64, 7, 12), (16, 64, 2, 6), (18, 64, 6, 2), (33, 64, 28, 28)])
def test_bn_relu_maxpool_forward_and_backward_odd_shapes(ops, N, C, H, W):
    """MaskPre's BatchNorm(train) -> ReLU -> MaxPool(3, 2, 1) block by itself, away from the 56 / 28 maps of the
    golden: non-square maps, odd sizes (the plane-in-LDS backward needs H * W % 4 == 0; 7 x 9 takes the other
    one), few channels; 16 images or more of 64 channels or more take the backward whose pooling adjoint also
    keeps BatchNorm's sums (2 x 2 blocks on even maps, the element form otherwise).  Against autograd of the same block (forward 1e-5, gradients at the 1e-4 gate)."""
    x = torch.randn(N, C, H, W, generator=_g(110)) * 1.5 + 0.3
    gamma = torch.rand(C, generator=_g(111)) + 0.5
    beta = torch.randn(C, generator=_g(112)) * 0.2
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.max_pool2d(F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)), 3, 2, 1)
    go = torch.randn(y.shape, generator=_g(113))
    y.backward(go)
    xd = _dev(x)
    mean, var = ops.bn_stats(xd)
    out = ops.bn_relu_maxpool(xd, mean, var, _dev(gamma), _dev(beta), 1e-5)
    _close(out, y, atol=1e-5, rtol=1e-5)
    gx, gg, gb = ops.bn_relu_maxpool_backward(xd, mean, var, _dev(gamma), _dev(beta), _dev(go), 1e-5)
    _close(gx, xr.grad)
    _close(gg, gr.grad)
    _close(gb, br.grad)


@pytest.mark.parametrize('N,C,H,W', [(2, 16, 10, 6), (1, 32, 20, 12), (3, 16, 7, 9)])
def test_deform_conv_backward_non_square(ops, N, C, H, W):
    x = torch.randn(N, C, H, W, generator=_g(120), requires_grad=True)
    w = (torch.randn(C, C, 3, 3, generator=_g(121)) / (9 * C) ** 0.5).requires_grad_(True)
    off = (torch.randn(N, 36, H, W, generator=_g(122)) * 1.5).requires_grad_(True)
    y = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2)
    go = torch.randn(y.shape, generator=_g(123))
    y.backward(go)
    gx, goff, gw = ops.deform_conv_backward(_dev(x), _dev(off), _dev(w), _dev(go), 2)
    _close(gx, x.grad)
    _close(goff, off.grad)
    _close(gw, w.grad)
