"""Parity of every HIP operator (through the C ABI) against the CPU oracle.
fp32 tolerance of BASELINE.json's north_star: atol = rtol = 1e-4; index /
integer-valued outputs bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi
from oracle import ref_model, ref_ops

pytestmark = pytest.mark.gpu

TOL = dict(atol=1e-4, rtol=1e-4)


def _g(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def _dev(t):
    return t.cuda().contiguous()


def _close(a, b, **kw):
    kw = kw or TOL
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(a, b, **kw)


@pytest.fixture(scope='module')
def ops():
    from dynamask_amd import ops as o
    return o


def _random_rois(n, B, img_h, img_w, seed):
    from dynamask_amd import synth
    per = (n + B - 1) // B
    return synth.make_rois(B, per, img_h, img_w, seed=seed)[:n] if B == 1 else synth.make_rois(B, per, img_h, img_w, seed=seed)


def test_roi_align_multilevel_golden_rois(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_head.npz'))
    hi = gi.head_inputs()
    feats = [_dev(f) for f in hi['feats'][:4]]
    out, lv = ops.roi_align(feats, _dev(hi['rois']), 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32], return_levels=True)
    assert np.array_equal(lv.cpu().numpy(), g['levels'])          # level indices bit-exact
    _close(out, g['ins_feats'])


def test_roi_align_random_rois_all_levels(ops):
    from dynamask_amd import synth
    feats = synth.make_fpn(2, 640, 800, 32, seed=7)
    rois = synth.make_rois(2, 40, 640, 800, seed=8, max_size=800.0)
    ref = ref_ops.single_roi_extractor(feats[:4], rois, 14, (4, 8, 16, 32))
    lv_ref = ref_ops.map_roi_levels(rois, 4)
    assert len(set(lv_ref.tolist())) == 4
    out, lv = ops.roi_align([_dev(f) for f in feats[:4]], _dev(rois), 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32],
                            return_levels=True)
    assert np.array_equal(lv.cpu().numpy(), lv_ref.numpy())
    _close(out, ref)


def test_roi_align_single_level_56_and_sampling_ratio(ops):
    from dynamask_amd import synth
    feats = synth.make_fpn(2, 256, 320, 16, seed=9)
    rois = torch.cat([synth.make_rois(2, 6, 256, 320, seed=10, max_size=400.0), gi.head_inputs()['rois']], 0)
    rois = rois[torch.argsort(rois[:, 0], stable=True)]
    ref = ref_ops.single_roi_extractor([feats[0]], rois, 56, (4,))
    out = ops.roi_align([_dev(feats[0])], _dev(rois), 56, [1 / 4])
    _close(out, ref)
    ref2 = ref_ops.roi_align(feats[1], rois, 7, 1 / 8, sampling_ratio=2)
    out2 = ops.roi_align([_dev(feats[1])], _dev(rois), 7, [1 / 8], sampling_ratio=2)
    _close(out2, ref2)
    # empty RoI set
    assert ops.roi_align([_dev(feats[0])], _dev(rois[:0]), 14, [1 / 4]).shape == (0, 16, 14, 14)


@pytest.mark.parametrize('P,levels,sr', [(56, 1, 0), (28, 4, 0), (56, 1, 2), (24, 1, 0)])
def test_roi_align_adjoint_in_gather_form_matches_autograd_of_the_oracle(ops, P, levels, sr):
    """roi_align_bwd_gather_kernel (16 < P <= 64; round 5: what lets MaskPre's conv1 run on the P2 map in training, its
    weight gradient taken through the adjoint of the 56 x 56 extraction): one workgroup per (RoI, channel quad), a thread
    per footprint cell, one float atomic per cell.  Against autograd of the oracle's RoIAlign
    (single_level_roi_extractor.py:53-81 / mmcv RoIAlign backward): RoIs from 8 to 640 pixels (sampling grids 1 .. 3 at
    P = 56 on stride 4), boxes over the border (void samples, clamped taps), zero-size and flipped boxes (no gradient), a
    fixed sampling ratio with samples more than a pixel apart (the kernel's per-tap path), overlapping RoIs."""
    from dynamask_amd import synth
    from tolerances import assert_grad_close
    C = 8
    feats = [f.requires_grad_(True) for f in synth.make_fpn(2, 160, 224, C, seed=31)[:levels]]
    rois = synth.make_rois(2, 14, 160, 224, seed=32, max_size=640.0)
    extra = torch.tensor([[0., -30., -20., 60., 50.], [0., 150., 100., 400., 300.], [1., 50., 50., 50., 50.],
                          [1., 90., 80., 40., 30.], [1., 0., 0., 223., 159.], [0., 10., 10., 18., 18.]])
    rois = torch.cat([rois, extra], 0)
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous()
    strides = (4, 8, 16, 32)[:levels]
    if levels == 1:
        ref = ref_ops.roi_align(feats[0], rois, P, 1 / 4, sampling_ratio=sr)
    else:
        ref = ref_ops.single_roi_extractor(feats, rois, P, strides)
    go = torch.randn(ref.shape, generator=_g(33))
    ref.backward(go)
    grads = ops.roi_align_backward(_dev(go), [tuple(f.shape) for f in feats], _dev(rois), P, [1.0 / s_ for s_ in strides],
                                   sampling_ratio=sr)
    for lvl, (gr, f) in enumerate(zip(grads, feats)):
        ref_g = f.grad if f.grad is not None else torch.zeros_like(f)
        assert_grad_close(gr, ref_g, f'level {lvl}', rel=1e-4, zero=not bool(ref_g.abs().max() > 0))


def test_roi_align_adjoint_dense_fixed_grid_and_non_finite_boxes(ops):
    """ADVICE r5 on roi_align_bwd_gather_kernel: (a) a FIXED sampling ratio on boxes a fraction of a pixel wide puts the
    samples closer than an ulp of their coordinate -- no monotone order to build the tables on: per-tap path, against
    autograd of the oracle; (b) boxes with Inf / NaN coordinates or of absurd size contribute no gradient and the launch
    terminates (the per-tap path would loop over an unbounded sampling grid)."""
    from dynamask_amd import synth
    from tolerances import assert_grad_close
    P, C = 56, 8
    feat = synth.make_fpn(1, 160, 224, C, seed=35)[0].requires_grad_(True)
    rois = torch.tensor([[0., 100.0, 60.0, 100.004, 60.004],         # 0.001 feature pixels wide: bin / g ~ 1e-5
                         [0., 40.0, 33.9999, 40.02, 34.0001],        # straddles a pixel boundary
                         [0., 10.0, 10.0, 90.0, 70.0]])              # an ordinary box beside them
    ref = ref_ops.roi_align(feat, rois, P, 1 / 4, sampling_ratio=2)
    go = torch.randn(ref.shape, generator=_g(36))
    ref.backward(go)
    gr = ops.roi_align_backward(_dev(go), [tuple(feat.shape)], _dev(rois), P, [0.25], sampling_ratio=2)[0]
    assert_grad_close(gr, feat.grad, 'dense fixed grid', rel=1e-4)
    inf, nan = float('inf'), float('nan')
    bad = torch.tensor([[0., 10., 10., inf, 50.], [0., -inf, 10., 50., 50.], [0., nan, 10., 50., 50.], [0., 0., 0., 3e9, 40.],
                        [0., 10.0, 10.0, 90.0, 70.0]])
    go2 = torch.randn(5, C, P, P, generator=_g(37))
    for sr in (0, 2):
        g_bad = ops.roi_align_backward(_dev(go2), [tuple(feat.shape)], _dev(bad), P, [0.25], sampling_ratio=sr)[0]
        g_ok = ops.roi_align_backward(_dev(go2[4:]), [tuple(feat.shape)], _dev(bad[4:]), P, [0.25], sampling_ratio=sr)[0]
        torch.cuda.synchronize()
        assert torch.isfinite(g_bad).all()
        assert_grad_close(g_bad, g_ok.cpu(), f'only the finite box contributes (sr={sr})', rel=1e-5)


def test_roi_align_backward(ops):
    from dynamask_amd import synth
    feats = [f.requires_grad_(True) for f in synth.make_fpn(2, 160, 224, 8, seed=11)[:4]]
    rois = synth.make_rois(2, 12, 160, 224, seed=12, max_size=400.0)
    ref = ref_ops.single_roi_extractor(feats, rois, 14, (4, 8, 16, 32))
    go = torch.randn(ref.shape, generator=_g(13))
    ref.backward(go)
    grads = ops.roi_align_backward(_dev(go), [tuple(f.shape) for f in feats], _dev(rois), 14,
                                   [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    from tolerances import assert_grad_close
    for lvl, (gr, f) in enumerate(zip(grads, feats)):
        ref_g = f.grad if f.grad is not None else torch.zeros_like(f)
        # (a level no RoI maps to: exact zeros on both sides)
        assert_grad_close(gr, ref_g, f'level {lvl}', rel=1e-4, zero=not bool(ref_g.abs().max() > 0))


@pytest.mark.parametrize('N,Cin,Cout,S,ks', [
    (7, 256, 256, 14, 3),     # instance convs
    (3, 64, 36, 56, 3),       # DCN offset conv, stage 2
    (5, 128, 36, 28, 3),      # DCN offset conv, stage 1
    (9, 48, 30, 7, 3),        # tiles spanning >2 images, ragged channels
    (7, 256, 126, 14, 1),     # fuse_transform_out
    (4, 64, 30, 56, 1),
    (2, 256, 64, 33, 1),      # odd spatial size
    (1, 40, 80, 5, 1),
])
def test_conv2d_single_source(ops, N, Cin, Cout, S, ks):
    x = torch.randn(N, Cin, S, S, generator=_g(20))
    w = torch.randn(Cout, Cin, ks, ks, generator=_g(21)) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=_g(22))
    for relu in (False, True):
        ref = F.conv2d(x, w, b, padding=ks // 2)
        if relu:
            ref = F.relu(ref)
        out = ops.conv2d(_dev(x), ops.pack_conv_weight(_dev(w)), _dev(b), Cout, ks, relu=relu)
        _close(out, ref)


def test_conv2d_fused_concat_and_channel_slice_output(ops):
    # fuse_conv[0]: cat[x(C), isf(C), sig(1), sig(1)] -> C ; written into a channel slice
    N, C, S = 6, 64, 28
    xs = [torch.randn(N, c, S, S, generator=_g(30 + i)) for i, c in enumerate((C, C, 1, 1))]
    w = torch.randn(C, 2 * C + 2, 1, 1, generator=_g(35)) / (2 * C) ** 0.5
    b = torch.randn(C, generator=_g(36))
    ref = F.relu(F.conv2d(torch.cat(xs, 1), w, b))
    out = torch.full((N, C + 2, S, S), -7.0).cuda()
    ops.conv2d([_dev(t) for t in xs], ops.pack_conv_weight(_dev(w), src_channels=[C, C, 1, 1]), _dev(b), C, 1, relu=True,
               out=out, out_ch_offset=1)
    _close(out[:, 1:C + 1], ref)
    assert torch.all(out[:, 0] == -7.0) and torch.all(out[:, C + 1] == -7.0)


@pytest.mark.parametrize('cins,Cout,S,nbig', [([256, 256, 1, 1], 256, 14, 256), ([256], 126, 14, 440), ([128, 128, 1, 1], 128, 28, 64)])
def test_conv1x1_rows_do_not_depend_on_launch_size(ops, cins, Cout, S, nbig):
    # up to 1.25 tiles of 128 x 128 per CU a 1x1 launch takes 128 x 32 tiles with 32-channel chunks, above that 128 x 128
    # tiles with 16-channel chunks (conv_igemm.hip: conv2d_launch): the same products in the same order -- the same bits
    xs = [(torch.randn(nbig, c, S, S, generator=_g(140 + i)) * 0.7).cuda() for i, c in enumerate(cins)]
    w = (torch.randn(Cout, sum(cins), 1, 1, generator=_g(145)) / sum(cins) ** 0.5).cuda()
    b = torch.randn(Cout, generator=_g(146)).cuda()
    wq = ops.pack_conv_weight(w, src_channels=cins)
    big = ops.conv2d(xs, wq, b, Cout, 1, relu=True)
    _close(big[:5], F.relu(F.conv2d(torch.cat([t[:5].cpu() for t in xs], 1), w.cpu(), b.cpu())))
    for n in (1, 3, 16, 50):
        assert torch.equal(ops.conv2d([t[:n].contiguous() for t in xs], wq, b, Cout, 1, relu=True), big[:n])


def test_conv2d_on_fpn_map(ops):
    x = torch.randn(2, 256, 40, 56, generator=_g(40)) * 0.5
    w = torch.randn(128, 256, 1, 1, generator=_g(41)) / 16
    b = torch.randn(128, generator=_g(42)) * 0.1
    out = ops.conv2d(_dev(x), ops.pack_conv_weight(_dev(w)), _dev(b), 128, 1, relu=True)
    _close(out, F.relu(F.conv2d(x, w, b)))


def test_conv2d_bwd_data_weights_packing(ops):
    # data gradient of a 3x3 conv == conv with transposed, 180-degree-rotated weights
    N, Cin, Cout, S = 3, 24, 40, 14
    x = torch.randn(N, Cin, S, S, generator=_g(45), requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=_g(46)) / 15
    y = F.conv2d(x, w, padding=1)
    go = torch.randn(y.shape, generator=_g(47))
    y.backward(go)
    gx = ops.conv2d(_dev(go), ops.pack_conv_weight(_dev(w), transpose_flip=True), None, Cin, 3)
    _close(gx, x.grad)


def test_point_sample(ops):
    hi = gi.head_inputs()
    feat = hi['feats'][2][:, :48].contiguous()      # P4
    for S, scale in ((14, 0.25), (28, 1 / 16)):
        ref = ref_ops.simple_roi_align(feat, hi['rois'], S, scale)
        out = ops.point_sample(_dev(feat), _dev(hi['rois']), S, scale)
        _close(out, ref)


def test_class_logits(ops):
    N, C, S, nc = 7, 128, 28, 80
    x = torch.randn(N, C, S, S, generator=_g(50))
    wi = torch.randn(nc, C, 1, 1, generator=_g(51)) / C ** 0.5
    wd = torch.randn(nc, C, 1, 1, generator=_g(52)) / C ** 0.5
    bi = torch.randn(nc, generator=_g(53))
    bd = torch.randn(nc, generator=_g(54))
    labels = torch.randint(0, nc, (N,), generator=_g(55))
    ar = torch.arange(N)
    ri = F.conv2d(x, wi, bi)[ar, labels][:, None]
    rd = F.conv2d(x, wd, bd)[ar, labels][:, None]
    sig = torch.zeros(N, 6, S, S).cuda()
    oi, od = ops.class_logits(_dev(x), _dev(wi.view(nc, C)), _dev(bi), _dev(wd.view(nc, C)), _dev(bd), _dev(labels),
                              sig_out=sig, sig_ch_offset=4)
    _close(oi, ri)
    _close(od, rd)
    _close(sig[:, 4:5], ri.sigmoid())
    _close(sig[:, 5:6], rd.sigmoid())


@pytest.mark.parametrize('N,C,H,W', [(7, 128, 14, 14), (3, 30, 7, 6), (130, 5, 2, 2), (1, 64, 28, 28)])
def test_class_logits_of_the_upsampled_stage_without_the_upsampled_tensor(ops, N, C, H, W):
    """dm_class_logits_up2x_fwd = F.interpolate(x2, bilinear, align_corners=False) -> relu -> the two class-gathered 1x1
    logits (dynamask_head.py:120-122 then :110-113 of the next stage), against torch and against the two-kernel path."""
    nc = 80
    x = torch.randn(N, C, H, W, generator=_g(56))
    wi = torch.randn(nc, C, 1, 1, generator=_g(57)) / C ** 0.5
    wd = torch.randn(nc, C, 1, 1, generator=_g(58)) / C ** 0.5
    bi, bd = torch.randn(nc, generator=_g(59)), torch.randn(nc, generator=_g(49))
    labels = torch.randint(0, nc, (N,), generator=_g(48))
    up = F.relu(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False))
    ar = torch.arange(N)
    ri = F.conv2d(up, wi, bi)[ar, labels][:, None]
    rd = F.conv2d(up, wd, bd)[ar, labels][:, None]
    args = (_dev(wi.view(nc, C)), _dev(bi), _dev(wd.view(nc, C)), _dev(bd), _dev(labels))
    oi, od = ops.class_logits_up2x(_dev(x), *args)
    assert tuple(oi.shape) == (N, 1, 2 * H, 2 * W)
    _close(oi, ri)
    _close(od, rd)
    # the two-kernel path: same interpolated values (dm_up2x_interp), same channel-sum order -> the same bits
    ti, td = ops.class_logits(ops.upsample2x(_dev(x), align_corners=False, relu=True), *args)
    assert torch.equal(oi, ti) and torch.equal(od, td)


def test_class_logits_up2x_refuses_odd_widths(ops):
    x = torch.randn(2, 4, 6, 7).cuda()
    assert not ops.class_logits_up2x_supported(x)
    w, b = torch.randn(3, 4).cuda(), torch.randn(3).cuda()
    with pytest.raises(Exception):
        ops.class_logits_up2x(x, w, b, w, b, torch.zeros(2, dtype=torch.int64).cuda())


@pytest.mark.parametrize('N,C,S', [(7, 256, 14), (5, 128, 28), (3, 64, 56), (2, 16, 9)])
def test_deform_conv(ops, N, C, S):
    x = torch.randn(N, C, S, S, generator=_g(60))
    w = torch.randn(C, C, 3, 3, generator=_g(61)) / (9 * C) ** 0.5
    off = torch.randn(N, 36, S, S, generator=_g(62)) * 1.5
    off[0, :, 0, 0] = 40.0          # far outside: all taps void
    off[-1, :, -1, -1] = -3.25
    ref = F.relu(ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2))
    out = ops.deform_conv(_dev(x), _dev(off), ops.pack_conv_weight(_dev(w)), C, 2, relu=True)
    _close(out, ref)
    # known answer: zero offsets == plain conv
    out0 = ops.deform_conv(_dev(x), torch.zeros_like(off).cuda(), ops.pack_conv_weight(_dev(w)), C, 2)
    _close(out0, F.conv2d(x, w, padding=1))


@pytest.mark.parametrize('N,C,S', [(3, 64, 56), (4, 128, 28), (2, 64, 32)])
def test_deform_conv_large_offsets_leave_the_lds_band(ops, N, C, S):
    """The 28 x 28 / 56 x 56 kernel gathers from a 16-row LDS band around the tile and sends samples that fall
    outside it to global memory: offsets of up to +-25 pixels make most taps of some pixels take that road,
    others (offset 0, +-5) stay inside, and the far corners clamp at every border."""
    x = torch.randn(N, C, S, S, generator=_g(66))
    w = torch.randn(C, C, 3, 3, generator=_g(67)) / (9 * C) ** 0.5
    off = torch.randn(N, 36, S, S, generator=_g(68)) * 8.0
    off[0, :18] = 0.0                                         # group 0 of image 0: plain taps
    off[1, 18:] = torch.randn(36 - 18, S, S, generator=_g(69)) * 2.0
    off[-1, ::2, S // 2] = 25.0                               # one row of pixels looks 25 rows down
    off[-1, ::2, S // 2 + 1] = -25.0
    ref = F.relu(ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2))
    out = ops.deform_conv(_dev(x), _dev(off), ops.pack_conv_weight(_dev(w)), C, 2, relu=True)
    _close(out, ref)


def test_deform_conv_rows_do_not_depend_on_launch_size(ops):
    # few RoIs run on 64 x 64 tiles, many on the 128 x 128 LDS-gather kernel (14 x 14) / 128 x 64 tiles:
    # the same bits either way (the early-exit path relies on rows being independent of the batch)
    for C, S in ((256, 14), (128, 28), (64, 56)):
        x = torch.randn(64, C, S, S, generator=_g(63)).cuda()
        off = (torch.randn(64, 36, S, S, generator=_g(64)) * 1.5).cuda()
        wq = ops.pack_conv_weight((torch.randn(C, C, 3, 3, generator=_g(65)) / (9 * C) ** 0.5).cuda())
        big = ops.deform_conv(x, off, wq, C, 2, relu=True)
        for n in (3, 16):
            assert torch.equal(ops.deform_conv(x[:n].contiguous(), off[:n].contiguous(), wq, C, 2, relu=True), big[:n])


def test_upsample2x(ops):
    x = torch.randn(5, 7, 14, 14, generator=_g(70))
    _close(ops.upsample2x(_dev(x), align_corners=False, relu=True),
           F.relu(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)), atol=1e-5, rtol=1e-5)
    y = torch.randn(4, 1, 56, 56, generator=_g(71))
    _close(ops.upsample2x(_dev(y), align_corners=True),
           F.interpolate(y, scale_factor=2, mode='bilinear', align_corners=True), atol=1e-5, rtol=1e-5)


def test_boundary_merge_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_merge.npz'))
    ips = [_dev(t.clone()) for t in gi.merge_inputs()['ips']]
    preds = ips[1:]
    for i in range(len(preds) - 1):
        ops.boundary_merge_(preds[i], preds[i + 1])
    _close(preds[-1], g['merged'], atol=1e-5, rtol=1e-5)


def test_boundary_merge_chain_has_the_bits_of_the_launches_it_replaces(ops, golden_dir):
    """dm_boundary_merge_chain (round 6): both merges of dynamask_roi_head.py:138-149 in one launch, with and without the
    final align_corners x2 upsample folded in -- against g5 (the reference's own merge) and bit for bit against
    dm_upsample2x_bilinear_fwd + dm_boundary_merge x 2, also at sizes where the bands are ragged."""
    g = np.load(os.path.join(golden_dir, 'g5_merge.npz'))
    ips = [_dev(t.clone()) for t in gi.merge_inputs()['ips']]
    p28, p56, p112 = ips[1], ips[2], ips[3]
    keep56 = p56.clone()
    out = ops.boundary_merge_chain(p28, p56, p112.clone())
    _close(out, g['merged'], atol=1e-5, rtol=1e-5)
    assert torch.equal(p56, keep56)                                             # the 2S logits are not written back
    seq56, seq112 = p56.clone(), p112.clone()
    ops.boundary_merge_(p28, seq56)
    ops.boundary_merge_(seq56, seq112)
    assert torch.equal(out, seq112)
    gen = torch.Generator().manual_seed(91)
    for S, n in ((28, 7), (14, 3), (7, 2), (5, 4)):
        a = _dev(torch.randn(n, 1, S, S, generator=gen) * 2)
        b = _dev(torch.randn(n, 1, 2 * S, 2 * S, generator=gen) * 2)
        fin = _dev(torch.randn(n, 1, 2 * S, 2 * S, generator=gen) * 2)
        a[0, 0, :2, :2] = 0.0                                                  # exact ties of the sigmoid threshold
        got = ops.boundary_merge_chain(a, b, fin)
        assert tuple(got.shape) == (n, 1, 4 * S, 4 * S)
        fine = ops.upsample2x(fin, align_corners=True)
        b2 = b.clone()
        ops.boundary_merge_(a, b2)
        ops.boundary_merge_(b2, fine)
        assert torch.equal(got, fine), S
        # ... and against the oracle's merge of the same three tensors
        ref = ref_model.boundary_merge([None, a.cpu(), b.cpu(), F.interpolate(fin.cpu(), scale_factor=2, mode='bilinear',
                                                                              align_corners=True)])
        bad = (got.cpu() - ref).abs() > 1e-4 + 1e-4 * ref.abs()
        assert bad.float().mean() < 2e-3, (S, float(bad.float().mean()))      # (threshold ties of |logit| ~ 1e-7 may flip a 3x3 block)
    assert ops.boundary_merge_chain(a[:0], b[:0], fin[:0]).shape[0] == 0


def test_dcn_with_chained_1x1_has_the_bits_of_the_two_launches(ops):
    """dm_deform_conv_tout_fwd (round 6): DCN 3x3 + ReLU + fuse_transform_out (1x1 + bias + ReLU) in one launch, the second
    GEMM on the accumulators in registers -- bit for bit the two launches, at the 28 x 28 / 128-channel and 56 x 56 /
    64-channel stage shapes with offsets that leave the LDS band, into a channel slice of a wider tensor, with and
    without keeping the DCN output; shapes the kernel does not take are refused."""
    gen = torch.Generator().manual_seed(94)
    for C, S, n in ((128, 28, 40), (64, 56, 12), (64, 56, 9)):
        m2 = C // 2 - 2
        x = _dev(torch.randn(n, C, S, S, generator=gen))
        off = torch.randn(n, 36, S, S, generator=gen) * 1.5
        off[0, :, :4, :4] = 9.0                                                     # far samples: the slow pass of the band kernel
        off = _dev(off)
        w = _dev(torch.randn(C, C, 3, 3, generator=gen) / (9 * C) ** 0.5)
        w2 = _dev(torch.randn(m2, C, 1, 1, generator=gen) / C ** 0.5)
        b2 = _dev(torch.randn(m2, generator=gen))
        wp, w2p, w2t = ops.pack_conv_weight(w), ops.pack_conv_weight(w2), ops.pack_tout_weight(w2)
        assert ops.deform_conv_tout_supported(x, C, m2)
        d_ref = ops.deform_conv(x, off, wp, C, 2, relu=True)
        t_ref = torch.full((n, C // 2, S, S), 7.0, device='cuda')
        ops.conv2d(d_ref, w2p, b2, m2, 1, relu=True, out=t_ref, out_ch_offset=0)
        for keep in (False, True):
            t = torch.full((n, C // 2, S, S), 7.0, device='cuda')
            d = ops.deform_conv_tout(x, off, wp, C, 2, w2t, b2, m2, t, keep_dcn=keep)
            assert torch.equal(t, t_ref), (C, S, keep)                               # incl. the two channels behind m2: untouched
            assert (d is None) if not keep else torch.equal(d, d_ref)
        buf = torch.empty_like(d_ref)                                                # relu(DCN) into a caller's buffer
        t = torch.full((n, C // 2, S, S), 7.0, device='cuda')
        assert ops.deform_conv_tout(x, off, wp, C, 2, w2t, b2, m2, t, dcn_out=buf) is buf
        assert torch.equal(buf, d_ref) and torch.equal(t, t_ref)
    xs = _dev(torch.randn(4, 64, 56, 56, generator=gen))                            # a handful of RoIs: another wave layout
    assert not ops.deform_conv_tout_supported(xs, 64, 30)
    assert not ops.deform_conv_tout_supported(_dev(torch.randn(40, 256, 14, 14, generator=gen)), 256, 126)
    with pytest.raises(RuntimeError, match='not supported'):
        ops.deform_conv_tout(xs, _dev(torch.zeros(4, 36, 56, 56)), ops.pack_conv_weight(_dev(torch.randn(64, 64, 3, 3, generator=gen))), 64, 2,
                             ops.pack_tout_weight(_dev(torch.randn(30, 64, 1, 1, generator=gen))), _dev(torch.zeros(30)), 30,
                             torch.zeros(4, 32, 56, 56, device='cuda'))


def test_grouped_1x1_convs_equal_their_own_launches(ops):
    """dm_conv1x1_group_fwd (round 6): the three FPN-wide semantic convolutions in one launch -- bit for bit what three
    dm_conv2d_fwd launches produce (256 -> 256 / 128 / 64 on maps of three sizes, two images), and the 2- and 1-problem forms."""
    gen = torch.Generator().manual_seed(93)
    xs = [_dev(torch.randn(2, 256, h, w, generator=gen)) for h, w in ((13, 21), (25, 42), (50, 84))]
    couts = [256, 128, 64]
    ws = [_dev(torch.randn(c, 256, 1, 1, generator=gen) / 16) for c in couts]
    bs = [_dev(torch.randn(c, generator=gen)) for c in couts]
    wq = [ops.pack_conv_weight(w) for w in ws]
    ref = [ops.conv2d(x, q, b, c, 1, relu=True) for x, q, b, c in zip(xs, wq, bs, couts)]
    for k in (3, 2, 1):
        got = ops.conv1x1_group(xs[:k], wq[:k], bs[:k], couts[:k], relu=True)
        for g_, r_ in zip(got, ref):
            assert torch.equal(g_, r_), k
    for x, w, b, r_ in zip(xs, ws, bs, ref):
        _close(r_, F.relu(F.conv2d(x.cpu(), w.cpu(), b.cpu())), atol=1e-4, rtol=1e-4)
    got = ops.conv1x1_group(xs[1:], wq[1:], [None, bs[2]], couts[1:], relu=False)           # a problem without bias
    assert torch.equal(got[0], ops.conv2d(xs[1], wq[1], None, 128, 1)) and torch.equal(got[1], ops.conv2d(xs[2], wq[2], bs[2], 64, 1))


def test_stage_head_launch_equals_point_sample_and_class_logits(ops):
    """dm_stage_head_fwd (round 6) = dm_point_sample_fwd + dm_class_logits_fwd in one launch: same bits, at the three
    stage shapes, with RoIs of two images, an out-of-range batch index and a label outside [0, classes)."""
    gen = torch.Generator().manual_seed(92)
    for C, S, Cs, H, W in ((256, 14, 256, 50, 84), (128, 28, 128, 100, 168), (64, 56, 64, 40, 60)):
        n = 9
        sem = _dev(torch.randn(2, Cs, H, W, generator=gen))
        x = _dev(torch.randn(n, C, S, S, generator=gen))
        rois = torch.cat([torch.randint(0, 2, (n, 1), generator=gen).float(), torch.rand(n, 2, generator=gen) * 100,
                          torch.rand(n, 2, generator=gen) * 200 + 100], 1)
        rois[3, 0] = 5.0                                                           # no such image: zero rows
        rois = _dev(rois)
        labels = torch.randint(0, 80, (n,), generator=gen)
        labels[1] = 93
        labels = labels.cuda()
        wi, bi, wd, bd = (_dev(torch.randn(80, C, generator=gen)), _dev(torch.randn(80, generator=gen)),
                          _dev(torch.randn(80, C, generator=gen)), _dev(torch.randn(80, generator=gen)))
        sig_a = torch.zeros(n, C // 2, S, S, device='cuda')
        sig_b = torch.zeros_like(sig_a)
        ps = ops.point_sample(sem, rois, S, 0.25)
        ip, dp = ops.class_logits(x, wi, bi, wd, bd, labels, sig_out=sig_a, sig_ch_offset=C // 2 - 2)
        ps2, ip2, dp2 = ops.stage_head(sem, rois, S, 0.25, x, wi, bi, wd, bd, labels, sig_out=sig_b, sig_ch_offset=C // 2 - 2)
        assert torch.equal(ps, ps2) and torch.equal(ip, ip2) and torch.equal(dp, dp2) and torch.equal(sig_a, sig_b)


def test_gumbel_selector_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_gumbel.npz'))
    logits = gi.gumbel_logits()
    torch.manual_seed(gi.GUMBEL_SEED)
    U = torch.rand(logits.shape)
    y, hot, idx = ops.gumbel_select(_dev(logits), _dev(U), 0.5)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), g['index'])      # selection indices bit-exact
    _close(hot, g['y_hard'], atol=1e-6, rtol=0)
    y_ref, _ = ref_model.gumbel_select(logits, U, 0.5)
    # y_hard of the reference = (one_hot - y).detach() + y  -> value == one_hot; soft part vs oracle
    _close(y, F.softmax((logits + (-torch.log(-torch.log(U + 1e-20) + 1e-20))) / 0.5, -1), atol=1e-5, rtol=1e-4)
    # U == 0.5 everywhere -> index = argmax(logits)   (SURVEY 8c known answer v)
    _, _, idx2 = ops.gumbel_select(_dev(logits), torch.full_like(logits, 0.5).cuda(), 0.5)
    assert np.array_equal(idx2.cpu().numpy(), logits.argmax(-1).numpy())


def test_detail_target_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g1_losses.npz'))
    li = gi.loss_inputs()
    for i in range(4):
        out = ops.detail_target(_dev(li['targets'][i]))
        assert np.array_equal(out.cpu().numpy()[:, None], g[f'detail_target{i}'])   # bit-exact {0,1}


def test_mask_loss_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g1_losses.npz'))
    li = gi.loss_inputs()
    ip, dp, t = li['ips'][1], li['dps'][1], li['targets'][1]
    w = li['mask_labels'][:, 1].contiguous()
    dt = ref_model.detail_target(t).squeeze(1)
    n_el = ip.numel()
    # golden eps-BCE was taken against the instance target itself
    sums0, _, _, _ = ops.mask_loss(_dev(ip), _dev(dp), _dev(t), _dev(t), _dev(w), need_grad=False)
    _close(sums0[0] / n_el, g['bce_stage1'], atol=1e-5, rtol=1e-4)
    _close(sums0[1] / n_el, g['epsbce_stage1'], atol=1e-5, rtol=1e-4)
    sums, per_roi, gi_, gd_ = ops.mask_loss(_dev(ip), _dev(dp), _dev(t), _dev(dt), _dev(w))
    _close(sums[1] / n_el, ref_model.mask_cross_entropy(dp.squeeze(1), dt, w.view(-1, 1, 1)), atol=1e-5, rtol=1e-4)
    # the stage's sum against the per-RoI sums added in float64 (round 3: atol 1e-3 on a sum of ~1e4)
    s64 = float((per_roi.double() * _dev(w).double()).sum())
    assert abs(float(sums[1]) - s64) <= 2e-6 * abs(s64), (float(sums[1]), s64)
    ipr = ip.clone().requires_grad_(True)
    dpr = dp.clone().requires_grad_(True)
    (ref_model.binary_cross_entropy(ipr.squeeze(1), t) * n_el).backward()
    (ref_model.mask_cross_entropy(dpr.squeeze(1), dt, w.view(-1, 1, 1)) * n_el).backward()
    _close(gi_, ipr.grad, atol=1e-5, rtol=1e-4)
    _close(gd_, dpr.grad, atol=1e-5, rtol=1e-4)


def test_deconv_and_carafe(ops):
    x = torch.relu(torch.randn(3, 256, 14, 14, generator=_g(80)))
    w = torch.randn(256, 256, 2, 2, generator=_g(81)) / 16
    b = torch.randn(256, generator=_g(82)) * 0.1
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2))
    out = ops.deconv2x2(_dev(x), ops.pack_deconv_weight(_dev(w)), _dev(b), 256, relu=True)
    _close(out, ref)
    enc = torch.randn(3, 100, 14, 14, generator=_g(83))
    mask = F.pixel_shuffle(enc, 2)
    mask = F.softmax(mask.view(3, 1, 25, 28, 28), dim=2).view(3, 25, 28, 28)
    ref2 = ref_ops.carafe_reassemble(x, mask, 5, 1, 2)
    _close(ops.carafe(_dev(x), _dev(enc), 5, 1, 2), ref2)


def test_ops_refuse_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.upsample2x(torch.zeros(1, 1, 4, 4))


# ------------------------------------------------------------------ K19 RLE (8f rank 2)
def test_rle_encode_canvas_matches_oracle_and_round_trips(ops):
    rng = np.random.default_rng(3)
    cases = [np.zeros((5, 7), np.uint8), np.ones((5, 7), np.uint8), (rng.random((37, 53)) < 0.5).astype(np.uint8),
             (rng.random((64, 64)) < 0.02).astype(np.uint8)]
    blob = np.zeros((300, 417), np.uint8)
    yy, xx = np.mgrid[:300, :417]
    blob[((yy - 140) / 90.0) ** 2 + ((xx - 200) / 150.0) ** 2 < 1] = 1
    blob[100:120, 180:260] = 0
    cases.append(blob)
    for m in cases:
        got = ops.rle_encode(torch.from_numpy(m[None]).cuda())
        ref = ref_ops.rle_encode(m)
        assert got[0] == ref, (m.shape, got[0]['counts'][:40], ref['counts'][:40])
        assert (ref_ops.rle_decode(got[0]) == m).all()
    # a batch: segments > 1 per mask (4096-pixel segments), packed offsets
    batch = np.stack([(rng.random((130, 97)) < p).astype(np.uint8) for p in (0.0, 0.3, 1.0, 0.9, 0.01)])
    got = ops.rle_encode(torch.from_numpy(batch).cuda())
    for g, m in zip(got, batch):
        assert g == ref_ops.rle_encode(m)
    assert ops.rle_encode(torch.zeros((0, 4, 4), dtype=torch.uint8, device='cuda')) == []


def test_paste_rle_equals_rle_of_pasted_canvas(ops):
    g = torch.Generator().manual_seed(5)
    N, S, H, W = 9, 28, 211, 307
    masks = torch.randn(N, 1, S, S, generator=g) * 3
    ctr = torch.rand(N, 2, generator=g) * torch.tensor([W, H])
    wh = torch.rand(N, 2, generator=g) * 150 + 4
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    boxes[0] = torch.tensor([-20.0, -30.0, 80.0, 60.0])        # sticks out of the image
    boxes[1] = torch.tensor([10.0, 10.0, 10.0, 40.0])           # zero width
    canvas = ops.paste_masks(masks.cuda(), boxes.cuda(), H, W, 0.5, apply_sigmoid=True)
    via_canvas = ops.rle_encode(canvas)
    fused = ops.paste_rle(masks.cuda(), boxes.cuda(), H, W, 0.5, apply_sigmoid=True)
    assert fused == via_canvas
    cv = canvas.cpu().numpy()
    for n in range(N):
        assert (ref_ops.rle_decode(fused[n]) == cv[n]).all()
        assert fused[n] == ref_ops.rle_encode(cv[n])


def test_roi_align_tile_kernel_corner_cases(ops):
    """Every route through the LDS tile kernel (P = 14 and 7): merged stencils g = 1..4, the
    run-time grid (clipped slivers, g up to 15), RoIs hanging over every border, one-pixel and
    inverted RoIs, a footprint too large for the buffer (single level, whole map -> global path),
    channel counts that leave a short last batch."""
    from dynamask_amd import synth
    feats = synth.make_fpn(2, 800, 1344, 24, seed=21)
    r = [
        [0, 10.0, 10.0, 26.0, 26.0],            # g = 1, level 0
        [0, 100.0, 50.0, 190.0, 150.0],         # g = 2
        [1, 300.0, 100.0, 410.0, 260.0],        # g = 3 on level 0 (tall)
        [0, 5.0, 0.0, 20.0, 799.0],             # sliver: 800 tall, 15 wide -> level 0, g = 15
        [1, 0.0, 700.0, 1343.0, 712.0],         # flat sliver along the bottom border
        [0, -40.0, -30.0, 60.0, 45.0],          # hangs over the top-left corner
        [1, 1300.0, 760.0, 1400.0, 850.0],      # hangs over the bottom-right corner
        [0, 500.0, 300.0, 500.5, 300.5],        # sub-pixel RoI
        [1, 600.0, 400.0, 590.0, 390.0],        # inverted (negative size) -> empty grid
        [0, 0.0, 0.0, 1343.0, 799.0],           # whole image (level 3)
        [1, 200.0, 100.0, 1000.0, 700.0],       # large, level 3
        [0, 64.0, 64.0, 175.0, 175.0],          # just below the level-1 threshold
        [1, 64.0, 64.0, 176.5, 176.5],          # just above it
    ]
    rois = torch.tensor(r, dtype=torch.float32)
    fd = [_dev(f) for f in feats[:4]]
    for P in (14, 7):
        ref = ref_ops.single_roi_extractor(feats[:4], rois, P, (4, 8, 16, 32))
        out = ops.roi_align(fd, _dev(rois), P, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
        _close(out, ref)
    # single level + huge RoIs: footprint > 2048 pixels -> direct global path inside the same kernel
    ref = ref_ops.roi_align(feats[0], rois, 14, 1 / 4, sampling_ratio=0)
    out = ops.roi_align([fd[0]], _dev(rois), 14, [1 / 4])
    _close(out, ref)
    # fixed sampling ratio whose samples are more than a pixel apart -> per-sample path in LDS
    ref = ref_ops.roi_align(feats[1], rois, 14, 1 / 8, sampling_ratio=2)
    out = ops.roi_align([fd[1]], _dev(rois), 14, [1 / 8], sampling_ratio=2)
    _close(out, ref)
    # banded kernel (16 < P <= 64): 56x56 on P2 as the semantic extractor does (huge RoIs -> many bands,
    # the whole-image RoI is too wide for a one-row band -> global path), 28x28 multi-level, fixed ratio
    ref = ref_ops.roi_align(feats[0], rois, 56, 1 / 4, sampling_ratio=0)
    out = ops.roi_align([fd[0]], _dev(rois), 56, [1 / 4])
    _close(out, ref)
    ref = ref_ops.single_roi_extractor(feats[:4], rois, 28, (4, 8, 16, 32))
    out = ops.roi_align(fd, _dev(rois), 28, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    _close(out, ref)
    ref = ref_ops.roi_align(feats[2], rois, 20, 1 / 16, sampling_ratio=3)
    out = ops.roi_align([fd[2]], _dev(rois), 20, [1 / 16], sampling_ratio=3)
    _close(out, ref)
    # 20 channels: 5 quads -> the last batch of a workgroup is short; 6 channels: C % 4 != 0 -> old kernel
    for C in (20, 6):
        f2 = [f[:, :C].contiguous() for f in feats[:4]]
        ref = ref_ops.single_roi_extractor(f2, rois, 14, (4, 8, 16, 32))
        out = ops.roi_align([_dev(f) for f in f2], _dev(rois), 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
        _close(out, ref)


def test_conv3x3_tail_couts_vs_torch(ops):
    """Cout in (32, 36]: 32 couts on 32x32x2 MFMA tiles + the last <= 4 on v_mfma_f32_4x4x1
    (the DCN offset convs); ragged pixel counts, bias + ReLU, a concat of two sources."""
    g = torch.Generator().manual_seed(17)
    for cout, cins, S, nb in ((36, [64], 14, 5), (33, [24, 16], 9, 3), (35, [8], 28, 2), (36, [256], 14, 7)):
        xs = [torch.randn(nb, c, S, S, generator=g) for c in cins]
        w = torch.randn(cout, sum(cins), 3, 3, generator=g) / (9 * sum(cins)) ** 0.5
        b = torch.randn(cout, generator=g)
        ref = F.relu(F.conv2d(torch.cat(xs, 1), w, b, padding=1))
        wq = ops.pack_conv_weight(_dev(w), src_channels=cins)
        out = ops.conv2d([_dev(x) for x in xs], wq, _dev(b), cout, 3, relu=True)
        _close(out, ref)


def test_roi_level_adversarial_sweep_around_every_threshold(ops):
    """FPN level indices must be bit-exact (north star): RoIs whose scale sits within +-64 ulps of every
    level threshold sqrt(w*h)/56 + 1e-6 = 2^k, on both sides, against the reference formula evaluated by
    torch on the CPU (floor(log2(.)), single_level_roi_extractor.py:32-51) and against a float64 ideal
    of the same fp32 input.  Note what the sweep pins: just below 8 (and 16) the correctly rounded fp32
    log2 already returns 3.0 (4.0), so the level changes ONE ulp below the power of two."""
    feats = [torch.zeros(1, 4, 128 >> l, 128 >> l) for l in range(4)]
    for L in (4, 3):            # DM_MAX_LEVELS = 4; L = 3 also checks the clamp at the top level
        boxes, side = [], []
        for k in range(1, L):
            w = np.float32(56.0 * 2 ** k)
            # h with sqrt(w*h)/56 + 1e-6 ~ 2^k, then +-64 ulps of h (each ulp of h moves t by ~half an ulp)
            h0 = np.float32((np.float64(2.0 ** k) - 1e-6) ** 2 * 56.0 ** 2 / np.float64(w))
            h = h0
            for _ in range(64):
                h = np.nextafter(h, np.float32(0))
            for _ in range(129):
                boxes.append([0.0, 3.0, 5.0, 3.0 + float(w), 5.0 + float(h)])
                h = np.nextafter(h, np.float32(1e9))
        # 3 + w and 5 + h round, so x2 - x1 / y2 - y1 are not exactly w / h: the expectation is recomputed from
        # the differences the kernel will actually see
        rois = torch.tensor(boxes, dtype=torch.float32)
        ref = ref_ops.map_roi_levels(rois, L)
        ww = (rois[:, 3] - rois[:, 1]).numpy()
        hh = (rois[:, 4] - rois[:, 2]).numpy()
        t = (np.sqrt(ww * hh) / np.float32(56.0) + np.float32(1e-6)).astype(np.float32)
        ideal = np.clip(np.floor(np.log2(t.astype(np.float64)).astype(np.float32)), 0, L - 1).astype(np.int64)
        np.testing.assert_array_equal(ref.numpy(), ideal)              # torch's CPU log2 is correctly rounded here
        _, lv = ops.roi_align([_dev(f) for f in feats[:L]], _dev(rois), 7, [1 / 4 / 2 ** l for l in range(L)],
                              return_levels=True)
        np.testing.assert_array_equal(lv.cpu().numpy().astype(np.int64), ref.numpy())
        for k in range(1, L):                                          # the sweep straddles every threshold
            seg = ref.numpy()[(k - 1) * 129:k * 129]
            assert seg.min() == k - 1 and seg.max() == k, (L, k, seg.min(), seg.max())


def test_roi_align_on_the_reference_tests_own_roi(ops):
    """tests/test_models/test_roi_extractor.py:40 of the reference (shape-only there): that RoI on those FPN
    shapes, 7x7, sampling_ratio 2 and adaptive, multi-level extraction vs the oracle and the f64 brute force."""
    rois = torch.tensor([[0.0000, 587.8285, 52.1405, 886.2484, 341.5644]])
    g = torch.Generator().manual_seed(40)
    feats = [torch.rand(1, 8, 200 >> l, 336 >> l, generator=g) for l in range(4)]
    for sr in (2, 0):
        out, lv = ops.roi_align([_dev(f) for f in feats], _dev(rois), 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], sampling_ratio=sr,
                                return_levels=True)
        assert lv.tolist() == [2]
        ref = ref_ops.single_roi_extractor(feats, rois, 7, (4, 8, 16, 32), sampling_ratio=sr)
        _close(out, ref)
        bf = ref_ops.roi_align_bruteforce_f64(feats[2], rois, 7, 1 / 16, sr, True)
        _close(out, bf.float(), atol=1e-5, rtol=1e-5)


def test_deform_conv_integer_shifts_at_every_border(ops):
    """Whole-pixel offsets: DCN == conv of the shifted, zero-extended image, borders included (both DCN kernels:
    the LDS-gather build at 14x14 and the global-gather build at 28x28)."""
    import torch.nn.functional as F
    for S, C in ((14, 32), (28, 16)):
        g = torch.Generator().manual_seed(41)
        x = torch.randn(3, C, S, S, generator=g)
        w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
        wq = ops.pack_conv_weight(_dev(w))
        for dh, dw in ((-1, 0), (1, 0), (0, -1), (0, 1), (2, 2), (-2, -3), (S, 0), (0, -S)):
            off = torch.zeros(3, 36, S, S)
            off[:, 0::2] = float(dh)
            off[:, 1::2] = float(dw)
            out = ops.deform_conv(_dev(x), _dev(off), wq, C, 2)
            pad = S + 2
            shifted = torch.roll(F.pad(x, (pad, pad, pad, pad)), shifts=(-dh, -dw), dims=(2, 3))
            exp = F.conv2d(shifted, w, padding=1)[:, :, pad:pad + S, pad:pad + S]
            _close(out, exp, atol=1e-5, rtol=1e-5)


def test_fc_rows_do_not_depend_on_batch_size_and_match_torch(ops):
    """dm_fc_fwd: a row of the output has the same bits whether 8 or 512 rows share the launch and from run
    to run (the K split is a function of K alone, partial sums are added in a fixed order) -- the selector
    logits of an RoI, hence its exit, must not depend on how many RoIs are in the batch."""
    g = torch.Generator().manual_seed(8)
    for K, M, relu in ((3136, 512, True), (12544, 1024, True), (1024, 81, False), (512, 4, False)):
        x = torch.randn(512, K, generator=g)
        w = torch.randn(M, K, generator=g) / K ** 0.5
        b = torch.randn(M, generator=g)
        xd, wd, bd = _dev(x), _dev(w), _dev(b)
        full = ops.fc(xd, wd, bd, relu=relu)
        assert torch.equal(full, ops.fc(xd, wd, bd, relu=relu))                       # run to run
        for n in (1, 8, 200):
            assert torch.equal(full[:n], ops.fc(xd[:n].contiguous(), wd, bd, relu=relu)), (K, M, n)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        ref = ref.relu() if relu else ref
        _close(full, ref.float(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('N,C,Cout,H,W', [(2, 16, 64, 16, 64), (3, 48, 128, 40, 28), (2, 64, 64, 33, 32), (1, 32, 128, 17, 60),
                                          (70, 16, 64, 24, 28)])
def test_deform_conv_band_kernel_odd_shapes(ops, N, C, Cout, H, W):
    """Shapes that take the 28x28 / 56x56 kernel but are neither: the minimum height (band = whole map), the widest
    map, non-square maps whose tiles start mid-row, C != Cout, 8 channels per deformable group, and a batch large
    enough for the many-RoI build (70 x 24 x 28) next to the few-RoI one."""
    x = torch.randn(N, C, H, W, generator=_g(80))
    w = torch.randn(Cout, C, 3, 3, generator=_g(81)) / (9 * C) ** 0.5
    off = torch.randn(N, 36, H, W, generator=_g(82)) * 2.5
    off[0, :, 0, :] = 9.0                       # first row looks 9 rows / columns away: the slow pass
    ref = F.relu(ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2))
    out = ops.deform_conv(_dev(x), _dev(off), ops.pack_conv_weight(_dev(w)), Cout, 2, relu=True)
    _close(out, ref)


# ------------------------------------------------------------------ split-K entry points (round 4)
@pytest.mark.parametrize('N,cins,cout,S,ks', [(3, [256], 256, 14, 3), (9, [256, 256, 2], 256, 14, 1), (5, [256], 36, 14, 3),
                                              (2, [128], 128, 28, 3), (1, [64], 64, 56, 3), (4, [256], 126, 14, 1)])
def test_conv2d_split_k_vs_oracle_and_unsplit(ops, N, cins, cout, S, ks):
    """dm_conv2d_fwd_ws: a launch of few workgroups splits its K loop (bare sums to a workspace, fixed-order reduce with
    bias + ReLU).  Against F.conv2d, against the unsplit launch (another association of the same products: 1e-5 of the
    scale), the same bits on every run, concat sources, the 36-cout tail build and an out_ch_offset destination."""
    srcs = [torch.randn(N, c, S, S, generator=_g(700 + i)) for i, c in enumerate(cins)]
    cin = sum(cins)
    w = torch.randn(cout, cin, ks, ks, generator=_g(710)) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, generator=_g(711))
    ref = F.relu(F.conv2d(torch.cat(srcs, 1), w, b, padding=ks // 2))
    wq = ops.pack_conv_weight(_dev(w), src_channels=cins)
    ds = [_dev(t) for t in srcs]
    plain = ops.conv2d(ds, wq, _dev(b), cout, ks, relu=True)
    assert int(ops.lib().dm_conv2d_splitk_floats(N, S, S, cout, ks)) > 0
    with ops.splitk_scope():
        a = ops.conv2d(ds, wq, _dev(b), cout, ks, relu=True)
        a2 = ops.conv2d(ds, wq, _dev(b), cout, ks, relu=True)
        wide = torch.full((N, cout + 3, S, S), -7.0).cuda()
        ops.conv2d(ds, wq, _dev(b), cout, ks, relu=True, out=wide, out_ch_offset=2)
    _close(a, ref)
    assert torch.equal(a, a2)
    assert torch.equal(wide[:, 2:2 + cout], a) and float(wide[:, :2].max()) == -7.0 and float(wide[:, -1].min()) == -7.0
    scale = float(plain.abs().max())
    assert float((a - plain).abs().max()) <= 1e-5 * max(scale, 1.0)


@pytest.mark.parametrize('N,C,S', [(3, 256, 14), (16, 256, 14), (5, 128, 28), (2, 64, 56)])
def test_deform_conv_split_k_vs_oracle_and_unsplit(ops, N, C, S):
    """dm_deform_conv_fwd_ws: the three DCN forward kernels with a channel range per split (calls of up to 24 RoIs)."""
    x = torch.randn(N, C, S, S, generator=_g(720))
    w = torch.randn(C, C, 3, 3, generator=_g(721)) / (9 * C) ** 0.5
    off = torch.randn(N, 36, S, S, generator=_g(722)) * 1.5
    off[0, :, 0, 0] = 30.0
    ref = F.relu(ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2))
    wq = ops.pack_conv_weight(_dev(w))
    plain = ops.deform_conv(_dev(x), _dev(off), wq, C, 2, relu=True)
    assert int(ops.lib().dm_deform_conv_splitk_floats(N, C, S, S, C)) > 0
    with ops.splitk_scope():
        a = ops.deform_conv(_dev(x), _dev(off), wq, C, 2, relu=True)
        a2 = ops.deform_conv(_dev(x), _dev(off), wq, C, 2, relu=True)
    _close(a, ref)
    assert torch.equal(a, a2)
    assert not torch.equal(a, plain), 'the call did not split: the test exercises nothing'
    scale = float(plain.abs().max())
    assert float((a - plain).abs().max()) <= 1e-5 * max(scale, 1.0)


# ------------------------------------------------------------------ the ordered RoI walk (default from 192 RoIs on)
@pytest.mark.parametrize('P,n', [(14, 192), (14, 300), (14, 513), (14, 1024), (7, 260), (14, 1)])
def test_roi_align_ordered_walk_gives_the_bits_and_levels_of_the_unordered_kernel(ops, P, n):
    """dm_roi_align_fwd_ws (the default for 14 x 14 extractions of 192 .. 1024 RoIs): roi_order_kernel ranks the RoIs by
    (image, level, position) and the tile kernel walks them in that order, writing every RoI's rows where
    dm_roi_align_fwd writes them.  Compared with the UNORDERED kernel (ops.ROI_WORKSPACE = False -- ADVICE r4: the old test
    compared the ordered path with itself): the same bits and levels, with duplicate boxes (equal keys but for the index),
    zero-size boxes, boxes outside the image, NaN boxes and images beyond the four the key has bits for.  P = 7 never
    takes the ordered path (P * P < 128): it must still honour the workspace argument."""
    from dynamask_amd import synth
    B = 6
    feats = [_dev(f) for f in synth.make_fpn(B, 160, 224, 32, seed=40)[:4]]
    per = max(1, (n + B - 1) // B)
    rois = synth.make_rois(B, per, 160, 224, seed=41)[:n].clone()
    g = _g(42)
    if n >= 16:
        dup = torch.randint(0, n, (n // 8,), generator=g)
        rois[dup] = rois[(dup + 3) % n]                                   # duplicates (in other rows)
        rois[5, 3:] = rois[5, 1:3]                                        # zero size
        rois[6, 1:] = torch.tensor([-500., -400., -300., -350.])          # outside the image
        rois[7, 1:] = torch.tensor([float('nan'), 10., 50., 60.])         # NaN corner
        rois[8, 1:] = float('nan')
    rois = rois.contiguous()
    scales = (0.25, 0.125, 0.0625, 0.03125)
    was = (ops.ROI_WORKSPACE, ops.ROI_WORKSPACE_MIN)
    try:
        ops.ROI_WORKSPACE = False
        ref, lv_ref = ops.roi_align(feats, _dev(rois), P, scales, return_levels=True)
        ops.ROI_WORKSPACE, ops.ROI_WORKSPACE_MIN = True, 1
        import os
        os.environ['DM_ROI_SORT_MIN'] = '1'
        ops.lib().dm_reload_env_knobs()
        out, lv = ops.roi_align(feats, _dev(rois), P, scales, return_levels=True)
        out2, _ = ops.roi_align(feats, _dev(rois), P, scales, return_levels=True)
    finally:
        os.environ.pop('DM_ROI_SORT_MIN', None)
        ops.lib().dm_reload_env_knobs()
        ops.ROI_WORKSPACE, ops.ROI_WORKSPACE_MIN = was
    assert torch.equal(lv, lv_ref)
    nan_ref = torch.isnan(ref)
    assert torch.equal(torch.isnan(out), nan_ref)
    assert torch.equal(out.masked_fill(nan_ref, 0.), ref.masked_fill(nan_ref, 0.))
    assert torch.equal(out2.masked_fill(nan_ref, 0.), out.masked_fill(nan_ref, 0.))
    if n >= 16:
        good = torch.ones(n, dtype=torch.bool)
        good[[7, 8]] = False
        assert torch.isfinite(out[good.to(out.device)]).all()


def test_split_k_entry_points_with_a_workspace_smaller_than_they_ask_for(ops):
    """The C ABI takes whatever workspace the caller owns: with room for two splits instead of eight a call splits in two,
    with room for less than two it runs unsplit -- the result stays the oracle's either way."""
    import ctypes
    from dynamask_amd.ops import _p, _ptr_array, _int_array, _stream
    N, C, S = 4, 256, 14
    x = torch.randn(N, C, S, S, generator=_g(730)).cuda()
    w = torch.randn(C, C, 3, 3, generator=_g(731)) / (9 * C) ** 0.5
    b = torch.randn(C, generator=_g(732)).cuda()
    ref = F.relu(F.conv2d(x.cpu(), w, b.cpu(), padding=1))
    wq = ops.pack_conv_weight(w.cuda())
    per = N * C * S * S
    strides = (ctypes.c_longlong * 1)(int(x.stride(0)))
    for room in (2 * per, per + 7, 0):
        out = torch.empty_like(x)
        ws = torch.empty((max(room, 1),), device='cuda')
        rc = ops.lib().dm_conv2d_fwd_ws(_ptr_array([x]), _int_array([C]), strides, 1, N, S, S, _p(wq), _p(b), C, 3, 1, _p(out), C, 0,
                                        _p(ws) if room else None, room, _stream())
        assert rc == 0
        _close(out, ref)
    off = (torch.randn(N, 36, S, S, generator=_g(733)) * 1.2).cuda()
    dref = F.relu(ref_ops.deform_conv2d(x.cpu(), off.cpu(), w, 1, 1, 1, 2))
    for room in (2 * per, per + 7, 0):
        out = torch.empty_like(x)
        ws = torch.empty((max(room, 1),), device='cuda')
        rc = ops.lib().dm_deform_conv_fwd_ws(_p(x), _p(off), N, C, S, S, _p(wq), C, 2, 1, _p(out), _p(ws) if room else None, room,
                                             _stream())
        assert rc == 0
        _close(out, dref)
