"""Parity at BASELINE.json's full sizes (512 RoIs, 1333x800 FPN) through
size-independent properties, plus edge cases (empty / single / ragged RoI sets)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi
from tolerances import assert_grad_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dynamask_amd import ops as o
    return o


@pytest.fixture(scope='module')
def full():
    from dynamask_amd import synth
    dev = torch.device('cuda')
    feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
    rois = synth.make_rois(1, 512, 800, 1333, seed=1).to(dev)
    labels = synth.make_labels(512, seed=2).to(dev)
    return feats, rois, labels


def _head():
    from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors, synth  # noqa: F401
    cfg = dict(type='DynaMaskRoIHead',
               mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG))
    m = registry.build_head(cfg)
    m.load_state_dict({**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6)})
    return m.cuda().eval()


def test_roialign_constant_and_level_partition_full_size(ops, full):
    feats, rois, _ = full
    const = [torch.full_like(f, 3.25) for f in feats[:4]]
    out, lv = ops.roi_align(const, rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32], return_levels=True)
    assert torch.allclose(out, torch.full_like(out, 3.25), atol=1e-5)        # average of a constant map
    assert set(lv.cpu().tolist()) == {0, 1, 2, 3}
    # linearity in the features
    a = ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    b = ops.roi_align([2.0 * f for f in feats[:4]], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    assert torch.allclose(b, 2.0 * a, atol=1e-5, rtol=1e-5)


def test_zero_offset_dcn_equals_conv_kernel_full_size(ops):
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(512, 256, 14, 14, device='cuda', generator=g)
    w = torch.randn(256, 256, 3, 3, device='cuda', generator=g) / 48
    wq = ops.pack_conv_weight(w)
    y_conv = ops.conv2d(x, wq, None, 256, 3)
    y_dcn = ops.deform_conv(x, torch.zeros(512, 36, 14, 14, device='cuda'), wq, 256, 2)
    # two different kernels, same fp32 fma order per output (K walked tap-major, channel quads)
    assert torch.allclose(y_conv, y_dcn, atol=1e-5, rtol=1e-5)
    # and both agree with MIOpen's conv on a slice (independent implementation)
    ref = F.conv2d(x[:16], w, padding=1)
    assert torch.allclose(y_conv[:16], ref, atol=1e-4, rtol=1e-4)
    # 512 RoIs = 3.06 rounds of 128 x 128 tiles: the last 2048 pixels go to a second launch with
    # 128 x 32 tiles.  Same products in the same order: rows do not depend on where the split falls
    assert torch.allclose(y_conv[-16:], F.conv2d(x[-16:], w, padding=1), atol=1e-4, rtol=1e-4)
    assert torch.equal(y_conv[-16:], ops.conv2d(x[-16:].contiguous(), wq, None, 256, 3))
    assert torch.equal(y_conv[:336], ops.conv2d(x[:336].contiguous(), wq, None, 256, 3))     # 2.01 rounds: other split
    with ops.overlapped_streams():      # scheduling hint (flag bit 3): one launch, same bits
        assert torch.equal(y_conv, ops.conv2d(x, wq, None, 256, 3))
    assert torch.equal(y_dcn[-16:], ops.deform_conv(x[-16:].contiguous(), torch.zeros(16, 36, 14, 14, device='cuda'), wq, 256, 2))


def test_head_is_roi_permutation_equivariant_and_stream_invariant(full):
    feats, rois, labels = full
    m = _head()
    with torch.no_grad():
        m.num_streams = 1
        one = m._mask_forward(feats, rois, labels)
        m.num_streams = 2
        two = m._mask_forward(feats, rois, labels)
        perm = torch.randperm(512, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4))
        m.num_streams = 1
        shuf = m._mask_forward(feats, rois[perm].contiguous(), labels[perm].contiguous())
    for k in ('stage_instance_preds', 'stage_detail_preds'):
        for i in range(4):
            assert torch.equal(one[k][i], two[k][i])                          # chunking over streams changes nothing
            assert torch.allclose(one[k][i][perm], shuf[k][i], atol=1e-5, rtol=1e-5)
    assert one['stage_instance_preds'][3].shape == (512, 1, 112, 112)
    assert all(torch.isfinite(t).all() for t in one['stage_instance_preds'])


def test_edge_cases_empty_single_and_ragged_rois():
    hi = gi.head_inputs()
    m = _head()
    feats = [f.cuda() for f in hi['feats']]
    with torch.no_grad():
        # no detections: the reference guards this in simple_test_mask (dynamask_roi_head.py:122-123)
        res = m.simple_test_mask(feats, [dict(ori_shape=(256, 320, 3), scale_factor=1.0)],
                                 torch.zeros(0, 5).cuda(), torch.zeros(0, dtype=torch.long).cuda())
        assert len(res) == 80 and all(len(r) == 0 for r in res)
        assert m.mask_roi_extractor(feats[:4], torch.zeros(0, 5).cuda()).shape == (0, 256, 14, 14)
        # a single RoI, and RoIs that only live in the second image of the batch
        r1 = m._mask_forward(feats, hi['rois'][:1].cuda(), hi['labels'][:1].cuda())
        sel = hi['rois'][:, 0] == 1
        r2 = m._mask_forward(feats, hi['rois'][sel].cuda().contiguous(), hi['labels'][sel].cuda().contiguous())
        full_ = m._mask_forward(feats, hi['rois'].cuda(), hi['labels'].cuda())
    assert torch.allclose(r1['stage_instance_preds'][3], full_['stage_instance_preds'][3][:1], atol=1e-5)
    assert torch.allclose(r2['stage_detail_preds'][2], full_['stage_detail_preds'][2][sel.cuda()], atol=1e-5)


def test_dynamic_inference_full_size_rows_identical_to_fixed_path(full):
    """512 RoIs, exits spread over the four resolutions: every RoI's logits at its own exit equal
    the all-exits path bit for bit (RoIs never interact), whatever the bucket sizes.  (Split-K of the launches that
    leave the chip idle -- here the 1x1 convolutions on the FPN maps -- is off: it is an inference-entry-point choice
    that depends on the launch's size, tests/test_path_gpu.py ``no_splitk``.)"""
    from dynamask_amd import ops as _ops
    was = _ops.CONV_SPLITK[0]
    _ops.CONV_SPLITK[0] = False
    try:
        _dynamic_rows_identical(full)
    finally:
        _ops.CONV_SPLITK[0] = was


def _dynamic_rows_identical(full):
    feats, rois, labels = full
    m = _head()
    m.num_streams = 1
    exits = (torch.arange(512) * 7 + 3) % 4
    with torch.no_grad():
        ref = m._mask_forward(feats, rois, labels)['stage_instance_preds']
        res = m.dynamic_mask_logits(feats, rois[:, 1:].contiguous(), labels, merge=False, exits=exits)
    order = res['order'].cpu().tolist()
    assert res['n_ge'] == [int((exits >= k).sum()) for k in range(4)]
    bad = 0
    for p, j in enumerate(order):
        e = int(exits[j])
        bad += int(not torch.equal(res['preds'][e][p], ref[e][j]))
    assert bad == 0


def test_paste_rle_round_trip_full_size(ops):
    """100 detections on a 1333x800 canvas: decode(device RLE) == pasted bitmap, fused == canvas form."""
    from oracle import ref_ops
    g = torch.Generator().manual_seed(11)
    N, S, H, W = 100, 112, 800, 1333
    masks = torch.randn(N, 1, S, S, generator=g) * 2 + 0.5
    ctr = torch.rand(N, 2, generator=g) * torch.tensor([W, H])
    wh = torch.rand(N, 2, generator=g) * 400 + 10
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    canvas = ops.paste_masks(masks.cuda(), boxes.cuda(), H, W, 0.5, apply_sigmoid=True)
    fused = ops.paste_rle(masks.cuda(), boxes.cuda(), H, W, 0.5, apply_sigmoid=True)
    assert fused == ops.rle_encode(canvas)
    cv = canvas.cpu().numpy()
    for n in (0, 17, 99):
        assert (ref_ops.rle_decode(fused[n]) == cv[n]).all()
    # total foreground area from the run lengths == popcount of the canvas
    area = [sum(ref_ops.rle_from_string(r['counts'])[1::2]) for r in fused]
    assert area == cv.reshape(N, -1).sum(1).tolist()


def test_nms_idempotent_and_separated_full_size(ops):
    """5000 boxes: the kept set is NMS-stable (running NMS on it keeps everything) and every
    suppressed box overlaps some higher-scored kept box by more than the threshold."""
    g = torch.Generator().manual_seed(12)
    n = 5000
    ctr = torch.rand(n, 2, generator=g) * torch.tensor([1333.0, 800.0])
    wh = torch.rand(n, 2, generator=g) * 200 + 8
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda()
    scores = torch.rand(n, generator=g).cuda()
    dets, keep = ops.nms(boxes, scores, 0.5)
    assert (dets[1:, 4] <= dets[:-1, 4]).all()                     # descending scores
    d2, k2 = ops.nms(dets[:, :4].contiguous(), dets[:, 4].contiguous(), 0.5)
    assert k2.tolist() == list(range(len(keep)))                   # idempotent
    kb = boxes[keep]
    def iou(a, b):
        lt = torch.maximum(a[:, None, :2], b[None, :, :2]); rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        aa = (a[:, 2:] - a[:, :2]).prod(-1); ab = (b[:, 2:] - b[:, :2]).prod(-1)
        return inter / (aa[:, None] + ab[None] - inter)
    mask = torch.ones(n, dtype=torch.bool, device='cuda'); mask[keep] = False
    sup = boxes[mask]
    m = iou(sup, kb)
    higher = scores[mask][:, None] <= scores[keep][None]
    assert ((m > 0.5) & higher).any(1).all()


# ---------------------------------------------------------------------------------------------
# BASELINE configs[2] and configs[4] at their real sizes against the oracle (these two
# were only timed, never checked).
# ---------------------------------------------------------------------------------------------
def test_config2_training_step_at_full_size_matches_oracle():
    """configs[2]: training step of the mask path, 2 images x 128 positive RoIs on 1333x800 FPN maps,
    dynamic 14/28/56/112 selection + BCE backward.  Loss, selector indices (bit-exact, with the top-2
    margins they had) and a gradient slice -- parameters late in the graph, so that the CPU oracle's
    autograd stays cheap -- against the oracle."""
    from dynamask_amd import synth
    from oracle import ref_model
    B, per = 2, 128
    feats = synth.make_fpn(B, 800, 1333, 256, seed=10)
    rois = synth.make_rois(B, per, 800, 1333, seed=11)
    labels = synth.make_labels(B * per, seed=12)
    targets = synth.make_targets(B * per, seed=13)
    noise = synth.make_gumbel_noise(B * per, seed=14)
    sd = {**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6)}
    keys = ['mask_head.stages.2.fuse_transform_out.weight', 'mask_head.final_instance_logits.weight',
            'mask_head.final_detail_logits.bias', 'mask_predictor.fc2.weight', 'mask_predictor.fc1.bias',
            'mask_predictor.bn2.weight', 'mask_predictor.bn2.bias']
    # Parameters in front of a max-pool: of the 26 M pooling windows of a step a few have two taps equal to the
    # last fp32 bit; which one wins (and so which PIXEL receives the gradient) depends on how BatchNorm's affine was
    # rounded -- torch CPU, torch GPU and this kernel each round it their own way.  So the test (1) reads the
    # device's choice of tap per window (dm_bn_relu_maxpool_argmax), (2) counts the RoIs in which any choice
    # differs from the oracle's own arg-max and checks that every such window is a genuine tie in the oracle's
    # activations (its two taps closer than the forward gate of 1e-4), (3) runs the oracle's backward with the device's
    # choices and compares these parameters at the same 1e-4 gate as all others.
    flip_keys = ['mask_predictor.conv2.weight', 'mask_predictor.bn1.weight']
    from dynamask_amd import ops as _ops
    choice = {}
    orig_pool = _ops.bn_relu_maxpool

    def spy(x, mean, var, gamma, beta, eps=1e-5):
        choice['pool%d' % (len(choice) + 1)] = _ops.bn_relu_maxpool_argmax(x, mean, var, gamma, beta, eps).cpu()
        return orig_pool(x, mean, var, gamma, beta, eps)
    m = _head()
    m.load_state_dict(sd, strict=True)
    m.train()
    _ops.bn_relu_maxpool = spy
    try:
        res = m._mask_forward_train([f.cuda() for f in feats], rois.cuda(), labels.cuda(), [t.cuda() for t in targets],
                                    noise=noise.cuda())
    finally:
        _ops.bn_relu_maxpool = orig_pool
    assert sorted(choice) == ['pool1', 'pool2']
    loss = res['loss_mask']['loss_masks']
    loss.backward()
    torch.cuda.synchronize()
    sdo = {k: (v.clone().requires_grad_(True) if k in keys + flip_keys else v) for k, v in sd.items()}
    record = {}
    loss_ref, _, ind_ref, logits_ref = ref_model.mask_forward_train(sdo, feats, rois, labels, targets, noise,
                                                                     pool_choice=choice, pool_record=record)
    loss_ref.backward()
    # (2) windows whose tap differs from the oracle's own arg-max: few RoIs, and each a tie in the oracle's z
    flipped_rois, flipped_windows, all_windows = set(), 0, 0
    for name in ('pool1', 'pool2'):
        z, own = record[name]
        dev_idx = choice[name].long()
        diff = (dev_idx != own).nonzero()
        zf = z.flatten(2)
        a = zf.gather(2, dev_idx.flatten(2)).view_as(own)[dev_idx != own]
        b = zf.gather(2, own.flatten(2)).view_as(own)[dev_idx != own]
        # "tied" = closer than the forward parity gate: the device's and the oracle's activations themselves agree
        # only to 1e-4 (conv1 sums 256 products per pixel in another order), so two taps closer than that have no
        # defined order
        gap = (b - a).abs()
        assert bool((gap <= 1e-4 + 1e-4 * b.abs()).all()), \
            f'{name}: a device choice is not among the window maxima (worst gap {float(gap.max()):.3e})'
        flipped_rois |= set(diff[:, 0].tolist())
        flipped_windows += len(diff)
        all_windows += own.numel()
        print(f'{name}: {len(diff)} of {own.numel()} windows take another tap (worst gap between the two taps '
              f'{float(gap.max()) if len(diff) else 0.0:.2e}), in {len(set(diff[:, 0].tolist()))} RoIs')
    # measured: 59 of 26.5 M windows (2e-6), spread over 45 of the 256 RoIs (each RoI has 103 k windows)
    assert flipped_windows <= 1e-5 * all_windows, f'{flipped_windows} windows take another tap ({len(flipped_rois)} RoIs)'

    # selector: indices bit-exact; report how decisive the choices were
    assert torch.equal(res['mask_index'].cpu().long(), ind_ref.long())
    y = (logits_ref.detach() - torch.log(-torch.log(noise + 1e-20) + 1e-20)) / 0.5
    top2 = y.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    print(f'selector: 256/256 indices equal; top-2 margin min {float(margin.min()):.3e}, median {float(margin.median()):.3e}; '
          f'exit histogram {torch.bincount(ind_ref, minlength=4).tolist()}')
    np.testing.assert_allclose(res['mask_logits'].detach().cpu().numpy(), logits_ref.detach().numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(float(loss.detach()), float(loss_ref.detach()), atol=1e-4, rtol=1e-4)
    named = dict(m.named_parameters())
    for k in keys:
        assert_grad_close(named[k].grad, sdo[k].grad, k)
    for k in flip_keys:          # (3) same gate as everything else once the oracle routes through the device's taps
        assert_grad_close(named[k].grad, sdo[k].grad, k)


def test_config4_carafe_head_on_2048x1024_maps_matches_oracle(ops):
    """configs[4]: FCNMaskHead + CARAFE on Cityscapes-shaped 2048x1024 input (P2 = 256x512), 512 RoIs.
    RoIAlign14 and the head against the oracle on the 48 RoIs with the largest footprints plus the first
    16 (the LDS tile kernel's fallbacks live at the large end); at the full 512, rows do not depend on
    the batch they were computed in (bit for bit) and every output is finite."""
    from dynamask_amd import registry, mask_heads, synth  # noqa: F401
    from oracle import ref_model, ref_ops
    feats = synth.make_fpn(1, 1024, 2048, 256, seed=20)
    rois = synth.make_rois(1, 510, 1024, 2048, seed=21)
    rois = torch.cat([rois, torch.tensor([[0, 0.0, 0.0, 2047.0, 1023.0], [0, 100.0, 50.0, 1900.0, 1000.0]])])   # whole image
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    lv = ref_ops.map_roi_levels(rois, 4)
    foot = area / (4.0 * 2 ** lv.float()) ** 2                       # footprint in feature pixels of its level
    sel = torch.unique(torch.cat([torch.argsort(foot, descending=True)[:48], torch.arange(16)]))
    fd = [f.cuda() for f in feats[:4]]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    full = ops.roi_align(fd, rois.cuda(), 14, scales)
    sub = ops.roi_align(fd, rois[sel].contiguous().cuda(), 14, scales)
    assert torch.equal(full[sel.cuda()], sub)
    ref_ins = ref_ops.single_roi_extractor(feats[:4], rois[sel], 14, (4, 8, 16, 32))
    np.testing.assert_allclose(sub.cpu().numpy(), ref_ins.numpy(), atol=1e-4, rtol=1e-4)
    assert set(lv[sel].tolist()) >= {2, 3} and float(foot[sel].max()) > 2000          # 64 x 32 level-3 pixels: the whole image
    for up in ('carafe', 'deconv'):
        cfg = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
        cfg.pop('loss_mask')
        if up == 'carafe':
            cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                       encoder_dilation=1, compressed_channels=64)
        fsd = synth.init_fcn_head_state(seed=7, upsample=up, test_mode=True)
        hsd = {k[len('mask_head.'):]: v for k, v in fsd.items()}
        head = registry.build_head(cfg)
        head.load_state_dict(hsd, strict=True)
        head = head.cuda().eval()
        with torch.no_grad():
            out_full = head(full)
            out_sub = head(sub)
        assert out_full.shape == (512, 80, 28, 28) and bool(torch.isfinite(out_full).all())
        assert torch.equal(out_full[sel.cuda()], out_sub)
        with torch.no_grad():
            ref = ref_model.fcn_mask_head_forward(hsd, ref_ins, upsample=up)
        np.testing.assert_allclose(out_sub.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=1e-4, err_msg=up)
