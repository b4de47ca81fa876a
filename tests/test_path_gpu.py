"""End-to-end parity of the mask-head path (registry-built modules -> C ABI ->
HIP kernels) against golden vectors produced by the reference's own modules and
against the CPU oracle.  atol = rtol = 1e-4 fp32; selection indices bit-exact."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import ref_model, ref_ops
from tolerances import assert_grad_close

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def _close(a, b, **kw):
    kw = kw or TOL
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(a, b, **kw)


def _dev(t):
    return t.cuda().contiguous()


def _roi_head(train=False):
    from dynamask_amd import registry, roi_head  # noqa: F401  (registers the classes)
    from dynamask_amd import losses, mask_heads, roi_extractors  # noqa: F401
    cfg = dict(type='DynaMaskRoIHead',
               mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG),
               train_cfg=registry.ConfigDict(flops=[0.23, 0.62, 1.01, 1.4], Lambda=0.3, mask_size=28),
               test_cfg=registry.ConfigDict(mask_thr_binary=0.5))
    m = registry.build_head(cfg)
    m.load_state_dict({**gi.head_state(), **gi.mask_pre_state()}, strict=True)
    m = m.cuda()
    m.train(train)
    return m


def test_mask_forward_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_head.npz'))
    hi = gi.head_inputs()
    m = _roi_head()
    with torch.no_grad():
        res = m._mask_forward([_dev(f) for f in hi['feats']], _dev(hi['rois']), _dev(hi['labels']))
    for i in range(4):
        _close(res['stage_instance_preds'][i], g[f'ip{i}'])
        _close(res['stage_detail_preds'][i], g[f'dp{i}'])


def test_fixed_28_exit_matches_full_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_head.npz'))
    hi = gi.head_inputs()
    m = _roi_head()
    with torch.no_grad():
        res = m._mask_forward([_dev(f) for f in hi['feats']], _dev(hi['rois']), _dev(hi['labels']), last_stage=1)
    assert len(res['stage_instance_preds']) == 2
    _close(res['stage_instance_preds'][1], g['ip1'])
    _close(res['stage_detail_preds'][1], g['dp1'])


def test_inference_merge_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_merge.npz'))
    m = _roi_head()
    merged = m.merge_stage_preds([_dev(t.clone()) for t in gi.merge_inputs()['ips']])
    _close(merged, g['merged'], atol=1e-4, rtol=1e-4)


def test_simple_test_mask_logits_vs_oracle():
    hi = gi.head_inputs()
    m = _roi_head()
    sel = hi['rois'][:, 0] == 0
    boxes = hi['rois'][sel][:, 1:]
    labels = hi['labels'][sel]
    with torch.no_grad():
        out = m.simple_test_mask_logits([_dev(f) for f in hi['feats']], _dev(boxes), _dev(labels))
        sd = gi.head_state()
        rois = torch.cat([torch.zeros(len(boxes), 1), boxes], 1)
        ips, _ = ref_model.mask_forward(sd, hi['feats'], rois, labels)
        ref = ref_model.boundary_merge(ips)
    # The merge thresholds sigmoid >= 0.5 on logits that differ by ~1e-6 between CPU and GPU: a logit within that of 0
    # may fall on either side and move its 3x3 boundary block.  Instead of tolerating a fraction of the pixels (round 3:
    # 1e-3 of them), the ties are PROVEN: the oracle's merge is run with the threshold at -1e-4, 0 and +1e-4 on the
    # logit; where the three agree no tie is involved and the product must agree to 1e-4; everywhere else it must agree
    # with one of the three.
    import torch.nn.functional as F

    def merge_thr(stage_preds, thr):
        preds = [p.clone() for p in stage_preds[1:]]
        for idx in range(len(preds) - 1):
            inst = preds[idx].squeeze(1) >= thr
            nb = (ref_model.generate_block_target(inst, boundary_width=1) != 1).unsqueeze(1)
            nb = F.interpolate(nb.float(), preds[idx + 1].shape[-2:], mode='bilinear', align_corners=True) >= 0.5
            pre_pred = F.interpolate(preds[idx], preds[idx + 1].shape[-2:], mode='bilinear', align_corners=True)
            preds[idx + 1][nb] = pre_pred[nb]
        return preds[-1]
    lo, hi_ = merge_thr(ips, -1e-4), merge_thr(ips, 1e-4)
    got = out.cpu()

    def near(a, b):
        return (a - b).abs() <= 1e-4 + 1e-4 * b.abs()
    certain = near(lo, ref) & near(hi_, ref)
    assert bool(near(got, ref)[certain].all()), 'a pixel no threshold tie can reach differs from the oracle'
    tied = ~certain
    assert bool((near(got, ref) | near(got, lo) | near(got, hi_))[tied].all()), 'a tie-affected pixel matches no side of its tie'
    print(f'merge: {int(tied.sum())} of {tied.numel()} pixels within reach of a |logit| < 1e-4 tie, '
          f'{int((~near(got, ref)).sum())} of them on the other side than the oracle')
    assert float(tied.float().mean()) < 1e-2


def test_fused_inference_launches_give_the_bits_of_the_unfused_sequence():
    """Round 6: simple_test_mask_logits with the stage heads and the merge tail fused (the default) against the launch
    sequence of round 5 (DM_FUSED_STAGE_HEAD / DM_FUSED_MERGE_TAIL switches), one stream and two, eager and as the
    bucketed HIP graph: bit for bit."""
    from dynamask_amd import mask_heads, roi_head
    hi = gi.head_inputs()
    m = _roi_head()
    feats = [_dev(f) for f in hi['feats']]
    boxes, labels = _dev(hi['rois'][:, 1:].contiguous()), _dev(hi['labels'])
    sel = hi['rois'][:, 0] == 0
    boxes, labels = boxes[_dev(sel)].contiguous(), labels[_dev(sel)].contiguous()
    outs = {}
    from dynamask_amd import ops as _ops
    split_was = _ops.CONV_SPLITK[0]
    _ops.CONV_SPLITK[0] = False      # bit equality holds for one order of sums: a split-K launch (chosen per launch shape) adds its
    #                                  channel ranges in another association than the grouped launch, which never splits
    with torch.no_grad():
        for split_min in (64, 2):                 # one chain / two chains on two streams
            m.stream_split_min = m.stream_split_min_graph = split_min
            for fused in (False, True):
                mask_heads.FUSED_STAGE_HEAD[0] = fused
                mask_heads.GROUPED_SEMANTIC_MAPS[0] = fused
                mask_heads.FUSED_DCN_TOUT[0] = fused
                roi_head.FUSED_MERGE_TAIL[0] = fused
                outs[(split_min, fused)] = m.simple_test_mask_logits(feats, boxes, labels).clone()
            m.enable_inference_graphs(True)
            outs[(split_min, 'graph')] = m.simple_test_mask_logits(feats, boxes, labels).clone()
            m.enable_inference_graphs(False)
        _ops.CONV_SPLITK[0] = split_was
        mask_heads.FUSED_STAGE_HEAD[0] = mask_heads.GROUPED_SEMANTIC_MAPS[0] = mask_heads.FUSED_DCN_TOUT[0] = roi_head.FUSED_MERGE_TAIL[0] = True
        m.stream_split_min = m.stream_split_min_graph = 64
        with_split = m.simple_test_mask_logits(feats, boxes, labels).clone()
    ref = outs[(64, False)]
    assert tuple(ref.shape) == (boxes.shape[0], 1, 112, 112)
    for k, v in outs.items():
        assert torch.equal(v, ref), k
    # with split-K on (the default) the sums associate differently: rounding only, except where a merge threshold ties
    far = (with_split - ref).abs() > 1e-4 + 1e-4 * ref.abs()
    assert float(far.float().mean()) < 1e-3


def test_fcn_mask_head_matches_reference_golden(golden_dir):
    from dynamask_amd import registry, mask_heads  # noqa: F401
    g = np.load(os.path.join(golden_dir, 'g6_fcn.npz'))
    x = _dev(gi.fcn_input())
    for up in ('deconv', 'carafe', 'bilinear'):
        cfg = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
        if up == 'carafe':
            cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                       encoder_dilation=1, compressed_channels=64)
        elif up == 'bilinear':
            cfg['upsample_cfg'] = dict(type='bilinear', scale_factor=2)
        cfg.pop('loss_mask')
        head = registry.build_head(cfg)
        head.load_state_dict({k[len('mask_head.'):]: v for k, v in gi.fcn_state(up).items()}, strict=True)
        head = head.cuda()
        with torch.no_grad():
            _close(head(x), g[up])


def test_mask_pre_and_selector_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g2_maskpre.npz'))
    x = _dev(gi.mask_pre_input())
    m = _roi_head(train=False)
    with torch.no_grad():
        _close(m.mask_predictor(x), g['logits_eval'])
    m.train(True)
    with torch.no_grad():
        logits = m.mask_predictor(x)
    _close(logits, g['logits_train'])
    _close(m.mask_predictor.bn1.running_mean, g['bn1_running_mean'], atol=1e-4, rtol=1e-4)
    _close(m.mask_predictor.bn1.running_var, g['bn1_running_var'], atol=1e-4, rtol=1e-4)
    _close(m.mask_predictor.bn2.running_var, g['bn2_running_var'], atol=1e-4, rtol=1e-4)
    assert int(m.mask_predictor.bn1.num_batches_tracked) == 1
    # selector on the golden logits: indices bit-exact, with the top-2 margin reported
    U = torch.rand(4, 4, generator=torch.Generator().manual_seed(5))
    from dynamask_amd import ops
    _, hot, idx = ops.gumbel_select(logits.contiguous(), _dev(U), 0.5)
    _, idx_ref = ref_model.gumbel_select(torch.from_numpy(g['logits_train']), U, 0.5)
    assert np.array_equal(idx.cpu().numpy(), idx_ref.numpy())


def test_mask_pre_on_the_map_matches_the_reference_order_in_training():
    """train_path.MaskPreMapFn (round 5: conv1 on the P2 map, 128 channels extracted, conv1's weight gradient through the
    adjoint of the 56 x 56 extraction) against MaskPreFn on the extracted [N, 256, 56, 56] tensor (the reference's order
    of operations, base_roi_head.py:10-27 + dynamask_roi_head.py:59, itself pinned by golden g2; the map form is also what
    the forward_train golden g11 runs through): logits, running statistics (conv1's bias re-enters the running mean) and
    every parameter gradient; boxes over the border included (void samples count as 0 before and after the bias)."""
    from dynamask_amd import ops, synth, train_path
    from tolerances import assert_grad_close
    torch.manual_seed(0)
    feat = synth.make_fpn(2, 192, 256, 256, seed=50)[0]                     # [2, 256, 48, 64]
    rois = synth.make_rois(2, 7, 192, 256, seed=51, max_size=300.0)
    rois = torch.cat([rois, torch.tensor([[0., -20., -10., 90., 70.], [1., 200., 150., 300., 230.]])], 0)
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous()
    outs = {}
    for name in ('reference order', 'map'):
        m = _roi_head(train=True)
        mp = m.mask_predictor
        with torch.no_grad():
            mp.conv1.bias.add_(torch.linspace(-0.5, 0.5, 128, device='cuda'))      # a bias worth shifting the running mean by
        if name == 'map':
            logits = train_path.MaskPreMapFn.apply(mp, _dev(feat), _dev(rois), 56, 0.25, 0, *list(mp.parameters()))
        else:
            x = ops.roi_align([_dev(feat)], _dev(rois), 56, [0.25])
            logits = train_path.MaskPreFn.apply(mp, x, *list(mp.parameters()))
        (logits * torch.arange(1, 5, dtype=torch.float32, device='cuda')).square().sum().backward()
        outs[name] = (logits.detach(), {k: p.grad.clone() for k, p in mp.named_parameters()},
                      mp.bn1.running_mean.clone(), mp.bn1.running_var.clone(), mp.bn2.running_mean.clone())
    la, ga, rma, rva, rm2a = outs['reference order']
    lb, gb, rmb, rvb, rm2b = outs['map']
    _close(lb, la, atol=1e-5, rtol=1e-5)
    _close(rmb, rma, atol=1e-6, rtol=1e-5)
    _close(rvb, rva, atol=1e-6, rtol=1e-5)
    _close(rm2b, rm2a, atol=1e-6, rtol=1e-5)
    # y1 differs by ~1e-6 between the two orders, and of the 1.8 M pooling windows behind it one or two have two candidates
    # that close (DESIGN section 2 "Max-pool ties"): the window's gradient then goes to another pixel of the SAME channel,
    # which moves that channel's row of conv1.weight (and its BatchNorm parameters) and nothing else.  Parameters in
    # front of the first pool are therefore compared per output channel: all but at most two channels at the gate,
    # those within 3 % of the tensor's scale.
    flipped = set()
    for k in ga:
        if k in ('conv1.bias', 'conv2.bias'):        # rounding residues in both (train-mode BatchNorm cancels a conv's bias)
            floor = 1e-5 * float(ga[k.replace('bias', 'weight')].abs().max())
            assert float(ga[k].abs().max()) <= floor and float(gb[k].abs().max()) <= floor
            continue
        if k in ('conv1.weight', 'bn1.weight', 'bn1.bias'):
            a2, b2 = ga[k].reshape(128, -1), gb[k].reshape(128, -1)
            scale = float(a2.abs().max())
            bad = ((b2 - a2).abs() > 1e-4 * min(scale, 1.0) + 1e-4 * a2.abs()).any(dim=1)
            flipped |= set(bad.nonzero().flatten().tolist())
            assert float((b2 - a2).abs().max()) <= 0.03 * scale, k
            continue
        assert_grad_close(gb[k], ga[k], k)
    assert len(flipped) <= 2, f'channels of conv1 / bn1 beyond the gate: {sorted(flipped)}'
    # ... and a channel beyond the gate must actually HOLD a tie (a per-channel bug of that size would pass
    # otherwise): relu(bn1(conv1(x))) of that channel recomputed from the extracted tensor, and among its 3 x 3 / stride-2
    # pooling windows at least one whose two largest candidates are closer than the ~1e-6 the two orders differ by
    if flipped:
        import torch.nn.functional as F
        m = _roi_head(train=True)
        mp = m.mask_predictor
        with torch.no_grad():
            mp.conv1.bias.add_(torch.linspace(-0.5, 0.5, 128, device='cuda'))
            x = ops.roi_align([_dev(feat)], _dev(rois), 56, [0.25])
            for c in sorted(flipped):
                y = F.conv2d(x, mp.conv1.weight[c:c + 1], mp.conv1.bias[c:c + 1])            # [N, 1, 56, 56]
                z = F.relu((y - y.mean()) / torch.sqrt(y.var(unbiased=False) + mp.bn1.eps) * mp.bn1.weight[c] + mp.bn1.bias[c])
                win = F.unfold(F.pad(z, (1, 1, 1, 1), value=float('-inf')), kernel_size=3, stride=2)   # [N, 9, 784]
                top2 = win.topk(2, dim=1).values
                live = top2[:, 0] > 0                                                          # (a window of zeros routes no gradient)
                gap = (top2[:, 0] - top2[:, 1])[live]
                print(f'channel {c}: smallest top-2 gap of a live pooling window {float(gap.min()):.3g}')
                assert float(gap.min()) < 2e-5 * max(float(z.max()), 1.0), f'channel {c} differs beyond the gate without a pooling tie'


def test_dyna_loss_and_grads_match_reference_golden(golden_dir):
    from dynamask_amd import registry
    from dynamask_amd import losses  # noqa: F401
    g = np.load(os.path.join(golden_dir, 'g1_losses.npz'))
    li = gi.loss_inputs()
    loss_mod = registry.build_loss(dict(type='DynaCrossEntropyLoss', **gi.LOSS_CFG)).cuda()
    ips = [_dev(t).requires_grad_(True) for t in li['ips']]
    dps = [_dev(t).requires_grad_(True) for t in li['dps']]
    ml = _dev(li['mask_labels']).requires_grad_(True)
    out = loss_mod(ips, dps, [_dev(t) for t in li['targets']], ml)
    loss = out['loss_masks']
    loss.backward()
    _close(loss, g['loss_masks'], atol=1e-4, rtol=1e-4)
    assert_grad_close(ml.grad, g['grad_mask_labels'], 'mask_labels')
    for i in range(4):
        # Quirk Q2 / start_stage: only the last stage's instance BCE and the first three detail BCEs reach the loss
        assert_grad_close(dps[i].grad, g[f'grad_dp{i}'], f'dp{i}', zero=(i == 3))
        assert_grad_close(ips[i].grad, g[f'grad_ip{i}'], f'ip{i}', zero=(i < 3))


def test_mask_forward_train_loss_vs_oracle():
    hi = gi.head_inputs()
    n = hi['rois'].shape[0]
    m = _roi_head(train=True)
    U = torch.rand(n, 4, generator=torch.Generator().manual_seed(9))
    tg = gi.head_targets(n)
    with torch.no_grad():
        res = m._mask_forward_train([_dev(f) for f in hi['feats']], _dev(hi['rois']), _dev(hi['labels']),
                                    [_dev(t) for t in tg], noise=_dev(U))
        sd = {**gi.head_state(), **gi.mask_pre_state()}
        loss_ref, ml_ref, idx_ref, logits_ref = ref_model.mask_forward_train(sd, hi['feats'], hi['rois'],
                                                                             hi['labels'], tg, U)
    _close(res['mask_logits'], logits_ref)
    assert np.array_equal(res['mask_index'].cpu().numpy(), idx_ref.numpy())       # resolution selection bit-exact
    _close(res['loss_mask']['loss_masks'], loss_ref)


def test_training_slice_grads_match_reference_golden(golden_dir):
    """Forward + hand-sequenced backward of extractor + head + loss vs gradients
    autograd produced for the reference's own modules (g7)."""
    g = np.load(os.path.join(golden_dir, 'g7_head_train.npz'))
    hi = gi.head_inputs()
    m = _roi_head(train=True)
    feats = [_dev(f).requires_grad_(True) for f in hi['feats']]
    n = hi['rois'].shape[0]
    res = m._mask_forward(feats, _dev(hi['rois']), _dev(hi['labels']))
    ml = _dev(gi.head_mask_labels(n)).requires_grad_(True)
    loss = m.mask_head.loss_func(res['stage_instance_preds'], res['stage_detail_preds'],
                                 [_dev(t) for t in gi.head_targets(n)], ml)['loss_masks']
    loss.backward()
    _close(loss, g['loss'])
    assert_grad_close(ml.grad, g['grad_mask_labels'], 'mask_labels')
    named = dict(m.mask_head.named_parameters())
    for k in gi.GRAD_KEYS:
        assert named[k].grad is not None, k
        assert_grad_close(gi.grad_slice(named[k].grad), g['grad.' + k], k)
    for i in range(4):
        assert feats[i].grad is not None, i
        assert_grad_close(gi.feat_grad_slice(feats[i].grad), g[f'grad_feat{i}'], f'feat{i}')


def test_mask_pre_and_selector_backward_match_reference_golden(golden_dir):
    g2 = np.load(os.path.join(golden_dir, 'g2_maskpre.npz'))
    g3 = np.load(os.path.join(golden_dir, 'g3_gumbel.npz'))
    from dynamask_amd import train_path
    m = _roi_head(train=True)
    mp = m.mask_predictor
    x = _dev(gi.mask_pre_input())
    logits = train_path.MaskPreFn.apply(mp, x, *list(mp.parameters()))
    _close(logits, g2['logits_train'])
    logits.square().sum().backward()
    assert_grad_close(mp.fc2.weight.grad, g2['grad_fc2_w'], 'fc2.weight')
    assert_grad_close(mp.conv1.bias.grad, g2['grad_conv1_b'], 'conv1.bias', cancels=1e-5)   # train-mode BN removes it
    assert_grad_close(mp.bn1.weight.grad, g2['grad_bn1_w'], 'bn1.weight')
    assert_grad_close(mp.conv2.weight.grad, g2['grad_conv2_w'], 'conv2.weight')
    # straight-through selector gradient
    lg = _dev(gi.gumbel_logits()).requires_grad_(True)
    torch.manual_seed(gi.GUMBEL_SEED)
    U = torch.rand(lg.shape)
    hot, idx = train_path.GumbelSelectFn.apply(lg, _dev(U), 0.5)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), g3['index'])
    (hot * torch.arange(1, 5, dtype=torch.float32, device='cuda')).sum().backward()
    assert_grad_close(lg.grad, g3['grad_logits'], 'selector logits')


def test_full_training_step_runs_and_updates_every_parameter():
    hi = gi.head_inputs()
    n = hi['rois'].shape[0]
    m = _roi_head(train=True)
    feats = [_dev(f).requires_grad_(True) for f in hi['feats']]
    U = torch.rand(n, 4, generator=torch.Generator().manual_seed(9))
    res = m._mask_forward_train(feats, _dev(hi['rois']), _dev(hi['labels']), [_dev(t) for t in gi.head_targets(n)],
                                noise=_dev(U))
    res['loss_mask']['loss_masks'].backward()
    missing = [k for k, p in m.named_parameters() if p.grad is None and 'fuse_kernel' not in k]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    assert feats[0].grad is not None and feats[4].grad is None       # P6 is never read by the path


def test_mask_targets_and_paste_match_reference_golden(golden_dir):
    """SURVEY 8f rows 1-2 on the device: get_targets and get_seg_masks (bit-exact masks
    up to pixels whose interpolated value sits within 1e-6 of the threshold)."""
    from dynamask_amd.registry import ConfigDict
    m = _roi_head()
    g9 = np.load(os.path.join(golden_dir, 'g9_targets.npz'))
    ti = gi.target_inputs()
    tg = m.mask_head.get_targets([_dev(t['boxes']) for t in ti], [_dev(t['inds']) for t in ti],
                                 [t['masks'].cuda() for t in ti])
    for i in range(4):
        ne = tg[i].cpu().numpy().astype(np.uint8) != g9[f't{i}']
        # {0,1} targets = (RoIAlign of the bitmap >= 0.5): a flip needs a sample average within an ulp of 0.5
        print(f'mask targets {g9[f"t{i}"].shape}: {int(ne.sum())} of {ne.size} pixels differ from the reference')
        assert ne.mean() < 2e-4, (i, ne.mean())
    g8 = np.load(os.path.join(golden_dir, 'g8_paste.npz'))
    pi = gi.paste_inputs()
    for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
        segs = m.mask_head.get_seg_masks(_dev(pi['logits']), _dev(pi['det_bboxes']), torch.zeros(5, dtype=torch.long).cuda(),
                                         ConfigDict(mask_thr_binary=0.5), pi['ori_shape'], sf, rescale)
        ref = g8[f'seg_rescale{int(rescale)}_sf{sf}']
        assert segs[0].shape == ref[0].shape and segs[0].dtype == np.bool_
        # reference golden (CPU run): exact for regular boxes; the zero-width box (index 4) follows the
        # reference's whole-canvas GPU path here, checked against the oracle's restatement of it
        ne = np.stack(segs[:4]).astype(np.uint8) != ref[:4]
        full = ref_model.get_seg_masks(pi['logits'], pi['det_bboxes'], pi['ori_shape'], sf, rescale, device_type='cuda')
        ne4 = segs[4] != full[4].numpy()
        print(f'paste rescale={rescale} sf={sf}: {int(ne.sum())} of {ne.size} pixels differ from the reference golden, '
              f'{int(ne4.sum())} of {ne4.size} (zero-width box) from the oracle')
        assert ne.mean() < 2e-5, ne.mean()
        assert ne4.mean() < 2e-4, ne4.mean()


def test_fcn_mask_head_callers_match_reference_golden(golden_dir):
    """FCNMaskHead.get_seg_masks / get_seg_rles / get_targets (fcn_mask_head.py:128-237) on the device against g14 =
    the reference's own methods: per-class grouping and order exact, bitmaps exact up to pixels whose interpolated
    value sits within an ulp of the threshold; the zero-width box follows the reference's whole-canvas GPU path
    (the golden ran on CPU, one tight region per mask) and is checked against the oracle's restatement of that path."""
    import types
    from dynamask_amd import registry
    from dynamask_amd.registry import ConfigDict
    g = np.load(os.path.join(golden_dir, 'g14_fcn_callers.npz'))
    pi = gi.fcn_paste_inputs()
    order = [i for c in range(80) for i in range(7) if int(pi['det_labels'][i]) == c]       # detection index of every flattened row
    for ag in (False, True):
        head = registry.build_head(dict(type='FCNMaskHead', **dict(gi.FCN_HEAD_CFG, class_agnostic=ag))).cuda()
        logits = pi['logits'] if not ag else pi['logits'][torch.arange(7), pi['det_labels']][:, None]
        for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
            key = f'ag{int(ag)}_rescale{int(rescale)}_sf{sf}'
            segs = head.get_seg_masks(_dev(logits), _dev(pi['det_bboxes']), pi['det_labels'].cuda(), ConfigDict(mask_thr_binary=0.5),
                                      pi['ori_shape'], sf, rescale)
            assert len(segs) == 80 and [len(c) for c in segs] == g['counts_' + key].tolist()
            flat = np.stack([m for c in segs for m in c])
            assert flat.dtype == np.bool_
            ref = g['seg_' + key]
            regular = [k for k, i in enumerate(order) if i != 6]
            ne = flat[regular].astype(np.uint8) != ref[regular]
            full = ref_model.fcn_get_seg_masks(logits, pi['det_bboxes'], pi['det_labels'], pi['ori_shape'], sf, rescale,
                                               class_agnostic=ag, device_type='cuda')
            ne6 = flat[order.index(6)] != [m for c in full for m in c][order.index(6)].numpy()
            print(f'fcn paste {key}: {int(ne.sum())} of {ne.size} pixels differ from the reference golden, {int(ne6.sum())} (zero-width box) from the oracle')
            assert ne.mean() < 2e-5 and ne6.mean() < 2e-4
            # the RLE form of the same call decodes to the same bitmaps
            rles = head.get_seg_rles(_dev(logits), _dev(pi['det_bboxes']), pi['det_labels'].cuda(), ConfigDict(mask_thr_binary=0.5),
                                     pi['ori_shape'], sf, rescale)
            assert [len(c) for c in rles] == [len(c) for c in segs]
            for r, m in zip([r for c in rles for r in c], flat):
                assert r == ref_ops.rle_encode(m.astype(np.uint8))
        # ndarray of probabilities (multi-scale testing): no sigmoid
        segs = head.get_seg_masks(logits.sigmoid().numpy(), _dev(pi['det_bboxes']), pi['det_labels'].cuda(), ConfigDict(mask_thr_binary=0.5),
                                  pi['ori_shape'], 1.0, True)
        flat = np.stack([m for c in segs for m in c])
        ne = flat[regular].astype(np.uint8) != g[f'seg_ag{int(ag)}_ndarray'][regular]
        assert ne.mean() < 2e-5
        assert head.get_seg_masks(_dev(logits[:0]), _dev(pi['det_bboxes'][:0]), pi['det_labels'][:0].cuda(), ConfigDict(mask_thr_binary=0.5),
                                  pi['ori_shape'], 1.0, True) == [[] for _ in range(80)]
    head = registry.build_head(dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)).cuda()
    ti = gi.target_inputs()
    res = [types.SimpleNamespace(pos_bboxes=_dev(t['boxes']), pos_assigned_gt_inds=_dev(t['inds'])) for t in ti]
    for size in (28, 14):
        tg = head.get_targets(res, [t['masks'].cuda() for t in ti], ConfigDict(mask_size=size))
        assert tg.dtype == torch.float32 and tuple(tg.shape) == g[f'targets{size}'].shape
        ne = tg.cpu().numpy().astype(np.uint8) != g[f'targets{size}']
        print(f'fcn mask targets {size}: {int(ne.sum())} of {ne.size} differ')
        assert ne.mean() < 2e-4
    with pytest.raises(NotImplementedError):
        head.loss(None, None, None)


def test_standard_roi_head_with_fcn_mask_head_matches_the_oracle():
    """StandardRoIHead (standard_roi_head.py + MaskTestMixin.simple_test_mask, test_mixins.py:151-176) over FCNMaskHead --
    BASELINE configs[4]'s RoI head: ``_mask_forward`` against the oracle's RoIAlign + FCNMaskHead forward (1e-4),
    ``simple_test_mask`` against the oracle's get_seg_masks of those logits (per class, in detection order; bitmaps up to
    threshold ties), its RLE form, ``simple_test`` end to end, and the training entry point: bbox losses and mask targets
    are produced, the mask loss raises as the fork's does (Quirk Q5)."""
    from dynamask_amd import registry, roi_head, bbox_heads, losses, mask_heads, roi_extractors  # noqa: F401  (register the classes)
    from dynamask_amd.registry import ConfigDict
    up = 'carafe'
    mcfg = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
    mcfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3, encoder_dilation=1,
                                compressed_channels=64)
    m = registry.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
        bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
        mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG), mask_head=mcfg,
        train_cfg=registry._to_cfgdict(gi.RCNN_TRAIN_CFG), test_cfg=ConfigDict(**gi.RCNN_TEST_CFG)))
    fsd = gi.fcn_state(up)
    m.load_state_dict({**fsd, **gi.bbox_head_state(), **gi.mask_pre_state()}, strict=True)
    m = m.cuda().eval()
    hi = gi.head_inputs()
    feats = [_dev(f) for f in hi['feats']]
    sel = hi['rois'][:, 0] == 0
    rois, labels = hi['rois'][sel].contiguous(), hi['labels'][sel].contiguous()
    sdo = {k[len('mask_head.'):]: v for k, v in fsd.items()}
    ref_feats = ref_ops.single_roi_extractor(hi['feats'][:4], rois, 14, (4, 8, 16, 32))
    ref_pred = ref_model.fcn_mask_head_forward(sdo, ref_feats, upsample=up)
    with torch.no_grad():
        res = m._mask_forward(feats, _dev(rois))
    assert set(res) == {'mask_pred', 'mask_feats'}
    _close(res['mask_feats'], ref_feats.numpy())
    _close(res['mask_pred'], ref_pred.numpy())
    det = torch.cat([rois[:, 1:], torch.full((len(rois), 1), 0.7)], 1)
    metas = [dict(img_shape=(256, 320, 3), ori_shape=(256, 320, 3), scale_factor=1.0)]
    segs = m.simple_test_mask(feats, metas, _dev(det), labels.cuda(), rescale=False)
    ref_segs = ref_model.fcn_get_seg_masks(ref_pred, det, labels, (256, 320, 3), 1.0, False, device_type='cuda')
    assert len(segs) == 80 and [len(c) for c in segs] == [len(c) for c in ref_segs]
    flat, rflat = np.stack([b for c in segs for b in c]), np.stack([b.numpy() for c in ref_segs for b in c])
    ne = flat != rflat
    print(f'StandardRoIHead.simple_test_mask: {int(ne.sum())} of {ne.size} pixels differ from the oracle')
    assert flat.dtype == np.bool_ and ne.mean() < 2e-4
    rles = m.simple_test_mask(feats, metas, _dev(det), labels.cuda(), rescale=False, encode=True)
    for r, b in zip([r for c in rles for r in c], flat):
        assert r == ref_ops.rle_encode(b.astype(np.uint8))
    assert m.simple_test_mask(feats, metas, _dev(det[:0]), labels[:0].cuda()) == [[] for _ in range(80)]
    # simple_test: boxes first, then the masks of the kept detections
    props = _dev(gi.bbox_inputs()[1][:, 1:].contiguous())
    bb, sg = m.simple_test(feats, [props], metas, rescale=False)
    assert len(bb) == 80 and [len(b) for b in bb] == [len(s_) for s_ in sg]
    # training entry point (standard_roi_head.py:70-134)
    ti = gi.train_inputs()
    m.train()
    tf = [f.cuda() for f in ti['feats']]
    with pytest.raises(NotImplementedError, match='Q5'):
        m.forward_train(tf, ti['img_metas'], [p.cuda() for p in ti['proposals']], [b.cuda() for b in ti['gt_bboxes']],
                        [l.cuda() for l in ti['gt_labels']], None, [t.cuda() for t in ti['gt_masks']])
    m.mask_head.loss = lambda pred, tgt, lab: {'loss_mask': (pred[torch.arange(len(lab)), lab] - tgt).abs().mean()}   # a stand-in loss
    losses = m.forward_train(tf, ti['img_metas'], [p.cuda() for p in ti['proposals']], [b.cuda() for b in ti['gt_bboxes']],
                             [l.cuda() for l in ti['gt_labels']], None, [t.cuda() for t in ti['gt_masks']])
    assert set(losses) == {'loss_cls', 'acc', 'loss_bbox', 'loss_mask'} and all(torch.isfinite(v).all() for v in losses.values())


def test_simple_test_mask_end_to_end():
    hi = gi.head_inputs()
    m = _roi_head()
    from dynamask_amd.registry import ConfigDict
    m.test_cfg = ConfigDict(mask_thr_binary=0.5)
    sel = hi['rois'][:, 0] == 0
    det = torch.cat([hi['rois'][sel][:, 1:], torch.ones(int(sel.sum()), 1)], 1)
    labels = hi['labels'][sel]
    with torch.no_grad():
        res = m.simple_test_mask([_dev(f) for f in hi['feats']], [dict(ori_shape=(256, 320, 3), scale_factor=1.0)],
                                 _dev(det), _dev(labels), rescale=False)
    assert len(res) == 80 and sum(len(r) for r in res) == int(sel.sum())
    for c in labels.tolist():
        assert res[c][0].shape == (256, 320) and res[c][0].dtype == np.bool_


# ------------------------------------------------------------------ dynamic inference (8f rank 3)
def _dyn_case(seed=3, n=37):
    from dynamask_amd import synth
    feats = synth.make_fpn(1, 256, 320, 256, seed=seed)
    rois = synth.make_rois(1, n, 256, 320, seed=seed + 1)
    labels = synth.make_labels(n, seed=seed + 2)
    exits = torch.randint(0, 4, (n,), generator=torch.Generator().manual_seed(seed + 3))
    return feats, rois, labels, exits


@pytest.fixture
def no_splitk():
    """Launches of few workgroups split their K loop in the inference entry points (dm_conv2d_fwd_ws): another
    association of the same products per RoI COUNT.  Tests of bitwise row identity across RoI counts switch it off."""
    from dynamask_amd import ops
    was = ops.CONV_SPLITK[0]
    ops.CONV_SPLITK[0] = False
    yield
    ops.CONV_SPLITK[0] = was


def test_small_inference_calls_split_k_and_stay_within_rounding_of_the_unsplit_sums():
    """dm_conv2d_fwd_ws: at the detection counts of real inference the 14 x 14 convolutions are a few
    workgroups walking K = 2304 alone; with split-K the same call must give the unsplit logits up to the rounding of
    another association (1e-5 of the logit scale), the same bits on every run, and the oracle's within 1e-4."""
    from dynamask_amd import ops
    feats, rois, labels, _ = _dyn_case(seed=21, n=9)
    m = _roi_head()
    m.num_streams = 1
    fd = [_dev(f) for f in feats]
    assert ops.CONV_SPLITK[0]
    with torch.no_grad():
        a = m._mask_forward(fd, _dev(rois), _dev(labels))['stage_instance_preds']
        b = m._mask_forward(fd, _dev(rois), _dev(labels))['stage_instance_preds']
        ops.CONV_SPLITK[0] = False
        try:
            c = m._mask_forward(fd, _dev(rois), _dev(labels))['stage_instance_preds']
        finally:
            ops.CONV_SPLITK[0] = True
        sd = {**gi.head_state(), **gi.mask_pre_state()}
        ips, _ = ref_model.mask_forward(sd, feats, rois, labels)
    differs = False
    for k in range(4):
        assert torch.equal(a[k], b[k])
        scale = float(c[k].abs().max())
        assert float((a[k] - c[k]).abs().max()) <= 1e-5 * max(scale, 1.0), k
        differs = differs or not torch.equal(a[k], c[k])
        _close(a[k], ips[k])
    assert differs, 'nine RoIs did not split any launch: the test exercises nothing'


def test_dynamic_exit_rows_are_bit_identical_to_the_fixed_path(no_splitk):
    """RoIs never interact inside the head: RoI j leaving at exit e must carry exactly the
    logits the all-exits path computes for it at e (bitwise; split-K off: see ``no_splitk``)."""
    feats, rois, labels, exits = _dyn_case()
    m = _roi_head()
    m.num_streams = 1
    fd = [_dev(f) for f in feats]
    with torch.no_grad():
        full = m._mask_forward(fd, _dev(rois), _dev(labels))['stage_instance_preds']
        res = m.dynamic_mask_logits(fd, _dev(rois[:, 1:]), _dev(labels), merge=False, exits=exits)
    order = res['order'].cpu().tolist()
    assert res['n_ge'] == [int((exits >= k).sum()) for k in range(4)]
    assert sorted(order) == list(range(len(rois)))
    for p, j in enumerate(order):
        e = int(exits[j])
        assert torch.equal(res['preds'][e][p], full[e][j]), (p, j, e)
    # degenerate distributions: everyone at one exit, empty deeper stages
    for e in (0, 3):
        with torch.no_grad():
            r1 = m.dynamic_mask_logits(fd, _dev(rois[:, 1:]), _dev(labels), merge=False, exits=torch.full((len(rois),), e))
        assert r1['n_ge'] == [len(rois)] * (e + 1) + [0] * (3 - e)
        assert torch.equal(r1['preds'][e], full[e])
        assert all(r1['preds'][k].shape[0] == 0 for k in range(e + 1, 4))


def test_dynamic_merge_and_selector_vs_oracle():
    feats, rois, labels, exits = _dyn_case(seed=11, n=21)
    m = _roi_head()
    fd = [_dev(f) for f in feats]
    sd = {**gi.head_state(), **gi.mask_pre_state()}
    with torch.no_grad():
        ips, _ = ref_model.mask_forward(sd, feats, rois, labels)
        ref = ref_model.dynamic_exit_logits(ips, exits, merge=True)
        res = m.dynamic_mask_logits(fd, _dev(rois[:, 1:]), _dev(labels), merge=True, exits=exits)
    flips = 0
    for p, j in enumerate(res['order'].cpu().tolist()):
        e = int(exits[j])
        got = res['preds'][e][p].cpu()
        diff = (got - ref[j]).abs()
        flips += int((diff > 1e-4 + 1e-4 * ref[j].abs()).sum())
    assert flips <= 20, flips          # sigmoid>=0.5 side flips of the merge mask on ~1e-6 logit differences
    # selector: eval-mode MaskPre on RoIAlign56(P2), no sampling -> argmax of the oracle's logits
    with torch.no_grad():
        sem = ref_ops.single_roi_extractor([feats[0]], rois, 56, (4,))
        logits = ref_model.mask_pre(sd, sem, training=False)
        res2 = m.dynamic_mask_logits(fd, _dev(rois[:, 1:]), _dev(labels))
    top2 = logits.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    sure = margin > 1e-4
    assert torch.equal(res2['exits'].cpu()[sure], logits.argmax(1)[sure])
    # explicit Gumbel noise goes through the same sampler as training
    U = torch.rand(len(rois), 4, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        res3 = m.dynamic_mask_logits(fd, _dev(rois[:, 1:]), _dev(labels), noise=_dev(U))
    _, ind = ref_model.gumbel_select(logits, U, 0.5)
    g = -torch.log(-torch.log(U + 1e-20) + 1e-20)
    t2 = ((logits + g) / 0.5).topk(2, dim=1).values
    sure3 = (t2[:, 0] - t2[:, 1]) > 1e-3
    assert torch.equal(res3['exits'].cpu()[sure3], ind[sure3])


def test_dynamic_test_mask_pastes_each_detection_from_its_own_exit():
    feats, rois, labels, exits = _dyn_case(seed=21, n=15)
    m = _roi_head()
    from dynamask_amd.registry import ConfigDict
    m.test_cfg = ConfigDict(mask_thr_binary=0.5)
    fd = [_dev(f) for f in feats]
    det = torch.cat([rois[:, 1:], torch.ones(len(rois), 1)], 1)
    res = m.dynamic_test_mask(fd, [dict(ori_shape=(256, 320, 3), scale_factor=1.0)], _dev(det), _dev(labels),
                              rescale=False, exits=exits, merge=False)
    assert len(res) == 80 and sum(len(r) for r in res) == len(rois)
    with torch.no_grad():
        full = m._mask_forward(fd, _dev(rois), _dev(labels))['stage_instance_preds']
    seen = {}
    for j, c in enumerate(labels.tolist()):
        k = seen.get(c, 0)
        seen[c] = k + 1
        e = int(exits[j])
        ref = ref_model.get_seg_masks(full[e][j:j + 1].cpu(), det[j:j + 1], (256, 320, 3), 1.0, False)[0]
        ref = np.asarray(ref)
        got = res[c][k]
        assert got.shape == (256, 320) and got.dtype == np.bool_
        assert (got != ref).mean() < 2e-4, (j, e, (got != ref).sum())


def test_simple_test_mask_encoded_equals_encoding_of_the_bitmaps():
    """simple_test_mask(encode=True) == encode_mask_results(simple_test_mask()) (apis/test.py:52-57)."""
    hi = gi.head_inputs()
    m = _roi_head()
    from dynamask_amd.registry import ConfigDict
    m.test_cfg = ConfigDict(mask_thr_binary=0.5)
    sel = hi['rois'][:, 0] == 0
    det = torch.cat([hi['rois'][sel][:, 1:], torch.ones(int(sel.sum()), 1)], 1)
    labels = hi['labels'][sel]
    fd = [_dev(f) for f in hi['feats']]
    metas = [dict(ori_shape=(256, 320, 3), scale_factor=1.0)]
    with torch.no_grad():
        bitmaps = m.simple_test_mask(fd, metas, _dev(det), _dev(labels), rescale=False)
        rles = m.simple_test_mask(fd, metas, _dev(det), _dev(labels), rescale=False, encode=True)
    assert len(rles) == 80
    for c in range(80):
        assert len(rles[c]) == len(bitmaps[c])
        for r, b in zip(rles[c], bitmaps[c]):
            assert r == ref_ops.rle_encode(b.astype(np.uint8))


# ------------------------------------------------------------------ bbox branch (8f rank 4)
def _bbox_roi_head():
    from dynamask_amd import registry, bbox_heads, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    from dynamask_amd.registry import ConfigDict
    cfg = dict(type='DynaMaskRoIHead',
               bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
               bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
               mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG),
               test_cfg=ConfigDict(**gi.RCNN_TEST_CFG))
    m = registry.build_head(cfg)
    m.load_state_dict({**gi.head_state(), **gi.mask_pre_state(), **gi.bbox_head_state()}, strict=True)
    return m.cuda().eval()


def test_bbox_head_and_decode_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_bbox.npz'))
    m = _bbox_roi_head()
    x, rois = gi.bbox_inputs()
    with torch.no_grad():
        cls_score, bbox_pred = m.bbox_head(_dev(x))
    # library fp32 GEMM over K = 12544: summation order differs from the CPU reference
    _close(cls_score, g['cls_score'], atol=1e-4, rtol=1e-4)
    _close(bbox_pred, g['bbox_pred'], atol=1e-4, rtol=1e-4)
    cs, bp = _dev(torch.from_numpy(g['cls_score'])), _dev(torch.from_numpy(g['bbox_pred']))
    b, s = m.bbox_head.get_bboxes(_dev(rois), cs, bp, gi.BBOX_IMG_SHAPE, 1.0)
    _close(b, g['bboxes'], atol=1e-4, rtol=1e-4)
    _close(s, g['scores'], atol=1e-4, rtol=1e-4)
    sf = np.array([1.25, 1.6, 1.25, 1.6], dtype=np.float32)
    b1, _ = m.bbox_head.get_bboxes(_dev(rois), cs, bp, gi.BBOX_IMG_SHAPE, sf, rescale=True)
    _close(b1, g['bboxes_rescaled'], atol=1e-4, rtol=1e-4)
    d, lab = m.bbox_head.get_bboxes(_dev(rois), cs, bp, gi.BBOX_IMG_SHAPE, 1.0, cfg=m.test_cfg)
    assert d.shape == (100, 5)
    same = (lab.cpu().numpy() == g['det_labels']) & (np.abs(d.cpu().numpy() - g['det_bboxes']).max(1) < 1e-3)
    # an IoU within an ulp of 0.5 may fall on the other side: one such flip replaces a detection and shifts the rest
    print(f'multiclass_nms: {int(same.sum())} of {same.size} kept detections equal the reference golden position by position')
    assert same.mean() > 0.97, same.mean()
    r = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    dl = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    from dynamask_amd.bbox_heads import DeltaXYWHBBoxCoder
    _close(DeltaXYWHBBoxCoder().decode(_dev(r), _dev(dl), max_shape=(32, 32)), g['doc_example'], atol=1e-6)


def test_nms_matches_oracle():
    from dynamask_amd import ops
    g = torch.Generator().manual_seed(4)
    for n in (1, 5, 64, 65, 300, 1500):
        ctr = torch.rand(n, 2, generator=g) * 300
        wh = torch.rand(n, 2, generator=g) * 90 + 5
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        scores = torch.rand(n, generator=g)
        for off in (0, 1):
            dets, keep = ops.nms(boxes.cuda(), scores.cuda(), 0.5, offset=off)
            rd, rk = ref_model.nms(boxes, scores, 0.5, offset=off)
            assert keep.cpu().tolist() == rk.tolist(), (n, off)
            _close(dets, rd.numpy(), atol=1e-4, rtol=1e-4)
    d0, k0 = ops.nms(torch.zeros((0, 4), device='cuda'), torch.zeros((0,), device='cuda'), 0.5)
    assert d0.shape == (0, 5) and k0.numel() == 0


def test_simple_test_boxes_then_masks():
    m = _bbox_roi_head()
    hi = gi.head_inputs()
    fd = [_dev(f) for f in hi['feats']]
    _, rois = gi.bbox_inputs()
    metas = [dict(img_shape=gi.BBOX_IMG_SHAPE, ori_shape=gi.BBOX_IMG_SHAPE, scale_factor=1.0)]
    bbox_results, segm_results = m.simple_test(fd, [_dev(rois[:, 1:])], metas, rescale=False)
    assert len(bbox_results) == 80 and len(segm_results) == 80
    n = sum(len(b) for b in bbox_results)
    assert 0 < n <= 100 and sum(len(s) for s in segm_results) == n
    for b, s in zip(bbox_results, segm_results):
        assert b.shape[1:] == (5,) and len(s) == len(b)
        for mask in s:
            assert mask.shape == (256, 320) and mask.dtype == np.bool_
    # the detections themselves: oracle on the same extractor output
    with torch.no_grad():
        feats7 = m.bbox_roi_extractor(fd[:4], _dev(rois))
        cs, bp = ref_model.bbox_head_forward(gi.bbox_head_state(), feats7.cpu())
        d, lab = ref_model.get_bboxes(rois, cs, bp, gi.BBOX_IMG_SHAPE, 1.0, cfg=gi.RCNN_TEST_CFG)
    got = np.concatenate([b for b in bbox_results if len(b)])
    assert abs(len(got) - len(d)) <= 2


def test_mask_pre_conv1_commutes_with_roialign():
    """MaskPre.forward_from_map (inference shortcut): conv1(RoIAlign56(x)) == RoIAlign56(W1 x) + b1."""
    feats, rois, labels, _ = _dyn_case(seed=31, n=23)
    rois = torch.cat([rois, torch.tensor([[0, -20.0, -10.0, 90.0, 70.0], [0, 250.0, 200.0, 330.0, 260.0]])])  # over the borders
    m = _roi_head()
    p2 = _dev(feats[0])
    with torch.no_grad():
        direct = m.mask_predictor(m.semantic_roi_extractor([p2], _dev(rois)))
        fast = m.mask_predictor.forward_from_map(p2, _dev(rois), m.semantic_roi_extractor)
        sd = {**gi.head_state(), **gi.mask_pre_state()}
        ref = ref_model.mask_pre(sd, ref_ops.single_roi_extractor([feats[0]], rois, 56, (4,)), training=False)
    _close(fast, direct.cpu().numpy(), atol=1e-4, rtol=1e-4)
    _close(fast, ref.numpy(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize('up', ['deconv', 'carafe', 'bilinear', 'nearest'])
def test_fcn_mask_head_backward_matches_oracle_autograd(up):
    """FCNMaskHead (fcn_mask_head.py:117-126) forward AND backward for every upsample type: gradients of
    all parameters and of the RoI features against autograd of the oracle (CARAFE: mmcv's carafe backward +
    kernel_normalizer backward restated through F.softmax / pixel_shuffle).  atol = rtol = 1e-4 on the
    tensor's scale."""
    from dynamask_amd import registry, mask_heads  # noqa: F401
    cfg = dict(type='FCNMaskHead', **gi.FCN_HEAD_CFG)
    cfg.pop('loss_mask')
    if up == 'carafe':
        cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                   encoder_dilation=1, compressed_channels=64)
    elif up != 'deconv':
        cfg['upsample_cfg'] = dict(type=up, scale_factor=2)
    sd = {k[len('mask_head.'):]: v for k, v in gi.fcn_state(up).items()}
    if up == 'carafe':      # the reference's std 0.001 encoder gives a uniform kernel: make the softmax matter
        sd['upsample.content_encoder.weight'] = sd['upsample.content_encoder.weight'] * 12.0
    head = registry.build_head(cfg)
    head.load_state_dict(sd, strict=True)
    head = head.cuda().train()
    x = gi.fcn_input()
    G = torch.randn(3, 80, 28, 28, generator=torch.Generator().manual_seed(7)) * 0.1
    xg = _dev(x).requires_grad_(True)
    out = head(xg)
    (out * _dev(G)).sum().backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    ref = ref_model.fcn_mask_head_forward(sdo, xo, upsample=up)
    (ref * G).sum().backward()
    _close(out, ref)

    def scaled(got, want, what):
        got, want = got.detach().cpu().numpy(), want.detach().numpy()
        tol = 1e-4 * max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= tol, (what, float(np.abs(got - want).max()), tol)
    scaled(xg.grad, xo.grad, 'x')
    for k, p in head.named_parameters():
        assert p.grad is not None, k
        scaled(p.grad, sdo[k].grad, k)


class _FakePolygonMasks:
    """The part of mmdet's PolygonMasks get_targets touches: .masks (list over objects of lists of arrays), size."""

    def __init__(self, masks, height, width):
        self.masks, self.height, self.width = masks, height, width


def test_polygon_mask_targets_match_the_oracle_and_the_reference_known_answers(golden_dir):
    """DynaMaskHead.get_targets with polygon annotations (dynamask_head.py:248-262 -> structures.py:469-503, 583-599 ->
    pycocotools): the device rasteriser against oracle/ref_poly.py (the restated rleFrPoly, pinned by the reference's
    own test bitmaps) -- bit-exact, every supervision size, polygons that reach far outside their boxes, several
    parts per object, degenerate edges, boxes beyond the image -- and against those bitmaps directly."""
    from dynamask_amd import ops
    from oracle import ref_poly as rp
    g = np.load(os.path.join(golden_dir, 'g12_polygon_truth.npz'))
    dev = torch.device('cuda')
    # (a) the reference's known answers: crop_and_resize with the box = the canvas is resize
    for polys, hw, size, truth in (([g['poly1']], 5, 10, g['truth1']), ([g['poly2a'], g['poly2b']], 3, 6, g['truth2'])):
        packed = ops.pack_polygons([polys], dev)
        box = torch.tensor([[0., 0., hw, hw]], device=dev)
        t = ops.polygon_mask_targets(packed, box, torch.zeros(1, dtype=torch.long, device=dev), size)
        assert np.array_equal(t[0].cpu().numpy().astype(np.uint8), truth)
    # (b) random scenes against the oracle
    rng = np.random.default_rng(7)
    H, W = 96, 128
    objs = []
    for o in range(9):
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(3, 12))
            cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(2, 60)
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            rad = r * rng.uniform(0.4, 1.0, k)
            p = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).reshape(-1)
            if o % 3 == 0:
                p = np.round(p)                       # integer vertices: ties in every rounding
            if o % 4 == 1 and k > 3:
                p[2:4] = p[0:2]                       # duplicate vertex
            parts.append(p)
        objs.append(parts)
    objs.append([])                                   # an object without polygons
    masks = _FakePolygonMasks(objs, H, W)
    n = 40
    x1, y1 = rng.uniform(-10, W - 4, n), rng.uniform(-10, H - 4, n)
    boxes = np.stack([x1, y1, x1 + rng.uniform(0.5, 90, n), y1 + rng.uniform(0.5, 90, n)], 1).astype(np.float32)
    boxes[0] = [3, 4, 3.2, 4.1]                       # thinner than one pixel: width / height floor at 1
    boxes[1] = [0, 0, W, H]
    inds = rng.integers(0, len(objs), n)
    head = type('H', (), {})()
    from dynamask_amd.mask_heads import DynaMaskHead
    head.stage_sup_size = [14, 28, 56, 112]
    tgt = DynaMaskHead.get_targets(head, [torch.from_numpy(boxes).to(dev)], [torch.from_numpy(inds).to(dev)], [masks])
    for size, t in zip(head.stage_sup_size, tgt):
        ref = rp.polygon_mask_targets(objs, H, W, boxes, inds, size, events=rp.fr_poly_points_literal)
        got = t.cpu().numpy()
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), (size, int((got != ref).sum()))
        assert 0.02 < ref.mean() < 0.9                # the scene is neither empty nor full


@pytest.mark.parametrize('size', [5, 33, 256])
def test_polygon_mask_targets_other_sizes(size):
    """dm_polygon_mask_targets at sizes the head does not use (odd, tiny, the largest it accepts)."""
    from dynamask_amd import ops
    from oracle import ref_poly as rp
    dev = torch.device('cuda')
    objs = [[np.array([3.0, 2.0, 30.5, 4.0, 41.0, 25.0, 22.0, 38.5, 5.0, 21.0])],
            [np.array([10.0, 10.0, 20.0, 10.0, 20.0, 20.0, 10.0, 20.0]), np.array([15.0, 15.0, 44.0, 18.0, 30.0, 44.0])]]
    boxes = np.array([[0, 0, 48, 40], [8, 8, 24, 24], [12.5, 9.25, 47.0, 39.5]], np.float32)
    inds = np.array([0, 1, 1])
    got = ops.polygon_mask_targets(ops.pack_polygons(objs, dev), torch.from_numpy(boxes).to(dev),
                                   torch.from_numpy(inds).to(dev), size).cpu().numpy()
    ref = rp.polygon_mask_targets(objs, 40, 48, boxes, inds, size, events=rp.fr_poly_events)
    assert np.array_equal(got, ref)
    assert 0.05 < ref.mean() < 0.95


def test_bucketed_inference_graphs_replay_the_eager_launch_sequence(no_splitk):
    """DynaMaskRoIHead.enable_inference_graphs(): simple_test_mask_logits through a HIP graph per bucket of detection
    counts (16 / 24 / 32 / 48 / 64 / 80 / 100, padded with empty boxes) gives the bits of the eager call (with split-K off: a bucket
    pads the RoI count, and the split of a launch depends on it -- ``no_splitk``), captures once per bucket
    and map storage, follows a parameter update, and leaves counts above the largest bucket to the eager path."""
    from dynamask_amd import ops, synth
    m = _roi_head().eval()
    feats = [_dev(f) for f in synth.make_fpn(1, 608, 1024, 256, seed=3)]
    rois = synth.make_rois(1, 128, 608, 1024, seed=4)
    labels = _dev(synth.make_labels(128, seed=5))
    boxes = _dev(rois[:, 1:])
    with torch.no_grad():
        eager = {n: m.simple_test_mask_logits(feats, boxes[:n], labels[:n]).clone() for n in (1, 16, 17, 100, 128)}
        gl = m.enable_inference_graphs(True)
        for n in (1, 16, 17, 100):
            got = m.simple_test_mask_logits(feats, boxes[:n], labels[:n])
            assert got.shape == eager[n].shape and torch.equal(got, eager[n]), n
        assert gl.captures == 3 and gl.replays == 4            # buckets 16 (n = 1, 16), 24 (17), 100
        assert torch.equal(m.simple_test_mask_logits(feats, boxes[:16], labels[:16]), eager[16]) and gl.captures == 3
        # the graph's static RoI buffer: a call with fewer boxes than the last one zeroes the rows it leaves (empty boxes), and
        # the [n, 5] form (batch index in column 0) followed by the boxes form leaves no batch index behind
        assert torch.equal(m.simple_test_mask_logits(feats, boxes[:1], labels[:1]), eager[1])
        static_rois = next(v[1] for k, v in gl._graphs.items() if k[0] == 16)
        assert torch.equal(static_rois[0, 1:], boxes[0]) and not static_rois[1:].any() and not static_rois[:, 0].any()
        rois5 = torch.cat([torch.zeros(16, 1, device=boxes.device), boxes[:16]], 1)
        assert torch.equal(gl(feats, rois5, labels[:16]), eager[16])
        static_rois[:, 0] = 7.0                                  # (as a multi-image caller of the [n, 5] form would leave it)
        assert torch.equal(m.simple_test_mask_logits(feats, boxes[:16], labels[:16]), eager[16]) and not static_rois[:, 0].any()
        gl.replays -= 3
        assert torch.equal(m.simple_test_mask_logits(feats, boxes, labels), eager[128]) and gl.replays == 5      # eager
        # a parameter update (the fused SGD step bumps the epoch) invalidates the packed weights the graph holds
        with torch.no_grad():
            m.mask_head.final_instance_logits.weight.mul_(2.0)
        ops.WEIGHT_EPOCH[0] += 1
        m.enable_inference_graphs(False)
        ref = m.simple_test_mask_logits(feats, boxes[:16], labels[:16]).clone()
        m._mask_graphs = gl
        assert torch.equal(m.simple_test_mask_logits(feats, boxes[:16], labels[:16]), ref) and gl.captures == 4
        assert not torch.equal(ref, eager[16])
        # an in-place update that does NOT bump the epoch (torch.optim step, load_state_dict, copy_): a conv weight the
        # graph reads through its packed copy -- the key carries every parameter's (address, version)
        with torch.no_grad():
            m.mask_head.instance_convs[0].conv.weight.mul_(1.25)
        m.enable_inference_graphs(False)
        ref2 = m.simple_test_mask_logits(feats, boxes[:16], labels[:16]).clone()
        m._mask_graphs = gl
        assert torch.equal(m.simple_test_mask_logits(feats, boxes[:16], labels[:16]), ref2) and gl.captures == 5
        assert not torch.equal(ref2, ref)
        m.enable_inference_graphs(False)
