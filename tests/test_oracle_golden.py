"""Oracle (oracle/ref_model.py) vs golden vectors produced by the REFERENCE's
own modules (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import torch

import golden_inputs as gi
from oracle import ref_model, ref_ops

TOL = dict(atol=1e-5, rtol=1e-5)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(a, b, **kw):
    kw = kw or TOL
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, **kw)


def test_losses_match_reference(golden_dir):
    g = _load(golden_dir, 'g1_losses.npz')
    li = gi.loss_inputs()
    ips = [t.clone().requires_grad_(True) for t in li['ips']]
    dps = [t.clone().requires_grad_(True) for t in li['dps']]
    ml = li['mask_labels'].clone().requires_grad_(True)
    loss = ref_model.dyna_loss(ips, dps, li['targets'], ml)
    loss.backward()
    _close(loss, g['loss_masks'])
    _close(ml.grad, g['grad_mask_labels'])
    for i in range(4):
        _close(dps[i].grad, g[f'grad_dp{i}'])
        gip = ips[i].grad if ips[i].grad is not None else torch.zeros_like(ips[i])
        _close(gip, g[f'grad_ip{i}'])
        # integer-valued stencil output: bit exact
        assert np.array_equal(ref_model.detail_target(li['targets'][i]).numpy(), g[f'detail_target{i}'])
    for bw in (1, 2, 3):
        assert np.array_equal(ref_model.generate_block_target(li['targets'][1], bw).numpy(), g[f'block_target_bw{bw}'])
    _close(ref_model.binary_cross_entropy(li['ips'][1].squeeze(1), li['targets'][1]), g['bce_stage1'])
    _close(ref_model.mask_cross_entropy(li['dps'][1].squeeze(1), li['targets'][1],
                                        li['mask_labels'][:, 1].view(-1, 1, 1)), g['epsbce_stage1'])


def test_mask_pre_matches_reference(golden_dir):
    g = _load(golden_dir, 'g2_maskpre.npz')
    sd = gi.mask_pre_state()
    x = gi.mask_pre_input()
    _close(ref_model.mask_pre(sd, x, training=True), g['logits_train'], atol=2e-5, rtol=1e-4)
    _close(ref_model.mask_pre(sd, x, training=False), g['logits_eval'], atol=2e-5, rtol=1e-4)


def test_gumbel_selector_matches_reference(golden_dir):
    g = _load(golden_dir, 'g3_gumbel.npz')
    logits = gi.gumbel_logits().requires_grad_(True)
    torch.manual_seed(gi.GUMBEL_SEED)
    U = torch.rand(logits.shape)          # the draw the reference makes (dynamask_roi_head.py:90)
    y, ind = ref_model.gumbel_select(logits, U, 0.5)
    assert np.array_equal(ind.numpy(), g['index'])
    _close(y, g['y_hard'])
    (y * torch.arange(1, 5, dtype=torch.float32)).sum().backward()
    _close(logits.grad, g['grad_logits'])


def test_head_forward_matches_reference(golden_dir):
    g = _load(golden_dir, 'g4_head.npz')
    hi = gi.head_inputs()
    sd = gi.head_state()
    assert np.array_equal(ref_ops.map_roi_levels(hi['rois'], 4).numpy(), g['levels'])
    assert set(g['levels'].tolist()) == {0, 1, 2, 3}
    ins = ref_ops.single_roi_extractor(hi['feats'][:4], hi['rois'], 14, (4, 8, 16, 32))
    _close(ins, g['ins_feats'])
    with torch.no_grad():
        ips, dps = ref_model.mask_forward(sd, hi['feats'], hi['rois'], hi['labels'])
    for i in range(4):
        _close(ips[i], g[f'ip{i}'], atol=2e-5, rtol=1e-4)
        _close(dps[i], g[f'dp{i}'], atol=2e-5, rtol=1e-4)


def test_boundary_merge_matches_reference(golden_dir):
    g = _load(golden_dir, 'g5_merge.npz')
    mi = gi.merge_inputs()
    _close(ref_model.boundary_merge(mi['ips']), g['merged'])


def test_fcn_head_matches_reference(golden_dir):
    g = _load(golden_dir, 'g6_fcn.npz')
    x = gi.fcn_input()
    for up in ('deconv', 'carafe', 'bilinear'):
        sd = gi.fcn_state(up)
        with torch.no_grad():
            out = ref_model.fcn_mask_head_forward(sd, x, pre='mask_head.', upsample=up)
        _close(out, g[up], atol=2e-5, rtol=1e-4)


def test_head_train_slice_matches_reference(golden_dir):
    g = _load(golden_dir, 'g7_head_train.npz')
    hi = gi.head_inputs()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in gi.head_state().items()}
    feats = [f.clone().requires_grad_(True) for f in hi['feats']]
    ips, dps = ref_model.mask_forward(sd, feats, hi['rois'], hi['labels'])
    n = hi['rois'].shape[0]
    ml = gi.head_mask_labels(n).clone().requires_grad_(True)
    loss = ref_model.dyna_loss(ips, dps, gi.head_targets(n), ml,
                               fuse_kernel=sd['mask_head.loss_func.detail_target.fuse_kernel'])
    loss.backward()
    _close(loss, g['loss'], atol=1e-5, rtol=1e-5)
    _close(ml.grad, g['grad_mask_labels'], atol=1e-5, rtol=1e-4)
    for k in gi.GRAD_KEYS:
        _close(gi.grad_slice(sd['mask_head.' + k].grad), g['grad.' + k], atol=2e-6, rtol=1e-3)
    for i in range(4):
        _close(gi.feat_grad_slice(feats[i].grad), g[f'grad_feat{i}'], atol=2e-6, rtol=1e-3)


def test_paste_and_targets_match_reference(golden_dir):
    g8 = _load(golden_dir, 'g8_paste.npz')
    pi = gi.paste_inputs()
    for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
        # the goldens ran on CPU: the reference then pastes one mask per chunk with skip_empty=True
        out = ref_model.get_seg_masks(pi['logits'], pi['det_bboxes'], pi['ori_shape'], sf, rescale, device_type='cpu')
        assert np.array_equal(out.numpy().astype(np.uint8), g8[f'seg_rescale{int(rescale)}_sf{sf}'])
        # whole-canvas (GPU) path: identical except for the degenerate zero-width box (index 4)
        out2 = ref_model.get_seg_masks(pi['logits'], pi['det_bboxes'], pi['ori_shape'], sf, rescale, device_type='cuda')
        assert np.array_equal(out2.numpy()[:4], out.numpy()[:4])
    g9 = _load(golden_dir, 'g9_targets.npz')
    ti = gi.target_inputs()
    tg = ref_model.get_targets([t['boxes'] for t in ti], [t['inds'] for t in ti], [t['masks'] for t in ti])
    for i in range(4):
        assert np.array_equal(tg[i].numpy().astype(np.uint8), g9[f't{i}'])


def test_fcn_callers_oracle_matches_reference_golden(golden_dir):
    """g14: the reference's own FCNMaskHead.get_seg_masks / get_targets (fcn_mask_head.py:128-237, mask_target.py)."""
    g = _load(golden_dir, 'g14_fcn_callers.npz')
    pi = gi.fcn_paste_inputs()
    for ag in (False, True):
        logits = pi['logits'] if not ag else pi['logits'][torch.arange(7), pi['det_labels']][:, None]
        for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
            segs = ref_model.fcn_get_seg_masks(logits, pi['det_bboxes'], pi['det_labels'], pi['ori_shape'], sf, rescale,
                                               class_agnostic=ag, device_type='cpu')
            key = f'ag{int(ag)}_rescale{int(rescale)}_sf{sf}'
            assert np.array_equal(np.array([len(c) for c in segs], np.int32), g['counts_' + key])
            flat = np.stack([m.numpy() for c in segs for m in c]).astype(np.uint8)
            assert np.array_equal(flat, g['seg_' + key])
        segs = ref_model.fcn_get_seg_masks(logits.sigmoid().numpy(), pi['det_bboxes'], pi['det_labels'], pi['ori_shape'], 1.0, True,
                                           class_agnostic=ag, device_type='cpu')
        assert np.array_equal(np.stack([m.numpy() for c in segs for m in c]).astype(np.uint8), g[f'seg_ag{int(ag)}_ndarray'])
    ti = gi.target_inputs()
    for size in (28, 14):
        tg = ref_model.fcn_get_targets([t['boxes'] for t in ti], [t['inds'] for t in ti], [t['masks'] for t in ti], mask_size=size)
        assert np.array_equal(tg.numpy().astype(np.uint8), g[f'targets{size}'])


def test_bbox_branch_oracle_matches_reference_golden(golden_dir):
    """g10: Shared2FCBBoxHead.forward / BBoxHead.get_bboxes / delta2bbox / multiclass_nms of the
    reference (the NMS inside is a stand-in: mmcv is third-party, see make_golden_bbox.py)."""
    import golden_inputs as gi
    from oracle import ref_model
    g = np.load(os.path.join(golden_dir, 'g10_bbox.npz'))
    sd = gi.bbox_head_state()
    x, rois = gi.bbox_inputs()
    with torch.no_grad():
        cls_score, bbox_pred = ref_model.bbox_head_forward(sd, x)
        np.testing.assert_allclose(cls_score.numpy(), g['cls_score'], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(bbox_pred.numpy(), g['bbox_pred'], rtol=1e-4, atol=1e-4)
        cs, bp = torch.from_numpy(g['cls_score']), torch.from_numpy(g['bbox_pred'])
        b, s = ref_model.get_bboxes(rois, cs, bp, gi.BBOX_IMG_SHAPE, 1.0)
        np.testing.assert_allclose(b.numpy(), g['bboxes'], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(s.numpy(), g['scores'], rtol=1e-5, atol=1e-7)
        sf = np.array([1.25, 1.6, 1.25, 1.6], dtype=np.float32)
        b1, _ = ref_model.get_bboxes(rois, cs, bp, gi.BBOX_IMG_SHAPE, sf, rescale=True)
        np.testing.assert_allclose(b1.numpy(), g['bboxes_rescaled'], rtol=1e-5, atol=1e-4)
        d, lab = ref_model.get_bboxes(rois, cs, bp, gi.BBOX_IMG_SHAPE, 1.0, cfg=gi.RCNN_TEST_CFG)
        np.testing.assert_allclose(d.numpy(), g['det_bboxes'], rtol=1e-5, atol=1e-4)
        assert (lab.numpy() == g['det_labels']).all()
        r = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
        dl = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
        np.testing.assert_allclose(ref_model.delta2bbox(r, dl, max_shape=(32, 32)).numpy(), g['doc_example'], atol=1e-6)
        # the reference's own docstring values (delta_xywh_bbox_coder.py:155-160)
        np.testing.assert_allclose(g['doc_example'], [[0, 0, 1, 1], [0.1409, 0.1409, 2.8591, 2.8591],
                                                       [0, 0.3161, 4.1945, 0.6839], [5, 5, 5, 5]], atol=1e-4)


# ----------------------------------------------------------- training entry point (g11)
def _g11(golden_dir):
    return np.load(os.path.join(golden_dir, 'g11_train.npz'))


def test_assigner_sampler_targets_match_reference(golden_dir):
    """IoU matrix, MaxIoUAssigner (three configurations), RandomSampler under the reference's seed and
    BBoxHead.get_targets: indices bit-exact, floats to 1e-6."""
    g = _g11(golden_dir)
    ti = gi.train_inputs()
    a, s = gi.RCNN_TRAIN_CFG['assigner'], gi.RCNN_TRAIN_CFG['sampler']
    torch.manual_seed(gi.TRAIN_SEED)
    samples = []
    for i in range(2):
        ov = ref_model.bbox_overlaps(ti['gt_bboxes'][i], ti['proposals'][i])
        np.testing.assert_array_equal(ov.numpy(), g[f'overlaps{i}'])          # same fp32 ops in the same order
        np.testing.assert_array_equal(ref_model.bbox_overlaps(ti['gt_bboxes'][i], ti['proposals'][i], 'iof').numpy(),
                                      g[f'iof{i}'])
        gi_, mo, lab = ref_model.max_iou_assign(ov, a['pos_iou_thr'], a['neg_iou_thr'], a['min_pos_iou'], True, True,
                                                ti['gt_labels'][i])
        np.testing.assert_array_equal(gi_.numpy(), g[f'gt_inds{i}'])
        np.testing.assert_array_equal(mo.numpy(), g[f'max_overlaps{i}'])
        np.testing.assert_array_equal(lab.numpy(), g[f'assigned_labels{i}'])
        sm = ref_model.random_sample(gi_, lab, ti['proposals'][i], ti['gt_bboxes'][i], ti['gt_labels'][i], s['num'],
                                     s['pos_fraction'])
        for k in ('pos_inds', 'neg_inds', 'pos_assigned_gt_inds', 'pos_is_gt'):
            np.testing.assert_array_equal(sm[k].numpy(), g[f'{k}{i}'])
        samples.append(sm)
    ov0 = ref_model.bbox_overlaps(ti['gt_bboxes'][0], ti['proposals'][0])
    np.testing.assert_array_equal(ref_model.max_iou_assign(ov0, 0.7, (0.1, 0.3), 0.3, True, False)[0].numpy(), g['alt_gt_inds'])
    ov1 = ref_model.bbox_overlaps(ti['gt_bboxes'][1], ti['proposals'][1])
    np.testing.assert_array_equal(ref_model.max_iou_assign(ov1, 0.5, 0.5, 0.5, False, True, ti['gt_labels'][1])[0].numpy(),
                                  g['nolq_gt_inds'])
    lab, lw, bt, bw = ref_model.bbox_targets(samples)
    np.testing.assert_array_equal(lab.numpy(), g['labels'])
    np.testing.assert_array_equal(lw.numpy(), g['label_weights'])
    np.testing.assert_allclose(bt.numpy(), g['bbox_targets'], atol=1e-6, rtol=1e-6)
    np.testing.assert_array_equal(bw.numpy(), g['bbox_weights'])


def test_bbox_losses_match_reference(golden_dir):
    g = _g11(golden_dir)
    cs = torch.from_numpy(g['in_cls_score']).requires_grad_(True)
    bp = torch.from_numpy(g['in_bbox_pred']).requires_grad_(True)
    tg = [torch.from_numpy(g[k]) for k in ('labels', 'label_weights', 'bbox_targets', 'bbox_weights')]
    lc, acc, lb = ref_model.bbox_loss(cs, bp, *tg, loss_weight_cls=2.0, loss_weight_bbox=2.0)
    (lc * 1.5 + lb * 0.5).backward()
    np.testing.assert_allclose(lc.item(), g['loss_cls_alone'], rtol=1e-6)
    np.testing.assert_allclose(acc.item(), g['acc_alone'].item(), rtol=1e-6)
    np.testing.assert_allclose(lb.item(), g['loss_bbox_alone'], rtol=1e-6)
    np.testing.assert_allclose(cs.grad.numpy(), g['grad_cls_score'], atol=1e-7, rtol=1e-5)
    np.testing.assert_allclose(bp.grad.numpy(), g['grad_bbox_pred'], atol=1e-7, rtol=1e-5)
    tg[0] = torch.full_like(tg[0], 80)
    assert ref_model.bbox_loss(cs.detach(), bp.detach(), *tg)[2].item() == g['loss_bbox_no_pos'] == 0.0


def test_forward_train_matches_reference(golden_dir):
    """The whole DynaMaskRoIHead.forward_train: four losses and gradients of bbox-head, mask-head, MaskPre
    parameters and of the FPN maps."""
    g = _g11(golden_dir)
    ti = gi.train_inputs()
    sd = {k: v.clone().requires_grad_(True) if v.is_floating_point() else v
          for k, v in {**gi.head_state(), **gi.mask_pre_state(), **gi.bbox_train_head_state()}.items()}
    feats = [f.clone().requires_grad_(True) for f in ti['feats']]
    torch.manual_seed(gi.TRAIN_SEED)
    losses, samples, _, _ = ref_model.forward_train(sd, feats, ti['proposals'], ti['gt_bboxes'], ti['gt_labels'],
                                                    ti['gt_masks'], gi.RCNN_TRAIN_CFG)
    for i in range(2):
        np.testing.assert_array_equal(samples[i]['pos_inds'].numpy(), g[f'pos_inds{i}'])
    for k in ('loss_cls', 'acc', 'loss_bbox', 'loss_masks'):
        np.testing.assert_allclose(float(losses[k].detach()), float(g['ft.' + k].reshape(-1)[0]), rtol=2e-5, atol=1e-6)
    (losses['loss_cls'] + losses['loss_bbox'] + losses['loss_masks']).backward()
    for k in gi.BBOX_GRAD_KEYS:
        np.testing.assert_allclose(gi.grad_slice(sd['bbox_head.' + k].grad).numpy(), g['ft.grad.bbox_head.' + k],
                                   atol=1e-5, rtol=1e-4)
    for k in gi.GRAD_KEYS:
        np.testing.assert_allclose(gi.grad_slice(sd['mask_head.' + k].grad).numpy(), g['ft.grad.mask_head.' + k],
                                   atol=1e-5, rtol=1e-4)
    for k in ('conv1.weight', 'bn1.weight', 'fc2.weight'):
        np.testing.assert_allclose(gi.grad_slice(sd['mask_predictor.' + k].grad).numpy(), g['ft.grad.mask_predictor.' + k],
                                   atol=1e-5, rtol=1e-4)
    for i in range(4):
        if g[f'ft.grad_feat{i}'].size > 1:
            np.testing.assert_allclose(gi.feat_grad_slice(feats[i].grad).numpy(), g[f'ft.grad_feat{i}'], atol=1e-5, rtol=1e-4)


def test_assigner_ignore_regions_match_reference_golden(golden_dir):
    """g13: the reference's MaxIoUAssigner with gt_bboxes_ignore (max_iou_assigner.py:107-118), both IoF conventions."""
    g = np.load(os.path.join(golden_dir, 'g13_assign_ignore.npz'))
    b, gts, ign, lab = (torch.from_numpy(g[k]) for k in ('bboxes', 'gts', 'ign', 'labels'))
    for name, wrt in (('cand', True), ('region', False)):
        ov = ref_model.ignore_overlaps(ref_model.bbox_overlaps(gts, b), b, ign, 0.5, wrt)
        gi_, mo, lb = ref_model.max_iou_assign(ov, 0.5, 0.5, 0.5, gt_labels=lab)
        assert np.array_equal(gi_.numpy(), g[f'{name}_gt_inds'])
        np.testing.assert_array_equal(mo.numpy(), g[f'{name}_max_overlaps'])
        assert np.array_equal(lb.numpy(), g[f'{name}_labels'])
    assert (g['cand_gt_inds'] == -1).sum() > 0 and (g['noign_gt_inds'] == -1).sum() == 0
