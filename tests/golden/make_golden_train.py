"""Golden vectors for the TRAINING entry point of the RoI head, from the REFERENCE's own
modules run on the CPU: ``MaxIoUAssigner`` + ``BboxOverlaps2D``, ``RandomSampler`` /
``SamplingResult``, ``BBoxHead.get_targets`` / ``loss`` (``CrossEntropyLoss``, ``L1Loss``,
``accuracy``, ``bbox2delta``) and the whole ``DynaMaskRoIHead.forward_train``
(dynamask_roi_head.py:21-73) with gradients.  Reference files are loaded by path with the
stand-ins of make_golden.py (mmcv's four operators delegate to oracle/ref_ops.py: those stay
parity-unpinned; everything else in this fixture is the reference's own arithmetic).

Run ONLY in the authoring container:  python tests/golden/make_golden_train.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import golden_inputs as gi  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def load_train_reference():
    R = {}
    mg.install_standins()
    core = mg._pkg('mmdet.core')
    core.auto_fp16 = mg._identity_decorator
    core.force_fp32 = mg._identity_decorator
    core.multi_apply = lambda f, *a, **k: tuple(map(list, zip(*map(lambda *x: f(*x, **k), *a))))
    core.multiclass_nms = None
    mu = mg._pkg('mmdet.utils')
    um = mg._load('mmdet.utils.util_mixins', 'mmdet/utils/util_mixins.py')
    mu.util_mixins = um
    bb = mg._pkg('mmdet.core.bbox')
    bld = mg._load('mmdet.core.bbox.builder', 'mmdet/core/bbox/builder.py')
    core.build_assigner, core.build_sampler, core.build_bbox_coder = bld.build_assigner, bld.build_sampler, bld.build_bbox_coder
    bb.demodata = mg._load('mmdet.core.bbox.demodata', 'mmdet/core/bbox/demodata.py')
    mg._pkg('mmdet.core.bbox.iou_calculators')
    icb = mg._load('mmdet.core.bbox.iou_calculators.builder', 'mmdet/core/bbox/iou_calculators/builder.py')
    ic = mg._load('mmdet.core.bbox.iou_calculators.iou2d_calculator', 'mmdet/core/bbox/iou_calculators/iou2d_calculator.py')
    sys.modules['mmdet.core.bbox.iou_calculators'].build_iou_calculator = icb.build_iou_calculator
    R['iou'] = ic
    mg._pkg('mmdet.core.bbox.assigners')
    mg._load('mmdet.core.bbox.assigners.assign_result', 'mmdet/core/bbox/assigners/assign_result.py')
    mg._load('mmdet.core.bbox.assigners.base_assigner', 'mmdet/core/bbox/assigners/base_assigner.py')
    R['assigner'] = mg._load('mmdet.core.bbox.assigners.max_iou_assigner', 'mmdet/core/bbox/assigners/max_iou_assigner.py')
    mg._pkg('mmdet.core.bbox.samplers')
    mg._load('mmdet.core.bbox.samplers.sampling_result', 'mmdet/core/bbox/samplers/sampling_result.py')
    mg._load('mmdet.core.bbox.samplers.base_sampler', 'mmdet/core/bbox/samplers/base_sampler.py')
    R['sampler'] = mg._load('mmdet.core.bbox.samplers.random_sampler', 'mmdet/core/bbox/samplers/random_sampler.py')
    mg._pkg('mmdet.core.bbox.coder')
    mg._load('mmdet.core.bbox.coder.base_bbox_coder', 'mmdet/core/bbox/coder/base_bbox_coder.py')
    R['coder'] = mg._load('mmdet.core.bbox.coder.delta_xywh_bbox_coder', 'mmdet/core/bbox/coder/delta_xywh_bbox_coder.py')
    ref = mg.load_reference()          # builder, losses, extractors, heads, roi heads (re-installs the stand-ins)
    core.build_assigner, core.build_sampler, core.build_bbox_coder = bld.build_assigner, bld.build_sampler, bld.build_bbox_coder
    core.multi_apply = lambda f, *a, **k: tuple(map(list, zip(*map(lambda *x: f(*x, **k), *a))))
    R.update(ref)
    srh = sys.modules['mmdet.models.roi_heads.standard_roi_head']      # imported the names while they were placeholders
    srh.build_assigner, srh.build_sampler = bld.build_assigner, bld.build_sampler
    losses = sys.modules['mmdet.models.losses']
    acc = mg._load('mmdet.models.losses.accuracy', 'mmdet/models/losses/accuracy.py')
    losses.accuracy = acc.accuracy
    R['l1'] = mg._load('mmdet.models.losses.smooth_l1_loss', 'mmdet/models/losses/smooth_l1_loss.py')
    mg._pkg('mmdet.models.roi_heads.bbox_heads')
    mg._load('mmdet.models.roi_heads.bbox_heads.bbox_head', 'mmdet/models/roi_heads/bbox_heads/bbox_head.py')
    R['convfc'] = mg._load('mmdet.models.roi_heads.bbox_heads.convfc_bbox_head',
                           'mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py')
    R['structures'] = mg._load('mmdet.core.mask.structures', 'mmdet/core/mask/structures.py')
    return R


def build_reference_roi_head(R):
    cfg = types.SimpleNamespace()
    train_cfg = mg_config(gi.RCNN_TRAIN_CFG)
    roi_cls = R['roi'].DynaMaskRoIHead
    # the registries of the stand-in builder hold what the loaded reference files registered
    head = roi_cls(bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
                   bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
                   mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                   mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG), train_cfg=train_cfg, test_cfg=None)
    sd = {**gi.head_state(), **gi.mask_pre_state(), **gi.bbox_train_head_state()}
    head.load_state_dict(sd, strict=True)
    head.train()
    del cfg
    return head


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def mg_config(d):
    return _Cfg({k: (mg_config(v) if isinstance(v, dict) else v) for k, v in d.items()})


def main():
    torch.set_num_threads(4)
    R = load_train_reference()
    ti = gi.train_inputs()
    out = {}
    # ---------------- pieces: IoU matrix, assignment, sampling, bbox targets
    asg = R['assigner'].MaxIoUAssigner(**{k: v for k, v in gi.RCNN_TRAIN_CFG['assigner'].items() if k != 'type'})
    smp = R['sampler'].RandomSampler(**{k: v for k, v in gi.RCNN_TRAIN_CFG['sampler'].items() if k != 'type'})
    torch.manual_seed(gi.TRAIN_SEED)
    srs = []
    for i in range(2):
        ov = R['iou'].bbox_overlaps(ti['gt_bboxes'][i], ti['proposals'][i][:, :4])
        out[f'overlaps{i}'] = _np(ov)
        out[f'iof{i}'] = _np(R['iou'].bbox_overlaps(ti['gt_bboxes'][i], ti['proposals'][i][:, :4], mode='iof'))
        ar = asg.assign(ti['proposals'][i], ti['gt_bboxes'][i], None, ti['gt_labels'][i])
        out[f'gt_inds{i}'], out[f'max_overlaps{i}'], out[f'assigned_labels{i}'] = _np(ar.gt_inds), _np(ar.max_overlaps), _np(ar.labels)
        sr = smp.sample(ar, ti['proposals'][i], ti['gt_bboxes'][i], ti['gt_labels'][i])
        out[f'pos_inds{i}'], out[f'neg_inds{i}'] = _np(sr.pos_inds), _np(sr.neg_inds)
        out[f'pos_assigned_gt_inds{i}'] = _np(sr.pos_assigned_gt_inds)
        out[f'pos_is_gt{i}'] = _np(sr.pos_is_gt)
        srs.append(sr)
    # the other branches of the assigner
    asg2 = R['assigner'].MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.3, gt_max_assign_all=False,
                                        match_low_quality=True)
    ar2 = asg2.assign(ti['proposals'][0], ti['gt_bboxes'][0], None, None)
    out['alt_gt_inds'] = _np(ar2.gt_inds)
    asg3 = R['assigner'].MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    out['nolq_gt_inds'] = _np(asg3.assign(ti['proposals'][1], ti['gt_bboxes'][1], None, ti['gt_labels'][1]).gt_inds)

    head = build_reference_roi_head(R)
    tc = mg_config(gi.RCNN_TRAIN_CFG)
    lab, lw, bt, bw = head.bbox_head.get_targets(srs, ti['gt_bboxes'], ti['gt_labels'], tc)
    out['labels'], out['label_weights'], out['bbox_targets'], out['bbox_weights'] = _np(lab), _np(lw), _np(bt), _np(bw)

    # ---------------- bbox losses alone, with gradients wrt the predictions
    g = torch.Generator().manual_seed(304)
    n = lab.shape[0]
    cs = (torch.randn(n, 81, generator=g) * 2).requires_grad_(True)
    bp = (torch.randn(n, 320, generator=g) * 0.5).requires_grad_(True)
    out['in_cls_score'], out['in_bbox_pred'] = _np(cs), _np(bp)
    rois = R['roi'].bbox2roi([r.bboxes for r in srs])
    ls = head.bbox_head.loss(cs, bp, rois, lab, lw, bt, bw)
    (ls['loss_cls'] * 1.5 + ls['loss_bbox'] * 0.5).backward()
    out['loss_cls_alone'], out['acc_alone'], out['loss_bbox_alone'] = _np(ls['loss_cls']), _np(ls['acc']), _np(ls['loss_bbox'])
    out['grad_cls_score'], out['grad_bbox_pred'] = _np(cs.grad), _np(bp.grad)
    # no positive row: loss_bbox = bbox_pred.sum() * 0
    ls0 = head.bbox_head.loss(cs.detach(), bp.detach(), rois, torch.full_like(lab, 80), lw, bt, bw)
    out['loss_bbox_no_pos'] = _np(ls0['loss_bbox'])

    # ---------------- the whole forward_train with gradients
    gtm = [R['structures'].BitmapMasks(m.numpy(), m.shape[1], m.shape[2]) for m in ti['gt_masks']]
    feats = [f.clone().requires_grad_(True) for f in ti['feats']]
    for p in head.parameters():
        p.grad = None
    torch.manual_seed(gi.TRAIN_SEED)
    losses = head.forward_train(feats, ti['img_metas'], ti['proposals'], ti['gt_bboxes'], ti['gt_labels'], None, gtm)
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    for k, v in losses.items():
        out['ft.' + k] = _np(v)
    named = dict(head.named_parameters())
    for k in gi.BBOX_GRAD_KEYS:
        out['ft.grad.bbox_head.' + k] = _np(gi.grad_slice(named['bbox_head.' + k].grad))
    for k in gi.GRAD_KEYS:
        out['ft.grad.mask_head.' + k] = _np(gi.grad_slice(named['mask_head.' + k].grad))
    for k in ('conv1.weight', 'bn1.weight', 'fc2.weight'):
        out['ft.grad.mask_predictor.' + k] = _np(gi.grad_slice(named['mask_predictor.' + k].grad))
    for i in range(4):
        out[f'ft.grad_feat{i}'] = (_np(gi.feat_grad_slice(feats[i].grad)) if feats[i].grad is not None
                                   else np.zeros(1, np.float32))
    np.savez_compressed(os.path.join(HERE, 'g11_train.npz'), **out)
    print('g11:', {k: (v.shape if v.ndim else float(v)) for k, v in out.items() if k.startswith('ft.') and 'grad' not in k})
    print('pos per image', [len(out[f'pos_inds{i}']) for i in range(2)], 'neg', [len(out[f'neg_inds{i}']) for i in range(2)])


if __name__ == '__main__':
    main()
