"""Training entry point of the RoI head on the MI355X, through the registry classes and the C ABI:
assigner / sampler / bbox targets / bbox losses / ``DynaMaskRoIHead.forward_train`` against golden g11
(produced by the reference's own modules, tests/golden/make_golden_train.py) and, at the config's real
sizes (1000 proposals, 512 samples per image), against the oracle.  Indices bit-exact; floats 1e-4."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import ref_model
from tolerances import assert_grad_close

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def _g11(golden_dir):
    return np.load(os.path.join(golden_dir, 'g11_train.npz'))


def _eq(t, ref):
    np.testing.assert_array_equal(t.detach().cpu().numpy(), ref)


def _close(t, ref, **kw):
    np.testing.assert_allclose(t.detach().cpu().numpy(), ref, **(kw or TOL))


def _full_roi_head(train_cfg=None):
    from dynamask_amd import bbox_heads, losses, mask_heads, registry, roi_extractors, roi_head  # noqa: F401
    cfg = dict(type='DynaMaskRoIHead',
               bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
               bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
               mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG),
               train_cfg=registry._to_cfgdict(train_cfg or gi.RCNN_TRAIN_CFG), test_cfg=None)
    m = registry.build_head(cfg)
    m.load_state_dict({**gi.head_state(), **gi.mask_pre_state(), **gi.bbox_train_head_state()}, strict=True)
    m = m.cuda().train()
    m.bbox_sampler.cpu_rng = True          # the reference run that made the golden drew on the CPU generator
    return m


def test_overlaps_assigner_sampler_targets_match_reference_golden(golden_dir):
    g = _g11(golden_dir)
    ti = gi.train_inputs()
    m = _full_roi_head()
    from dynamask_amd import assigners
    torch.manual_seed(gi.TRAIN_SEED)
    srs = []
    for i in range(2):
        props, gtb, gtl = ti['proposals'][i].cuda(), ti['gt_bboxes'][i].cuda(), ti['gt_labels'][i].cuda()
        calc = assigners.BboxOverlaps2D()
        _eq(calc(gtb, props), g[f'overlaps{i}'])                      # bit-exact: thresholds decide on these
        _eq(calc(gtb, props, mode='iof'), g[f'iof{i}'])
        ar = m.bbox_assigner.assign(props, gtb, None, gtl)
        _eq(ar.gt_inds, g[f'gt_inds{i}'])
        _eq(ar.max_overlaps, g[f'max_overlaps{i}'])
        _eq(ar.labels, g[f'assigned_labels{i}'])
        sr = m.bbox_sampler.sample(ar, props, gtb, gtl)
        for k in ('pos_inds', 'neg_inds', 'pos_assigned_gt_inds', 'pos_is_gt'):
            _eq(getattr(sr, k), g[f'{k}{i}'])
        srs.append(sr)
    a2 = assigners.MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.3, gt_max_assign_all=False)
    _eq(a2.assign(ti['proposals'][0].cuda(), ti['gt_bboxes'][0].cuda()).gt_inds, g['alt_gt_inds'])
    a3 = assigners.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    _eq(a3.assign(ti['proposals'][1].cuda(), ti['gt_bboxes'][1].cuda(), None, ti['gt_labels'][1].cuda()).gt_inds,
        g['nolq_gt_inds'])
    lab, lw, bt, bw = m.bbox_head.get_targets(srs, None, None, m.train_cfg)
    _eq(lab, g['labels'])
    _eq(lw, g['label_weights'])
    _close(bt, g['bbox_targets'], atol=1e-6, rtol=1e-5)
    _eq(bw, g['bbox_weights'])


def test_assigner_empty_cases():
    from dynamask_amd import assigners
    a = assigners.MaxIoUAssigner(0.5, 0.5)
    boxes = torch.tensor([[0., 0., 10., 10.], [10., 10., 20., 20.]]).cuda()
    r = a.assign(boxes, torch.zeros((0, 4)).cuda(), None, torch.zeros((0,), dtype=torch.long).cuda())
    assert r.gt_inds.tolist() == [0, 0] and r.labels.tolist() == [-1, -1]          # no gt: all background
    r = a.assign(torch.zeros((0, 4)).cuda(), boxes)
    assert r.gt_inds.numel() == 0
    # the reference docstring example (max_iou_assigner.py:82-88)
    r = a.assign(boxes, torch.tensor([[0., 0., 10., 9.]]).cuda())
    assert r.gt_inds.tolist() == [1, 0]


def test_bbox_losses_and_gradients_match_reference_golden(golden_dir):
    g = _g11(golden_dir)
    m = _full_roi_head()
    cs = torch.from_numpy(g['in_cls_score']).cuda().requires_grad_(True)
    bp = torch.from_numpy(g['in_bbox_pred']).cuda().requires_grad_(True)
    tg = [torch.from_numpy(g[k]).cuda() for k in ('labels', 'label_weights', 'bbox_targets', 'bbox_weights')]
    ls = m.bbox_head.loss(cs, bp, None, *tg)
    (ls['loss_cls'] * 1.5 + ls['loss_bbox'] * 0.5).backward()
    _close(ls['loss_cls'], g['loss_cls_alone'], atol=1e-5, rtol=1e-5)
    assert float(ls['acc']) == float(g['acc_alone'].reshape(-1)[0])
    _close(ls['loss_bbox'], g['loss_bbox_alone'], atol=1e-6, rtol=1e-5)
    _close(cs.grad, g['grad_cls_score'], atol=1e-6, rtol=1e-4)
    _close(bp.grad, g['grad_bbox_pred'], atol=1e-7, rtol=1e-5)
    tg[0] = torch.full_like(tg[0], 80)                                              # no positive row
    ls0 = m.bbox_head.loss(cs.detach(), bp.detach(), None, *tg)
    assert float(ls0['loss_bbox']) == 0.0 == float(g['loss_bbox_no_pos'])


def test_forward_train_matches_reference_golden(golden_dir):
    """DynaMaskRoIHead.forward_train called with the reference's argument list
    (x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks) -- two_stage.py:161-164."""
    g = _g11(golden_dir)
    ti = gi.train_inputs()
    m = _full_roi_head()
    feats = [f.cuda().requires_grad_(True) for f in ti['feats']]
    gt_masks = [t.cuda() for t in ti['gt_masks']]
    torch.manual_seed(gi.TRAIN_SEED)
    losses = m.forward_train(feats, ti['img_metas'], [p.cuda() for p in ti['proposals']], [b.cuda() for b in ti['gt_bboxes']],
                             [l.cuda() for l in ti['gt_labels']], None, gt_masks)
    assert set(losses) == {'loss_cls', 'acc', 'loss_bbox', 'loss_masks'}
    for k in ('loss_cls', 'loss_bbox', 'loss_masks'):
        _close(losses[k].reshape(-1), g['ft.' + k].reshape(-1))
    assert float(losses['acc']) == float(g['ft.acc'].reshape(-1)[0])
    sum(v for k, v in losses.items() if 'loss' in k).backward()
    named = dict(m.named_parameters())
    worst = 0.0
    for pre, keys in (('bbox_head.', gi.BBOX_GRAD_KEYS), ('mask_head.', gi.GRAD_KEYS),
                      ('mask_predictor.', ('conv1.weight', 'bn1.weight', 'fc2.weight'))):
        for k in keys:
            got, ref = gi.grad_slice(named[pre + k].grad).cpu().numpy(), g['ft.grad.' + pre + k]
            worst = max(worst, float(np.abs(got - ref).max()))
            # (the golden's slice of fc_reg.weight covers rows of classes that no sampled positive has: exact zeros)
            assert_grad_close(got, ref, pre + k, zero=not np.any(ref))
    # FPN-map gradients: a cell is a sum over every RoI and sample that touches it, added in another order here (atomics)
    # than in the reference.  The fp64 triangle: the oracle's forward_train once more in float64 (same fp32 geometry,
    # same sampled boxes and selector indices -- asserted), and the product may be as far from it as the reference's own
    # fp32 run is, or 1e-4 of the map's peak, whichever is larger (round 3 allowed a flat 1e-3 of the peak).
    from tolerances import assert_close_via_f64
    sd64 = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() else v)
            for k, v in {**gi.head_state(), **gi.mask_pre_state(), **gi.bbox_train_head_state()}.items()}
    f64 = [f.clone().double().requires_grad_(True) for f in ti['feats']]
    torch.manual_seed(gi.TRAIN_SEED)
    l64, s64, _, _ = ref_model.forward_train(sd64, f64, ti['proposals'], ti['gt_bboxes'], ti['gt_labels'], ti['gt_masks'],
                                             gi.RCNN_TRAIN_CFG)
    for i in range(2):
        np.testing.assert_array_equal(s64[i]['pos_inds'].numpy(), g[f'pos_inds{i}'])
    assert abs(float(l64['loss_masks']) - float(g['ft.loss_masks'].reshape(-1)[0])) < 1e-5
    (l64['loss_cls'] + l64['loss_bbox'] + l64['loss_masks']).backward()
    for i in range(4):
        if g[f'ft.grad_feat{i}'].size > 1:
            e, re_, sc = assert_close_via_f64(gi.feat_grad_slice(feats[i].grad), g[f'ft.grad_feat{i}'],
                                              gi.feat_grad_slice(f64[i].grad), f'feat{i}')
            print(f'forward_train: d/dP{i + 2}: product - f64 {e:.3g}, reference(fp32) - f64 {re_:.3g}, scale {sc:.3g}')
    print('forward_train: worst parameter-gradient abs error', worst)


def test_reference_signature_mask_forward_train_equals_tensor_form(golden_dir):
    """_mask_forward_train(x, sampling_results, bbox_feats, gt_bboxes, gt_masks, gt_labels, img_metas)
    (dynamask_roi_head.py:48) unpacks to the tensor-level call of round 1."""
    ti = gi.train_inputs()
    m = _full_roi_head()
    feats = [f.cuda() for f in ti['feats']]
    torch.manual_seed(gi.TRAIN_SEED)
    srs = []
    for i in range(2):
        props, gtb, gtl = ti['proposals'][i].cuda(), ti['gt_bboxes'][i].cuda(), ti['gt_labels'][i].cuda()
        srs.append(m.bbox_sampler.sample(m.bbox_assigner.assign(props, gtb, None, gtl), props, gtb, gtl))
    gt_masks = [t.cuda() for t in ti['gt_masks']]
    n = sum(len(s.pos_inds) for s in srs)
    noise = torch.rand(n, 4).cuda()
    r1 = m._mask_forward_train(feats, srs, None, None, gt_masks, None, ti['img_metas'], noise=noise)
    from dynamask_amd.roi_head import bbox2roi
    tg = m.mask_head.get_targets([s.pos_bboxes for s in srs], [s.pos_assigned_gt_inds for s in srs], gt_masks)
    r2 = m._mask_forward_train(feats, bbox2roi([s.pos_bboxes for s in srs]).contiguous(),
                               torch.cat([s.pos_gt_labels for s in srs]), tg, noise=noise)
    assert torch.equal(r1['mask_index'], r2['mask_index'])
    _close(r1['loss_mask']['loss_masks'], r2['loss_mask']['loss_masks'].detach().cpu().numpy(), atol=1e-6, rtol=1e-6)
    assert 'loss_flops' in r1 and float(r1['loss_flops']['loss_flops']) >= 0.0      # attached, not summed (Q3)


def test_assign_sample_targets_at_config_size_match_oracle():
    """configs/dynamask sizes: 1000 proposals, 512 samples (25 % positives) per image, 15 / 7 gts."""
    from dynamask_amd import assigners, bbox_heads, registry  # noqa: F401
    g = torch.Generator().manual_seed(77)
    a_cfg = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=True, ignore_iof_thr=-1)
    asg = assigners.MaxIoUAssigner(**a_cfg)
    smp = assigners.RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True, cpu_rng=True)
    coder = bbox_heads.DeltaXYWHBBoxCoder(target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    for G in (15, 7):
        c = torch.rand(G, 2, generator=g) * torch.tensor([1100.0, 600.0]) + 100
        wh = torch.rand(G, 2, generator=g) * 300 + 30
        gtb = torch.cat([c - wh / 2, c + wh / 2], 1)
        gtl = torch.randint(0, 80, (G,), generator=g)
        near = gtb[torch.randint(0, G, (600,), generator=g)] + torch.randn(600, 4, generator=g) * 15
        far = torch.rand(400, 2, generator=g) * torch.tensor([1200.0, 700.0])
        far = torch.cat([far, far + torch.rand(400, 2, generator=g) * 250 + 8], 1)
        props = torch.cat([near, far, ])
        props = torch.cat([props, torch.rand(1000, 1, generator=g)], 1)
        ov_ref = ref_model.bbox_overlaps(gtb, props)
        gi_ref, mo_ref, lab_ref = ref_model.max_iou_assign(ov_ref, 0.5, 0.5, 0.5, True, True, gtl)
        ar = asg.assign(props.cuda(), gtb.cuda(), None, gtl.cuda())
        _eq(ar.gt_inds, gi_ref.numpy())
        _eq(ar.max_overlaps, mo_ref.numpy())
        _eq(ar.labels, lab_ref.numpy())
        torch.manual_seed(5)
        sm_ref = ref_model.random_sample(gi_ref, lab_ref, props, gtb, gtl, 512, 0.25)
        torch.manual_seed(5)
        sr = smp.sample(ar, props.cuda(), gtb.cuda(), gtl.cuda())
        _eq(sr.pos_inds, sm_ref['pos_inds'].numpy())
        _eq(sr.neg_inds, sm_ref['neg_inds'].numpy())
        assert len(sr.pos_inds) == 128 and len(sr.neg_inds) == 384
        _close(coder.encode(sr.pos_bboxes, sr.pos_gt_bboxes),
               ref_model.bbox2delta(sm_ref['pos_bboxes'], sm_ref['pos_gt_bboxes'], stds=(0.1, 0.1, 0.2, 0.2)).numpy(),
               atol=1e-5, rtol=1e-5)


def test_clip_grad_norm_on_flat_group():
    """optimizer_config grad_clip(max_norm=35, norm_type=2): dm_sumsq + dm_clip_scale vs torch."""
    from dynamask_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4162462, generator=g) * 0.05).cuda()
    ss = ops.sumsq(x)
    ref = float((x.double() ** 2).sum())
    assert abs(float(ss) - ref) / ref < 1e-5
    y = x.clone()
    ops.clip_scale_(y, ss, 35.0)
    coef = 35.0 / (ref ** 0.5 + 1e-6)
    torch.testing.assert_close(y, x * min(1.0, coef), atol=1e-7, rtol=1e-5)
    assert coef < 1.0                                                   # the clip was active
    z = x.clone()
    ops.clip_scale_(z, ss, 1e6)                                         # inactive: untouched, bit for bit
    assert torch.equal(z, x)


def test_forward_train_with_an_image_without_ground_truth():
    """One of the two images has no GT box: everything of it is background; the mask branch sees only the
    other image's positives.  And a batch with no GT at all: zero mask loss, bbox losses still defined."""
    ti = gi.train_inputs()
    m = _full_roi_head()
    feats = [f.cuda() for f in ti['feats']]
    props = [p.cuda() for p in ti['proposals']]
    gtb = [ti['gt_bboxes'][0].cuda(), torch.zeros((0, 4)).cuda()]
    gtl = [ti['gt_labels'][0].cuda(), torch.zeros((0,), dtype=torch.long).cuda()]
    gtm = [ti['gt_masks'][0].cuda(), torch.zeros((0, gi.IMG_H, gi.IMG_W), dtype=torch.uint8).cuda()]
    torch.manual_seed(3)
    losses = m.forward_train(feats, ti['img_metas'], props, gtb, gtl, None, gtm)
    assert all(torch.isfinite(v).all() for v in losses.values())
    sum(v for k, v in losses.items() if 'loss' in k).backward()
    gtb0 = [torch.zeros((0, 4)).cuda() for _ in range(2)]
    gtl0 = [torch.zeros((0,), dtype=torch.long).cuda() for _ in range(2)]
    gtm0 = [torch.zeros((0, gi.IMG_H, gi.IMG_W), dtype=torch.uint8).cuda() for _ in range(2)]
    losses0 = m.forward_train(feats, ti['img_metas'], props, gtb0, gtl0, None, gtm0)
    assert float(losses0['loss_masks'].detach()) == 0.0 and float(losses0['loss_bbox'].detach()) == 0.0
    assert torch.isfinite(losses0['loss_cls']).all()


def test_side_streams_come_from_one_shared_pool():
    """Hardware queues are few (4 per process unless GPU_MAX_HW_QUEUES says otherwise): the inference path, the
    training step and the gradient all-reduce must draw their side streams from one pool of ``streams.POOL``,
    or streams meant to run beside each other end up sharing a queue (training step 23.7 -> 25.5 ms after an
    inference pass in the same process, round 2)."""
    from dynamask_amd import streams, train_path, registry, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    from dynamask_amd.dist import FlatParamGroup
    dev = torch.device('cuda', torch.cuda.current_device())
    pool = {streams.side(dev, i).cuda_stream for i in range(8)}
    assert len(pool) == streams.POOL
    used = {train_path.side_stream(dev, w).cuda_stream for w in ('leaf', 'selector', 'coord', 'bbox')}
    head = registry.build_head(dict(type='DynaMaskRoIHead',
                                    mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                                    mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
    used |= {s.cuda_stream for s in head._side_streams(4, dev)}
    grp = FlatParamGroup([torch.nn.Parameter(torch.zeros(8, device=dev))])
    used.add(grp._stream.cuda_stream)
    assert used <= pool
    assert torch.cuda.default_stream(dev).cuda_stream not in pool


def test_training_step_on_side_streams_matches_the_single_stream_step(monkeypatch):
    """The step overlaps its branches on three side streams (leaf work, selector, coordinate gradient).  Two steps at
    the benchmark's size (2 x 128 RoIs on the 800 x 1333 pyramid) with and without them must give the same parameters
    up to the last-bit noise of the atomically accumulated weight gradients -- a missing stream dependency or a tensor
    recycled under a side stream shows up as a gross difference."""
    from dynamask_amd import synth, registry, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    dev = torch.device('cuda')
    B, per, H, W = 2, 128, 800, 1333
    feats = [f.to(dev) for f in synth.make_fpn(B, H, W, 256, seed=10)]
    rois = synth.make_rois(B, per, H, W, seed=11).to(dev)
    labels = synth.make_labels(B * per, seed=12).to(dev)
    targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
    noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)

    def run(side):
        monkeypatch.setenv('DM_TRAIN_SIDE_STREAM', '1' if side else '0')
        m = registry.build_head(dict(type='DynaMaskRoIHead',
                                     mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                                     mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
        m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
        m = m.to(dev).train()
        grp = FlatParamGroup(mask_path_parameters(m))
        out = []
        for _ in range(2):
            grp.zero_grad()
            res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            out.append((float(res['loss_mask']['loss_masks'].detach()), grp.flat_grad.clone()))
            grp.all_reduce_async()
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        torch.cuda.synchronize()
        return out, grp.flat_param.clone()
    (a0, a1), pa = run(True)
    (b0, b1), pb = run(False)
    assert abs(a0[0] - b0[0]) <= 1e-6 * abs(b0[0])
    scale = float(b0[1].abs().max())
    assert float((a0[1] - b0[1]).abs().max()) <= 2e-5 * scale, 'first-step gradients differ'
    assert abs(a1[0] - b1[0]) <= 1e-5 * abs(b1[0])
    torch.testing.assert_close(pa, pb, atol=2e-6, rtol=1e-4)


def test_direct_mask_forward_after_a_weight_update_reads_refreshed_packs(monkeypatch):
    """``_mask_forward`` under grad WITHOUT the RoI head's up-front PACK_PLAN.refresh() (no selector branch:
    ``_INPUTS_READY`` is None), right after an in-place weight update.  The packs are then refreshed inside
    MaskHeadFn.issue; side streams that read them must be ordered behind that refresh.  Loss and gradients must equal
    the single-stream run of the same two calls."""
    from dynamask_amd import synth, registry, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    dev = torch.device('cuda')
    B, per, H, W = 2, 96, 800, 1333
    feats = [f.to(dev) for f in synth.make_fpn(B, H, W, 256, seed=20)]
    rois = synth.make_rois(B, per, H, W, seed=21).to(dev)
    labels = synth.make_labels(B * per, seed=22).to(dev)

    def run(side):
        monkeypatch.setenv('DM_TRAIN_SIDE_STREAM', '1' if side else '0')
        m = registry.build_head(dict(type='DynaMaskRoIHead',
                                     mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                                     mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
        m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
        m = m.to(dev).train()
        outs = []
        for it in range(3):
            for p_ in m.mask_head.parameters():
                p_.grad = None
            res = m._mask_forward(feats, rois, labels)
            loss = sum(t.square().mean() for t in res['stage_instance_preds']) + sum(t.square().mean() for t in res['stage_detail_preds'])
            loss.backward()
            g = torch.cat([p_.grad.flatten() for p_ in m.mask_head.parameters() if p_.grad is not None])
            outs.append((float(loss.detach()), g.clone()))
            with torch.no_grad():          # an in-place update that bumps _version only
                for p_ in m.mask_head.parameters():
                    p_.mul_(1.0 + 0.05 * (it + 1))
        torch.cuda.synchronize()
        return outs
    a, b = run(True), run(False)
    assert abs(a[1][0] - a[0][0]) > 1e-3 * abs(a[0][0]), 'the update must change the loss'
    for (la, ga), (lb, gb) in zip(a, b):
        assert abs(la - lb) <= 1e-5 * abs(lb)
        assert float((ga - gb).abs().max()) <= 5e-5 * float(gb.abs().max())


def test_deterministic_mode_gives_bit_identical_training_runs(monkeypatch):
    """DM_DETERMINISTIC (ops.DETERMINISTIC): every cross-workgroup accumulation of the step goes through 64-bit
    fixed-point cells, so two runs of a 2-step training loop at the benchmark's size -- and a third one without the
    side streams -- end in torch.equal parameters, losses and gradients.  The default mode (float atomics) agrees with
    it to the usual last-bit noise."""
    from dynamask_amd import ops, synth, registry, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    dev = torch.device('cuda')
    B, per, H, W = 2, 128, 800, 1333
    feats = [f.to(dev) for f in synth.make_fpn(B, H, W, 256, seed=10)]
    rois = synth.make_rois(B, per, H, W, seed=11).to(dev)
    labels = synth.make_labels(B * per, seed=12).to(dev)
    targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
    noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)

    def run(det, side=True):
        monkeypatch.setenv('DM_TRAIN_SIDE_STREAM', '1' if side else '0')
        ops.DETERMINISTIC[0] = det
        m = registry.build_head(dict(type='DynaMaskRoIHead',
                                     mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                                     mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
        m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
        m = m.to(dev).train()
        grp = FlatParamGroup(mask_path_parameters(m))
        out = []
        for _ in range(2):
            grp.zero_grad()
            res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            out.append((res['loss_mask']['loss_masks'].detach().clone(), grp.flat_grad.clone()))
            grp.all_reduce_async()
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        torch.cuda.synchronize()
        return out, grp.flat_param.clone()
    try:
        a, pa = run(True)
        b, pb = run(True)
        c, pc = run(True, side=False)
        d, pd = run(False)
    finally:
        ops.DETERMINISTIC[0] = False
    for x, px, what in ((b, pb, 'second run'), (c, pc, 'run without side streams')):
        for s in range(2):
            assert torch.equal(a[s][0], x[s][0]), f'{what}: loss of step {s}'
            assert torch.equal(a[s][1], x[s][1]), f'{what}: gradients of step {s}'
        assert torch.equal(pa, px), f'{what}: parameters'
    # the default mode: float atomics, and (round 5) MaskPre's conv1 on the P2 map with its weight gradient through the
    # adjoint of the 56 x 56 extraction -- another association of the same products (the deterministic mode keeps the
    # reference's order): 1e-4 of the gradient's scale, was 2e-5 for the atomics' arrival order alone
    scale = float(a[0][1].abs().max())
    assert float((a[0][1] - d[0][1]).abs().max()) <= 1e-4 * scale
    torch.testing.assert_close(pa, pd, atol=1e-5, rtol=1e-4)


def test_assigner_ignore_regions_match_reference_golden(golden_dir):
    """MaxIoUAssigner with gt_bboxes_ignore (max_iou_assigner.py:107-118) on the device against the reference's own
    assigner (g13): indices and labels bit-exact, both IoF conventions; without regions nothing is ignored."""
    from dynamask_amd.assigners import MaxIoUAssigner
    g = np.load(os.path.join(golden_dir, 'g13_assign_ignore.npz'))
    dev = torch.device('cuda')
    b, gts, ign, lab = (torch.from_numpy(g[k]).to(dev) for k in ('bboxes', 'gts', 'ign', 'labels'))
    for name, wrt in (('cand', True), ('region', False)):
        asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5, ignore_wrt_candidates=wrt)
        ar = asg.assign(b, gts, ign, lab)
        assert np.array_equal(ar.gt_inds.cpu().numpy(), g[f'{name}_gt_inds'])
        np.testing.assert_array_equal(ar.max_overlaps.cpu().numpy(), g[f'{name}_max_overlaps'])
        assert np.array_equal(ar.labels.cpu().numpy(), g[f'{name}_labels'])
    asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5)
    assert np.array_equal(asg.assign(b, gts, None, lab).gt_inds.cpu().numpy(), g['noign_gt_inds'])
    assert np.array_equal(asg.assign(b, gts, ign[:0], lab).gt_inds.cpu().numpy(), g['noign_gt_inds'])


def test_ce_loss_and_accuracy_known_answers_of_the_reference_tests():
    """/root/reference/tests/test_models/test_losses.py:7-50 (inputs and expected values as data): CrossEntropyLoss
    with and without class weights, use_mask + use_sigmoid refused, top-1 accuracy incl. the empty prediction."""
    from dynamask_amd import registry, losses
    dev = torch.device('cuda')
    with pytest.raises(AssertionError):
        registry.build_loss(dict(type='CrossEntropyLoss', use_mask=True, use_sigmoid=True, loss_weight=1.0))
    fake_pred = torch.tensor([[100., -100.]], device=dev)
    fake_label = torch.tensor([1], device=dev)
    loss_cls = registry.build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, class_weight=[0.8, 0.2], loss_weight=1.0))
    assert torch.allclose(loss_cls(fake_pred, fake_label).cpu(), torch.tensor(40.))
    loss_cls = registry.build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    assert torch.allclose(loss_cls(fake_pred, fake_label).cpu(), torch.tensor(200.))
    assert losses.accuracy(torch.empty(0, 4, device=dev), torch.empty(0, device=dev)).item() == 0
    pred = torch.tensor([[0.2, 0.3, 0.6, 0.5], [0.1, 0.1, 0.2, 0.6], [0.9, 0.0, 0.0, 0.1], [0.4, 0.7, 0.1, 0.1],
                         [0.0, 0.0, 0.99, 0]], device=dev)
    assert losses.accuracy(pred, torch.tensor([2, 3, 0, 1, 2], device=dev)).item() == 100
    assert losses.accuracy(pred, torch.tensor([2, 3, 0, 0, 1], device=dev)).item() == 60
