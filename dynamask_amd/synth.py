"""Synthetic workload of the mask-head path.

Host-side generators only (CPU ``torch.Generator`` with fixed seeds, then the
caller copies to the device): FPN feature pyramids, RoIs whose sqrt(area) is
log-uniform so that all four FPN levels are hit, class labels, blob mask
targets, explicit Gumbel noise, and reference-style weight initialisers
producing ``state_dict``s with the reference's key names.
"""
import math

import torch

FPN_STRIDES = (4, 8, 16, 32, 64)


def _gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def fpn_shapes(img_h, img_w, strides=FPN_STRIDES):
    """Spatial sizes of P2..P6 for an image padded to a multiple of 32
    (Pad size_divisor=32, configs/dynamask/coco/r50-dynamask-1x.py:169;
    necks/fpn.py:190-199: P6 = max_pool(P5, 1, stride 2))."""
    ph = int(math.ceil(img_h / 32.0) * 32)
    pw = int(math.ceil(img_w / 32.0) * 32)
    out = []
    for s in strides[:4]:
        out.append((ph // s, pw // s))
    h5, w5 = out[-1]
    out.append(((h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1))
    return out


def make_fpn(batch, img_h=800, img_w=1333, channels=256, seed=0):
    g = _gen(seed)
    return [torch.randn(batch, channels, h, w, generator=g) * 0.5 for (h, w) in fpn_shapes(img_h, img_w)]


def make_rois(batch, n_per_img, img_h=800, img_w=1333, seed=1, min_size=16.0, max_size=800.0):
    """[N,5] = (batch_idx, x1, y1, x2, y2), grouped by image (bbox2roi order)."""
    g = _gen(seed)
    rois = []
    for b in range(batch):
        n = n_per_img
        cx = torch.rand(n, generator=g) * img_w
        cy = torch.rand(n, generator=g) * img_h
        s = torch.exp(torch.rand(n, generator=g) * (math.log(max_size) - math.log(min_size)) + math.log(min_size))
        ar = torch.exp((torch.rand(n, generator=g) * 2 - 1) * math.log(3.0))
        w = s * torch.sqrt(ar)
        h = s / torch.sqrt(ar)
        x1 = (cx - w / 2).clamp(0, img_w - 1)
        x2 = (cx + w / 2).clamp(0, img_w - 1)
        y1 = (cy - h / 2).clamp(0, img_h - 1)
        y2 = (cy + h / 2).clamp(0, img_h - 1)
        x2 = torch.maximum(x2, x1 + 1.0)
        y2 = torch.maximum(y2, y1 + 1.0)
        rois.append(torch.stack([torch.full((n,), float(b)), x1, y1, x2, y2], dim=1))
    return torch.cat(rois, 0)


def make_labels(n, num_classes=80, seed=2):
    return torch.randint(0, num_classes, (n,), generator=_gen(seed))


def make_gumbel_noise(n, seed=4):
    return torch.rand(n, 4, generator=_gen(seed))


def make_targets(n, sizes=(14, 28, 56, 112), seed=3):
    """Blob targets: union of 3 random ellipses rasterised at 112^2, then
    area-resampled (avg-pool, >=0.5) to the coarser stages."""
    g = _gen(seed)
    S = sizes[-1]
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32) + 0.5,
                            torch.arange(S, dtype=torch.float32) + 0.5, indexing='ij')
    full = torch.zeros(n, S, S)
    for _ in range(3):
        cx = (torch.rand(n, generator=g) * 0.6 + 0.2) * S
        cy = (torch.rand(n, generator=g) * 0.6 + 0.2) * S
        ax = (torch.rand(n, generator=g) * 0.3 + 0.08) * S
        ay = (torch.rand(n, generator=g) * 0.3 + 0.08) * S
        d = ((xx[None] - cx[:, None, None]) / ax[:, None, None]) ** 2 + \
            ((yy[None] - cy[:, None, None]) / ay[:, None, None]) ** 2
        full = torch.maximum(full, (d <= 1.0).float())
    out = []
    for s in sizes:
        if s == S:
            out.append(full.clone())
        else:
            k = S // s
            out.append((torch.nn.functional.avg_pool2d(full[:, None], k)[:, 0] >= 0.5).float())
    return out


def make_train_batch(batch=2, img_h=800, img_w=1333, n_props=1000, n_gt=(15, 7), seed=30):
    """What ``forward_train`` receives per image at the reference's training shape (configs/dynamask/coco/
    r50-dynamask-1x.py:109-134: 1000 RPN proposals with a score column, sampler 512 x 0.25): ``n_gt[i]`` ground-truth
    boxes with labels and [G, H, W] uint8 bitmaps (an ellipse inside each box), 60 % of the proposals jittered copies of
    the GT boxes (so that the sampler finds its 128 positives), the rest scattered."""
    g = _gen(seed)
    out = dict(proposals=[], gt_bboxes=[], gt_labels=[], gt_masks=[],
               img_metas=[dict(img_shape=(img_h, img_w, 3), pad_shape=(img_h, img_w, 3)) for _ in range(batch)])
    yy, xx = torch.meshgrid(torch.arange(img_h, dtype=torch.float32) + 0.5, torch.arange(img_w, dtype=torch.float32) + 0.5,
                            indexing='ij')
    for b in range(batch):
        G = n_gt[b % len(n_gt)]
        c = torch.rand(G, 2, generator=g) * torch.tensor([img_w - 300.0, img_h - 300.0]) + 150
        wh = torch.rand(G, 2, generator=g) * 300 + 40
        gtb = torch.cat([c - wh / 2, c + wh / 2], 1)
        gtb[:, 0::2] = gtb[:, 0::2].clamp(0, img_w - 1)
        gtb[:, 1::2] = gtb[:, 1::2].clamp(0, img_h - 1)
        cx, cy = (gtb[:, 0] + gtb[:, 2]) / 2, (gtb[:, 1] + gtb[:, 3]) / 2
        ax, ay = (gtb[:, 2] - gtb[:, 0]) / 2, (gtb[:, 3] - gtb[:, 1]) / 2
        m = ((((xx[None] - cx[:, None, None]) / ax[:, None, None]) ** 2
              + ((yy[None] - cy[:, None, None]) / ay[:, None, None]) ** 2) <= 1.0).to(torch.uint8)
        n_near = int(0.6 * n_props)
        near = gtb[torch.randint(0, G, (n_near,), generator=g)] + torch.randn(n_near, 4, generator=g) * 12
        far = make_rois(1, n_props - n_near, img_h, img_w, seed=seed + 100 + b)[:, 1:]
        props = torch.cat([near, far])
        props[:, 0::2] = props[:, 0::2].clamp(0, img_w - 1)
        props[:, 1::2] = props[:, 1::2].clamp(0, img_h - 1)
        props[:, 2] = torch.maximum(props[:, 2], props[:, 0] + 1)
        props[:, 3] = torch.maximum(props[:, 3], props[:, 1] + 1)
        out['proposals'].append(torch.cat([props, torch.rand(n_props, 1, generator=g)], 1))
        out['gt_bboxes'].append(gtb)
        out['gt_labels'].append(torch.randint(0, 80, (G,), generator=g))
        out['gt_masks'].append(m)
    return out


# ------------------------------------------------------------ weight initialisers
def _kaiming_fan_out(shape, g):
    fan_out = shape[0] * shape[2] * shape[3]
    return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)


def init_dynamask_head_state(seed=5, prefix='mask_head.', in_channels=256, num_classes=(80, 80, 80, 1),
                             num_convs_instance=2, sem_channels=256, test_mode=False):
    """state_dict of DynaMaskHead at the reference initialisers
    (dynamask_head.py:92-100,215-218; DCN U(+-1/sqrt(9C)), zero offset conv:
    mmdet/ops/dcn/deform_conv.py:230-235,273-275).

    test_mode=True additionally draws small non-zero biases and a non-zero
    offset conv (offsets ~N(0, 0.5px)) so every term of the path is exercised.
    """
    g = _gen(seed)
    sd = {}

    def bias(n):
        return torch.randn(n, generator=g) * 0.1 if test_mode else torch.zeros(n)

    C = in_channels
    for i in range(num_convs_instance):
        sd[f'{prefix}instance_convs.{i}.conv.weight'] = _kaiming_fan_out((C, C, 3, 3), g)
        sd[f'{prefix}instance_convs.{i}.conv.bias'] = bias(C)
    cin = C
    for s in range(3):
        p = f'{prefix}stages.{s}.'
        cout = cin // 2
        sd[p + 'semantic_transform_in.weight'] = _kaiming_fan_out((cin, sem_channels, 1, 1), g)
        sd[p + 'semantic_transform_in.bias'] = bias(cin)
        for nm in ('instance_logits', 'detail_logits'):
            sd[p + nm + '.weight'] = _kaiming_fan_out((num_classes[s], cin, 1, 1), g)
            sd[p + nm + '.bias'] = bias(num_classes[s])
        sd[p + 'fuse_conv.0.weight'] = _kaiming_fan_out((cin, 2 * cin + 2, 1, 1), g)
        sd[p + 'fuse_conv.0.bias'] = bias(cin)
        stdv = 1.0 / math.sqrt(cin * 9)
        sd[p + 'fuse_conv.1.weight'] = (torch.rand((cin, cin, 3, 3), generator=g) * 2 - 1) * stdv
        if test_mode:
            # offsets = conv_offset(x); x is O(1) post-ReLU over 9*cin taps
            sd[p + 'fuse_conv.1.conv_offset.weight'] = torch.randn((36, cin, 3, 3), generator=g) * (0.5 / math.sqrt(9 * cin))
            sd[p + 'fuse_conv.1.conv_offset.bias'] = torch.randn(36, generator=g) * 0.2
        else:
            sd[p + 'fuse_conv.1.conv_offset.weight'] = torch.zeros((36, cin, 3, 3))
            sd[p + 'fuse_conv.1.conv_offset.bias'] = torch.zeros(36)
        sd[p + 'fuse_transform_out.weight'] = _kaiming_fan_out((cout - 2, cin, 1, 1), g)
        sd[p + 'fuse_transform_out.bias'] = bias(cout - 2)
        cin = cout
    for nm in ('final_instance_logits', 'final_detail_logits'):
        sd[prefix + nm + '.weight'] = _kaiming_fan_out((num_classes[3], cin, 1, 1), g)
        sd[prefix + nm + '.bias'] = bias(num_classes[3])
    sd[prefix + 'loss_func.detail_target.fuse_kernel'] = torch.tensor([[7. / 10], [3. / 10]]).reshape(1, 2, 1, 1)
    return sd


def init_mask_pre_state(seed=6, prefix='mask_predictor.'):
    """state_dict of MaskPre at torch's default initialisers
    (base_roi_head.py:10-18)."""
    g = _gen(seed)
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    sd[prefix + 'conv1.weight'] = uni((128, 256, 1, 1), 256)
    sd[prefix + 'conv1.bias'] = uni((128,), 256)
    sd[prefix + 'conv2.weight'] = uni((16, 128, 3, 3), 128 * 9)
    sd[prefix + 'conv2.bias'] = uni((16,), 128 * 9)
    sd[prefix + 'fc1.weight'] = uni((512, 3136), 3136)
    sd[prefix + 'fc1.bias'] = uni((512,), 3136)
    sd[prefix + 'fc2.weight'] = uni((4, 512), 512)
    sd[prefix + 'fc2.bias'] = uni((4,), 512)
    for i, c in ((1, 128), (2, 16)):
        sd[f'{prefix}bn{i}.weight'] = torch.ones(c) + torch.randn(c, generator=g) * 0.05
        sd[f'{prefix}bn{i}.bias'] = torch.randn(c, generator=g) * 0.05
        sd[f'{prefix}bn{i}.running_mean'] = torch.zeros(c)
        sd[f'{prefix}bn{i}.running_var'] = torch.ones(c)
        sd[f'{prefix}bn{i}.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    return sd


def init_fcn_head_state(seed=7, prefix='mask_head.', in_channels=256, num_convs=4, num_classes=80,
                        upsample='deconv', test_mode=False):
    """state_dict of FCNMaskHead (fcn_mask_head.py:59-104; init :106-115)."""
    g = _gen(seed)
    sd = {}
    C = in_channels

    def bias(n):
        return torch.randn(n, generator=g) * 0.1 if test_mode else torch.zeros(n)

    for i in range(num_convs):
        sd[f'{prefix}convs.{i}.conv.weight'] = _kaiming_fan_out((C, C, 3, 3), g)
        sd[f'{prefix}convs.{i}.conv.bias'] = bias(C)
    if upsample == 'deconv':
        # ConvTranspose2d weight [Cin, Cout, 2, 2]; kaiming fan_out = Cin*k*k for this layout
        sd[prefix + 'upsample.weight'] = torch.randn((C, C, 2, 2), generator=g) * math.sqrt(2.0 / (C * 4))
        sd[prefix + 'upsample.bias'] = bias(C)
    elif upsample == 'carafe':
        a = math.sqrt(6.0 / (C + 64))
        sd[prefix + 'upsample.channel_compressor.weight'] = (torch.rand((64, C, 1, 1), generator=g) * 2 - 1) * a
        sd[prefix + 'upsample.channel_compressor.bias'] = bias(64)
        std = 0.05 if test_mode else 0.001
        sd[prefix + 'upsample.content_encoder.weight'] = torch.randn((100, 64, 3, 3), generator=g) * std
        sd[prefix + 'upsample.content_encoder.bias'] = bias(100)
    sd[prefix + 'conv_logits.weight'] = _kaiming_fan_out((num_classes, C, 1, 1), g)
    sd[prefix + 'conv_logits.bias'] = bias(num_classes)
    return sd


def init_bbox_head_state(seed=8, prefix='bbox_head.', in_channels=256, roi_feat=7, fc_out=1024, num_classes=80):
    """state_dict of Shared2FCBBoxHead (convfc_bbox_head.py:189-205).  Test-mode scales: the
    reference's init (xavier / N(0, 0.01) / N(0, 0.001)) gives near-uniform class scores, so the
    predictors are drawn larger to spread the softmax and the box deltas."""
    g = _gen(seed)
    K = in_channels * roi_feat * roi_feat
    sd = {}
    sd[prefix + 'shared_fcs.0.weight'] = torch.randn((fc_out, K), generator=g) * math.sqrt(2.0 / (K + fc_out))
    sd[prefix + 'shared_fcs.0.bias'] = torch.randn(fc_out, generator=g) * 0.05
    sd[prefix + 'shared_fcs.1.weight'] = torch.randn((fc_out, fc_out), generator=g) * math.sqrt(2.0 / (2 * fc_out))
    sd[prefix + 'shared_fcs.1.bias'] = torch.randn(fc_out, generator=g) * 0.05
    sd[prefix + 'fc_cls.weight'] = torch.randn((num_classes + 1, fc_out), generator=g) * 0.25
    sd[prefix + 'fc_cls.bias'] = torch.randn(num_classes + 1, generator=g) * 0.1
    sd[prefix + 'fc_reg.weight'] = torch.randn((4 * num_classes, fc_out), generator=g) * 0.05
    sd[prefix + 'fc_reg.bias'] = torch.randn(4 * num_classes, generator=g) * 0.05
    return sd


# ------------------------------------------------------------------ the reference's config values
# configs/dynamask/coco/r50-dynamask-1x.py:60-91
MASK_ROI_EXTRACTOR_CFG = dict(
    roi_layer=dict(type='RoIAlign', output_size=14, sampling_ratio=0),
    out_channels=256, featmap_strides=[4, 8, 16, 32])
LOSS_CFG = dict(
    stage_instance_loss_weight=[0.5, 0.75, 0.75, 1.0],
    stage_detail_loss_weight=[0.5, 0.5, 0.5, 0.5],
    detail_loss_weight=1.0, cb_loss_weight=0.8, boundary_width=2, start_stage=4)
MASK_HEAD_CFG = dict(
    num_convs_instance=2, num_convs_semantic=4,
    conv_in_channels_instance=256, conv_in_channels_semantic=256,
    conv_kernel_size_instance=3, conv_kernel_size_semantic=3,
    conv_out_channels_instance=256, conv_out_channels_semantic=256,
    conv_cfg=None, norm_cfg=None, semantic_out_stride=[16, 8, 4],
    mask_use_sigmoid=True, pre_upsample_last_stage=False,
    stage_num_classes=[80, 80, 80, 1], stage_sup_size=[14, 28, 56, 112],
    upsample_cfg=dict(type='bilinear', scale_factor=2),
    loss_cfg=dict(type='DynaCrossEntropyLoss', **LOSS_CFG))
# configs/_base_/models/mask_rcnn_r50_fpn.py:57-68
FCN_HEAD_CFG = dict(num_convs=4, in_channels=256, conv_out_channels=256, num_classes=80,
                    loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))
# configs/dynamask/coco/r50-dynamask-1x.py:41-59,141-150
BBOX_HEAD_CFG = dict(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                     target_stds=[0.1, 0.1, 0.2, 0.2]),
                     reg_class_agnostic=False,
                     loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0),
                     loss_bbox=dict(type='L1Loss', loss_weight=2.0))
BBOX_ROI_EXTRACTOR_CFG = dict(roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0), out_channels=256,
                              featmap_strides=[4, 8, 16, 32])
# configs/dynamask/coco/r50-dynamask-1x.py:118-134
RCNN_TRAIN_CFG = dict(
    assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=True,
                  ignore_iof_thr=-1),
    sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
    flops=[0.23, 0.62, 1.01, 1.4], Lambda=0.3, mask_size=28, pos_weight=-1, debug=False)
RCNN_TEST_CFG = dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100, mask_thr_binary=0.5)
