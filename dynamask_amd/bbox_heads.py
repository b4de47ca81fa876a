"""Bbox branch of the RoI head at inference: the reference's
``Shared2FCBBoxHead`` (roi_heads/bbox_heads/convfc_bbox_head.py:9-205, bbox_head.py:13-217),
``DeltaXYWHBBoxCoder`` (core/bbox/coder/delta_xywh_bbox_coder.py) and ``multiclass_nms``
(core/post_processing/bbox_nms.py) under their registry names, kwargs and ``state_dict`` keys.

The 7x7 RoI extraction is the same HIP kernel as the mask branch's; softmax + box decoding and
the NMS suppression matrix are HIP kernels (``dm_bbox_decode``, ``dm_nms_mask``); the four fully
connected layers run on ``dm_fc_fwd`` (fp32 MFMA GEMM with split-K over the long 12544 axis).
Training losses of the bbox branch are not built."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .registry import HEADS, Registry, build_from_cfg

BBOX_CODERS = Registry('bbox_coder')


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    """core/bbox/coder/delta_xywh_bbox_coder.py:9-60 (decode only; encode belongs to training)."""

    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.)):
        self.means = tuple(target_means)
        self.stds = tuple(target_stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        nb = pred_bboxes.size(1) // 4
        out, _ = ops.bbox_decode(bboxes.contiguous(), None, pred_bboxes.contiguous(), nb, self.means, self.stds,
                                 wh_ratio_clip, max_shape)
        return out

    def encode(self, bboxes, gt_bboxes):
        raise NotImplementedError('bbox targets belong to the training of the bbox branch (not built)')


class _FC(nn.Module):
    """nn.Linear parameters; the product runs on the fp32 MFMA FC kernel (dm_fc_fwd)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.xavier_uniform_(self.weight)

    def run(self, x, relu=False):
        return ops.fc(x.contiguous(), self.weight.detach(), self.bias.detach(), relu=relu)


@HEADS.register_module()
class Shared2FCBBoxHead(nn.Module):
    """convfc_bbox_head.py:189-205 (= ConvFCBBoxHead with 2 shared FCs) on top of BBoxHead
    (bbox_head.py:13-60).  ``state_dict`` keys: shared_fcs.{0,1}.{weight,bias}, fc_cls.*, fc_reg.*."""

    def __init__(self, fc_out_channels=1024, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7,
                 in_channels=256, num_classes=80, bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                                                  target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False, loss_cls=None, loss_bbox=None,
                 conv_out_channels=256, conv_cfg=None, norm_cfg=None):
        super().__init__()
        if with_avg_pool or not with_cls or not with_reg or conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('configs/dynamask use the plain Shared2FCBBoxHead')
        self.roi_feat_size = (roi_feat_size, roi_feat_size) if isinstance(roi_feat_size, int) else tuple(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.reg_class_agnostic = reg_class_agnostic
        self.fc_out_channels = fc_out_channels
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls_cfg, self.loss_bbox_cfg = loss_cls, loss_bbox
        self.shared_fcs = nn.ModuleList([_FC(in_channels * self.roi_feat_area, fc_out_channels),
                                         _FC(fc_out_channels, fc_out_channels)])
        self.fc_cls = _FC(fc_out_channels, num_classes + 1)
        self.fc_reg = _FC(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        """bbox_head.py:62-70 + convfc_bbox_head.py:128-136."""
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)
        for fc in self.shared_fcs:
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def forward(self, x):
        """convfc_bbox_head.py:138-186 for the Shared2FC configuration."""
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = fc.run(x, relu=True)
        return self.fc_cls.run(x), self.fc_reg.run(x)

    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """bbox_head.py:186-223."""
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scale = (1.0, 1.0)
        if rescale:
            if isinstance(scale_factor, float):
                scale = (scale_factor, scale_factor)
            else:
                sf = [float(v) for v in np.asarray(scale_factor).reshape(-1)]
                scale = (sf[0], sf[1])
                assert len(sf) == 4 and sf[2] == sf[0] and sf[3] == sf[1], 'scale_factor is [w, h, w, h]'
        coder = self.bbox_coder
        bboxes, scores = ops.bbox_decode(rois.contiguous(), None if cls_score is None else cls_score.contiguous(),
                                         None if bbox_pred is None else bbox_pred.contiguous(), self.num_classes,
                                         coder.means, coder.stds, 16 / 1000, img_shape, scale,
                                         class_agnostic=self.reg_class_agnostic or bbox_pred is None)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)

    def loss(self, *args, **kwargs):
        raise NotImplementedError('the training losses of the bbox branch are not built (SURVEY 8f rank 4: inference)')


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.nms.batched_nms (mmcv 1.0.5): boxes of different classes are moved apart by
    (max coordinate + 1) * class so that one NMS handles all classes."""
    cfg = dict(nms_cfg)
    cfg.pop('type', 'nms')
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)
        boxes_for_nms = boxes + offsets[:, None]
    dets, keep = ops.nms(boxes_for_nms.contiguous(), scores.contiguous(), cfg.get('iou_threshold', cfg.get('iou_thr', 0.5)))
    return torch.cat([boxes[keep], dets[:, -1:]], 1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """core/post_processing/bbox_nms.py:5-68 -> (dets [k, 5], labels [k]), labels 0-based; the
    last score column (background) is ignored."""
    n, ncls = multi_scores.shape[0], multi_scores.shape[1] - 1
    fg = multi_scores[:, :ncls]
    cand = (fg > score_thr).nonzero(as_tuple=False)          # row-major: the reference's masked_select order
    if cand.shape[0] == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    ri, ci = cand[:, 0], cand[:, 1]
    per_class = multi_bboxes.view(n, -1, 4)
    boxes = per_class[ri, ci] if per_class.shape[1] > 1 else per_class[ri, 0]
    scores = fg[ri, ci] if score_factors is None else (fg * score_factors[:, None])[ri, ci]
    dets, keep = batched_nms(boxes, scores, ci, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, ci[keep]


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:76-96."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    bboxes = bboxes.cpu().numpy()
    labels = labels.cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
