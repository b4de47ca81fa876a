"""Bbox branch of the RoI head at inference: the reference's
``Shared2FCBBoxHead`` (roi_heads/bbox_heads/convfc_bbox_head.py:9-205, bbox_head.py:13-217),
``DeltaXYWHBBoxCoder`` (core/bbox/coder/delta_xywh_bbox_coder.py) and ``multiclass_nms``
(core/post_processing/bbox_nms.py) under their registry names, kwargs and ``state_dict`` keys.

The 7x7 RoI extraction is the same HIP kernel as the mask branch's; softmax + box decoding and
the NMS suppression matrix are HIP kernels (``dm_bbox_decode``, ``dm_nms_mask``); the four fully
connected layers run on ``dm_fc_fwd`` (fp32 MFMA GEMM with split-K over the long 12544 axis).
Training (second half): ``get_targets`` (bbox2delta = ``dm_bbox_encode``),
``loss`` (softmax CE + accuracy = ``dm_softmax_ce_fwd_bwd``, L1 on the positive rows' class
columns = ``dm_l1_loss_fwd_bwd``) and the backward of the FC stack (``BBoxHeadFn``: data and
weight gradients as fp32-MFMA GEMMs through the implicit-GEMM kernels on [N, C, 1, 1] tensors)."""
import numpy as np
import torch
import torch.nn as nn

from . import hazard, ops
from .registry import HEADS, Registry, build_from_cfg

BBOX_CODERS = Registry('bbox_coder')


def build_bbox_coder(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    """core/bbox/coder/delta_xywh_bbox_coder.py:9-60."""

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
        """delta_xywh_bbox_coder.py:33-50 -> bbox2delta."""
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return ops.bbox_encode(bboxes.float().contiguous(), gt_bboxes.float().contiguous(), self.means, self.stds)


class _FC(nn.Module):
    """nn.Linear parameters; the product runs on the fp32 MFMA FC kernel (dm_fc_fwd)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.xavier_uniform_(self.weight)
        from .mask_heads import _Packed
        self._pk = _Packed()      # transposed packed weights of the backward's data-gradient GEMM

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
        self.reg_decoded_bbox = reg_decoded_bbox
        if reg_decoded_bbox:
            raise NotImplementedError('reg_decoded_bbox=False in configs/dynamask')
        self.loss_cls_cfg, self.loss_bbox_cfg = loss_cls, loss_bbox
        # bbox_head.py:47-48 (no parameters; built when configured so that inference-only configs stay light)
        from .registry import build_loss
        self.loss_cls = build_loss(loss_cls) if loss_cls is not None else None
        self.loss_bbox = build_loss(loss_bbox) if loss_bbox is not None else None
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
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return BBoxHeadFn.apply(self, x, *list(self.parameters()))
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = fc.run(x, relu=True)
        return self.fc_cls.run(x), self.fc_reg.run(x)

    # ------------------------------------------------------------------ training
    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        """bbox_head.py:85-116."""
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        num_samples = num_pos + num_neg
        labels = pos_bboxes.new_full((num_samples,), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(num_samples)
        bbox_targets = pos_bboxes.new_zeros(num_samples, 4)
        bbox_weights = pos_bboxes.new_zeros(num_samples, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            pos_weight = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
            label_weights[:num_pos] = pos_weight
            bbox_targets[:num_pos, :] = self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes)
            bbox_weights[:num_pos, :] = 1
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        """bbox_head.py:118-141."""
        outs = [self._get_target_single(r.pos_bboxes, r.neg_bboxes, r.pos_gt_bboxes, r.pos_gt_labels, rcnn_train_cfg)
                for r in sampling_results]
        labels, label_weights, bbox_targets, bbox_weights = (list(t) for t in zip(*outs))
        if concat:
            labels = torch.cat(labels, 0)
            label_weights = torch.cat(label_weights, 0)
            bbox_targets = torch.cat(bbox_targets, 0)
            bbox_weights = torch.cat(bbox_weights, 0)
            # every sampled row carries a weight > 0 (_get_target_single: pos_weight, made 1 when <= 0, for positives; 1 for
            # negatives): the count the classification loss normalises by is the row count, known on the host -- ``loss``
            # below takes it from here instead of reading it back from the device (a host sync per step)
            label_weights._dm_num_weighted = int(label_weights.shape[0])
        return labels, label_weights, bbox_targets, bbox_weights

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights, reduction_override=None):
        """bbox_head.py:143-184 -> {'loss_cls', 'acc', 'loss_bbox'}."""
        from .losses import accuracy
        losses = dict()
        if cls_score is not None:
            known = getattr(label_weights, '_dm_num_weighted', None)      # set by get_targets: all of its rows are weighted
            avg_factor = max(float(known) if known is not None else torch.sum(label_weights > 0).float().item(), 1.)
            if cls_score.numel() > 0:
                losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg_factor,
                                                   reduction_override=reduction_override)
                losses['acc'] = accuracy(cls_score, labels)
        if bbox_pred is not None and bbox_pred.shape[0] == 0:
            losses['loss_bbox'] = bbox_pred.sum() * 0            # no sampled RoI at all (bbox_head.py:181-182)
        elif bbox_pred is not None:
            # the reference gathers bbox_pred[pos, labels[pos]] and calls loss_bbox on the gathered rows
            # (avg_factor = number of samples); rows with a background label contribute nothing, and
            # with no positive row the loss is `bbox_pred.sum() * 0`: both are what the fused kernel gives
            losses['loss_bbox'] = self.loss_bbox.forward_pos(bbox_pred, labels, bbox_targets, bbox_weights,
                                                             self.num_classes, avg_factor=bbox_targets.size(0),
                                                             reduction_override=reduction_override)
        return losses

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



class BBoxHeadFn(torch.autograd.Function):
    """Shared2FCBBoxHead.forward with a hand-sequenced backward.  Inputs: (head, bbox_feats
    [N, C, 7, 7], *head.parameters()); outputs (cls_score, bbox_pred).  nn.Linear's three
    products -- y = x W^T, dx = dy W, dW = dy^T x -- are the forward FC kernel and, for the
    gradients, the implicit-GEMM conv kernels on [N, C, 1, 1] tensors (one "pixel" per sample)."""

    @staticmethod
    def forward(ctx, head, x, *params):
        n = x.shape[0]
        flat = x.detach().reshape(n, -1).contiguous()
        acts = [flat]
        for fc in head.shared_fcs:
            acts.append(fc.run(acts[-1], relu=True))
        cls_score, bbox_pred = head.fc_cls.run(acts[-1]), head.fc_reg.run(acts[-1])
        ctx.head, ctx.acts, ctx.x_shape, ctx.need_x = head, acts, tuple(x.shape), x.requires_grad
        return cls_score, bbox_pred

    @staticmethod
    @hazard.backward_node
    def backward(ctx, g_cls, g_reg):
        head, acts = ctx.head, ctx.acts
        from . import hazard
        hazard.engine_handoff(g_cls, g_reg)
        n = acts[0].shape[0]
        pg = {}

        def fc_bwd(fc, gy, xin, need_data=True, out=None, accumulate=False):
            from .train_path import params_grad
            gy4 = gy.contiguous().view(n, fc.out_features, 1, 1)
            x4 = xin.view(n, fc.in_features, 1, 1)
            params_grad(fc.weight, fc.bias, gy4, x4, 1, pg, (fc.out_features, fc.in_features, 1, 1))
            if not need_data:
                return None
            wq = fc._pk.get('flip', fc.weight, lambda t: ops.pack_conv_weight(
                t.view(fc.out_features, fc.in_features, 1, 1), transpose_flip=True))
            return ops.conv2d(gy4, wq, None, fc.in_features, 1, out=out, accumulate=accumulate)

        h = acts[-1]
        g_h = None
        for fc, g in ((head.fc_cls, g_cls), (head.fc_reg, g_reg)):
            if g is None:
                continue
            if g_h is None:
                g_h = fc_bwd(fc, g, h)
            else:
                fc_bwd(fc, g, h, out=g_h, accumulate=True)
        g_x = None
        if g_h is not None:
            for i in reversed(range(len(head.shared_fcs))):
                ops.relu_backward_(g_h, acts[i + 1].view_as(g_h))
                need = i > 0 or ctx.need_x
                g_h = fc_bwd(head.shared_fcs[i], g_h, acts[i], need_data=need)
            if ctx.need_x:
                g_x = g_h.reshape(ctx.x_shape)
        from .train_path import _join_caller_after_backward
        _join_caller_after_backward(acts[0].device)
        return (None, g_x, *[pg.get(p) for p in head.parameters()])


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
