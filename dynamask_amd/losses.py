"""Losses behind the reference's LOSSES registry.

``DynaCrossEntropyLoss`` / ``DetailTarget`` mirror
``mmdet/models/losses/cross_entropy_loss.py:363-487`` (constructor kwargs,
``forward(stage_instance_preds, stage_detail_preds, stage_instance_targets,
mask_labels) -> {'loss_masks': scalar}``, the ``detail_target.fuse_kernel``
parameter key).  Reference quirks kept by default: Q2 only the
last stage's instance BCE reaches the loss; the eps-BCE normaliser
``sum(mask_labels[:, idx])`` is a detached host scalar in the reference -- here
it stays on the device (no ``.item()`` sync per stage).
"""
import torch
import torch.nn as nn

from . import ops
from .registry import LOSSES


class DetailTarget(nn.Module):
    def __init__(self):
        super().__init__()
        self.fuse_kernel = nn.Parameter(torch.tensor([[7. / 10], [3. / 10]], dtype=torch.float32).reshape(1, 2, 1, 1))

    def forward(self, gtmasks):
        return ops.detail_target(gtmasks.contiguous(), self.fuse_kernel.detach().reshape(-1).contiguous()).unsqueeze(1)


class _DynaLossFn(torch.autograd.Function):
    """One fused fwd+bwd kernel per stage; autograd only carries the
    pre-computed gradients."""

    @staticmethod
    def forward(ctx, mask_labels, detail_w, cb_w, start_stage, fuse, n_stage, *tensors):
        ips = tensors[:n_stage]
        dps = tensors[n_stage:2 * n_stage]
        tgts = tensors[2 * n_stage:3 * n_stage]
        ml = mask_labels.detach().contiguous()
        used = [idx for idx in range(n_stage) if idx <= start_stage]
        # per stage: the detail target and ONE fused kernel (+ its finishing workgroup) that applies the stage's
        # normalisers where the values are produced (dm_mask_loss_stage); round 2 finished every stage with ~30 small
        # tensor operations -- 130 launches between the forward and the backward, with the GPU idle behind them
        terms = torch.zeros((2,), device=ml.device, dtype=torch.float32)      # [mean BCE of the last stage, detail terms]
        grad_ml = torch.zeros_like(ml)
        g_dps = [None] * n_stage
        gi_last, last_idx = None, -1
        for idx in used:
            ip = ips[idx].detach().contiguous()
            dp = dps[idx].detach().contiguous()
            tg = tgts[idx].detach().contiguous()
            dt = ops.detail_target(tg, fuse)
            gi, gd = ops.mask_loss_stage(ip, dp, tg, dt, ml, idx, detail_w[idx], terms, grad_ml,
                                         want_inst_grad=(idx == used[-1]))
            g_dps[idx] = gd
            if gi is not None:
                gi_last, last_idx = gi, idx
        cb, g_cb = ops.class_balance(ml)
        grad_ml.add_(g_cb, alpha=cb_w)
        loss = terms.sum() + cb_w * cb
        ctx.n_stage = n_stage
        ctx.last_idx = last_idx
        ctx.save_for_backward(grad_ml, gi_last, *[g for g in g_dps if g is not None])
        ctx.dp_mask = [g is not None for g in g_dps]
        ctx.shapes = [t.shape for t in ips]
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        grad_ml, gi_last = saved[0], saved[1]
        gd = list(saved[2:])
        n = ctx.n_stage
        g_ips = [None] * n
        g_ips[ctx.last_idx] = (gi_last * g).view(ctx.shapes[ctx.last_idx])
        g_dps = []
        k = 0
        for i in range(n):
            if ctx.dp_mask[i]:
                g_dps.append((gd[k] * g).view(ctx.shapes[i]))
                k += 1
            else:
                g_dps.append(None)
        return (grad_ml * g, None, None, None, None, None, *g_ips, *g_dps, *([None] * n))


@LOSSES.register_module()
class DynaCrossEntropyLoss(nn.Module):
    def __init__(self, stage_instance_loss_weight=[1.0, 1.0, 1.0, 1.0], stage_detail_loss_weight=[1.0, 1.0, 1.0, 1.0],
                 detail_loss_weight=1.0, cb_loss_weight=1.0, boundary_width=2, start_stage=1):
        super().__init__()
        self.stage_instance_loss_weight = stage_instance_loss_weight
        self.stage_detail_loss_weight = stage_detail_loss_weight
        self.detail_loss_weight = detail_loss_weight
        self.cb_loss_weight = cb_loss_weight
        self.boundary_width = boundary_width
        self.start_stage = start_stage
        self.detail_target = DetailTarget()

    def forward(self, stage_instance_preds, stage_detail_preds, stage_instance_targets, mask_labels):
        n = len(stage_instance_preds)
        n_used = sum(1 for idx in range(n) if idx <= self.start_stage)
        assert len(self.stage_instance_loss_weight) == n_used      # cross_entropy_loss.py:482
        fuse = self.detail_target.fuse_kernel.detach().reshape(-1).contiguous()      # stays on the device (no .tolist() sync)
        loss = _DynaLossFn.apply(mask_labels, list(self.stage_detail_loss_weight), float(self.cb_loss_weight),
                                 int(self.start_stage), fuse, n, *stage_instance_preds, *stage_detail_preds,
                                 *stage_instance_targets)
        return {'loss_masks': loss}


class _SoftmaxCEFn(torch.autograd.Function):
    """cross_entropy (cross_entropy_loss.py:9-38): fused forward + gradient (dm_softmax_ce_fwd_bwd)."""

    @staticmethod
    def forward(ctx, cls_score, label, weight, scale):
        loss, _, grad = ops.softmax_ce(cls_score.detach().contiguous(), label.long().contiguous(),
                                       None if weight is None else weight.float().contiguous(), scale, need_grad=True)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class _L1PosFn(torch.autograd.Function):
    """L1 loss over the positive rows' class columns of ``bbox_pred`` (bbox_head.py:159-182 +
    smooth_l1_loss.py:29-42): gather, |.|, weights, reduction and scatter of the gradient in one
    kernel (dm_l1_loss_fwd_bwd)."""

    @staticmethod
    def forward(ctx, bbox_pred, labels, targets, weights, num_classes, scale):
        loss, grad = ops.l1_loss_pos(bbox_pred.detach().contiguous(), labels.long().contiguous(), targets.contiguous(),
                                     weights.contiguous(), num_classes, scale, need_grad=True)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


def accuracy(pred, target, topk=1, thresh=None):
    """losses/accuracy.py:4-49 for top-1 without threshold -> percent, device scalar."""
    if topk != 1 or thresh is not None:
        raise NotImplementedError('BBoxHead.loss asks for plain top-1 accuracy')
    if pred.size(0) == 0:
        return pred.new_tensor(0.)
    _, acc, _ = ops.softmax_ce(pred.detach().contiguous(), target.long().contiguous(), None, 1.0, need_grad=False)
    return acc.reshape(1)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """losses/cross_entropy_loss.py:157-216.  The softmax form (``use_sigmoid=False,
    use_mask=False``: the bbox head's ``loss_cls``) is built; the fork's ``use_mask=True``
    path raises in the reference itself (Quirk Q5) and ``use_sigmoid=True`` belongs to the RPN."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.class_weight, self.loss_weight = reduction, class_weight, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if self.use_mask:
            raise NotImplementedError('CrossEntropyLoss(use_mask=True) is broken in the reference fork (SURVEY Q5); '
                                      'only FCNMaskHead.forward is on the path')
        if self.use_sigmoid:
            raise NotImplementedError('use_sigmoid=True is the RPN\'s form; the RoI head uses the softmax form')
        if self.class_weight is not None:
            # F.cross_entropy(..., weight=class_weight, reduction='none') scales sample i by class_weight[label_i]
            # (cross_entropy_loss.py:9-38): a gather, handed to the kernel as the per-sample weight
            if weight is not None:
                raise NotImplementedError('class_weight together with per-sample weights is not used by the RoI head')
            weight = cls_score.new_tensor(self.class_weight)[label.long()].contiguous()
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        n = cls_score.shape[0]
        if reduction == 'mean':
            scale = self.loss_weight / (float(avg_factor) if avg_factor is not None else float(n))
        elif reduction == 'sum':
            if avg_factor is not None:
                raise ValueError('avg_factor can not be used with reduction="sum"')
            scale = self.loss_weight
        else:
            raise NotImplementedError("reduction='none' is not used by the RoI head")
        return _SoftmaxCEFn.apply(cls_score, label, weight, scale)


@LOSSES.register_module()
class L1Loss(nn.Module):
    """losses/smooth_l1_loss.py:94-136.  ``forward`` is the generic elementwise form on already
    gathered rows; ``BBoxHead.loss`` uses ``forward_pos`` (gather + loss + scatter fused)."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def _scale(self, numel, avg_factor, reduction_override):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if reduction == 'mean':
            return self.loss_weight / (float(avg_factor) if avg_factor is not None else float(numel))
        if reduction == 'sum' and avg_factor is None:
            return self.loss_weight
        raise NotImplementedError("reduction 'none' / 'sum' with avg_factor is not used by the RoI head")

    def forward_pos(self, bbox_pred, labels, bbox_targets, bbox_weights, num_classes, avg_factor=None,
                    reduction_override=None):
        if avg_factor is None and (reduction_override or self.reduction) == 'mean':
            # the reference takes the mean over the rows it has GATHERED (positives only, bbox_head.py:166-180): that
            # count lives on the device; the RoI head always passes avg_factor, so this form is not supported here
            raise NotImplementedError("L1Loss.forward_pos: 'mean' without avg_factor would need the number of positive "
                                      "rows on the host; pass avg_factor (bbox_head.py:175 does)")
        scale = self._scale(bbox_pred.shape[0] * 4, avg_factor, reduction_override)
        return _L1PosFn.apply(bbox_pred, labels, bbox_targets, bbox_weights, num_classes, scale)

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """Rows already gathered ([n, 4]): every row counts as positive of a 1-class head."""
        assert pred.size() == target.size() and target.numel() > 0
        n = pred.shape[0]
        lab = torch.zeros((n,), device=pred.device, dtype=torch.long)
        w = weight if weight is not None else torch.ones_like(pred)
        scale = self._scale(pred.numel(), avg_factor, reduction_override)
        return _L1PosFn.apply(pred.reshape(n, -1), lab, target.reshape(n, -1), w.reshape(n, -1), 1, scale)
