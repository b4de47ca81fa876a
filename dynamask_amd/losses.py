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
        fk = self.fuse_kernel.detach().reshape(-1).tolist()
        return ops.detail_target(gtmasks.contiguous(), fk).unsqueeze(1)


class _DynaLossFn(torch.autograd.Function):
    """One fused fwd+bwd kernel per stage; autograd only carries the
    pre-computed gradients."""

    @staticmethod
    def forward(ctx, mask_labels, detail_w, cb_w, start_stage, fuse, n_stage, *tensors):
        ips = tensors[:n_stage]
        dps = tensors[n_stage:2 * n_stage]
        tgts = tensors[2 * n_stage:3 * n_stage]
        N = mask_labels.shape[0]
        ml = mask_labels.detach().contiguous()
        loss_mask = None
        gi_last = None
        total_detail = torch.zeros((), device=ml.device, dtype=torch.float32)
        grad_ml = torch.zeros_like(ml)
        g_dps = []
        last_idx = -1
        for idx in range(n_stage):
            ip = ips[idx].detach().contiguous()
            dp = dps[idx].detach().contiguous()
            tg = tgts[idx].detach().contiguous()
            if idx > start_stage:
                g_dps.append(None)
                continue
            w = ml[:, idx].contiguous()
            dt = ops.detail_target(tg, fuse)
            sums, per_roi, gi, gd = ops.mask_loss(ip, dp, tg, dt, w, need_grad=True)
            n_el = float(ip.numel())
            den = w.sum() + 1e-5                      # device scalar, detached (reference: .item())
            scale = (N / n_el) / den
            loss_mask = sums[0] / n_el
            gi_last, last_idx = gi / n_el, idx
            total_detail = total_detail + detail_w[idx] * sums[1] * scale
            g_dps.append(gd * (detail_w[idx] * scale))
            grad_ml[:, idx] = per_roi * (detail_w[idx] * scale)
        cb, g_cb = ops.class_balance(ml)
        grad_ml += cb_w * g_cb
        ctx.n_stage = n_stage
        ctx.last_idx = last_idx
        ctx.save_for_backward(grad_ml, gi_last, *[g for g in g_dps if g is not None])
        ctx.dp_mask = [g is not None for g in g_dps]
        ctx.shapes = [t.shape for t in ips]
        return loss_mask + total_detail + cb_w * cb

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
        fuse = self.detail_target.fuse_kernel.detach().reshape(-1).tolist()
        loss = _DynaLossFn.apply(mask_labels, list(self.stage_detail_loss_weight), float(self.cb_loss_weight),
                                 int(self.start_stage), fuse, n, *stage_instance_preds, *stage_detail_preds,
                                 *stage_instance_targets)
        return {'loss_masks': loss}


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Accepted so that stock FCNMaskHead configs build; the fork's
    ``use_mask=True`` loss path raises in the reference itself (Quirk Q5)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        self.use_sigmoid, self.use_mask, self.loss_weight = use_sigmoid, use_mask, loss_weight

    def forward(self, *a, **k):
        raise NotImplementedError('CrossEntropyLoss(use_mask=True) is broken in the reference fork (SURVEY Q5); '
                                  'only FCNMaskHead.forward is on the path')
