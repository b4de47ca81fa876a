"""Mask heads behind the reference's HEADS registry.

``DynaMaskHead`` / ``SFMStage`` mirror
``mmdet/models/roi_heads/mask_heads/dynamask_head.py:54-244`` and
``FCNMaskHead`` mirrors ``mask_heads/fcn_mask_head.py:19-126``: same
constructor kwargs, same ``forward`` signatures, same ``state_dict`` keys
(SURVEY App. D), so reference checkpoints load.  The arithmetic is entirely in
libdynamask_hip.so; these classes only own parameters and sequence launches:

  * torch.cat never runs: concat sources are walked in the conv kernel's K loop,
    and producers write straight into channel slices of the consumer's input;
  * the 80-class logits conv + gather (dynamask_head.py:110-111) is a per-RoI
    gathered dot product (1/80 of the work);
  * bias, ReLU, sigmoid ride in the producing kernels' epilogues.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .registry import HEADS, UPSAMPLE_LAYERS, build_loss


class _Packed:
    """Cache of kernel-layout weights, refreshed when the parameter changes."""

    def __init__(self):
        self._c = {}

    def get(self, key, param, fn, job=None):
        """``job`` = (transpose_flip, src_channels | None, lo | None, hi | None): the pack is dm_conv_pack_weight of
        the parameter itself (or of its input-channel window lo:hi); such packs of contiguous device parameters are
        registered with ops.PACK_PLAN and refreshed together in one launch.  Anything else goes through ``fn``."""
        if job is not None and param.is_cuda and param.dim() == 4 and param.is_contiguous():
            e = self._c.get(key)
            if e is None or e[0] != 'plan' or e[1]['param']() is not param:
                e = ('plan', ops.PACK_PLAN.register(param, *job))
                self._c[key] = e
            with torch.no_grad():
                return ops.PACK_PLAN.get(e[1])
        ver = (param.data_ptr(), param._version, param.device, ops.WEIGHT_EPOCH[0])
        e = self._c.get(key)
        if e is None or e[0] != ver:
            with torch.no_grad():
                e = (ver, fn(param.detach().contiguous()))
            self._c[key] = e
        return e[1]


class _Conv(nn.Module):
    """Parameter holder named like nn.Conv2d (weight [Cout,Cin,k,k], bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')
        self._pk = _Packed()

    def packed(self, src_channels=None):
        key = tuple(src_channels) if src_channels is not None else (self.in_channels,)
        return self._pk.get(key, self.weight, lambda w: ops.pack_conv_weight(w, src_channels=list(key)),
                            job=(False, list(key), None, None))

    def run(self, srcs, relu=False, out=None, out_ch_offset=0):
        b = self.bias.detach() if self.bias is not None else None
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        wq = self.packed([s.shape[1] for s in srcs])
        return ops.conv2d(srcs, wq, b, self.out_channels, self.kernel_size, relu=relu, out=out,
                          out_ch_offset=out_ch_offset)


class ConvModule(nn.Module):
    """mmcv ConvModule without norm: conv(bias) + ReLU; keys ``conv.weight/bias``."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, dilation=1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU')):
        super().__init__()
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('conv_cfg / norm_cfg are None in configs/dynamask')
        if dilation != 1 or padding != kernel_size // 2:
            raise NotImplementedError('only "same" stride-1 convolutions are on the path')
        self.conv = _Conv(in_channels, out_channels, kernel_size)
        self.with_activation = act_cfg is not None

    def forward(self, x):
        return self.conv.run(x, relu=self.with_activation)


class DeformConv2dPack(nn.Module):
    """DCNv1 3x3 + its zero-initialised offset conv; keys ``weight``,
    ``conv_offset.weight/bias`` (mmdet/ops/dcn/deform_conv.py:189-275)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=1, deform_groups=1):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        if ks != 3:
            raise NotImplementedError('3x3 DCN only')
        self.in_channels, self.out_channels, self.deform_groups = in_channels, out_channels, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        stdv = 1.0 / math.sqrt(in_channels * 9)
        nn.init.uniform_(self.weight, -stdv, stdv)
        self.conv_offset = _Conv(in_channels, deform_groups * 18, 3)
        nn.init.zeros_(self.conv_offset.weight)
        self._pk = _Packed()

    def forward(self, x, relu=False):
        offset = self.conv_offset.run(x)
        wp = self._pk.get('w', self.weight, ops.pack_conv_weight, job=(False, None, None, None))
        return ops.deform_conv(x, offset, wp, self.out_channels, self.deform_groups, relu=relu)


class SFMStage(nn.Module):
    """dynamask_head.py:54-125."""

    def __init__(self, semantic_in_channel=256, semantic_out_channel=256, instance_in_channel=256,
                 instance_out_channel=256, out_size=14, num_classes=80, semantic_out_stride=4,
                 mask_use_sigmoid=False, upsample_cfg=dict(type='bilinear', scale_factor=2)):
        super().__init__()
        self.semantic_out_stride = semantic_out_stride
        self.mask_use_sigmoid = mask_use_sigmoid
        self.num_classes = num_classes
        self.out_size = out_size
        self.instance_in_channel = instance_in_channel
        self.instance_out_channel = instance_out_channel
        if upsample_cfg.get('type') != 'bilinear' or upsample_cfg.get('scale_factor') != 2:
            raise NotImplementedError('SFMStage upsample is bilinear x2 in configs/dynamask '
                                      '(a carafe cfg cannot be passed unchanged: SURVEY K17)')
        self.semantic_transform_in = _Conv(semantic_in_channel, semantic_out_channel, 1)
        self.spatial_scale = 1.0 / semantic_out_stride
        self.instance_logits = _Conv(instance_in_channel, num_classes, 1)
        self.detail_logits = _Conv(instance_in_channel, num_classes, 1)
        fuse_in_channel = instance_in_channel + semantic_out_channel + 2
        self.fuse_conv = nn.ModuleList([
            _Conv(fuse_in_channel, instance_in_channel, 1),
            DeformConv2dPack(instance_in_channel, instance_in_channel, kernel_size=(3, 3), stride=(1, 1), padding=1,
                             deform_groups=2)])
        self.fuse_transform_out = _Conv(instance_in_channel, instance_out_channel - 2, 1)

    def semantic_map(self, semantic_feat):
        """relu(semantic_transform_in(P_l)) on the whole FPN map (dynamask_head.py:104); it does
        not depend on the RoIs, so RoI chunks running on different streams share it."""
        return self.semantic_transform_in.run(semantic_feat, relu=True)

    def forward(self, instance_feats, semantic_feat, rois, roi_labels, upsample=True, sem=None, pred_out=None):
        return run_steps(self.steps(instance_feats, semantic_feat, rois, roi_labels, upsample, sem, pred_out))

    def steps(self, instance_feats, semantic_feat, rois, roi_labels, upsample=True, sem=None, pred_out=None):
        """``forward`` as a generator that yields after every launch (``run_steps`` exhausts it): a caller that runs
        several RoI chunks on several streams issues their launches in turn (roi_head._mask_forward_infer), so that
        no stream waits for the host -- or for the graph's node order -- to get through another stream's whole chain."""
        n, c, s = instance_feats.shape[0], self.instance_in_channel, self.out_size
        co = self.instance_out_channel
        # instance-wise semantic feats: relu(conv1x1) on the whole FPN map, then point sample
        if sem is None:
            sem = self.semantic_map(semantic_feat)
            yield
        # [fused_feats(co-2) | sigmoid(ip) | sigmoid(dp)] is assembled in place
        tail = torch.empty((n, co, s, s), device=instance_feats.device, dtype=torch.float32)
        nc = self.num_classes
        logit_args = (instance_feats, self.instance_logits.weight.detach().view(nc, c), self.instance_logits.bias.detach(),
                      self.detail_logits.weight.detach().view(nc, c), self.detail_logits.bias.detach(), roi_labels)
        if FUSED_STAGE_HEAD[0] and n > 0:
            # the point sample and the two class-gathered logits share no data: one launch (ops.stage_head, same bits)
            ins_sem, ip, dp = ops.stage_head(sem, rois, s, self.spatial_scale, *logit_args, sig_out=tail, sig_ch_offset=co - 2,
                                             out=pred_out)
            yield
        else:
            ins_sem = ops.point_sample(sem, rois, s, self.spatial_scale)
            yield
            ip, dp = ops.class_logits(*logit_args, sig_out=tail, sig_ch_offset=co - 2, out=pred_out)
            yield
        fused = self.fuse_conv[0].run([instance_feats, ins_sem, tail[:, co - 2:]], relu=True)
        yield
        dcn = self.fuse_conv[1]
        offset = dcn.conv_offset.run(fused)
        yield
        wp = dcn._pk.get('w', dcn.weight, ops.pack_conv_weight, job=(False, None, None, None))
        tout = self.fuse_transform_out
        if (FUSED_DCN_TOUT[0] and not torch.is_grad_enabled() and tout.bias is not None
                and ops.deform_conv_tout_supported(fused, dcn.out_channels, tout.out_channels)):
            # 28 x 28 / 56 x 56 at more than a handful of RoIs: the 1x1 runs on the DCN's accumulators, the DCN output
            # ([N, 64, 56, 56]: 80 MB at 100 RoIs, written once and read once) never exists
            w2t = tout._pk.get('tout', tout.weight, ops.pack_tout_weight)
            ops.deform_conv_tout(fused, offset, wp, dcn.out_channels, dcn.deform_groups, w2t, tout.bias.detach(),
                                 tout.out_channels, tail)
            yield
        else:
            fused = ops.deform_conv(fused, offset, wp, dcn.out_channels, dcn.deform_groups, relu=True)
            yield
            tout.run(fused, relu=True, out=tail, out_ch_offset=0)
            yield
        if upsample:
            tail = ops.upsample2x(tail, align_corners=False, relu=True)
            yield
        return ip, dp, tail


# inference launches fused in round 6 (A/B switches for tools/infer_bench.py and the equality tests; same bits either way)
import os as _os
FUSED_STAGE_HEAD = [_os.environ.get('DM_FUSED_STAGE_HEAD', '1') != '0']      # point sample + class logits: one launch per stage
FUSED_DCN_TOUT = [_os.environ.get('DM_FUSED_DCN_TOUT', '1') != '0']            # DCN + fuse_transform_out: one launch (28^2 / 56^2)
GROUPED_SEMANTIC_MAPS = [_os.environ.get('DM_GROUPED_SEM', '1') != '0']       # the stages' FPN-wide 1x1 convolutions: one launch


def run_steps(gen):
    """Exhaust a ``steps`` generator and hand back what it returns."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


def _paste_geometry(det_bboxes, ori_shape, scale_factor, rescale):
    """The canvas size and the boxes in canvas pixels, as both heads' ``get_seg_masks`` derive them
    (dynamask_head.py:293-305 = fcn_mask_head.py:176-186)."""
    import numpy as np
    bboxes = det_bboxes[:, :4]
    if rescale:
        img_h, img_w = ori_shape[:2]
    else:
        img_h = int(np.round(ori_shape[0] * scale_factor).astype(np.int32))
        img_w = int(np.round(ori_shape[1] * scale_factor).astype(np.int32))
        scale_factor = 1.0
    if not isinstance(scale_factor, (float, torch.Tensor)):
        scale_factor = bboxes.new_tensor(scale_factor)
    return (bboxes / scale_factor).contiguous(), int(img_h), int(img_w)


def _bitmaps_to_host(im_mask):
    """device -> host as the reference does (``im_mask[i].cpu().numpy()``), but ONE copy of all masks into a fresh pinned
    buffer (PCIe rate instead of the pageable-memory rate); returns the [N, h, w] bool array."""
    host = torch.empty(im_mask.shape, dtype=im_mask.dtype, pin_memory=True)
    host.copy_(im_mask, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


@HEADS.register_module()
class DynaMaskHead(nn.Module):
    """dynamask_head.py:128-244."""

    def __init__(self, num_convs_instance=2, num_convs_semantic=4, conv_in_channels_instance=256,
                 conv_in_channels_semantic=256, conv_kernel_size_instance=3, conv_kernel_size_semantic=3,
                 conv_out_channels_instance=256, conv_out_channels_semantic=256, conv_cfg=None, norm_cfg=None,
                 semantic_out_stride=[16, 8, 4], mask_use_sigmoid=False, pre_upsample_last_stage=False,
                 stage_num_classes=[80, 80, 80, 80], stage_sup_size=[14, 28, 56, 112],
                 upsample_cfg=dict(type='bilinear', scale_factor=2),
                 loss_cfg=dict(type='DynaCrossEntropyLoss')):
        super().__init__()
        self.num_convs_instance = num_convs_instance
        self.conv_kernel_size_instance = conv_kernel_size_instance
        self.conv_in_channels_instance = conv_in_channels_instance
        self.conv_out_channels_instance = conv_out_channels_instance
        self.num_convs_semantic = num_convs_semantic
        self.conv_kernel_size_semantic = conv_kernel_size_semantic
        self.conv_in_channels_semantic = conv_in_channels_semantic
        self.conv_out_channels_semantic = conv_out_channels_semantic
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.semantic_out_stride = semantic_out_stride
        self.stage_sup_size = stage_sup_size
        self.stage_num_classes = stage_num_classes
        self.pre_upsample_last_stage = pre_upsample_last_stage

        convs = []
        for i in range(num_convs_instance):
            cin = conv_in_channels_instance if i == 0 else conv_out_channels_instance
            convs.append(ConvModule(cin, conv_out_channels_instance, conv_kernel_size_instance, dilation=1, padding=1))
        self.instance_convs = nn.ModuleList(convs)
        self.loss_func = build_loss(loss_cfg)

        assert len(self.stage_sup_size) > 1
        self.stages = nn.ModuleList()
        out_channel = conv_out_channels_instance
        for idx, out_size in enumerate(self.stage_sup_size[:-1]):
            in_channel = out_channel
            out_channel = in_channel // 2
            self.stages.append(SFMStage(
                semantic_in_channel=conv_out_channels_semantic, semantic_out_channel=in_channel,
                instance_in_channel=in_channel, instance_out_channel=out_channel, out_size=out_size,
                num_classes=self.stage_num_classes[idx], semantic_out_stride=semantic_out_stride[-1],
                mask_use_sigmoid=mask_use_sigmoid, upsample_cfg=upsample_cfg))
        self.final_instance_logits = _Conv(out_channel, self.stage_num_classes[-1], 1)
        self.final_detail_logits = _Conv(out_channel, self.stage_num_classes[-1], 1)

    def init_weights(self):
        for m in [self.final_instance_logits, self.final_detail_logits]:
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.bias, 0)

    def semantic_maps(self, semantic_feats, last_stage=None):
        """Per-stage relu(semantic_transform_in(.)) maps (shared by all RoIs).  Without autograd, two or three of them
        are ONE launch (ops.conv1x1_group: no RoI enters them, and as three launches of 66 / 132 / 525 workgroups they
        headed the inference chain with ~100 us in which most of the chip idles); same bits either way."""
        n = len(self.stages) if last_stage is None else min(last_stage, len(self.stages))
        feats = [semantic_feats[-i - 3] for i in range(n)]
        convs = [self.stages[i].semantic_transform_in for i in range(n)]
        if (GROUPED_SEMANTIC_MAPS[0] and 2 <= n <= 3 and not torch.is_grad_enabled() and all(f.is_contiguous() for f in feats)
                and all(c.bias is not None for c in convs)):
            return ops.conv1x1_group(feats, [c.packed([c.in_channels]) for c in convs], [c.bias.detach() for c in convs],
                                     [c.out_channels for c in convs], relu=True)
        return [self.stages[i].semantic_map(feats[i]) for i in range(n)]

    def prepack(self, fused_dcn=None):
        """Refresh the kernel-layout weights of everything ``forward`` launches (except the semantic 1x1 convs, which
        ``semantic_maps`` owns) on the CURRENT stream.  The caches are filled by whoever asks first; a caller that is
        about to fork RoI chunks onto several streams calls this first, so that no stream reads a pack another
        stream is still writing.  ``fused_dcn``: per stage, whether the fused DCN kernel's layout is the one needed
        (default: all)."""
        for conv in self.instance_convs:
            conv.conv.packed([conv.conv.in_channels])
        for i, stage in enumerate(self.stages):
            c, dcn = stage.instance_in_channel, stage.fuse_conv[1]
            stage.fuse_conv[0].packed([c, stage.semantic_transform_in.out_channels, 2])
            dcn.conv_offset.packed([c])
            if fused_dcn is None or fused_dcn[i]:
                dcn._pk.get('w', dcn.weight, ops.pack_conv_weight, job=(False, None, None, None))
            stage.fuse_transform_out.packed([dcn.out_channels])
            if FUSED_DCN_TOUT[0] and not torch.is_grad_enabled() and stage.fuse_transform_out.weight.is_cuda:
                stage.fuse_transform_out._pk.get('tout', stage.fuse_transform_out.weight, ops.pack_tout_weight)

    def pred_sizes(self, last_stage=None, defer_final_up=False):
        """Spatial size of every (instance, detail) logit pair ``forward`` returns, in order."""
        n = len(self.stages) if last_stage is None else min(last_stage, len(self.stages))
        sizes = [self.stages[i].out_size for i in range(n)]
        if last_stage is not None and last_stage < len(self.stages):
            return sizes + [self.stages[last_stage].out_size]
        last = 2 * self.stages[-1].out_size
        return sizes + [last // 2 if (defer_final_up and not self.pre_upsample_last_stage) else last]

    def forward(self, instance_feats, semantic_feats, rois, roi_labels, last_stage=None, sems=None, pred_out=None):
        """Returns (stage_instance_preds, stage_detail_preds) as the reference.

        ``last_stage`` (extension, default None = all): stop after the logits of
        that exit (1 = the fixed 28x28 exit of BASELINE configs[1]).
        ``sems`` (extension): precomputed ``semantic_maps`` (multi-stream inference).
        ``pred_out`` (extension): one ``(instance, detail)`` pair of [N, 1, S, S] tensors per returned logit pair
        (``pred_sizes``) to write into -- the row slices of a chunked, multi-stream caller's buffers."""
        return run_steps(self.steps(instance_feats, semantic_feats, rois, roi_labels, last_stage, sems, pred_out))

    def steps(self, instance_feats, semantic_feats, rois, roi_labels, last_stage=None, sems=None, pred_out=None, extract=None,
              defer_final_up=False):
        """``forward`` as a generator that yields after every launch (see ``SFMStage.steps``).  ``extract``: a
        callable producing ``instance_feats`` -- the RoI extraction as the chain's first step.  ``defer_final_up``: hand
        back the last stage's logits at ITS resolution (2S) -- the caller folds their align_corners x2 upsample into the
        boundary merge (ops.boundary_merge_chain)."""
        if extract is not None:
            instance_feats = extract()
            yield
        po = (lambda i: None) if pred_out is None else (lambda i: pred_out[i])
        for conv in self.instance_convs:
            instance_feats = conv(instance_feats)
            yield
        stage_instance_preds, stage_detail_preds = [], []
        roi_labels = roi_labels.long().contiguous()
        fused_exit = False
        for idx, stage in enumerate(self.stages):
            if last_stage is not None and idx == last_stage:
                # exit here: only the logits of this resolution are needed
                c, nc = stage.instance_in_channel, stage.num_classes
                logits = ops.class_logits_up2x if fused_exit else ops.class_logits
                ip, dp = logits(instance_feats, stage.instance_logits.weight.detach().view(nc, c),
                                stage.instance_logits.bias.detach(), stage.detail_logits.weight.detach().view(nc, c),
                                stage.detail_logits.bias.detach(), roi_labels, out=po(idx))
                stage_instance_preds.append(ip)
                stage_detail_preds.append(dp)
                return stage_instance_preds, stage_detail_preds
            upsample_flag = self.pre_upsample_last_stage or idx < len(self.stages) - 1
            # the stage before the exit: its x2 upsample would only feed the exit's two logit maps -- they are computed from
            # the stage's own resolution instead (ops.class_logits_up2x), the upsampled tensor never exists
            fused_exit = (last_stage is not None and idx + 1 == last_stage and idx + 1 < len(self.stages) and upsample_flag
                          and not torch.is_grad_enabled() and ops.class_logits_up2x_supported(instance_feats))
            ip, dp, instance_feats = yield from stage.steps(
                instance_feats, semantic_feats[-idx - 3], rois, roi_labels, upsample_flag and not fused_exit,
                sem=None if sems is None else sems[idx], pred_out=po(idx))
            stage_instance_preds.append(ip)
            stage_detail_preds.append(dp)
        # (dynamask_head.py:236-237 clamps the labels to 0 for the class-agnostic last stage: the kernel clamps every
        # label into [0, num_classes - 1] itself, which for one class is that clamp -- no torch launch for it)
        nc = self.stage_num_classes[-1]
        c = self.final_instance_logits.in_channels
        direct = self.pre_upsample_last_stage or defer_final_up
        ip, dp = ops.class_logits(instance_feats, self.final_instance_logits.weight.detach().view(nc, c),
                                  self.final_instance_logits.bias.detach(),
                                  self.final_detail_logits.weight.detach().view(nc, c),
                                  self.final_detail_logits.bias.detach(), roi_labels,
                                  out=po(len(self.stages)) if direct else None)
        yield
        if not direct:
            fin = po(len(self.stages))
            ip = ops.upsample2x(ip, align_corners=True, out=None if fin is None else fin[0])
            yield
            dp = ops.upsample2x(dp, align_corners=True, out=None if fin is None else fin[1])
            yield
        stage_instance_preds.append(ip)
        stage_detail_preds.append(dp)
        return stage_instance_preds, stage_detail_preds


    def forward_dynamic(self, instance_feats, semantic_feats, rois, roi_labels, n_ge, sems=None):
        """Per-RoI early exit (the reference's intent, present there only
        as commented-out code, dynamask_roi_head.py:160-204).  RoIs must be ordered by exit,
        deepest first; ``n_ge[k]`` = number of RoIs whose exit is >= k (``n_ge[0] == N``), so
        the RoIs still alive at stage k are the prefix ``[:n_ge[k]]`` and no gather is needed.
        Stage k's full body runs only on the RoIs that continue; the ones that exit at k get
        just the two class-gathered logits.  Returns a list over k of instance logits
        ``[n_ge[k], 1, S_k, S_k]`` -- row j equals the fixed path's exit-k prediction of RoI j
        bit for bit (RoIs never interact inside the head)."""
        n_ge = list(n_ge) + [0]
        assert n_ge[0] == instance_feats.shape[0] and all(a >= b for a, b in zip(n_ge[:-1], n_ge[1:]))
        for conv in self.instance_convs:
            instance_feats = conv(instance_feats)
        roi_labels = roi_labels.long().contiguous()
        preds = []
        exit_ip = None      # logits of the RoIs that exit at the coming stage, computed from the previous stage's own resolution
        for idx, stage in enumerate(self.stages):
            n_here, n_cont = n_ge[idx], n_ge[idx + 1]
            if n_here == 0:
                preds.append(instance_feats.new_zeros((0, 1, stage.out_size, stage.out_size)))
                continue
            parts = []
            feats_here = instance_feats
            carried, exit_ip = exit_ip, None
            if n_cont > 0:
                upsample_flag = self.pre_upsample_last_stage or idx < len(self.stages) - 1
                n_keep = n_ge[idx + 2] if idx + 2 < len(n_ge) else 0        # RoIs that go on past the next stage
                fuse_next = (upsample_flag and idx + 1 < len(self.stages) and n_keep < n_cont
                             and ops.class_logits_up2x_supported(feats_here))
                ip, _, tail = stage(feats_here[:n_cont], semantic_feats[-idx - 3], rois[:n_cont], roi_labels[:n_cont],
                                    upsample_flag and not fuse_next, sem=None if sems is None else sems[idx])
                if fuse_next:
                    # the RoIs that exit at the next stage need its two logit maps only: from this stage's resolution, the
                    # upsampled features exist for the rows that go on (same bits as the fixed path's exit: same kernel)
                    nxt = self.stages[idx + 1]
                    c, nc = nxt.instance_in_channel, nxt.num_classes
                    exit_ip, _ = ops.class_logits_up2x(tail[n_keep:n_cont], nxt.instance_logits.weight.detach().view(nc, c),
                                                       nxt.instance_logits.bias.detach(), nxt.detail_logits.weight.detach().view(nc, c),
                                                       nxt.detail_logits.bias.detach(), roi_labels[n_keep:n_cont])
                    instance_feats = (ops.upsample2x(tail[:n_keep], align_corners=False, relu=True) if n_keep > 0
                                      else tail.new_zeros((0, tail.shape[1], 2 * tail.shape[2], 2 * tail.shape[3])))
                else:
                    instance_feats = tail
                parts.append(ip)
            if n_cont < n_here:
                if carried is not None:
                    assert carried.shape[0] == n_here - n_cont
                    parts.append(carried)
                else:
                    c, nc = stage.instance_in_channel, stage.num_classes
                    ip, _ = ops.class_logits(feats_here[n_cont:n_here], stage.instance_logits.weight.detach().view(nc, c),
                                             stage.instance_logits.bias.detach(),
                                             stage.detail_logits.weight.detach().view(nc, c),
                                             stage.detail_logits.bias.detach(), roi_labels[n_cont:n_here])
                    parts.append(ip)
            preds.append(parts[0] if len(parts) == 1 else torch.cat(parts))
        n_last = n_ge[len(self.stages)]
        s_last = self.stage_sup_size[-1]
        if n_last == 0:
            preds.append(instance_feats.new_zeros((0, 1, s_last, s_last)))
            return preds
        lab = roi_labels[:n_last]
        if self.stage_num_classes[-1] == 1:
            lab = lab.clamp(max=0)
        nc = self.stage_num_classes[-1]
        c = self.final_instance_logits.in_channels
        ip, _ = ops.class_logits(instance_feats, self.final_instance_logits.weight.detach().view(nc, c),
                                 self.final_instance_logits.bias.detach(),
                                 self.final_detail_logits.weight.detach().view(nc, c),
                                 self.final_detail_logits.bias.detach(), lab)
        if not self.pre_upsample_last_stage:
            ip = ops.upsample2x(ip, align_corners=True)
        preds.append(ip)
        return preds

    # ------------------------------------------------ callers either side of the path
    def get_targets(self, pos_bboxes_list, pos_assigned_gt_inds_list, gt_masks_list):
        """dynamask_head.py:246-271 with the GT bitmaps already on the device
        ([G, H, W] tensors per image): clip + RoIAlign(scale 1, adaptive grid) on the
        bitmaps + (>= 0.5), for every supervision size -- no device->host->device trip
        (the reference goes through numpy per image and size)."""
        per_stage = [[] for _ in self.stage_sup_size]
        for boxes, inds, masks in zip(pos_bboxes_list, pos_assigned_gt_inds_list, gt_masks_list):
            if boxes.shape[0] == 0:              # an image without positives (no GT): nothing to crop
                for i, size in enumerate(self.stage_sup_size):
                    per_stage[i].append(boxes.new_zeros((0, size, size)))
                continue
            if hasattr(masks, 'masks') and isinstance(masks.masks, (list, tuple)):
                # a PolygonMasks-like holder (list over objects of lists of vertex arrays; .height / .width): the
                # polygons go to the device once and every size is rasterised there (structures.py:469-503, 583-599)
                packed = ops.pack_polygons(masks.masks, boxes.device)
                b = boxes[:, :4].contiguous().float().clone()
                b[:, 0::2].clamp_(0, float(masks.width))
                b[:, 1::2].clamp_(0, float(masks.height))
                for i, size in enumerate(self.stage_sup_size):
                    per_stage[i].append(ops.polygon_mask_targets(packed, b, inds.long().contiguous(), size))
                continue
            if hasattr(masks, 'masks'):           # a BitmapMasks-like holder of a numpy array
                masks = torch.from_numpy(masks.masks).to(boxes.device)
            m = masks.to(torch.float32).contiguous()[:, None]
            maxh, maxw = m.shape[-2:]
            rois = ops.mask_target_rois(boxes[:, :4].contiguous().float(), inds.long().contiguous(), maxw, maxh)
            for i, size in enumerate(self.stage_sup_size):
                t = ops.roi_align([m], rois, size, [1.0], 0)
                per_stage[i].append(ops.threshold_ge(t, 0.5).squeeze(1))
        return [torch.cat(t) for t in per_stage]

    def get_seg_masks(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale):
        """dynamask_head.py:279-342: sigmoid -> paste into the image -> threshold ->
        list of (h, w) bool numpy arrays (one paste kernel for all detections)."""
        bboxes, img_h, img_w = _paste_geometry(det_bboxes, ori_shape, scale_factor, rescale)
        threshold = rcnn_test_cfg.mask_thr_binary
        if threshold < 0:
            raise NotImplementedError('visualisation mode (mask_thr_binary < 0) is not on the path')
        if mask_pred.shape[1] > 1:
            mask_pred = mask_pred[range(len(mask_pred)), det_labels][:, None]
        im_mask = ops.paste_masks(mask_pred.contiguous(), bboxes, img_h, img_w, threshold, apply_sigmoid=True)
        im = _bitmaps_to_host(im_mask)
        return [im[i] for i in range(len(im))]


    def get_seg_rles(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale):
        """``get_seg_masks`` followed by ``encode_mask_results`` (dynamask_head.py:279-342 +
        core/mask/utils.py:36-63) without the bitmaps: paste, threshold and run-length
        encoding run on the device, only run boundaries are copied to the host.  Returns one
        COCO RLE dict per detection -- what ``mask_util.encode(np.array(m[:, :, None],
        order='F'))[0]`` yields for the bitmap ``get_seg_masks`` would have returned."""
        bboxes, img_h, img_w = _paste_geometry(det_bboxes, ori_shape, scale_factor, rescale)
        threshold = rcnn_test_cfg.mask_thr_binary
        if threshold < 0:
            raise NotImplementedError('visualisation mode (mask_thr_binary < 0) is not on the path')
        if mask_pred.shape[1] > 1:
            mask_pred = mask_pred[range(len(mask_pred)), det_labels][:, None]
        return ops.paste_rle(mask_pred.contiguous(), bboxes, img_h, img_w, threshold, apply_sigmoid=True)


# ---------------------------------------------------------------- FCN mask head
@UPSAMPLE_LAYERS.register_module(name='deconv')
class _Deconv(nn.Module):
    """nn.ConvTranspose2d(k=2, s=2) parameter holder: weight [Cin, Cout, 2, 2]."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        if kernel_size != 2 or stride != 2:
            raise NotImplementedError('deconv upsample is 2x2 stride 2 on the path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 2, 2))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')
        self._pk = _Packed()

    def forward(self, x, relu=False):
        wp = self._pk.get('w', self.weight, ops.pack_deconv_weight)
        return ops.deconv2x2(x, wp, self.bias.detach(), self.out_channels, relu=relu)


@UPSAMPLE_LAYERS.register_module(name='carafe')
class CARAFEPack(nn.Module):
    """mmcv CARAFEPack: keys channel_compressor.*, content_encoder.*."""

    def __init__(self, channels, scale_factor, up_kernel=5, up_group=1, encoder_kernel=3, encoder_dilation=1,
                 compressed_channels=64):
        super().__init__()
        if encoder_kernel != 3 or encoder_dilation != 1:
            raise NotImplementedError('CARAFE encoder is 3x3, dilation 1 on the path')
        self.channels, self.scale_factor, self.up_kernel, self.up_group = channels, scale_factor, up_kernel, up_group
        self.channel_compressor = _Conv(channels, compressed_channels, 1)
        self.content_encoder = _Conv(compressed_channels, up_kernel * up_kernel * up_group * scale_factor ** 2, 3)
        self.init_weights()

    def init_weights(self):
        nn.init.xavier_uniform_(self.channel_compressor.weight)
        nn.init.zeros_(self.channel_compressor.bias)
        nn.init.normal_(self.content_encoder.weight, std=0.001)
        nn.init.zeros_(self.content_encoder.bias)

    def forward(self, x, relu=False):
        comp = self.channel_compressor.run(x)
        enc = self.content_encoder.run(comp)
        return ops.carafe(x, enc, self.up_kernel, self.up_group, self.scale_factor)


@UPSAMPLE_LAYERS.register_module(name='bilinear')
class _BilinearUp(nn.Module):
    def __init__(self, scale_factor=2, mode='bilinear', align_corners=False):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError('x2 upsampling only')
        self.align_corners = bool(align_corners)

    def forward(self, x, relu=False):
        return ops.upsample2x(x, align_corners=self.align_corners, relu=relu)


@UPSAMPLE_LAYERS.register_module(name='nearest')
class _NearestUp(nn.Module):
    """nn.Upsample(scale_factor=2, mode='nearest') (fcn_mask_head.py:88-96)."""

    def __init__(self, scale_factor=2, mode='nearest', align_corners=None):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError('x2 upsampling only')

    def forward(self, x, relu=False):
        assert not relu
        return ops.upsample2x_nearest(x)


def build_upsample_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    cls = UPSAMPLE_LAYERS.get(t)
    if cls is None:
        raise KeyError(f'unsupported upsample type {t!r}')
    return cls(**cfg)


@HEADS.register_module()
class FCNMaskHead(nn.Module):
    """fcn_mask_head.py:19-126.  Forward and (train_path.FCNMaskHeadFn) its backward for every upsample
    type; the fork's own ``loss`` is broken (``mask_cross_entropy`` changed its signature),
    so the head is differentiable but the reference cannot train it through ``loss_mask``."""

    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3, conv_out_channels=256,
                 num_classes=80, class_agnostic=False, upsample_cfg=dict(type='deconv', scale_factor=2),
                 conv_cfg=None, norm_cfg=None, loss_mask=None):
        super().__init__()
        self.upsample_cfg = dict(upsample_cfg)
        if self.upsample_cfg['type'] not in [None, 'deconv', 'nearest', 'bilinear', 'carafe']:
            raise ValueError(f'Invalid upsample method {self.upsample_cfg["type"]}, accepted methods are '
                             '"deconv", "nearest", "bilinear", "carafe"')
        self.num_convs = num_convs
        self.in_channels = in_channels
        self.conv_kernel_size = conv_kernel_size
        self.conv_out_channels = conv_out_channels
        self.upsample_method = self.upsample_cfg.get('type')
        self.scale_factor = self.upsample_cfg.pop('scale_factor', None)
        self.num_classes = num_classes
        self.class_agnostic = class_agnostic
        self.convs = nn.ModuleList()
        for i in range(num_convs):
            cin = in_channels if i == 0 else conv_out_channels
            self.convs.append(ConvModule(cin, conv_out_channels, conv_kernel_size, padding=(conv_kernel_size - 1) // 2,
                                         conv_cfg=conv_cfg, norm_cfg=norm_cfg))
        up_in = conv_out_channels if num_convs > 0 else in_channels
        cfg_ = dict(self.upsample_cfg)
        if self.upsample_method is None:
            self.upsample = None
        elif self.upsample_method == 'deconv':
            cfg_.update(in_channels=up_in, out_channels=conv_out_channels, kernel_size=self.scale_factor,
                        stride=self.scale_factor)
            self.upsample = build_upsample_layer(cfg_)
        elif self.upsample_method == 'carafe':
            cfg_.update(channels=up_in, scale_factor=self.scale_factor)
            self.upsample = build_upsample_layer(cfg_)
        elif self.upsample_method == 'bilinear':
            cfg_.update(scale_factor=self.scale_factor, mode='bilinear', align_corners=False)
            self.upsample = build_upsample_layer(cfg_)
        else:
            cfg_.update(scale_factor=self.scale_factor, mode='nearest', align_corners=None)
            self.upsample = build_upsample_layer(cfg_)
        out_channels = 1 if class_agnostic else num_classes
        logits_in = conv_out_channels if self.upsample_method == 'deconv' else up_in
        self.conv_logits = _Conv(logits_in, out_channels, 1)

    def init_weights(self):
        for m in [self.upsample, self.conv_logits]:
            if m is None:
                continue
            if isinstance(m, CARAFEPack):
                m.init_weights()
            elif hasattr(m, 'weight'):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .train_path import FCNMaskHeadFn
            return FCNMaskHeadFn.apply(self, x, *list(self.parameters()))
        for conv in self.convs:
            x = conv(x)
        if self.upsample is not None:
            x = self.upsample(x, relu=(self.upsample_method == 'deconv'))
        return self.conv_logits.run(x)

    # ------------------------------------------------ callers either side of the path
    def get_targets(self, sampling_results, gt_masks, rcnn_train_cfg):
        """fcn_mask_head.py:128-135 + core/mask/mask_target.py:7-62: per image, clip the positive proposals to the GT
        canvas, crop-and-resize the assigned GT bitmap to ``rcnn_train_cfg.mask_size`` (RoIAlign, scale 1, adaptive
        grid, aligned) and threshold at 0.5 -> float [N, S, S].  On the device: ``gt_masks`` are [G, H, W] tensors, or
        BitmapMasks- / PolygonMasks-like holders (``.masks``), as ``DynaMaskHead.get_targets`` takes them."""
        ms = rcnn_train_cfg.mask_size if hasattr(rcnn_train_cfg, 'mask_size') else rcnn_train_cfg['mask_size']
        size = ms if isinstance(ms, int) else ms[0]
        if not isinstance(ms, int) and ms[0] != ms[1]:
            raise NotImplementedError('square mask targets only (mask_size is an int in every config of the reference)')
        out = []
        for res, masks in zip(sampling_results, gt_masks):
            boxes, inds = res.pos_bboxes, res.pos_assigned_gt_inds
            if boxes.shape[0] == 0:
                out.append(boxes.new_zeros((0, size, size)))
                continue
            if hasattr(masks, 'masks') and isinstance(masks.masks, (list, tuple)):
                packed = ops.pack_polygons(masks.masks, boxes.device)
                b = boxes[:, :4].contiguous().float().clone()
                b[:, 0::2].clamp_(0, float(masks.width))
                b[:, 1::2].clamp_(0, float(masks.height))
                out.append(ops.polygon_mask_targets(packed, b, inds.long().contiguous(), size).float())
                continue
            if hasattr(masks, 'masks'):
                masks = torch.from_numpy(masks.masks).to(boxes.device)
            m = masks.to(torch.float32).contiguous()[:, None]
            maxh, maxw = m.shape[-2:]
            rois = ops.mask_target_rois(boxes[:, :4].contiguous().float(), inds.long().contiguous(), maxw, maxh)
            out.append(ops.threshold_ge(ops.roi_align([m], rois, size, [1.0], 0), 0.5).squeeze(1).float())
        return torch.cat(out) if out else out

    def loss(self, mask_pred, mask_targets, labels):
        """fcn_mask_head.py:137-149 cannot run in the fork: ``mask_cross_entropy`` lost its ``label`` argument
        (cross_entropy_loss.py:90-120), so ``CrossEntropyLoss(use_mask=True)`` raises there too."""
        raise NotImplementedError('FCNMaskHead.loss is broken in the reference fork itself (SURVEY App. C Q5)')

    def _selected(self, mask_pred, det_bboxes, det_labels):
        if isinstance(mask_pred, torch.Tensor):
            apply_sigmoid = True            # single-scale testing hands over logits (fcn_mask_head.py:168-169)
        else:                               # multi-scale testing: probabilities, already averaged, as an ndarray (:170-171)
            mask_pred, apply_sigmoid = det_bboxes.new_tensor(mask_pred), False
        if not self.class_agnostic:
            n = len(mask_pred)
            mask_pred = mask_pred[torch.arange(n, device=mask_pred.device), det_labels.to(mask_pred.device)][:, None]
        return mask_pred.contiguous(), apply_sigmoid

    def get_seg_masks(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale):
        """fcn_mask_head.py:151-237: (sigmoid ->) class select -> paste into the image -> threshold -> ``cls_segms``:
        one list per class holding the (h, w) bool arrays of that class's detections, in detection order.  The
        [n, classes, S, S] logits are gathered first (the sigmoid commutes with the selection), then one paste kernel
        runs for all detections and ONE device -> host copy carries the bitmaps."""
        bboxes, img_h, img_w = _paste_geometry(det_bboxes, ori_shape, scale_factor, rescale)
        threshold = rcnn_test_cfg.mask_thr_binary
        if threshold < 0:
            raise NotImplementedError('visualisation mode (mask_thr_binary < 0) is not on the path')
        cls_segms = [[] for _ in range(self.num_classes)]
        if len(mask_pred) == 0:
            return cls_segms
        sel, apply_sigmoid = self._selected(mask_pred, det_bboxes, det_labels)
        im = _bitmaps_to_host(ops.paste_masks(sel, bboxes, img_h, img_w, threshold, apply_sigmoid=apply_sigmoid))
        for i, lab in enumerate(det_labels.tolist()):
            cls_segms[lab].append(im[i])
        return cls_segms

    def get_seg_rles(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale):
        """``get_seg_masks`` followed by ``encode_mask_results`` (core/mask/utils.py:36-63) without the bitmaps: paste,
        threshold and run-length encoding on the device; returns ``cls_segms`` of COCO RLE dicts."""
        bboxes, img_h, img_w = _paste_geometry(det_bboxes, ori_shape, scale_factor, rescale)
        threshold = rcnn_test_cfg.mask_thr_binary
        if threshold < 0:
            raise NotImplementedError('visualisation mode (mask_thr_binary < 0) is not on the path')
        cls_segms = [[] for _ in range(self.num_classes)]
        if len(mask_pred) == 0:
            return cls_segms
        sel, apply_sigmoid = self._selected(mask_pred, det_bboxes, det_labels)
        rles = ops.paste_rle(sel, bboxes, img_h, img_w, threshold, apply_sigmoid=apply_sigmoid)
        for lab, r in zip(det_labels.tolist(), rles):
            cls_segms[lab].append(r)
        return cls_segms
