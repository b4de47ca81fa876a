"""Generate golden vectors from the REFERENCE's own Python modules.

Run ONLY in the authoring container (needs /root/reference):

    python tests/golden/make_golden.py

Loads the reference files of the hot path by file path (the ``mmdet`` package
itself cannot be imported: ``mmcv`` is absent -- an ordinary ImportError, not
an environment denial).  Third-party names the reference imports are provided
as *stand-ins that contain no DynaMask arithmetic*: a Registry, identity
decorators, ``nn.Conv2d``-based ``ConvModule``, ``nn.Upsample`` /
``nn.ConvTranspose2d`` for ``build_upsample_layer``.  The four mmcv operators
(RoIAlign, SimpleRoIAlign, DeformConv2dPack, CARAFEPack) delegate to
``oracle/ref_ops.py`` -- they are third-party code absent from the tree, so
parity for those ops stays "unpinned" (DESIGN.md); what these goldens PIN is
everything the reference itself owns: losses, DetailTarget,
generate_block_target, MaskPre, the gumbel selector, DynaMaskHead/SFMStage
control flow and channel plumbing, SingleRoIExtractor level mapping,
FCNMaskHead forward, the inference boundary merge.

Outputs: tests/golden/*.npz (inputs that are not re-derivable from a seed,
and expected outputs).  Inputs are produced by ``golden_inputs.py`` (shared
with the tests) from fixed seeds.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import ref_ops  # noqa: E402
import golden_inputs as gi  # noqa: E402


# ------------------------------------------------------------------ stand-ins
def _pkg(name):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    if '.' in name:
        parent, child = name.rsplit('.', 1)
        setattr(_pkg(parent), child, m)
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    parent, child = name.rsplit('.', 1)
    setattr(_pkg(parent), child, mod)
    spec.loader.exec_module(mod)
    return mod


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def get(self, key):
        return self.module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    cls = registry.get(args.pop('type'))
    return cls(**args)


class ConvModule(nn.Module):
    """Conv2d(bias=True) -> ReLU(inplace) (mmcv ConvModule with norm_cfg=None)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        assert conv_cfg is None and norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation)
        self.activate = nn.ReLU(inplace=True) if act_cfg is not None else None

    def forward(self, x):
        x = self.conv(x)
        return self.activate(x) if self.activate is not None else x


class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.aligned = aligned

    def forward(self, input, rois):
        return ref_ops.roi_align(input, rois, self.output_size[0], self.spatial_scale,
                                 self.sampling_ratio, self.aligned)


class SimpleRoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, aligned=True):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.aligned = aligned

    def forward(self, features, rois):
        return ref_ops.simple_roi_align(features, rois, self.output_size[0], self.spatial_scale, self.aligned)


class DeformConv2dPack(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False):
        super().__init__()
        assert not bias and groups == 1
        ks = _pair(kernel_size)
        self.deform_groups = deform_groups
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, *ks))
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 2 * ks[0] * ks[1], ks, _pair(stride),
                                     _pair(padding), bias=True)

    def forward(self, x):
        return ref_ops.deform_conv_pack(x, self.weight, self.conv_offset.weight, self.conv_offset.bias,
                                        self.deform_groups)


class CARAFEPack(nn.Module):
    def __init__(self, channels, scale_factor, up_kernel=5, up_group=1, encoder_kernel=3, encoder_dilation=1,
                 compressed_channels=64):
        super().__init__()
        self.cfg = dict(scale=scale_factor, up_kernel=up_kernel, up_group=up_group,
                        encoder_kernel=encoder_kernel, encoder_dilation=encoder_dilation)
        self.channel_compressor = nn.Conv2d(channels, compressed_channels, 1)
        self.content_encoder = nn.Conv2d(compressed_channels, up_kernel * up_kernel * up_group * scale_factor ** 2,
                                         encoder_kernel, padding=int((encoder_kernel - 1) * encoder_dilation / 2),
                                         dilation=encoder_dilation)

    def init_weights(self):
        pass

    def forward(self, x):
        return ref_ops.carafe_pack(x, self.channel_compressor.weight, self.channel_compressor.bias,
                                   self.content_encoder.weight, self.content_encoder.bias, **self.cfg)


def build_upsample_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t in ('bilinear', 'nearest'):
        cfg.setdefault('mode', t)
        return nn.Upsample(**cfg)
    if t == 'deconv':
        return nn.ConvTranspose2d(**cfg)
    if t == 'carafe':
        return CARAFEPack(**cfg)
    raise KeyError(t)


def _identity_decorator(*a, **k):
    def deco(f):
        return f
    return deco


def install_standins():
    mmcv = _pkg('mmcv')
    utils = _pkg('mmcv.utils')
    utils.Registry = Registry
    utils.build_from_cfg = build_from_cfg
    cnn = _pkg('mmcv.cnn')
    cnn.ConvModule = ConvModule
    cnn.build_upsample_layer = build_upsample_layer
    ops = _pkg('mmcv.ops')
    ops.RoIAlign = RoIAlign
    ops.SimpleRoIAlign = SimpleRoIAlign
    ops.DeformConv2dPack = DeformConv2dPack
    ops.Conv2d = nn.Conv2d
    ra = _pkg('mmcv.ops.roi_align')
    ra.roi_align = lambda inp, rois, out_shape, scale=1.0, sr=0, mode='avg', aligned=True: ref_ops.roi_align(
        inp, rois, out_shape if isinstance(out_shape, int) else out_shape[0], scale, sr, aligned)
    pc = _pkg('pycocotools')
    pcm = _pkg('pycocotools.mask')
    pc.mask = pcm
    mmcv.imresize = mmcv.imflip = mmcv.impad = mmcv.imrescale = mmcv.imrotate = mmcv.imshear = mmcv.imtranslate = None
    car = _pkg('mmcv.ops.carafe')
    car.CARAFEPack = CARAFEPack
    mmcv.ops = ops

    mu = _pkg('mmdet.utils')
    mu.get_root_logger = lambda *a, **k: None
    core = _pkg('mmdet.core')
    core.force_fp32 = _identity_decorator
    core.auto_fp16 = _identity_decorator
    for nm in ('mask_target', 'bbox2result', 'build_assigner', 'build_sampler'):
        setattr(core, nm, None)
    tr = _load('mmdet.core.bbox.transforms', 'mmdet/core/bbox/transforms.py')
    core.bbox2roi = tr.bbox2roi
    st = _pkg('mmdet.core.mask.structures')
    st.polygon_to_bitmap = None
    st.BitmapMasks = None
    _pkg('mmdet.models')
    _pkg('mmdet.models.losses')
    _pkg('mmdet.models.roi_heads')
    _pkg('mmdet.models.roi_heads.roi_extractors')
    _pkg('mmdet.models.roi_heads.mask_heads')
    tm = _pkg('mmdet.models.roi_heads.test_mixins')
    tm.BBoxTestMixin = type('BBoxTestMixin', (), {})
    tm.MaskTestMixin = type('MaskTestMixin', (), {})
    # the reference hard-codes CUDA placement (Quirk Q6)
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self


def load_reference():
    install_standins()
    R = {}
    R['builder'] = _load('mmdet.models.builder', 'mmdet/models/builder.py')
    _load('mmdet.models.losses.utils', 'mmdet/models/losses/utils.py')
    R['ce'] = _load('mmdet.models.losses.cross_entropy_loss', 'mmdet/models/losses/cross_entropy_loss.py')
    _load('mmdet.models.roi_heads.roi_extractors.base_roi_extractor',
          'mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py')
    R['ext'] = _load('mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor',
                     'mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py')
    R['base'] = _load('mmdet.models.roi_heads.base_roi_head', 'mmdet/models/roi_heads/base_roi_head.py')
    R['fcn'] = _load('mmdet.models.roi_heads.mask_heads.fcn_mask_head',
                     'mmdet/models/roi_heads/mask_heads/fcn_mask_head.py')
    R['head'] = _load('mmdet.models.roi_heads.mask_heads.dynamask_head',
                      'mmdet/models/roi_heads/mask_heads/dynamask_head.py')
    _load('mmdet.models.roi_heads.standard_roi_head', 'mmdet/models/roi_heads/standard_roi_head.py')
    R['roi'] = _load('mmdet.models.roi_heads.dynamask_roi_head', 'mmdet/models/roi_heads/dynamask_roi_head.py')
    return R


def _np(t):
    return t.detach().cpu().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    R = load_reference()
    ce = R['ce']

    # ------------------------------------------------------------- G1 losses
    li = gi.loss_inputs()
    ips = [t.clone().requires_grad_(True) for t in li['ips']]
    dps = [t.clone().requires_grad_(True) for t in li['dps']]
    ml = li['mask_labels'].clone().requires_grad_(True)
    loss_mod = ce.DynaCrossEntropyLoss(**gi.LOSS_CFG)
    out = loss_mod(ips, dps, li['targets'], ml)
    loss = out['loss_masks']
    loss.backward()
    g1 = {'loss_masks': _np(loss), 'grad_mask_labels': _np(ml.grad)}
    for i in range(4):
        g1[f'grad_ip{i}'] = _np(ips[i].grad) if ips[i].grad is not None else np.zeros(tuple(ips[i].shape), np.float32)
        g1[f'grad_dp{i}'] = _np(dps[i].grad)
        g1[f'detail_target{i}'] = _np(loss_mod.detail_target(li['targets'][i]))
    for bw in (1, 2, 3):
        g1[f'block_target_bw{bw}'] = _np(ce.generate_block_target(li['targets'][1], boundary_width=bw))
    g1['bce_stage1'] = _np(ce.binary_cross_entropy(li['ips'][1].squeeze(1), li['targets'][1]))
    g1['epsbce_stage1'] = _np(ce.mask_cross_entropy(li['dps'][1].squeeze(1), li['targets'][1],
                                                     class_weight=li['mask_labels'][:, 1].view(-1, 1, 1)))
    np.savez_compressed(os.path.join(HERE, 'g1_losses.npz'), **g1)
    print('g1 loss', float(loss))

    # ------------------------------------------------------------ G2 MaskPre
    mp = R['base'].MaskPre()
    sd = gi.mask_pre_state()
    mp.load_state_dict({k[len('mask_predictor.'):]: v for k, v in sd.items()}, strict=True)
    x = gi.mask_pre_input()
    mp.eval()
    with torch.no_grad():
        logits_eval = mp(x)        # with the initial running stats (before the train pass updates them)
    mp.train()
    xin = x.clone().requires_grad_(True)
    logits = mp(xin)
    logits.square().sum().backward()
    g2 = {'logits_train': _np(logits), 'grad_fc2_w': _np(mp.fc2.weight.grad), 'grad_conv1_b': _np(mp.conv1.bias.grad),
          'grad_bn1_w': _np(mp.bn1.weight.grad), 'grad_conv2_w': _np(mp.conv2.weight.grad),
          'bn1_running_mean': _np(mp.bn1.running_mean), 'bn1_running_var': _np(mp.bn1.running_var),
          'bn2_running_var': _np(mp.bn2.running_var)}
    g2['logits_eval'] = _np(logits_eval)
    np.savez_compressed(os.path.join(HERE, 'g2_maskpre.npz'), **g2)
    print('g2 logits', logits[0].tolist())

    # ----------------------------------------------------- G3 gumbel selector
    roi_cls = R['roi'].DynaMaskRoIHead
    sel = roi_cls.__new__(roi_cls)
    nn.Module.__init__(sel)
    sel.mask_predictor = lambda t: t
    glog = gi.gumbel_logits()
    torch.manual_seed(gi.GUMBEL_SEED)
    glog_r = glog.clone().requires_grad_(True)
    y = sel.get_mask_label(glog_r)
    (y * torch.arange(1, 5, dtype=torch.float32)).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'g3_gumbel.npz'), y_hard=_np(y), index=_np(y.argmax(-1)),
                        grad_logits=_np(glog_r.grad))
    print('g3 idx', y.argmax(-1).tolist())

    # ----------------------------------------- G4 extractor + DynaMaskHead forward
    hi = gi.head_inputs()
    ext = R['ext'].SingleRoIExtractor(**gi.MASK_ROI_EXTRACTOR_CFG)
    ins = ext(hi['feats'][:4], hi['rois'])
    lv = ext.map_roi_levels(hi['rois'], 4)
    head = R['head'].DynaMaskHead(**gi.MASK_HEAD_CFG)
    hsd = gi.head_state()
    missing = head.load_state_dict({k[len('mask_head.'):]: v for k, v in hsd.items()}, strict=True)
    head.eval()
    with torch.no_grad():
        ips4, dps4 = head(ins, hi['feats'], hi['rois'], hi['labels'])
    g4 = {'levels': _np(lv), 'ins_feats': _np(ins)}
    for i in range(4):
        g4[f'ip{i}'] = _np(ips4[i])
        g4[f'dp{i}'] = _np(dps4[i])
    np.savez_compressed(os.path.join(HERE, 'g4_head.npz'), **g4)
    print('g4 ip3 mean', float(ips4[3].mean()), 'keys', len(hsd), missing)

    # -------------------------------------------- G5 inference boundary merge
    rh = roi_cls.__new__(roi_cls)
    nn.Module.__init__(rh)
    captured = {}

    class _MH:
        stage_num_classes = [80, 80, 80, 1]

        def get_seg_masks(self, mask_pred, *a, **k):
            captured['merged'] = mask_pred.clone()
            return [None] * len(mask_pred)
    rh.mask_head = _MH()
    rh.test_cfg = None
    mi = gi.merge_inputs()
    rh._mask_forward = lambda x, rois, labels: dict(stage_instance_preds=[t.clone() for t in mi['ips']],
                                                    stage_detail_preds=None)
    n = mi['ips'][0].shape[0]
    det_bboxes = torch.cat([gi.head_inputs()['rois'][:n, 1:], torch.ones(n, 1)], dim=1)
    rh.simple_test_mask(None, [dict(ori_shape=(200, 300, 3), scale_factor=1.0)], det_bboxes,
                        torch.zeros(n, dtype=torch.long), rescale=False)
    np.savez_compressed(os.path.join(HERE, 'g5_merge.npz'), merged=_np(captured['merged']))
    print('g5 merged mean', float(captured['merged'].mean()))

    # ------------------------------------------------------ G6 FCNMaskHead fwd
    g6 = {}
    for up in ('deconv', 'carafe', 'bilinear'):
        cfg = dict(gi.FCN_HEAD_CFG)
        if up == 'carafe':
            cfg['upsample_cfg'] = dict(type='carafe', scale_factor=2, up_kernel=5, up_group=1, encoder_kernel=3,
                                       encoder_dilation=1, compressed_channels=64)
        elif up == 'bilinear':
            cfg['upsample_cfg'] = dict(type='bilinear', scale_factor=2)
        R['builder'].LOSSES.module_dict.setdefault('CrossEntropyLoss', R['ce'].CrossEntropyLoss)
        fh = R['fcn'].FCNMaskHead(**cfg)
        fsd = gi.fcn_state(up)
        fh.load_state_dict({k[len('mask_head.'):]: v for k, v in fsd.items()}, strict=True)
        with torch.no_grad():
            g6[up] = _np(fh(gi.fcn_input()))
    np.savez_compressed(os.path.join(HERE, 'g6_fcn.npz'), **g6)
    print('g6', {k: float(v.mean()) for k, v in g6.items()})

    # ------------------------------- G7 head training slice: loss + param grads
    head.train()
    for p in head.parameters():
        p.grad = None
    fe = [f.clone().requires_grad_(True) for f in hi['feats']]
    ins_t = ext(fe[:4], hi['rois'])
    ips_t, dps_t = head(ins_t, fe, hi['rois'], hi['labels'])
    n = hi['rois'].shape[0]
    tg = gi.head_targets(n)
    mlab = gi.head_mask_labels(n).clone().requires_grad_(True)
    ltrain = head.loss_func(ips_t, dps_t, tg, mlab)['loss_masks']
    ltrain.backward()
    g7 = {'loss': _np(ltrain), 'grad_mask_labels': _np(mlab.grad)}
    named = dict(head.named_parameters())
    for k in gi.GRAD_KEYS:
        g7['grad.' + k] = _np(gi.grad_slice(named[k].grad))
    for i in range(4):
        g7[f'grad_feat{i}'] = _np(gi.feat_grad_slice(fe[i].grad)) if fe[i].grad is not None else np.zeros(1, np.float32)
    np.savez_compressed(os.path.join(HERE, 'g7_head_train.npz'), **g7)
    print('g7 loss', float(ltrain))


    # ------------------------------------------ G8 paste (pure-torch reference code)
    pi = gi.paste_inputs()
    g8 = {}
    for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
        cfgt = types.SimpleNamespace(mask_thr_binary=0.5)
        segs = head.get_seg_masks(pi['logits'].clone(), pi['det_bboxes'].clone(), torch.zeros(5, dtype=torch.long), cfgt,
                                  pi['ori_shape'], sf, rescale)
        g8[f'seg_rescale{int(rescale)}_sf{sf}'] = np.stack(segs).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g8_paste.npz'), **g8)
    print('g8', {k: int(v.sum()) for k, v in g8.items()})

    # ---------------- G9 mask targets: reference BitmapMasks + DynaMaskHead.get_targets
    stm = _load('mmdet.core.mask.structures', 'mmdet/core/mask/structures.py')
    ti = gi.target_inputs()
    gtm = [stm.BitmapMasks(t['masks'].numpy(), t['masks'].shape[1], t['masks'].shape[2]) for t in ti]
    tg = head.get_targets([t['boxes'] for t in ti], [t['inds'] for t in ti], gtm)
    g9 = {f't{i}': _np(t).astype(np.uint8) for i, t in enumerate(tg)}
    np.savez_compressed(os.path.join(HERE, 'g9_targets.npz'), **g9)
    print('g9', {k: (v.shape, int(v.sum())) for k, v in g9.items()})


if __name__ == '__main__':
    main()
