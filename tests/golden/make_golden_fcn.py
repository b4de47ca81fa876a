"""Golden vectors for the callers either side of the REFERENCE's ``FCNMaskHead``
(mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:128-135 ``get_targets`` -> core/mask/mask_target.py, and :151-237
``get_seg_masks`` + ``_do_paste_mask``): the reference's own classes, loaded by path with the stand-ins of
make_golden.py (pure torch + numpy code; the only mmcv symbol behind it is the RoIAlign of ``BitmapMasks.crop_and_resize``,
which delegates to oracle/ref_ops.py as everywhere).  Run ONLY in the authoring container:

    python tests/golden/make_golden_fcn.py          ->  tests/golden/g14_fcn_callers.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_inputs as gi  # noqa: E402
import make_golden as mg  # noqa: E402


def main():
    R = mg.load_reference()
    mt = mg._load('mmdet.core.mask.mask_target', 'mmdet/core/mask/mask_target.py')
    R['fcn'].mask_target = mt.mask_target                      # the name fcn_mask_head.py imports from mmdet.core
    stm = mg._load('mmdet.core.mask.structures', 'mmdet/core/mask/structures.py')
    R['builder'].LOSSES.module_dict.setdefault('CrossEntropyLoss', R['ce'].CrossEntropyLoss)
    out = {}
    pi = gi.fcn_paste_inputs()
    for agnostic in (False, True):
        cfg = dict(gi.FCN_HEAD_CFG)
        cfg['class_agnostic'] = agnostic
        fh = R['fcn'].FCNMaskHead(**cfg)
        logits = pi['logits'] if not agnostic else pi['logits'][torch.arange(7), pi['det_labels']][:, None]
        for rescale, sf in ((False, 1.0), (True, 1.0), (True, 1.25)):
            cfgt = types.SimpleNamespace(mask_thr_binary=0.5)
            segs = fh.get_seg_masks(logits.clone(), pi['det_bboxes'].clone(), pi['det_labels'].clone(), cfgt, pi['ori_shape'], sf, rescale)
            assert len(segs) == 80
            # per class, in detection order -> flattened back to detection order with the class counts beside it
            counts = np.array([len(c) for c in segs], np.int32)
            flat = [m for c in segs for m in c]
            out[f'seg_ag{int(agnostic)}_rescale{int(rescale)}_sf{sf}'] = np.stack(flat).astype(np.uint8)
            out[f'counts_ag{int(agnostic)}_rescale{int(rescale)}_sf{sf}'] = counts
        # multi-scale testing hands over an ndarray of probabilities: no sigmoid (fcn_mask_head.py:168-171)
        probs = logits.sigmoid().numpy()
        segs = fh.get_seg_masks(probs, pi['det_bboxes'].clone(), pi['det_labels'].clone(), types.SimpleNamespace(mask_thr_binary=0.5),
                                pi['ori_shape'], 1.0, True)
        out[f'seg_ag{int(agnostic)}_ndarray'] = np.stack([m for c in segs for m in c]).astype(np.uint8)
    # get_targets: SamplingResult-like holders (pos_bboxes, pos_assigned_gt_inds), BitmapMasks, mask_size 28 and 14
    fh = R['fcn'].FCNMaskHead(**gi.FCN_HEAD_CFG)
    ti = gi.target_inputs()
    res = [types.SimpleNamespace(pos_bboxes=t['boxes'].clone(), pos_assigned_gt_inds=t['inds'].clone()) for t in ti]
    gtm = [stm.BitmapMasks(t['masks'].numpy(), t['masks'].shape[1], t['masks'].shape[2]) for t in ti]
    for size in (28, 14):
        tg = fh.get_targets(res, gtm, types.SimpleNamespace(mask_size=size))
        assert tg.dtype == torch.float32
        out[f'targets{size}'] = tg.numpy().astype(np.uint8)
    path = os.path.join(HERE, 'g14_fcn_callers.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: (v.shape, int(v.sum())) for k, v in out.items()})


if __name__ == '__main__':
    main()
