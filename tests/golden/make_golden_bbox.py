"""Golden vectors for the bbox branch from the REFERENCE's own modules:
Shared2FCBBoxHead.forward, BBoxHead.get_bboxes, DeltaXYWHBBoxCoder/delta2bbox and
multiclass_nms, loaded by file path with the stand-ins of make_golden.py.  mmcv's
``batched_nms`` is third-party code absent from the tree: the stand-in delegates to
``oracle/ref_model.batched_nms`` -- parity for the NMS itself therefore stays unpinned, what
this fixture pins is the reference's own arithmetic (FC stack, softmax, decode, clipping,
rescale, score threshold, class gathering, max_per_img truncation).

Run ONLY in the authoring container:  python tests/golden/make_golden_bbox.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import golden_inputs as gi  # noqa: E402
from oracle import ref_model  # noqa: E402


def main():
    mg.install_standins()
    core = mg._pkg('mmdet.core')
    core.auto_fp16 = mg._identity_decorator
    core.force_fp32 = mg._identity_decorator
    core.multi_apply = lambda f, *a, **k: tuple(map(list, zip(*map(f, *a))))
    bb = mg._pkg('mmdet.core.bbox')
    bld = mg._pkg('mmdet.core.bbox.builder')
    bld.BBOX_CODERS = mg.Registry('bbox_coder')
    bld.build_bbox_coder = lambda cfg, **kw: mg.build_from_cfg(cfg, bld.BBOX_CODERS, kw)
    core.build_bbox_coder = bld.build_bbox_coder
    mg._pkg('mmdet.core.bbox.coder')
    mg._load('mmdet.core.bbox.coder.base_bbox_coder', 'mmdet/core/bbox/coder/base_bbox_coder.py')
    coder = mg._load('mmdet.core.bbox.coder.delta_xywh_bbox_coder', 'mmdet/core/bbox/coder/delta_xywh_bbox_coder.py')
    nmsmod = mg._pkg('mmcv.ops.nms')
    nmsmod.batched_nms = lambda boxes, scores, idxs, cfg, class_agnostic=False: ref_model.batched_nms(boxes, scores, idxs, cfg)
    mg._pkg('mmdet.core.post_processing')
    pp = mg._load('mmdet.core.post_processing.bbox_nms', 'mmdet/core/post_processing/bbox_nms.py')
    core.multiclass_nms = pp.multiclass_nms
    builder = mg._load('mmdet.models.builder', 'mmdet/models/builder.py') if 'mmdet.models.builder' not in sys.modules \
        else sys.modules['mmdet.models.builder']
    builder.build_loss = lambda cfg: None
    losses = mg._pkg('mmdet.models.losses')
    losses.accuracy = lambda *a, **k: None
    mg._pkg('mmdet.models.roi_heads.bbox_heads')
    mg._load('mmdet.models.roi_heads.bbox_heads.bbox_head', 'mmdet/models/roi_heads/bbox_heads/bbox_head.py')
    cf = mg._load('mmdet.models.roi_heads.bbox_heads.convfc_bbox_head',
                  'mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py')

    head = cf.Shared2FCBBoxHead(**gi.BBOX_HEAD_CFG)
    sd = gi.bbox_head_state()
    head.load_state_dict({k[len('bbox_head.'):]: v for k, v in sd.items()}, strict=True)
    head.eval()
    x, rois = gi.bbox_inputs()
    out = {}
    with torch.no_grad():
        cls_score, bbox_pred = head(x)
        out['cls_score'], out['bbox_pred'] = cls_score.numpy(), bbox_pred.numpy()
        b0, s0 = head.get_bboxes(rois, cls_score, bbox_pred, gi.BBOX_IMG_SHAPE, 1.0, rescale=False, cfg=None)
        out['bboxes'], out['scores'] = b0.numpy(), s0.numpy()
        sf = np.array([1.25, 1.6, 1.25, 1.6], dtype=np.float32)
        b1, _ = head.get_bboxes(rois, cls_score, bbox_pred, gi.BBOX_IMG_SHAPE, sf, rescale=True, cfg=None)
        out['bboxes_rescaled'] = b1.numpy()
        from types import SimpleNamespace
        cfg = SimpleNamespace(**gi.RCNN_TEST_CFG)
        d, lab = head.get_bboxes(rois, cls_score, bbox_pred, gi.BBOX_IMG_SHAPE, 1.0, rescale=False, cfg=cfg)
        out['det_bboxes'], out['det_labels'] = d.numpy(), lab.numpy()
        # the docstring example of delta2bbox (delta_xywh_bbox_coder.py:146-160)
        r = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
        dl = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
        out['doc_example'] = coder.delta2bbox(r, dl, max_shape=(32, 32)).numpy()
    np.savez_compressed(os.path.join(HERE, 'g10_bbox.npz'), **out)
    print('g10_bbox:', {k: v.shape for k, v in out.items()}, 'kept', len(out['det_labels']))


if __name__ == '__main__':
    main()
