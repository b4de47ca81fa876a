"""Shape stress: the whole inference surface at odd RoI counts / image sizes, checked for
finiteness and (small sizes) against the oracle.  Not a benchmark."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_inputs as gi
from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors, bbox_heads, synth  # noqa: F401
from dynamask_amd.registry import ConfigDict
dev = torch.device('cuda')
cfg = dict(type='DynaMaskRoIHead',
           bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
           bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
           mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
           mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG), test_cfg=ConfigDict(**gi.RCNN_TEST_CFG))
m = registry.build_head(cfg)
m.load_state_dict({**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6),
                   **synth.init_bbox_head_state(seed=8)}, strict=True)
m = m.to(dev).eval()
for (H, W) in ((800, 1333), (608, 1024), (333, 500), (1024, 2048)):
    feats = [f.to(dev) for f in synth.make_fpn(1, H, W, 256, seed=H)]
    for N in (1, 7, 100, 129, 513, 1000):
        rois = synth.make_rois(1, N, H, W, seed=N).to(dev)
        labels = synth.make_labels(N, seed=N + 1).to(dev)
        with torch.no_grad():
            r = m._mask_forward(feats, rois, labels)
            ok = all(torch.isfinite(t).all().item() for t in r['stage_instance_preds'])
            d = m.dynamic_mask_logits(feats, rois[:, 1:].contiguous(), labels)
            ok &= all(torch.isfinite(t).all().item() for t in d['preds'])
            if N <= 513:
                metas = [dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=1.0)]
                bb, sg = m.simple_test(feats, [rois[:, 1:].contiguous()], metas, rescale=False, encode=True)
                ok &= sum(len(b) for b in bb) == sum(len(s) for s in sg)
        torch.cuda.synchronize()
        print(f'{H}x{W} N={N}: {"ok" if ok else "FAIL"}', flush=True)
