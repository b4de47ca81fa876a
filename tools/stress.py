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

# ---- the fused inference launches of round 6 at odd detection counts (both sides of the two-stream threshold, padded
# graph buckets): bits of the unfused launch sequence (split-K off: one order of sums), and the graph equals eager
from dynamask_amd import ops
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=3)]
ops.CONV_SPLITK[0] = False
for N in (1, 2, 7, 16, 17, 63, 64, 65, 99, 100, 101, 200):
    rois = synth.make_rois(1, N, 800, 1333, seed=100 + N).to(dev)
    boxes, labels = rois[:, 1:].contiguous(), synth.make_labels(N, seed=N + 5).to(dev)
    outs = []
    with torch.no_grad():
        for fused in (False, True):
            mask_heads.FUSED_STAGE_HEAD[0] = mask_heads.GROUPED_SEMANTIC_MAPS[0] = mask_heads.FUSED_DCN_TOUT[0] = fused
            roi_head.FUSED_MERGE_TAIL[0] = fused
            outs.append(m.simple_test_mask_logits(feats, boxes, labels).clone())
        m.enable_inference_graphs(True)
        outs.append(m.simple_test_mask_logits(feats, boxes, labels).clone())
        m.enable_inference_graphs(False)
    ok = torch.isfinite(outs[0]).all().item() and torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    print(f'fused inference N={N}: {"ok" if ok else "FAIL"} (unfused == fused: {torch.equal(outs[0], outs[1])}, graph == eager: {torch.equal(outs[1], outs[2])})', flush=True)
