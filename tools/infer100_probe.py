"""The 100-detection inference call (simple_test_mask_logits through the bucketed HIP graph) alone, a few replays, for
`rocprofv3 --kernel-trace` + tools/timeline.py: how much of its 2.2 ms is the GPU idle between dependent launches?
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/infer100_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
nd = int(os.environ.get('ND', '100'))
det, dl = rois[:nd, 1:].contiguous(), labels[:nd].contiguous()
with torch.no_grad():
    head.enable_inference_graphs(True)
    for _ in range(int(os.environ.get('REPS', '6'))):
        head.simple_test_mask_logits(feats, det, dl)
        torch.cuda.synchronize()
print('done')
