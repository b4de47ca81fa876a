"""How many RoI chunks on how many streams for the <= 100-detection inference call (graph-replayed full head to 112 x 112
+ boundary merge)?  Kernels at these sizes are bound by the K-loop latency of a lone workgroup per CU, not by
throughput: more, smaller launches side by side fill the chip."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench
from kbench import t
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return t(g.replay, iters=30)
for n in (16, 32, 64, 100):
    r, l = rois[:n].contiguous(), labels[:n].contiguous()
    row = f'N={n:4d}:'
    for ns in (1, 2, 3, 4):
        head.num_streams = ns
        head.stream_split_min = 1 if ns > 1 else 10 ** 9
        with torch.no_grad():
            v = graphed(lambda: head.merge_stage_preds(head._mask_forward(feats, r, l)['stage_instance_preds']))
        row += f'  {ns} stream(s) {v:.3f} ms'
    print(row, flush=True)
