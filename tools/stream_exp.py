"""Number of HIP streams over RoI chunks (and uneven splits) vs step time, eager and as a replayed HIP graph."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
dev = torch.device('cuda')
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]; rois = rois_c.to(dev); labels = labels_c.to(dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return t(g.replay, it=40)
cases = [(1, None), (2, None), (2, (0.35, 1.0)), (2, (0.4, 1.0)), (2, (0.45, 1.0)), (2, (0.652, 1.0)), (3, None), (3, (0.25, 0.6, 1.0)), (4, None)]
for ns, split in cases:
    head.num_streams = ns
    head.stream_split = split
    with torch.no_grad():
        a = t(lambda: head._mask_forward(feats, rois, labels, last_stage=1))
        ag = graphed(lambda: head._mask_forward(feats, rois, labels, last_stage=1))
        b = t(lambda: head._mask_forward(feats, rois, labels), it=8)
    print(f'streams {ns} split {split}: exit28 eager {a:.3f} ms  graph {ag:.3f} ms   full eager {b:.3f} ms', flush=True)
