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
for ns in (1, 2, 3, 4, 6):
    head.num_streams = ns
    with torch.no_grad():
        a = t(lambda: head._mask_forward(feats, rois, labels, last_stage=1))
        b = t(lambda: head._mask_forward(feats, rois, labels), it=8)
    print(f'streams {ns}: exit28 {a:.3f} ms   full {b:.3f} ms')
