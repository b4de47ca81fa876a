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
def one(last):
    with torch.no_grad():
        return head._mask_forward(feats, rois, labels, last_stage=last)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
halves = [(rois[:256].contiguous(), labels[:256].contiguous()), (rois[256:].contiguous(), labels[256:].contiguous())]
def two(last):
    cur = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for s, (r, l) in zip(streams, halves):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(head._mask_forward(feats, r, l, last_stage=last))
        for s in streams:
            cur.wait_stream(s)
    return outs
for last in (1, None):
    print('last_stage', last, 'one stream %.3f ms' % t(lambda: one(last)), 'two streams %.3f ms' % t(lambda: two(last)))
