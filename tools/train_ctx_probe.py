"""Why does bench.py's training leg run slower than tools/train_probe.py on the same box?  Runs bench.train_step_bench
after each of several preludes.   python tools/train_ctx_probe.py [none|eager|graph]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
os.environ['DM_BENCH_NO_RCCL'] = '1'
mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
head, sd = bench.build_head(dev)
if mode != 'none':
    feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
    feats = [f.to(dev) for f in feats_c]
    rois, labels = rois_c.to(dev), labels_c.to(dev)

    def step():
        with torch.no_grad():
            return head._mask_forward(feats, rois, labels, last_stage=1)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    if mode == 'graph':
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
r = bench.train_step_bench(head, dev, 0, 1)
print(mode, 'train step ms', r[0])
