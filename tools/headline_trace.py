"""Runs only the headline step of bench.py (eager launches) so that
`rocprofv3 --kernel-trace --stats -- python3 tools/headline_trace.py` lists exactly the
kernels of one 512-RoI pass of the fixed-28x28 mask path and nothing else."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
head, sd = bench.build_head(dev)
feats_c, rois_c, labels_c = bench.make_inputs(0, dev)
feats = [f.to(dev) for f in feats_c]
rois, labels = rois_c.to(dev), labels_c.to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
last = None if (len(sys.argv) > 2 and sys.argv[2] == 'full') else 1      # 'full': all stages to 112x112
with torch.no_grad():
    for _ in range(n):
        head._mask_forward(feats, rois, labels, last_stage=last)
torch.cuda.synchronize()
print('done', n)
