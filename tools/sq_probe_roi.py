"""The two RoIAlign forward kernels alone for rocprofv3 --pmc SQ_* passes (see sq_pmc.sh): 14x14 over P2..P5 (512 RoIs) and
56x56 on P2 (128 RoIs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth
dev = torch.device('cuda')
it = int(os.environ.get('PROBE_ITERS', '4'))
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=1)]
rois = synth.make_rois(1, 512, 800, 1333, seed=2).to(dev)
for _ in range(it):
    ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    ops.roi_align([feats[0]], rois[:128].contiguous(), 56, [1 / 4])
torch.cuda.synchronize()
print('done')
