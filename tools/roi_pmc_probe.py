"""One RoIAlign 14x14 variant alone for rocprofv3 --pmc passes: env ROI_SORT=0/1 (RoIs host-sorted by level, y, x),
DM_ROI_* knobs as usual, PROBE_ITERS launches (back to back, maps + output stay in the Infinity Cache)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth
ops.ROI_WORKSPACE = (os.environ.get('DM_ROI_SORT', '1') == '1' or os.environ.get('DM_ROI_PERSIST', '0') == '1')
dev = torch.device('cuda')
it = int(os.environ.get('PROBE_ITERS', '10'))
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, 512, 800, 1333, seed=1)
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
if os.environ.get('ROI_SORT', '0') == '1':
    _, lv = ops.roi_align(feats[:4], rois.to(dev), 14, scales, return_levels=True)
    lv = lv.cpu()
    cy, cx = (rois[:, 2] + rois[:, 4]) / 2, (rois[:, 1] + rois[:, 3]) / 2
    perm = torch.tensor(sorted(range(512), key=lambda i: (int(lv[i]), float(cy[i]), float(cx[i]))))
    rois = rois[perm].contiguous()
rois = rois.to(dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.roi_align(feats[:4], rois, 14, scales)
e0.record()
for _ in range(it):
    ops.roi_align(feats[:4], rois, 14, scales)
e1.record()
torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) / it * 1e3:.1f} us per call (eager, with the wrapper)')
