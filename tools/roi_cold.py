"""RoIAlign 14x14 (P2..P5 -> [512, 256, 14, 14], the headline's RoI batch) timed the three ways bench.py reports it:
from cold caches (a 512 MiB sweep before every call), 20 calls back to back in one graph, one call with one predecessor.
Prints a checksum of the output (sha1 of the bytes) so that a kernel change can be checked for identical bits.
ROI_N / ROI_SORT_OFF / ROI_P env: number of RoIs, the unordered kernel, output size."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import ops, synth

dev = torch.device('cuda')
N = int(os.environ.get('ROI_N', 512))
P = int(os.environ.get('ROI_P', 14))
feats = [f.to(dev) for f in synth.make_fpn(1, bench.IMG_H, bench.IMG_W, 256, seed=0)][:4]
rois = synth.make_rois(1, N, bench.IMG_H, bench.IMG_W, seed=1).to(dev)
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
if os.environ.get('ROI_SORT_OFF'):
    ops.ROI_WORKSPACE = False
call = lambda: ops.roi_align(feats, rois, P, scales)      # noqa: E731
out, lv = ops.roi_align(feats, rois, P, scales, return_levels=True)
torch.cuda.synchronize()
nbytes = bench.roialign_algorithmic_bytes(rois.cpu(), lv.cpu().long(), [tuple(f.shape[2:]) for f in feats], P=P)
print(f'{N} RoIs -> {tuple(out.shape)}; algorithmic bytes {nbytes / 1e6:.1f} MB; sha1 {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]}')
for rep in range(int(os.environ.get('ROI_REPS', 2))):
    cold = bench.time_kernel_cold(call)
    clean = bench.time_kernel_cold(call, dirty=False)
    warm = bench.time_kernel_graphed(call)
    single = bench.time_kernel_single_in_graph(call)
    f = lambda ms: nbytes / (ms * 1e-3) / 1e9 / 8000.0      # noqa: E731
    print(f'cold {cold * 1e3:6.1f} us ({f(cold):.3f} of 8 TB/s)   cold, evicted by reads {clean * 1e3:6.1f} us ({f(clean):.3f})   warm x20 {warm * 1e3:6.1f} us ({f(warm):.3f})   single {single * 1e3:6.1f} us ({f(single):.3f})', flush=True)
