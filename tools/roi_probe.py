"""RoIAlign 14x14 timing probe at the bench shape (512 RoIs, 1333x800 FPN)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth
dev = torch.device('cuda')
N = int(os.environ.get('RP_N', 512))
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, N, 800, 1333, seed=1)
if os.environ.get('RP_SORT'):
    import math
    def key(r):
        x1, y1, x2, y2 = r[1:].tolist()
        s = math.sqrt(max((x2 - x1) * (y2 - y1), 1e-6))
        lvl = min(3, max(0, int(math.floor(math.log2(s / 56 + 1e-6)))))
        cx, cy = int((x1 + x2) / 2) >> 4, int((y1 + y2) / 2) >> 4
        m = 0
        for b in range(8):
            m |= ((cx >> b) & 1) << (2 * b) | ((cy >> b) & 1) << (2 * b + 1)
        return (lvl, m)
    order = sorted(range(N), key=lambda i: key(rois[i]))
    if os.environ.get('RP_SORT') == '8':      # deal the sorted list into 8 contiguous groups, interleaved: RoI j -> group j % 8
        g = [order[i * (N // 8):(i + 1) * (N // 8)] for i in range(8)]
        order = [g[j % 8][j // 8] for j in range(N)]
    rois = rois[order].contiguous()
rois = rois.to(dev)
def t(fn, iters=30, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
call = lambda: ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
call(); torch.cuda.synchronize()
if os.environ.get('RP_NOGRAPH'):
    for _ in range(10): call()
    torch.cuda.synchronize(); print('done'); sys.exit(0)
# the Python wrapper costs more than the kernel: time 20 launches replayed as one HIP graph
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        out = call()
ms = t(g.replay, iters=5, warmup=2) / 20
print(f'ABL={os.environ.get("DM_ROI_ABL","0")} CT={os.environ.get("DM_ROI_CT","64")}: {ms*1e3:.1f} us')
