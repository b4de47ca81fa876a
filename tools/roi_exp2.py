"""RoIAlign 14x14 experiments, round 4: what RoI ORDER and cache state do to the launch.
  * RoIs as given / sorted on the HOST by (level, y, x) / by a Morton key of the box centre inside the level
  * back-to-back replays (maps + output = 194 MB stay in the 256 MB Infinity Cache) vs replays with a 1 GiB fill in between
usage: python tools/roi_exp2.py"""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import _lib, ops, synth
dev = torch.device('cuda')


def graph_us(call, reps=20, iters=7, between=None):
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            out = call()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2], out


def cold_us(call, junk, iters=9):
    """one launch between events, a 1 GiB fill before each (the maps and the output leave the caches)"""
    ts = []
    for _ in range(iters):
        junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def setenv(env):
    for k in ('DM_ROI_PERSIST', 'DM_ROI_SORT', 'DM_ROI_WORKSPACE', 'DM_ROI_ORDER', 'DM_ROI_CT', 'DM_ROI_NT14'):
        os.environ.pop(k, None)
    os.environ.update(env)
    ops.ROI_WORKSPACE = (env.get('DM_ROI_SORT', '1') == '1' or env.get('DM_ROI_PERSIST', '0') == '1')
    _lib.lib().dm_reload_env_knobs()


feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, 512, 800, 1333, seed=1)
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
_, lv = ops.roi_align(feats[:4], rois.to(dev), 14, scales, return_levels=True)
lv = lv.cpu().long()
cx, cy = (rois[:, 1] + rois[:, 3]) / 2, (rois[:, 2] + rois[:, 4]) / 2


def morton(x, y):
    k = 0
    for b in range(10):
        k |= ((int(x) >> b) & 1) << (2 * b) | ((int(y) >> b) & 1) << (2 * b + 1)
    return k


orders = {
    'as given': torch.arange(512),
    'level, y, x': torch.tensor(sorted(range(512), key=lambda i: (int(lv[i]), float(cy[i]), float(cx[i])))),
    'level, 64-px rows, x': torch.tensor(sorted(range(512), key=lambda i: (int(lv[i]), int(cy[i]) // (64 << int(lv[i])), float(cx[i])))),
    'level, morton': torch.tensor(sorted(range(512), key=lambda i: (int(lv[i]), morton(cx[i] / 8, cy[i] / 8)))),
    'morton only': torch.tensor(sorted(range(512), key=lambda i: morton(cx[i] / 8, cy[i] / 8))),
}
junk = torch.empty(1 << 28, device=dev)
for name, perm in orders.items():
    r = rois[perm].contiguous().to(dev)
    call = lambda: ops.roi_align(feats[:4], r, 14, scales)
    for env in ({}, {'DM_ROI_NT14': '1'}, {'DM_ROI_CT': '32'}, {'DM_ROI_CT': '32', 'DM_ROI_NT14': '1'}, {'DM_ROI_ORDER': '0'}, {'DM_ROI_PERSIST': '1'}):
        setenv(env)
        us, _ = graph_us(call)
        cold = cold_us(call, junk)
        print(f'{name:22s} {str(env):70s} warm {us:6.1f} us   cold {cold:6.1f} us', flush=True)
