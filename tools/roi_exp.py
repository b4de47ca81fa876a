"""RoIAlign experiments: 14x14 over P2..P5 (512 RoIs, bench shape) and 56x56 on P2 (128 RoIs of one image = the kbench
shape; 2 x 128 = the training step's), each timed as 20 launches replayed as one HIP graph, for a list of knob settings
(DM_ROI_PERSIST / DM_ROI_WPC / DM_ROI_ORDER / DM_ROI_CT / DM_ROI_BAND_ORDER / DM_ROI_UNITS: the library reads them once; the
sweep calls dm_reload_env_knobs() after every change; {} = the defaults: 14x14 by the tile kernel, 16 channels per
workgroup in the XCD-aware order; 56x56 full-height bands with column blocks; DM_ROI_PERSIST=1 = round 4's plan +
persistent kernels).  The first setting of a sweep is the reference for the bit comparison.
usage: python tools/roi_exp.py [14|56|all]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import _lib, ops, synth
dev = torch.device('cuda')
which = sys.argv[1] if len(sys.argv) > 1 else 'all'


def graph_us(call, reps=20, iters=7):
    call(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            out = call()
    for _ in range(8):          # the clocks ramp over the first replays: an unwarmed first variant reads 5 % slow
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2], out


def sweep(name, call, settings, ref=None):
    base = None
    for env in settings:
        for k in ('DM_ROI_PERSIST', 'DM_ROI_SORT', 'DM_ROI_SORT_MIN', 'DM_ROI_NT14', 'DM_ROI_WORKSPACE', 'DM_ROI_WPC', 'DM_ROI_ORDER', 'DM_ROI_CT', 'DM_ROI_BAND_ORDER', 'DM_ROI_UNITS', 'DM_ROI_NT', 'DM_ROI_UNIT_WGS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        ops.ROI_WORKSPACE = (env.get('DM_ROI_SORT', '1') == '1' or env.get('DM_ROI_PERSIST', '0') == '1')
        ops.ROI_WORKSPACE_MIN = int(env.get('DM_ROI_SORT_MIN', '192'))
        _lib.lib().dm_reload_env_knobs()
        us, out = graph_us(call)
        if base is None:
            base = out.clone()
        same = torch.equal(out, base)
        print(f'{name} {env or "default"}: {us:.1f} us  bits {"same" if same else "DIFFER max %.3g" % float((out - base).abs().max())}', flush=True)


if which in ('14', 'all'):
    feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
    rois = synth.make_rois(1, 512, 800, 1333, seed=1).to(dev)
    call = lambda: ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    sets = [{}, {'DM_ROI_SORT': '0'}, {'DM_ROI_SORT': '0', 'DM_ROI_CT': '32'}, {'DM_ROI_NT14': '1'}, {'DM_ROI_ORDER': '0'}, {'DM_ROI_CT': '16'}, {'DM_ROI_PERSIST': '1'}, {}]
    sweep('roialign14 512 RoIs', call, sets)
    # the other shapes the kernel serves: 7x7 at 1000 proposals (bbox branch), odd counts, one level, 2 images
    props = synth.make_rois(1, 1000, 800, 1333, seed=31).to(dev)
    call7 = lambda: ops.roi_align(feats[:4], props, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    sweep('roialign7 1000 RoIs', call7, [{}, {'DM_ROI_CT': '32'}, {'DM_ROI_CT': '64'}, {'DM_ROI_CT': '128'}, {'DM_ROI_CT': '256'}, {'DM_ROI_PERSIST': '1'}])
    r129 = synth.make_rois(1, 129, 800, 1333, seed=5).to(dev)
    call129 = lambda: ops.roi_align(feats[:4], r129, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    sweep('roialign14 129 RoIs', call129, [{'DM_ROI_SORT': '0'}, {'DM_ROI_SORT_MIN': '1'}, {'DM_ROI_PERSIST': '1'}])
    # slivers and large grids: clipped boxes along the image border (grids up to 15), everything on ONE level
    g = torch.Generator().manual_seed(7)
    n = 64
    x1 = torch.rand(n, generator=g) * 1200
    sl = torch.stack([torch.zeros(n), x1, torch.zeros(n), x1 + 4 + torch.rand(n, generator=g) * 40, torch.full((n,), 799.0)], 1)
    sl[n // 2:, 1], sl[n // 2:, 3] = 0.0, 1332.0
    sl[n // 2:, 2] = torch.rand(n // 2, generator=g) * 700
    sl[n // 2:, 4] = sl[n // 2:, 2] + 4 + torch.rand(n // 2, generator=g) * 60
    sl = sl.to(dev)
    calls = lambda: ops.roi_align(feats[:4], sl, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    sweep('roialign14 64 slivers', calls, [{'DM_ROI_SORT': '0'}, {'DM_ROI_SORT_MIN': '1'}, {'DM_ROI_PERSIST': '1'}])
    call1 = lambda: ops.roi_align([feats[1]], sl, 14, [1 / 8])
    sweep('roialign14 64 slivers on P3 only', call1, [{'DM_ROI_SORT': '0'}, {'DM_ROI_SORT_MIN': '1'}, {'DM_ROI_PERSIST': '1'}])
if which in ('56', 'all'):
    for B, per in ((1, 128), (2, 128)):
        feats = [f.to(dev) for f in synth.make_fpn(B, 800, 1333, 256, seed=10)]
        rois = synth.make_rois(B, per, 800, 1333, seed=11).to(dev)
        call = lambda: ops.roi_align([feats[0]], rois, 56, [1 / 4])
        out_mb = B * per * 256 * 3136 * 4 / 1e6
        print(f'roialign56: output {out_mb:.0f} MB')
        sweep(f'roialign56 {B}x{per} RoIs', call, [{}, {'DM_ROI_BAND_ORDER': '3'}, {'DM_ROI_BAND_ORDER': '5'}, {'DM_ROI_BAND_ORDER': '1'}, {'DM_ROI_UNITS': '1'}, {}])
