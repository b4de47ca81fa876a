"""RoIAlign experiments: 14x14 over P2..P5 (512 RoIs, bench shape) and 56x56 on P2 (128 RoIs of one image = the kbench
shape; 2 x 128 = the training step's), each timed as 20 launches replayed as one HIP graph, for a list of knob settings
(DM_ROI_ORDER / DM_ROI_CT / DM_ROI_BAND_ORDER / DM_ROI_UNITS_NOW are read by the library at every call;
{} = the defaults: 14x14 in the XCD-aware order with 16 channels per workgroup, 56x56 full-height bands with column blocks).
usage: python tools/roi_exp.py [14|56|all]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth
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
        for k in ('DM_ROI_ORDER', 'DM_ROI_CT', 'DM_ROI_BAND_ORDER', 'DM_ROI_UNITS_NOW', 'DM_ROI_NT', 'DM_ROI_UNIT_WGS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        us, out = graph_us(call)
        if base is None:
            base = out.clone()
        same = torch.equal(out, base)
        print(f'{name} {env or "default"}: {us:.1f} us  bits {"same" if same else "DIFFER max %.3g" % float((out - base).abs().max())}', flush=True)


if which in ('14', 'all'):
    feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
    rois = synth.make_rois(1, 512, 800, 1333, seed=1).to(dev)
    call = lambda: ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    sweep('roialign14 512 RoIs', call, [{}, {'DM_ROI_ORDER': '0', 'DM_ROI_CT': '32'}, {'DM_ROI_ORDER': '0', 'DM_ROI_CT': '16'},
                                       {'DM_ROI_ORDER': '1', 'DM_ROI_CT': '32'}, {'DM_ROI_ORDER': '1', 'DM_ROI_CT': '8'}])
if which in ('56', 'all'):
    for B, per in ((1, 128), (2, 128)):
        feats = [f.to(dev) for f in synth.make_fpn(B, 800, 1333, 256, seed=10)]
        rois = synth.make_rois(B, per, 800, 1333, seed=11).to(dev)
        call = lambda: ops.roi_align([feats[0]], rois, 56, [1 / 4])
        out_mb = B * per * 256 * 3136 * 4 / 1e6
        print(f'roialign56: output {out_mb:.0f} MB')
        sweep(f'roialign56 {B}x{per} RoIs', call, [{}, {'DM_ROI_BAND_ORDER': '3'}, {'DM_ROI_BAND_ORDER': '5'}, {'DM_ROI_BAND_ORDER': '1'}, {'DM_ROI_UNITS_NOW': '1'}, {}])
