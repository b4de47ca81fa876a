"""Per-workgroup phase timeline of the persistent RoIAlign kernel (csrc/roi_align.hip built with -DDM_ROI_STAMPS).
usage: python tools/roi_stamps_persist.py [CT[:WPC] ...]"""
import ctypes, os, subprocess, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import synth
so = '/tmp/libroi_stamps.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-shared', '-DDM_ROI_STAMPS',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'dynamask_amd', 'csrc'),
                       os.path.join(ROOT, 'dynamask_amd', 'csrc', 'roi_align.hip'), '-o', so])
L = ctypes.CDLL(so)
dev = torch.device('cuda')
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)][:4]
rois = synth.make_rois(1, 512, 800, 1333, seed=1).to(dev)
N = 512
out = torch.empty(N, 256, 14, 14, device=dev)
H = (ctypes.c_int * 4)(*[f.shape[2] for f in feats]); W = (ctypes.c_int * 4)(*[f.shape[3] for f in feats])
sc = (ctypes.c_float * 4)(1 / 4, 1 / 8, 1 / 16, 1 / 32)
fp = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
vp, ci, cf, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
L.dm_roi_align_fwd_ws.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, ci, ci, ci, cf, vp, vp, vp, ll, vp]
L.dm_roi_align_workspace_bytes.restype = ll
L.dm_roi_stamp_buffer.argtypes = [vp]
wsb = L.dm_roi_align_workspace_bytes(N, 14)
ws = torch.empty(wsb // 4, dtype=torch.int32, device=dev)
for spec in sys.argv[1:] or ['32', '64']:
    ct, wpc = (spec.split(':') + ['3'])[:2]
    os.environ['DM_ROI_PERSIST'] = '1'
    os.environ['DM_ROI_CT'] = ct
    os.environ['DM_ROI_WPC'] = wpc
    L.dm_reload_env_knobs()
    nwg = int(wpc) * 256
    stamps = torch.zeros(nwg, 32, dtype=torch.int64, device=dev)
    call = lambda: L.dm_roi_align_fwd_ws(fp, H, W, sc, 4, 1, 256, vp(rois.data_ptr()), N, 14, 0, 56.0, vp(out.data_ptr()), None, vp(ws.data_ptr()), wsb, None)
    L.dm_roi_stamp_buffer(None)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print(f'CT={ct} WPC={wpc}: 20 back-to-back calls (plan + persistent kernel) without stamps: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us each')
    # the plan kernel alone: a call with N RoIs but C = 0 is refused, so time the pair against a pair with 4 channels
    out4 = torch.empty(N, 4, 14, 14, device=dev)
    fp4 = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in feats])
    call4 = lambda: L.dm_roi_align_fwd_ws(fp4, H, W, sc, 4, 1, 4, vp(rois.data_ptr()), N, 14, 0, 56.0, vp(out4.data_ptr()), None, vp(ws.data_ptr()), wsb, None)
    os.environ['DM_ROI_CT'] = '4'; L.dm_reload_env_knobs()
    for _ in range(3):
        call4()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call4()
    e1.record(); torch.cuda.synchronize()
    print(f'  plan kernel + a 4-channel extraction (B = 1 treats the maps as 4-channel ones: timing only): {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per pair')
    os.environ['DM_ROI_CT'] = ct; L.dm_reload_env_knobs()
    L.dm_roi_stamp_buffer(vp(stamps.data_ptr()))
    rc = call(); torch.cuda.synchronize()
    assert rc == 0
    s = stamps.cpu().numpy().astype(np.float64)
    ok = s[:, 31] > 0
    t0 = s[ok, 0].min()
    print(f'  {ok.sum()} workgroups; first start -> last end {s[ok, 31].max() - t0:.0f} ticks; starts within {s[ok, 0].max() - t0:.0f}')
    print(f'  end of phase 1 (pipeline): median {np.median(s[ok, 30] - t0):.0f} p10 {np.percentile(s[ok, 30] - t0, 10):.0f} p90 {np.percentile(s[ok, 30] - t0, 90):.0f} max {(s[ok, 30] - t0).max():.0f}')
    print(f'  end of workgroup:          median {np.median(s[ok, 31] - t0):.0f} p10 {np.percentile(s[ok, 31] - t0, 10):.0f} p90 {np.percentile(s[ok, 31] - t0, 90):.0f} max {(s[ok, 31] - t0).max():.0f}')
    life = s[ok, 31] - s[ok, 0]
    print(f'  workgroup lifetime (own clock): mean {life.mean():.0f} median {np.median(life):.0f} p10 {np.percentile(life, 10):.0f} p90 {np.percentile(life, 90):.0f} min {life.min():.0f} max {life.max():.0f} ticks')
    ph2 = s[ok, 31] - s[ok, 30]
    print(f'  phase 2 (special units): mean {ph2.mean():.0f}, {(ph2 > 2000).sum()} workgroups spend more than 2000 ticks there, max {ph2.max():.0f}')
    def seg(a, b, name):
        m = ok & (s[:, a] > 0) & (s[:, b] > 0)
        if m.sum() == 0:
            return
        d = s[m, b] - s[m, a]
        print(f'  {name:52s} n={m.sum():5d} mean {d.mean():7.0f} median {np.median(d):7.0f} p90 {np.percentile(d, 90):7.0f}')
    seg(0, 1, 'start -> first header loaded')
    seg(1, 2, 'first setup (offsets)')
    seg(2, 3, 'first fetch + table + commit + barrier')
    for it in range(6):
        st = 4 + 4 * it
        seg(st - 1, st, f'round {it}: stores issued')
        seg(st, st + 1, f'round {it}: next batch set up + fetch issued')
        seg(st + 1, st + 2, f'round {it}: sampled')
        seg(st + 2, st + 3, f'round {it}: barrier, commit (fetch landed), barrier')
