"""Per-workgroup phase timeline of the RoIAlign tile kernel (debug build with DM_ROI_DBG)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamask_amd import ops, synth
dev = torch.device('cuda')
N = 512
CT = int(os.environ.get('DM_ROI_CT', 64))
feats = [f.to(dev) for f in synth.make_fpn(1, 800, 1333, 256, seed=0)]
rois = synth.make_rois(1, N, 800, 1333, seed=1).to(dev)
nblk = N * (256 // CT)
dbg = torch.zeros(nblk * 40, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
torch.cuda.synchronize()
os.environ['DM_ROI_DBG'] = str(dbg.data_ptr())
ops.roi_align(feats[:4], rois, 14, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk, 40)
n = d[:, 38]
t0 = d[:, 0].min()
start = (d[:, 0] - t0)
end = np.array([d[i, n[i] - 1] for i in range(nblk)]) - t0
life = end - (d[:, 0] - t0)
G = d[:, 39] >> 32; NQ = (d[:, 39] >> 16) & 0xffff; px = d[:, 39] & 0xffff
print('clock units; kernel span', end.max(), 'blocks', nblk)
print('block lifetime: mean %.0f  p50 %.0f  p90 %.0f  max %.0f' % (life.mean(), np.median(life), np.percentile(life, 90), life.max()))
setup = d[:, 1] - d[:, 0]; pro_issue = d[:, 2] - d[:, 1]; pro_wait = d[:, 3] - d[:, 2]
print('setup %.0f  prologue issue %.0f  prologue commit+sync %.0f' % (setup.mean(), pro_issue.mean(), pro_wait.mean()))
fi, sa, co, sy, last = [], [], [], [], []
for i in range(nblk):
    k = 3
    prev = d[i, 3]
    while k + 4 <= n[i] - 1:
        fi.append(d[i, k] - prev); sa.append(d[i, k + 1] - d[i, k]); co.append(d[i, k + 2] - d[i, k + 1]); sy.append(d[i, k + 3] - d[i, k + 2])
        prev = d[i, k + 3]; k += 4
    last.append(d[i, n[i] - 1] - prev)
print('per batch: fetch issue %.0f  sample %.0f  commit(wait+write) %.0f  barrier %.0f   (n=%d)' % (np.mean(fi), np.mean(sa), np.mean(co), np.mean(sy), len(fi)))
print('last sample %.0f' % np.mean(last))
for g in sorted(set(G.tolist())):
    m = G == g
    print('G=%d: blocks %d  life mean %.0f max %.0f  NQ mean %.1f px mean %.0f' % (g, m.sum(), life[m].mean(), life[m].max(), NQ[m].mean(), px[m].mean()))
order = np.argsort(start)
print('start times (every 256th block):', start[order][::256].tolist())
print('end-time percentiles:', [int(np.percentile(end, p)) for p in (50, 90, 99, 100)])
