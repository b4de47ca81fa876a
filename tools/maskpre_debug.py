import os, sys, time, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from dynamask_amd import synth, ops, roi_head
from dynamask_amd import train_path
torch.manual_seed(0)
N = 256
feats = synth.make_fpn(2, 800, 1333, 256, seed=10)
rois = synth.make_rois(2, 128, 800, 1333, seed=11)
x = ops.roi_align([feats[0].cuda()], rois.cuda(), 56, [1 / 4]).cpu()
print('x: max', x.abs().max().item(), 'exact zeros', (x == 0).float().mean().item())
sd = synth.init_mask_pre_state(seed=6)
mp = roi_head.MaskPre()
mp.load_state_dict({k[len('mask_predictor.'):]: v for k, v in sd.items()}, strict=True)
mp = mp.cuda().train()
rec = {}
orig = ops.conv2d_wgrad
def spy(dy, srcs, ks, dw=None):
    out = orig(dy, srcs, ks, dw)
    if ks == 3:
        rec['dy'], rec['x'], rec['dw'] = dy.clone(), (srcs if isinstance(srcs, torch.Tensor) else srcs[0]).clone(), out.clone()
    return out
ops.conv2d_wgrad = spy
gw = torch.randn(N, 4)
logits = train_path.MaskPreFn.apply(mp, x.cuda(), *list(mp.parameters()))
(logits * gw.cuda()).sum().backward()
torch.cuda.synchronize()
# fp64 reference of the whole MaskPre
p = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in mp.state_dict().items() if v.is_floating_point() and 'running' not in k}
xd = x.double()
y1 = F.conv2d(xd, p['conv1.weight'], p['conv1.bias'])
b1 = F.batch_norm(y1, None, None, p['bn1.weight'], p['bn1.bias'], True, 0.1, 1e-5)
p1 = F.max_pool2d(F.relu(b1), 3, 2, 1); p1.retain_grad()
y2 = F.conv2d(p1, p['conv2.weight'], p['conv2.bias'], padding=1); y2.retain_grad()
b2 = F.batch_norm(y2, None, None, p['bn2.weight'], p['bn2.bias'], True, 0.1, 1e-5)
p2 = F.max_pool2d(F.relu(b2), 3, 2, 1)
h = F.relu(F.linear(p2.reshape(N, 3136), p['fc1.weight'], p['fc1.bias']))
lg = F.linear(h, p['fc2.weight'], p['fc2.bias'])
(lg * gw.double()).sum().backward()
print('logits err', (logits.detach().cpu().double() - lg.detach()).abs().max().item())
print('p1 err', (rec['x'].cpu().double() - p1.detach()).abs().max().item(), 'max', p1.abs().max().item())
print('g_y2 err', (rec['dy'].cpu().double() - y2.grad).abs().max().item(), 'max', y2.grad.abs().max().item())
ref_own = torch.nn.grad.conv2d_weight(rec['x'].double(), (16, 128, 3, 3), rec['dy'].double(), padding=1)
print('wgrad vs own inputs', (rec['dw'].double() - ref_own).abs().max().item(), 'vs f64 graph', (rec['dw'].cpu().double() - p['conv2.weight'].grad).abs().max().item(),
      'max', p['conv2.weight'].grad.abs().max().item())
d = (rec['dy'].cpu().double() - y2.grad).abs()
idx = (d > 10 * d.median() + 1e-9).nonzero()
print('outlier count', len(idx), 'of', d.numel(), 'first', idx[:5].tolist(), 'median', d.median().item())
for k in ('conv1.weight', 'bn1.weight', 'bn1.bias', 'conv2.weight', 'bn2.weight', 'fc1.weight'):
    g = dict(mp.named_parameters())[k].grad.cpu().double()
    print(k, (g - p[k].grad).abs().max().item(), p[k].grad.abs().max().item())
print('---- localise')
gy_h = rec['dy'].cpu().double(); gy_r = y2.grad
d = (gy_h - gy_r).abs()
top = torch.topk(d.flatten(), 6).indices
z2 = F.relu(b2.detach())
for t in top.tolist():
    n, c, yy, xx = (t // (16 * 784)), (t // 784) % 16, (t // 28) % 28, t % 28
    print('loc', (n, c, yy, xx), 'hip', gy_h[n, c, yy, xx].item(), 'ref', gy_r[n, c, yy, xx].item(), 'z', z2[n, c, yy, xx].item())
    ys, xs = slice(max(yy - 2, 0), yy + 3), slice(max(xx - 2, 0), xx + 3)
    print(' z nbhd\n', z2[n, c, ys, xs].float())
    print(' hip g nbhd\n', gy_h[n, c, ys, xs].float())
    print(' ref g nbhd\n', gy_r[n, c, ys, xs].float())
print('per-channel max err', d.amax((0, 2, 3)).tolist())
print('per-roi err top', torch.topk(d.amax((1, 2, 3)), 8))
print('var2', v2 if 'v2' in dir() else None)
