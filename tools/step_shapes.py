"""Per-call listing of one training step: every `dynamask_amd.ops` call with its tensor shapes and its
synchronised duration (so the sum is larger than the pipelined step; the ranking is what matters).
  python tools/step_shapes.py > gpurun_out/step_shapes.txt"""
import os, sys, time, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import synth, ops
from dynamask_amd.dist import FlatParamGroup, mask_path_parameters

dev = torch.device('cuda')
head, sd = bench.build_head(dev)
B, per = 2, 128
feats = [f.to(dev) for f in synth.make_fpn(B, bench.IMG_H, bench.IMG_W, 256, seed=10)]
rois = synth.make_rois(B, per, bench.IMG_H, bench.IMG_W, seed=11).to(dev)
labels = synth.make_labels(B * per, seed=12).to(dev)
targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)
head.train()
grp = FlatParamGroup(mask_path_parameters(head))


def step():
    grp.zero_grad()
    res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
    res['loss_mask']['loss_masks'].backward()
    grp.all_reduce_async()
    grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)


for _ in range(3):
    step()
torch.cuda.synchronize()

log = []


def describe(a):
    if isinstance(a, torch.Tensor):
        return 'x'.join(map(str, a.shape)) or 'scalar'
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return '[' + ','.join(describe(t) for t in a) + ']'
    if isinstance(a, (int, float, bool, str, type(None))):
        return repr(a)
    return type(a).__name__


def wrap(name, fn):
    def inner(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        log.append((name, ' '.join(describe(a) for a in args) + ''.join(f' {k}={describe(v)}' for k, v in kw.items()), dt))
        return out
    return inner


for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not name.startswith('_') and getattr(fn, '__module__', '') == ops.__name__:
        setattr(ops, name, wrap(name, fn))

step()
torch.cuda.synchronize()
tot = sum(d for _, _, d in log)
print(f'{len(log)} ops calls, {tot:.2f} ms synchronised')
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, d in log:
    agg[(n, s)][0] += 1
    agg[(n, s)][1] += d
for (n, s), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{d:8.3f} ms  x{c:<3d} {n:28s} {s[:150]}')
