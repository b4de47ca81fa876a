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


def _cin(srcs):
    return srcs.shape[1] if isinstance(srcs, torch.Tensor) else sum(t.shape[1] for t in srcs)


def _first(srcs):
    return srcs if isinstance(srcs, torch.Tensor) else srcs[0]


def flops(name, args, kw):
    """Arithmetic of the GEMM-shaped calls (the leaf ones: composite wrappers are listed without a figure)."""
    try:
        if name == 'conv2d':
            x = _first(args[0]); N, _, H, W = x.shape
            return 2.0 * N * H * W * _cin(args[0]) * args[3] * args[4] ** 2
        if name == 'deform_conv':
            N, C, H, W = args[0].shape
            return 2.0 * N * H * W * C * args[3] * 9
        if name == 'conv2d_wgrad':
            N, Co, H, W = args[0].shape
            return 2.0 * N * H * W * Co * _cin(args[1]) * args[2] ** 2
        if name == 'fc':
            return 2.0 * args[0].shape[0] * args[0].shape[1] * args[1].shape[0]
    except Exception:
        pass
    return 0.0


NESTED = {'deform_conv_backward', 'deform_conv_backward_data', 'deform_conv_backward_weight', 'deform_col2im_coord'}


def wrap(name, fn):
    def inner(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        log.append((name, ' '.join(describe(a) for a in args) + ''.join(f' {k}={describe(v)}' for k, v in kw.items()), dt,
                    flops(name, args, kw)))
        return out
    return inner


for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not name.startswith('_') and getattr(fn, '__module__', '') == ops.__name__:
        setattr(ops, name, wrap(name, fn))

step()
torch.cuda.synchronize()
tot = sum(d for n, _, d, _ in log if n not in NESTED)
gemm_ms = sum(d for n, _, d, f in log if f > 0)
gemm_fl = sum(f for _, _, _, f in log)
print(f'{len(log)} ops calls, {tot:.2f} ms synchronised (composite wrappers not counted twice); GEMM-shaped calls: '
      f'{gemm_fl / 1e12:.3f} TFLOP in {gemm_ms:.2f} ms = {gemm_fl / gemm_ms / 1e9:.1f} TFLOP/s solo')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for n, s, d, f in log:
    agg[(n, s)][0] += 1
    agg[(n, s)][1] += d
    agg[(n, s)][2] += f
for (n, s), (c, d, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = f'{f / d / 1e9:6.1f} TF/s' if f > 0 else ('   (composite)' if n in NESTED else '            ')
    print(f'{d:8.3f} ms  x{c:<3d} {tf} {n:28s} {s[:150]}')
