"""What each component costs THE STEP (not itself): the training step timed with one component at a time replaced by a
no-op that only allocates its outputs (results are wrong on purpose; nothing is checked).  The difference to the shipped
step is the component's marginal cost among everything that runs beside it on the four streams -- the number a faster
kernel for that component is bounded by.  Windows of 8 steps, one process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import ops, synth
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


def window(k=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        grp.zero_grad()
        res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
        res['loss_mask']['loss_masks'].backward()
        grp.all_reduce_async()
        grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def measure():
    window(3)
    return sorted(window() for _ in range(3))[1]


base = measure()
print(f'shipped step                                   {base:6.2f} ms', flush=True)


def col2im(colgrad, offset, x_shape, deform_groups, out=None):
    return out if out is not None else torch.empty(x_shape, device=colgrad.device)


def coord(colgrad, x, offset, deform_groups, out=None):
    return out if out is not None else torch.empty_like(offset)


def psb(grad_out, feat_shape, rois, spatial_scale, grad_feat=None):
    return grad_feat if grad_feat is not None else torch.empty(feat_shape, device=grad_out.device)


def wgrad(dy, srcs, ksize, dw=None, db=None, want_bias=False):
    srcs = [srcs] if isinstance(srcs, torch.Tensor) else srcs
    cin = sum(s.shape[1] for s in srcs)
    if dw is None:
        dw = torch.empty((dy.shape[1], cin, ksize, ksize), device=dy.device)
    if db is None and want_bias:
        db = torch.empty((dy.shape[1],), device=dy.device)
    return (dw, db) if db is not None else dw


def im2col(x, offset, deform_groups, out=None):
    return out if out is not None else torch.empty((x.shape[0], 9 * x.shape[1], x.shape[2], x.shape[3]), device=x.device)


def bnpool_bwd(x, mean, var, gamma, beta, grad_out, eps=1e-5):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def roi_bwd(grad_out, feat_shapes, rois, output_size, spatial_scales, sampling_ratio=0, finest_scale=56.0):
    return tuple(torch.empty(s, device=grad_out.device) for s in feat_shapes)


cases = [('deform_col2im', col2im, 'DCN col2im (three stages)'),
         ('deform_coord_grad', coord, 'DCN coordinate gradient (three stages)'),
         ('point_sample_backward', psb, 'point-sample adjoint'),
         ('conv2d_wgrad', wgrad, 'ALL weight-gradient GEMMs + slab reduces'),
         ('deform_im2col', im2col, 'deformable im2col (forward 28 / 56, backward 14)'),
         ('bn_relu_maxpool_backward', bnpool_bwd, "MaskPre's BatchNorm + pool backward"),
         ('roi_align_backward', roi_bwd, 'RoIAlign adjoints (MaskPre conv1 on the map)')]
for name, fn, what in cases:
    orig = getattr(ops, name)
    setattr(ops, name, fn)
    try:
        t = measure()
        print(f'without {what:50s} {t:6.2f} ms   marginal cost {base - t:5.2f} ms', flush=True)
    except Exception as e:      # noqa: BLE001
        print(f'without {what}: failed ({type(e).__name__}: {e})', flush=True)
    finally:
        setattr(ops, name, orig)

# several at once: are the marginal costs additive?
combos = [('bn_relu_maxpool_backward', 'point_sample_backward', 'roi_align_backward', 'deform_coord_grad'),
          ('bn_relu_maxpool_backward', 'roi_align_backward'),
          ('point_sample_backward', 'deform_coord_grad')]
table = {n: f for n, f, _ in cases}
for combo in combos:
    origs = {n: getattr(ops, n) for n in combo}
    for n in combo:
        setattr(ops, n, table[n])
    try:
        t = measure()
        print(f'without {" + ".join(combo)}: {t:6.2f} ms   marginal cost {base - t:5.2f} ms', flush=True)
    finally:
        for n, f in origs.items():
            setattr(ops, n, f)
