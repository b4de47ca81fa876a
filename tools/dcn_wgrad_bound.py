"""Upper bound of what a gather-form DCN weight gradient could save: the step with the column matrix never kept and
(a) the backward's im2col + weight gradient as they are, (b) both skipped for the 28 x 28 / 56 x 56 stages."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import synth, ops, train_path
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
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        grp.zero_grad()
        res = head._mask_forward_train(feats, rois, labels, targets, noise=noise)
        res['loss_mask']['loss_masks'].backward()
        grp.all_reduce_async(); grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
def run(label):
    window(3)
    print(label, ' '.join(f'{window():.2f}' for _ in range(4)), flush=True)
run('as shipped (column matrix kept at 28 / 56)      ')
train_path._KEEP_COL_MIN_PIXELS = 10 ** 9
run('never kept (fused forward, im2col in backward)  ')
orig = ops.deform_conv_backward_weight
def skip(x, offset, grad_out, deform_groups, gw_accum=None, col=None):
    if x.shape[2] * x.shape[3] >= 784:
        return None if gw_accum is not None else torch.zeros(grad_out.shape[1], x.shape[1], 3, 3, device=x.device)
    return orig(x, offset, grad_out, deform_groups, gw_accum=gw_accum, col=col)
ops.deform_conv_backward_weight = skip
run('never kept, weight gradient at 28 / 56 skipped  ')
