"""cProfile of the host side of one training step (the step issues ~700 launches in ~29 ms)."""
import os, sys, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench
from dynamask_amd import synth
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
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(40)
