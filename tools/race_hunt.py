"""Run-to-run equality of the training step's gradients under DM_DETERMINISTIC=1, per parameter.

  python tools/race_hunt.py [--reps 40] [--perturb] [--hazard] [--no-side]

--perturb : random busy-wait kernels (torch.cuda._sleep) in front of every side-stream hand-off and between the
            chain's stages, so that the relative timing of the four streams changes from pass to pass: a result that
            depends on timing shows up as a differing tensor.
--hazard  : the first two passes run under the stream-hazard tracker (dynamask_amd/hazard.py) and its reports are printed.
(Round 5 used it with the bf16-split modes that have since been removed: their timing showed the dropped product of
profiles/r05_race_hunt.txt in 39 of 39 passes.)
"""
import argparse
import os
import random
import sys

os.environ.setdefault('DM_DETERMINISTIC', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch  # noqa: E402
import golden_inputs as gi  # noqa: E402
from dynamask_amd import hazard, ops, synth, registry, roi_head, mask_heads, roi_extractors, losses, train_path  # noqa: E402,F401
from dynamask_amd.dist import FlatParamGroup, mask_path_parameters  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=40)
ap.add_argument('--perturb', action='store_true')
ap.add_argument('--hazard', action='store_true')
ap.add_argument('--no-side', action='store_true')
ap.add_argument('--max-sleep', type=int, default=400000, help='cycles of the longest injected wait')
args = ap.parse_args()
if args.no_side:
    os.environ['DM_TRAIN_SIDE_STREAM'] = '0'
dev = torch.device('cuda')
B, per, H, W = 2, 128, 800, 1333
feats = [f.to(dev) for f in synth.make_fpn(B, H, W, 256, seed=10)]
rois = synth.make_rois(B, per, H, W, seed=11).to(dev)
labels = synth.make_labels(B * per, seed=12).to(dev)
targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)
m = registry.build_head(dict(type='DynaMaskRoIHead',
                             mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                             mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
m = m.to(dev).train()
params = mask_path_parameters(m)
names = {id(p): n for n, p in m.named_parameters()}
grp = FlatParamGroup(params)
rng = random.Random(1234)

if args.perturb:
    real_run = train_path._SideWork.run

    def run(self, fn, *tensors, after=None, alt=False):
        def delayed():
            if rng.random() < 0.5:
                torch.cuda._sleep(rng.randrange(args.max_sleep))
            return fn()
        r = real_run(self, delayed, *tensors, after=after, alt=alt)
        if self.enabled and rng.random() < 0.3:
            torch.cuda._sleep(rng.randrange(args.max_sleep))          # on the main stream
        return r
    train_path._SideWork.run = run


def one_pass():
    grp.zero_grad()
    res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
    loss = res['loss_mask']['loss_masks']
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), grp.flat_grad.clone()


if args.hazard:
    hazard.ENABLED[0] = True
    for _ in range(2):
        one_pass()
    hazard.ENABLED[0] = False
    print(f'hazard tracker: {hazard.TRACKER.launches} launches on {len(hazard.TRACKER.clock)} streams, '
          f'{len(hazard.reports())} report(s)')
    for r in hazard.reports():
        print('  ', r)
    hazard.reset()

ref_loss, ref = one_pass()
differ = {}
n_diff = 0
distinct = [ref]
seq = [0]
for rep in range(1, args.reps):
    loss, g = one_pass()
    for k, d in enumerate(distinct):
        if torch.equal(g, d):
            seq.append(k)
            break
    else:
        distinct.append(g)
        seq.append(len(distinct) - 1)
    if torch.equal(g, ref) and torch.equal(loss, ref_loss):
        continue
    n_diff += 1
    off = 0
    for p in grp.params:
        k = p.numel()
        d = (g[off:off + k] - ref[off:off + k]).abs().max().item()
        if d != 0.0:
            e = differ.setdefault(names.get(id(p), '?'), [0, 0.0])
            e[0] += 1
            e[1] = max(e[1], d)
            if e[0] <= 3:
                dd = (g[off:off + k] - ref[off:off + k]).view(p.shape[0], -1)
                nz = dd.nonzero()
                print(f'   pass {rep}: {names.get(id(p))} {tuple(p.shape)}: {nz.shape[0]} elements differ; rows {sorted(set(nz[:, 0].tolist()))[:20]} '
                      f'cols {sorted(set(nz[:, 1].tolist()))[:40]}; labels count of those rows {[int((labels == r).sum()) for r in sorted(set(nz[:, 0].tolist()))[:20]]}')
        off += k
print(f'mode: deterministic={ops.DETERMINISTIC[0]} '
      f'side_streams={os.environ.get("DM_TRAIN_SIDE_STREAM", "1")} perturb={args.perturb}')
print(f'{n_diff} of {args.reps - 1} repeated passes differ from the first; {len(distinct)} distinct results, sequence {"".join(chr(65 + min(k, 25)) for k in seq)}')
for n, (c, d) in sorted(differ.items()):
    print(f'  {n}: differs in {c} passes, max |diff| {d:.3e}')
