"""Where does DynaMaskRoIHead.forward_train wait for the GPU?  Runs bench.entry_points_bench's forward_train once under
torch's sync debug mode and prints every synchronising call with the package frame that made it.
  python tools/sync_trace.py"""
import os, sys, traceback, warnings, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dynamask_amd import bbox_heads, registry, synth  # noqa: F401
from dynamask_amd.registry import ConfigDict
dev = torch.device('cuda')
_, sd = bench.build_head(dev)
rh = registry.build_head(dict(
    type='DynaMaskRoIHead', bbox_roi_extractor=dict(type='SingleRoIExtractor', **synth.BBOX_ROI_EXTRACTOR_CFG),
    bbox_head=dict(type='Shared2FCBBoxHead', **synth.BBOX_HEAD_CFG),
    mask_roi_extractor=dict(type='SingleRoIExtractor', **synth.MASK_ROI_EXTRACTOR_CFG),
    mask_head=dict(type='DynaMaskHead', **synth.MASK_HEAD_CFG),
    train_cfg=registry._to_cfgdict(synth.RCNN_TRAIN_CFG), test_cfg=ConfigDict(**synth.RCNN_TEST_CFG)))
rh.load_state_dict({**sd, **synth.init_bbox_head_state(seed=8)}, strict=True)
rh = rh.to(dev).train()
feats = [f.to(dev) for f in synth.make_fpn(2, 800, 1333, 256, seed=40)]
tb = synth.make_train_batch(2, 800, 1333, seed=41)
args = (feats, tb['img_metas'], [p.to(dev) for p in tb['proposals']], [t.to(dev) for t in tb['gt_bboxes']],
        [t.to(dev) for t in tb['gt_labels']], None, [t.to(dev) for t in tb['gt_masks']])
def ft():
    losses = rh.forward_train(*args)
    sum(v for k, v in losses.items() if 'loss' in k).backward()
for _ in range(2):
    ft()
torch.cuda.synchronize()
seen = []
def showwarning(message, category, filename, lineno, file=None, line=None):
    if 'synchronizing' in str(message).lower():
        frames = [f for f in traceback.extract_stack() if 'dynamask_amd' in f.filename or f.filename.endswith('sync_trace.py')]
        seen.append(' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in reversed(frames[-4:])))
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
ft()
torch.cuda.set_sync_debug_mode('default')
print(f'{len(seen)} synchronising calls in forward_train + backward:')
for s in seen:
    print('  ', s)
