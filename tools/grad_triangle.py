"""fp64 triangle for the config[2] training step: |HIP - f64| and |f32 oracle - f64| per checked parameter."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import golden_inputs as gi
from dynamask_amd import synth, registry, roi_head, losses, mask_heads, roi_extractors  # noqa
from oracle import ref_model
B, per = 2, 128
feats = synth.make_fpn(B, 800, 1333, 256, seed=10)
rois = synth.make_rois(B, per, 800, 1333, seed=11)
labels = synth.make_labels(B * per, seed=12)
targets = synth.make_targets(B * per, seed=13)
noise = synth.make_gumbel_noise(B * per, seed=14)
sd = {**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6)}
keys = ['mask_head.stages.2.fuse_transform_out.weight', 'mask_head.final_instance_logits.weight',
        'mask_head.final_detail_logits.bias', 'mask_predictor.fc2.weight', 'mask_predictor.conv2.weight',
        'mask_predictor.bn2.weight', 'mask_predictor.fc1.weight', 'mask_predictor.bn2.bias']
cfg = dict(type='DynaMaskRoIHead', mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
           mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG))
m = registry.build_head(cfg); m.load_state_dict(sd, strict=True); m = m.cuda().train()
res = m._mask_forward_train([f.cuda() for f in feats], rois.cuda(), labels.cuda(), [t.cuda() for t in targets], noise=noise.cuda())
res['loss_mask']['loss_masks'].backward(); torch.cuda.synchronize()
named = dict(m.named_parameters())
from oracle import ref_ops
with torch.no_grad():
    t0 = time.time()
    ips, dps = ref_model.mask_forward(sd, feats, rois, labels)
    sem = ref_ops.single_roi_extractor([feats[0]], rois, 56, (4,))
    print('oracle head forward', time.time() - t0, flush=True)
mp_keys = [k for k in sd if k.startswith('mask_predictor.') and sd[k].is_floating_point() and 'running' not in k]
out = {}
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    sdo = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
    for k in mp_keys: sdo[k] = sdo[k].clone().requires_grad_(True)
    logits = ref_model.mask_pre(sdo, sem.to(dt), training=True)
    ml, ind = ref_model.gumbel_select(logits, noise.to(dt), 0.5)
    loss = ref_model.dyna_loss([t.to(dt) for t in ips], [t.to(dt) for t in dps], [t.to(dt) for t in targets], ml,
                               fuse_kernel=sd['mask_head.loss_func.detail_target.fuse_kernel'])
    loss.backward()
    out[dt] = ({k: sdo[k].grad for k in mp_keys}, float(loss.detach()), ind)
    print(dt, 'MaskPre+loss', time.time() - t0, 's loss', float(loss.detach()), flush=True)
print('hip loss', float(res['loss_mask']['loss_masks'].detach()), 'idx equal f32/f64:', torch.equal(out[torch.float32][2], out[torch.float64][2]),
      'hip idx == f64:', torch.equal(res['mask_index'].cpu().long(), out[torch.float64][2]))
for k in mp_keys:
    g64 = out[torch.float64][0][k]
    g32 = out[torch.float32][0][k].double()
    gh = named[k].grad.cpu().double()
    sc = float(g64.abs().max())
    print(f'{k:36s} max|g| {sc:.3e}  |hip-f64| {float((gh-g64).abs().max()):.3e}  |f32-f64| {float((g32-g64).abs().max()):.3e}  |hip-f32| {float((gh-g32).abs().max()):.3e}')
