"""Single-GPU hardware check of the multi-GPU coupling: a world-size-1 RCCL
process group, the flat-gradient all-reduce FORCED through it on the side stream, and the
fused SGD step (dm_sgd_momentum_step) with its 1/world scaling.  The 2-rank arithmetic is
covered on CPU over gloo (tests/test_dist_cpu.py); this test makes sure the RCCL call,
the stream hand-over and the HIP optimiser kernel execute on the MI355X in every round."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_world1():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_forced_allreduce_and_fused_sgd_on_rccl(rccl_world1):
    from dynamask_amd.dist import FlatParamGroup
    dev = rccl_world1
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(64, 96), nn.Linear(96, 8)).to(dev)
    ref = nn.Sequential(nn.Linear(64, 96), nn.Linear(96, 8)).to(dev)
    ref.load_state_dict(net.state_dict())
    opt = torch.optim.SGD(ref.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    grp = FlatParamGroup(net.parameters())
    assert dist.get_backend() == 'nccl' and grp.world_size == 1
    x = torch.randn(32, 64, device=dev)
    for it in range(3):
        opt.zero_grad()
        ref(x + it).square().mean().backward()
        opt.step()
        if it == 1:
            net.zero_grad()                    # set_to_none=True: autograd replaces the .grad views
        else:
            grp.zero_grad()
        net(x + it).square().mean().backward()
        before = grp.flat_grad.clone() if it != 1 else None
        grp.all_reduce_async(force=True)
        assert grp._work is not None, 'the RCCL all-reduce was skipped'
        grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        torch.cuda.synchronize()
        if before is not None:                 # sum over one rank = identity, bit for bit
            assert torch.equal(grp.flat_grad, before)
        for p, pr in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p.detach(), pr.detach(), atol=1e-6, rtol=1e-5)


def test_training_step_through_rccl_matches_unreduced_step(rccl_world1):
    """The mask-path training step (configs[2]) with the forced collective gives the same
    parameters as the step without it (world 1: sum = identity, scale = 1)."""
    import golden_inputs as gi
    from dynamask_amd import losses, mask_heads, registry, roi_extractors, roi_head, synth  # noqa: F401
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    dev = rccl_world1

    def run(force):
        cfg = dict(type='DynaMaskRoIHead',
                   mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                   mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG))
        m = registry.build_head(cfg)
        m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
        m = m.to(dev).train()
        grp = FlatParamGroup(mask_path_parameters(m))
        B, per = 2, 8
        feats = [f.to(dev) for f in synth.make_fpn(B, 256, 320, 256, seed=10)]
        rois = synth.make_rois(B, per, 256, 320, seed=11).to(dev)
        labels = synth.make_labels(B * per, seed=12).to(dev)
        targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
        noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)
        for _ in range(2):
            grp.zero_grad()
            res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            grp.all_reduce_async(force=force)
            assert (grp._work is not None) == force
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
        torch.cuda.synchronize()
        return grp.flat_param.clone(), float(res['loss_mask']['loss_masks'].detach())
    p0, l0 = run(False)
    p1, l1 = run(True)
    assert abs(l0 - l1) <= 1e-5 * abs(l0)      # second step's loss: the first step's atomics-ordered last bits show
    # weight-gradient GEMMs accumulate split-K partials with float atomics: run-to-run last-bit noise
    torch.testing.assert_close(p1, p0, atol=1e-6, rtol=1e-5)


def test_clip_grad_norm_on_the_flat_group_matches_torch(rccl_world1):
    """optimizer_config grad_clip(max_norm=35, norm_type=2) (config :274) through FlatParamGroup: same
    coefficient as torch.nn.utils.clip_grad_norm_, applied on the device without a host sync; a second
    parameter group's squared norm can be folded in (the detector's other parameters)."""
    from dynamask_amd.dist import FlatParamGroup
    dev = rccl_world1
    torch.manual_seed(1)
    net = nn.Sequential(nn.Linear(64, 96), nn.Linear(96, 8)).to(dev)
    ref = nn.Sequential(nn.Linear(64, 96), nn.Linear(96, 8)).to(dev)
    ref.load_state_dict(net.state_dict())
    grp = FlatParamGroup(net.parameters())
    x = torch.randn(32, 64, device=dev) * 5
    for max_norm, other in ((0.5, None), (1e6, None), (0.5, 7.0)):
        grp.zero_grad()
        ref.zero_grad()
        net(x).square().mean().backward()
        ref(x).square().mean().backward()
        grp.all_reduce_async(force=True)
        extra = None if other is None else torch.full((1,), other, device=dev)
        total = grp.clip_grad_norm_(max_norm, other_sumsq=extra)
        gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
        norm = float((gref.double().square().sum() + (other or 0.0)).sqrt())
        coef = min(1.0, max_norm / (norm + 1e-6))
        torch.testing.assert_close(float(total.sqrt()), norm, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(grp.flat_grad, gref * coef, rtol=1e-5, atol=1e-7)


def test_scale_grads_of_a_parameter_subset(rccl_world1):
    """The reference's optional OptimizerHook_ (OptimizerHook.py:27-29) multiplies the gradients of
    roi_head.mask_predictor by 0.05 between clipping and the step: FlatParamGroup.scale_grads_ on the device,
    adjacent parameters as one run, the other parameters untouched."""
    from dynamask_amd.dist import FlatParamGroup
    dev = rccl_world1
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(16, 24), nn.Linear(24, 8), nn.Linear(8, 4)).to(dev)
    grp = FlatParamGroup(list(net.parameters()))
    net(torch.randn(5, 16, device=dev)).square().sum().backward()
    before = [p.grad.clone() for p in net.parameters()]
    sub = list(net[0].parameters()) + [net[2].bias]
    grp.scale_grads_(sub, 0.05)
    for p, b in zip(net.parameters(), before):
        exp = b * 0.05 if any(p is q for q in sub) else b
        assert torch.equal(p.grad, exp)


def test_gloo_default_group_with_an_rccl_subgroup_as_bench_uses_at_n_gt_1():
    """At N > 1 bench.py keeps gloo as the default group (host barriers, max over ranks) and creates the RCCL
    communicator of the data-path collective as a SUBGROUP inside the training leg, destroyed again afterwards.  That
    combination of torch.distributed calls cannot be exercised at N > 1 on a one-GPU box; here it runs at world size 1
    in a process of its own (a process has one default group): gloo default + `new_group(backend='nccl', device_id=...)`,
    the flat-gradient all-reduce forced through the subgroup, the subgroup destroyed, the default group still usable."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, socket, sys, torch, torch.distributed as dist, torch.nn as nn
sys.path.insert(0, %r)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('gloo')
sub = dist.new_group(backend='nccl', device_id=dev)
from dynamask_amd.dist import FlatParamGroup
net = nn.Linear(32, 16).to(dev)
grp = FlatParamGroup(net.parameters(), process_group=sub)
grp.zero_grad(); net(torch.randn(4, 32, device=dev)).sum().backward()
before = grp.flat_grad.clone()
grp.all_reduce_async(force=True); assert grp._work is not None; grp.wait(); torch.cuda.synchronize()
assert torch.equal(grp.flat_grad, before) and dist.get_backend(sub) == 'nccl' and dist.get_backend() == 'gloo'
t = torch.tensor([3.0], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 3.0
torch.cuda.synchronize(); dist.barrier(); dist.destroy_process_group(sub)
dist.barrier(); dist.destroy_process_group()
print('subgroup ok')
''' % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'subgroup ok' in r.stdout, r.stderr[-3000:]
