"""The training step (configs[2] shape, four streams) under the stream-hazard tracker (DM_HAZARD / hazard.py).

Every C-ABI launch of two steps -- forward on two streams + selector stream + leaf stream, loss, the hand-sequenced
backward on four streams, flat-gradient all-reduce hook and fused SGD -- is checked for a cross-stream
read-after-write / write-after-read / write-after-write without an ordering event, and for tensors recycled by the
allocator under a side stream.  The reference runs on one stream (deform_conv_cuda_kernel.cu:265)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import golden_inputs as gi  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture
def tracker():
    from dynamask_amd import hazard
    prev = hazard.ENABLED[0]
    torch.cuda.synchronize()
    hazard.reset()
    hazard.ENABLED[0] = True
    yield hazard
    hazard.ENABLED[0] = prev
    hazard.reset()


def _build(dev):
    from dynamask_amd import synth, registry, roi_head, mask_heads, roi_extractors, losses  # noqa: F401
    m = registry.build_head(dict(type='DynaMaskRoIHead',
                                 mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
                                 mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG)))
    m.load_state_dict({**synth.init_dynamask_head_state(seed=5), **synth.init_mask_pre_state(seed=6)}, strict=True)
    return m.to(dev).train()


def test_training_step_has_no_cross_stream_hazard(tracker):
    from dynamask_amd import synth
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    dev = torch.device('cuda')
    B, per, H, W = 2, 128, 800, 1333
    feats = [f.to(dev) for f in synth.make_fpn(B, H, W, 256, seed=10)]
    rois = synth.make_rois(B, per, H, W, seed=11).to(dev)
    labels = synth.make_labels(B * per, seed=12).to(dev)
    targets = [t.to(dev) for t in synth.make_targets(B * per, seed=13)]
    noise = synth.make_gumbel_noise(B * per, seed=14).to(dev)
    m = _build(dev)
    grp = FlatParamGroup(mask_path_parameters(m))
    torch.cuda.synchronize()
    for _ in range(3):
        grp.zero_grad()
        res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
        res['loss_mask']['loss_masks'].backward()
        grp.all_reduce_async()
        grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
    torch.cuda.synchronize()
    assert tracker.TRACKER.launches > 600, 'the tracker did not see the step'
    assert len(tracker.TRACKER.clock) >= 4, 'the step is expected to use the main stream and three side streams'
    assert tracker.reports() == [], '\n'.join(tracker.reports())


def test_the_tracker_sees_a_planted_hazard(tracker):
    """A read on a side stream of what the main stream has just written, without a wait: must be reported; with the
    wait: must not."""
    from dynamask_amd import ops, streams
    dev = torch.device('cuda', torch.cuda.current_device())
    side = streams.side(dev, 0)
    x = torch.randn(64, 32, 14, 14, device=dev)
    g = torch.randn(64, 32, 14, 14, device=dev)
    torch.cuda.synchronize()
    ops.relu_backward_(g, x)                       # main writes g
    with torch.cuda.stream(side):
        ops.channel_sum(g)                         # side reads g: no wait
    assert any(r.startswith('read-after-write') for r in tracker.reports()), tracker.reports()
    tracker.reset()
    torch.cuda.synchronize()
    ops.relu_backward_(g, x)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ops.channel_sum(g)
    torch.cuda.current_stream(dev).wait_stream(side)
    assert tracker.reports() == []
    torch.cuda.synchronize()


def test_inference_on_two_streams_and_forward_train_have_no_hazard(tracker):
    """The 512-RoI inference split over two streams and the bbox + mask ``forward_train`` entry point."""
    from dynamask_amd import synth
    dev = torch.device('cuda')
    H, W = 800, 1333
    feats = [f.to(dev) for f in synth.make_fpn(1, H, W, 256, seed=20)]
    rois = synth.make_rois(1, 512, H, W, seed=21).to(dev)
    labels = synth.make_labels(512, seed=22).to(dev)
    m = _build(dev).eval()
    torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(2):
            m._mask_forward(feats, rois, labels)
    torch.cuda.synchronize()
    assert tracker.TRACKER.launches > 50
    assert tracker.reports() == [], '\n'.join(tracker.reports())
