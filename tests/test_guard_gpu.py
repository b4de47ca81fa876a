"""Memory-safety net: every device tensor the package allocates while these tests run --
outputs, scratch, intermediates, and the inputs the tests hand over -- sits between two guard bands of signalling
patterns.  After each pass the bands must be bit-for-bit intact (a kernel stored one past the end, or in front of the
start, of a buffer otherwise) and every result finite (float bands are NaN: an out-of-range READ that reaches a result
poisons it).  Shapes are the odd ones of tools/stress.py: RoI counts 1 / 7 / 129 / 513 on 333x500 ... 2048x1024
pyramids -- magic-number divisions, LDS band staging and the "last round" second launches all see ragged edges."""
import contextlib

import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu

GUARD = 256          # elements on each side (1 KB of fp32): keeps the interior 256-byte aligned
_PATTERN = {1: 0x5A, 2: 0x5A5A, 4: 0x7FC0DEAD, 8: 0x7FF8DEAD7FC0DEAD}      # 4 / 8 bytes: NaN bit patterns


class Arena:
    """Guarded allocation: replaces torch's tensor factories for CUDA tensors while active."""

    def __init__(self):
        self.blocks = []          # (parent int-view, n interior elements, shape, dtype)
        self.count = 0

    def _int_dtype(self, dtype):
        return {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[torch.empty((), dtype=dtype).element_size()]

    def alloc(self, shape, dtype, device, fill=None):
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        idt = self._int_dtype(dtype)
        esz = torch.empty((), dtype=dtype).element_size()
        pat = _PATTERN[esz]
        if esz == 1:
            raw = self._empty((n + 2 * GUARD,), dtype=idt, device=device)
            raw.fill_(pat)
        else:
            raw = self._full((n + 2 * GUARD,), pat if pat < 2 ** 63 else pat - 2 ** 64, dtype=torch.int64 if esz == 8 else idt,
                             device=device)
        inner = raw[GUARD:GUARD + n].view(dtype).view(shape)
        if fill is not None:
            inner.fill_(fill)
        self.blocks.append((raw, n, pat))
        self.count += 1
        return inner

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for k, (raw, n, pat) in enumerate(self.blocks):
            p = pat if pat < 2 ** 63 else pat - 2 ** 64
            lo, hi = raw[:GUARD], raw[GUARD + n:]
            if not bool((lo == p).all()) or not bool((hi == p).all()):
                bad.append((k, n, int((lo != p).sum()), int((hi != p).sum())))
        assert not bad, f'guard bands overwritten (block, elements, cells before, cells after): {bad[:8]}'
        return len(self.blocks)


@contextlib.contextmanager
def guarded(monkeypatch):
    arena = Arena()
    o_empty, o_zeros, o_full, o_ones = torch.empty, torch.zeros, torch.full, torch.ones
    o_empty_like, o_zeros_like = torch.empty_like, torch.zeros_like
    arena._empty, arena._full = o_empty, o_full

    def _is_cuda(device):
        return device is not None and torch.device(device).type == 'cuda'

    def _shape(args):
        return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else tuple(args)

    def mk(orig, fill):
        def f(*args, dtype=None, device=None, **kw):
            if not _is_cuda(device) or kw.get('out') is not None or kw.get('memory_format') is not None:
                return orig(*args, dtype=dtype, device=device, **kw)
            t = arena.alloc(_shape(args), dtype or torch.float32, device, fill)
            return t.requires_grad_(True) if kw.get('requires_grad') else t
        return f

    def full(size, fill_value, *, dtype=None, device=None, **kw):
        if not _is_cuda(device):
            return o_full(size, fill_value, dtype=dtype, device=device, **kw)
        if dtype is None:
            dtype = torch.float32 if isinstance(fill_value, float) else (torch.bool if isinstance(fill_value, bool) else torch.int64)
        return arena.alloc(size, dtype, device, fill_value)

    def like(fill):
        def f(t, *, dtype=None, device=None, **kw):
            device = device or t.device
            if not _is_cuda(device) or not t.is_contiguous():
                return (o_empty_like if fill is None else o_zeros_like)(t, dtype=dtype, device=device, **kw)
            return arena.alloc(t.shape, dtype or t.dtype, device, fill)
        return f

    def new(fill):
        def f(self, *args, dtype=None, device=None, **kw):
            device = device or self.device
            if not _is_cuda(device):
                return (o_empty if fill is None else o_zeros)(*args, dtype=dtype or self.dtype, device=device)
            return arena.alloc(_shape(args), dtype or self.dtype, device, fill)
        return f

    def new_full(self, size, fill_value, *, dtype=None, device=None, **kw):
        device = device or self.device
        if not _is_cuda(device):
            return o_full(size, fill_value, dtype=dtype or self.dtype, device=device)
        return arena.alloc(size, dtype or self.dtype, device, fill_value)

    monkeypatch.setattr(torch, 'empty', mk(o_empty, None))
    monkeypatch.setattr(torch, 'zeros', mk(o_zeros, 0))
    monkeypatch.setattr(torch, 'ones', mk(o_ones, 1))
    monkeypatch.setattr(torch, 'full', full)
    monkeypatch.setattr(torch, 'empty_like', like(None))
    monkeypatch.setattr(torch, 'zeros_like', like(0))
    monkeypatch.setattr(torch.Tensor, 'new_empty', new(None))
    monkeypatch.setattr(torch.Tensor, 'new_zeros', new(0))
    monkeypatch.setattr(torch.Tensor, 'new_full', new_full)
    try:
        yield arena
    finally:
        monkeypatch.undo()


def _put(arena, t):
    """Copy a host tensor into guarded device storage."""
    g = arena.alloc(t.shape, t.dtype, torch.device('cuda'))
    g.copy_(t)
    return g


def _finite(ts):
    return all(bool(torch.isfinite(t).all()) for t in ts if t.is_floating_point())


def _full_head():
    from dynamask_amd import registry, roi_head, losses, mask_heads, roi_extractors, bbox_heads, synth  # noqa: F401
    from dynamask_amd.registry import ConfigDict
    cfg = dict(type='DynaMaskRoIHead',
               bbox_roi_extractor=dict(type='SingleRoIExtractor', **gi.BBOX_ROI_EXTRACTOR_CFG),
               bbox_head=dict(type='Shared2FCBBoxHead', **gi.BBOX_HEAD_CFG),
               mask_roi_extractor=dict(type='SingleRoIExtractor', **gi.MASK_ROI_EXTRACTOR_CFG),
               mask_head=dict(type='DynaMaskHead', **gi.MASK_HEAD_CFG), test_cfg=ConfigDict(**gi.RCNN_TEST_CFG))
    m = registry.build_head(cfg)
    m.load_state_dict({**synth.init_dynamask_head_state(seed=5, test_mode=True), **synth.init_mask_pre_state(seed=6),
                       **synth.init_bbox_head_state(seed=8)}, strict=True)
    return m.cuda()


def test_arena_catches_a_store_past_the_end(monkeypatch):
    """The net itself: a deliberate one-past-the-end store must be reported."""
    with guarded(monkeypatch) as arena:
        t = torch.empty((5, 3), device='cuda')
        assert arena.count == 1 and t.is_contiguous() and t.data_ptr() % 256 == 0
        arena.check()
        t.view(-1).as_strided((16,), (1,))[15] = 1.0          # element 15 of a 15-element buffer
        with pytest.raises(AssertionError, match='guard bands overwritten'):
            arena.check()


@pytest.mark.parametrize('H,W', [(333, 500), (608, 1024), (800, 1333), (1024, 2048)])
def test_inference_surface_keeps_its_guard_bands(monkeypatch, H, W):
    """tools/stress.py as a test: mask head (all exits), per-RoI early exit with the selector, the whole
    simple_test (bbox branch, NMS, masks, device RLE) at RoI counts 1 / 7 / 129 / 513."""
    from dynamask_amd import synth
    m = _full_head().eval()
    with guarded(monkeypatch) as arena:
        feats = [_put(arena, f) for f in synth.make_fpn(1, H, W, 256, seed=H)]
        for N in (1, 7, 129, 513):
            rois = _put(arena, synth.make_rois(1, N, H, W, seed=N))
            labels = _put(arena, synth.make_labels(N, seed=N + 1))
            with torch.no_grad():
                r = m._mask_forward(feats, rois, labels)
                assert _finite(r['stage_instance_preds'] + r['stage_detail_preds']), (H, W, N)
                d = m.dynamic_mask_logits(feats, rois[:, 1:].contiguous(), labels)
                assert _finite(d['preds']), (H, W, N)
                # the fixed 28x28 exit and exits spread over the four resolutions: the logits-of-the-upsampled-stage kernel
                r1 = m._mask_forward(feats, rois, labels, last_stage=1)
                assert _finite(r1['stage_instance_preds']), (H, W, N)
                d2 = m.dynamic_mask_logits(feats, rois[:, 1:].contiguous(), labels, exits=(torch.arange(N) * 7 + 3) % 4)
                assert _finite(d2['preds']), (H, W, N)
                metas = [dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), scale_factor=1.0)]
                bb, sg = m.simple_test(feats, [rois[:, 1:].contiguous()], metas, rescale=False, encode=True)
                assert sum(len(b) for b in bb) == sum(len(s) for s in sg)
            n_blocks = arena.check()
        print(f'{H}x{W}: {n_blocks} guarded device buffers, all bands intact')


@pytest.mark.parametrize('H,W,per', [(333, 500, 7), (800, 1333, 129), (608, 1024, 37)])
def test_training_step_keeps_its_guard_bands(monkeypatch, H, W, per):
    """One training step of the mask path (forward, loss, hand-sequenced backward incl. the DCN / point-sample / RoIAlign
    adjoints, fused SGD) at odd RoI counts, default and deterministic accumulation."""
    from dynamask_amd import ops, synth
    from dynamask_amd.dist import FlatParamGroup, mask_path_parameters
    B = 2
    for det in (False, True):
        monkeypatch.setattr(ops, 'DETERMINISTIC', [det])
        m = _full_head().train()
        with guarded(monkeypatch) as arena:
            feats = [_put(arena, f).requires_grad_(True) for f in synth.make_fpn(B, H, W, 256, seed=H + 1)]
            rois = _put(arena, synth.make_rois(B, per, H, W, seed=per))
            labels = _put(arena, synth.make_labels(B * per, seed=per + 1))
            targets = [_put(arena, t) for t in synth.make_targets(B * per, seed=per + 2)]
            noise = _put(arena, synth.make_gumbel_noise(B * per, seed=per + 3))
            grp = FlatParamGroup(mask_path_parameters(m))
            grp.zero_grad()
            res = m._mask_forward_train(feats, rois, labels, targets, noise=noise)
            res['loss_mask']['loss_masks'].backward()
            grp.sgd_step(lr=0.02, momentum=0.9, weight_decay=1e-4)
            assert _finite([res['loss_mask']['loss_masks'].detach(), grp.flat_grad, grp.flat_param] + [f.grad for f in feats[:4]])
            n_blocks = arena.check()
        print(f'{H}x{W}, 2 x {per} RoIs, deterministic={det}: {n_blocks} guarded device buffers, all bands intact')
