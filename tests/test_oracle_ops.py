"""Known-answer tests that pin the oracle's restatement of the mmcv operators
(the third-party part: "parity unpinned" against mmcv itself)."""
import torch
import torch.nn.functional as F

from oracle import ref_ops


def _g(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def test_roi_align_constant_map():
    feat = torch.full((1, 3, 20, 30), 2.5)
    rois = torch.tensor([[0, 3.2, 4.1, 50.7, 60.3], [0, 0., 0., 8., 8.]])
    out = ref_ops.roi_align(feat, rois, 7, 0.25)
    assert torch.allclose(out, torch.full_like(out, 2.5), atol=1e-6)


def test_roi_align_linear_ramp():
    # value = 2*x + 3*y on the pixel grid; bilinear interpolation reproduces
    # a linear function, so each bin returns the ramp at the bin centre.
    H, W = 40, 50
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    feat = (2 * xx + 3 * yy)[None, None]
    rois = torch.tensor([[0, 8.0, 12.0, 36.0, 40.0]])   # fully inside, scale 1
    P = 7
    out = ref_ops.roi_align(feat, rois, P, 1.0, 0, True)[0, 0]
    bw = (36.0 - 8.0) / P
    bh = (40.0 - 12.0) / P
    cx = 8.0 - 0.5 + (torch.arange(P) + 0.5) * bw
    cy = 12.0 - 0.5 + (torch.arange(P) + 0.5) * bh
    exp = 2 * cx[None, :] + 3 * cy[:, None]
    assert torch.allclose(out, exp, atol=1e-4)


def test_roi_align_vs_float64_bruteforce():
    feat = torch.randn(2, 3, 12, 17, generator=_g(0))
    rois = torch.tensor([
        [0, 1.3, 2.2, 30.1, 20.7], [1, -5.0, -3.0, 70.0, 60.0], [0, 10.0, 10.0, 10.5, 10.2],
        [1, 20.0, 4.0, 66.0, 44.0], [0, 50.0, 30.0, 90.0, 70.0]])
    for P, sr in ((7, 0), (4, 2), (14, 0)):
        a = ref_ops.roi_align(feat, rois, P, 0.25, sr, True)
        b = ref_ops.roi_align_bruteforce_f64(feat, rois, P, 0.25, sr, True)
        assert torch.allclose(a.double(), b, atol=1e-5), (P, sr)


def test_simple_roi_align_identity_box():
    # RoI = whole feature map at scale 1 -> pixel centres -> identity
    feat = torch.randn(1, 4, 6, 6, generator=_g(1))
    rois = torch.tensor([[0, 0., 0., 6., 6.]])
    out = ref_ops.simple_roi_align(feat, rois, 6, 1.0)
    assert torch.allclose(out[0], feat[0], atol=1e-6)


def test_deform_conv_zero_offset_equals_conv2d():
    x = torch.randn(3, 8, 9, 11, generator=_g(2))
    w = torch.randn(6, 8, 3, 3, generator=_g(3))
    off = torch.zeros(3, 2 * 18, 9, 11)
    out = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 2)
    assert torch.allclose(out, F.conv2d(x, w, padding=1), atol=1e-5)


def test_deform_conv_integer_offset_is_shift():
    # offset (dh, dw) = (1, -2) everywhere == sampling a shifted image
    x = torch.randn(1, 4, 10, 10, generator=_g(4))
    w = torch.randn(5, 4, 3, 3, generator=_g(5))
    off = torch.zeros(1, 18, 10, 10)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    out = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 1)
    xs = torch.zeros(1, 4, 14, 14)
    xs[:, :, 2:12, 2:12] = x                      # zero-extended image
    # sample (h+1, w-2): build shifted copy then plain conv
    shifted = torch.zeros_like(xs)
    shifted[:, :, 0:13, 2:14] = xs[:, :, 1:14, 0:12]
    exp = F.conv2d(shifted, w, padding=1)[:, :, 2:12, 2:12]
    assert torch.allclose(out, exp, atol=1e-5)


def test_carafe_uniform_kernel_is_box_filter():
    x = torch.randn(1, 2, 5, 5, generator=_g(6))
    mask = torch.full((1, 25, 10, 10), 1.0 / 25)
    out = ref_ops.carafe_reassemble(x, mask, 5, 1, 2)
    box = F.avg_pool2d(F.pad(x, (2, 2, 2, 2)), 5, stride=1)
    exp = box.repeat_interleave(2, 2).repeat_interleave(2, 3)
    assert torch.allclose(out, exp, atol=1e-6)


def test_dynamic_exit_selection_restates_the_commented_reference_path():
    """dynamask_roi_head.py:160-204 (commented out there): RoI j takes exit mask_labels[j];
    with merge, exit 3 must coincide with the live boundary-merge path."""
    import torch
    from oracle import ref_model
    g = torch.Generator().manual_seed(0)
    ips = [torch.randn(5, 1, s, s, generator=g) for s in (14, 28, 56, 112)]
    exits = torch.tensor([3, 0, 2, 1, 3])
    raw = ref_model.dynamic_exit_logits(ips, exits, merge=False)
    for j, e in enumerate(exits.tolist()):
        assert torch.equal(raw[j], ips[e][j])
    merged = ref_model.dynamic_exit_logits(ips, exits, merge=True)
    full = ref_model.boundary_merge(ips)
    assert torch.equal(merged[0], full[0]) and torch.equal(merged[4], full[4])
    assert torch.equal(merged[1], ips[0][1]) and torch.equal(merged[3], ips[1][3])
    assert merged[2].shape == (1, 56, 56)
    two = ref_model.boundary_merge([None, ips[1][2:3], ips[2][2:3]])
    assert torch.equal(merged[2], two[0])


def test_rle_oracle_known_answers_and_round_trip():
    """COCO RLE restated from cocoapi's maskApi.c (pycocotools absent: parity unpinned);
    hand-derived vectors + encode/decode round trips."""
    import numpy as np
    from oracle import ref_ops as R
    assert R.rle_encode(np.ones((2, 2), np.uint8)) == {'size': [2, 2], 'counts': b'04'}
    assert R.rle_encode(np.zeros((2, 2), np.uint8)) == {'size': [2, 2], 'counts': b'4'}
    assert R.rle_counts(np.array([[0, 1], [1, 1]])) == [1, 3]            # column-major walk: 0,1,1,1
    assert R.rle_counts(np.array([[1, 0], [0, 0]])) == [0, 1, 3]
    # 40 = 0b01000 + (1 << 5): two groups, 8|0x20 -> 'X', 1 -> '1'
    assert R.rle_to_string([40]) == b'X1'
    # from the fourth count on the difference to the count two back is stored (5 - 3 = 2)
    assert R.rle_to_string([7, 3, 5]) == b'735' and R.rle_to_string([7, 3, 9, 5]) == b'739' + bytes([(2 & 0x1f) + 48])
    assert R.rle_from_string(R.rle_to_string([7, 3, 9, 1, 300, 2])) == [7, 3, 9, 1, 300, 2]
    rng = np.random.default_rng(1)
    for shape in ((1, 1), (3, 5), (17, 4), (40, 33)):
        for _ in range(10):
            m = (rng.random(shape) < rng.random()).astype(np.uint8)
            assert (R.rle_decode(R.rle_encode(m)) == m).all()


def test_roi_align_on_the_reference_tests_own_roi_matches_f64_bruteforce():
    """The one RoI the reference's tests hold for this op (tests/test_models/test_roi_extractor.py:40,
    FPN shapes 200x336 ... 25x42, output 7x7, sampling_ratio 2 -- asserted there for shape only): the
    oracle against the float64 sample-by-sample restatement, on the level the extractor picks."""
    rois = torch.tensor([[0.0000, 587.8285, 52.1405, 886.2484, 341.5644]])
    lvl = int(ref_ops.map_roi_levels(rois, 4)[0])
    assert lvl == 2                                   # sqrt(298.4 * 289.4) = 293.9 -> floor(log2(293.9 / 56)) = 2
    H, W = (200 >> lvl), (336 >> lvl)
    feat = torch.rand(1, 6, H, W, generator=_g(40))
    for sr in (2, 0):
        a = ref_ops.roi_align(feat, rois, 7, 1.0 / (4 << lvl), sr, True)
        b = ref_ops.roi_align_bruteforce_f64(feat, rois, 7, 1.0 / (4 << lvl), sr, True)
        assert torch.allclose(a.double(), b, atol=1e-6), sr


def test_deform_conv_integer_shifts_at_every_border():
    """Whole-pixel offsets turn DCN into a plain conv of the shifted, zero-extended image -- including the
    rows / columns where the shifted taps leave the map (deform_conv_cuda_kernel.cu:220-243 validity rule)."""
    x = torch.randn(2, 4, 9, 11, generator=_g(41))
    w = torch.randn(5, 4, 3, 3, generator=_g(42))
    for dh, dw in ((-1, 0), (1, 0), (0, -1), (0, 1), (2, 2), (-2, -3), (9, 0), (0, -11)):
        off = torch.zeros(2, 18, 9, 11)
        off[:, 0::2] = float(dh)
        off[:, 1::2] = float(dw)
        out = ref_ops.deform_conv2d(x, off, w, 1, 1, 1, 1)
        pad = 12
        xs = F.pad(x, (pad, pad, pad, pad))
        shifted = torch.roll(xs, shifts=(-dh, -dw), dims=(2, 3))           # sample (h + dh, w + dw)
        exp = F.conv2d(shifted, w, padding=1)[:, :, pad:pad + 9, pad:pad + 11]
        assert torch.allclose(out, exp, atol=1e-5), (dh, dw)


def test_oracle_assigner_on_the_reference_tests_known_answers():
    """/root/reference/tests/test_assigner.py:14-82 (inputs and expected assignments as data): plain, with an ignore
    region (IoF over the region), without ground truth."""
    import torch
    from oracle import ref_model
    bb = torch.tensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]], dtype=torch.float32)
    gt = torch.tensor([[0, 0, 10, 9], [0, 10, 10, 19]], dtype=torch.float32)
    gi_, _, lab = ref_model.max_iou_assign(ref_model.bbox_overlaps(gt, bb), 0.5, 0.5, gt_labels=torch.tensor([2, 3]))
    assert gi_.tolist() == [1, 0, 2, 0] and lab.tolist() == [2, -1, 3, -1]
    bb2 = bb.clone()
    bb2[3] = torch.tensor([30., 32., 40., 42.])
    ov = ref_model.ignore_overlaps(ref_model.bbox_overlaps(gt, bb2), bb2, torch.tensor([[30., 30., 40., 40.]]), 0.5, False)
    assert ref_model.max_iou_assign(ov, 0.5, 0.5)[0].tolist() == [1, 0, 2, -1]
    assert ref_model.max_iou_assign(torch.zeros(0, 4), 0.5, 0.5)[0].tolist() == [0, 0, 0, 0]


def test_oracle_bbox_overlaps_docstring_example():
    """iou2d_calculator.py:55-71."""
    import torch
    from oracle import ref_model
    b1 = torch.tensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]], dtype=torch.float32)
    b2 = torch.tensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]], dtype=torch.float32)
    torch.testing.assert_close(ref_model.bbox_overlaps(b1, b2), torch.tensor([[0.5, 0., 0.], [0., 0., 1.], [0., 0., 0.]]),
                               atol=1e-4, rtol=0)
