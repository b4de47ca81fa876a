"""Polygon rasterisation oracle (oracle/ref_poly.py) against the reference's own known answers
(tests/golden/g12_polygon_truth.npz <- /root/reference/tests/test_masks.py) and its closed form against the literal
restatement of cocoapi's rleFrPoly."""
import os

import numpy as np

from oracle import ref_poly as rp

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g12_polygon_truth.npz'))


def _cases():
    # test_polygon_mask_resize / _rescale: PolygonMasks.resize scales the vertices (5x5 -> 10x10, 3x3 -> 6x6)
    yield [G['poly1'] * 2.0], 10, 10, G['truth1']
    yield [G['poly1'] * 2.0], 10, 10, G['truth_rescale']
    yield [G['poly2a'] * 2.0, G['poly2b'] * 2.0], 6, 6, G['truth2']
    # test_polygon_mask_crop: vertices shifted by the box origin, canvas = box size (4 x 3)
    p = G['poly_crop'].copy()
    p[0::2] -= G['crop_bbox'][0]
    p[1::2] -= G['crop_bbox'][1]
    yield [p], 4, 3, G['truth_crop']


def test_oracle_reproduces_the_reference_tests_polygon_bitmaps():
    for events in (rp.fr_poly_points_literal, rp.fr_poly_events):
        for polys, h, w, truth in _cases():
            assert np.array_equal(rp.polygon_to_bitmap(polys, h, w, events), truth), events.__name__


def test_crop_and_resize_of_the_whole_canvas_is_resize():
    """PolygonMasks.crop_and_resize with the box = the canvas is PolygonMasks.resize (structures.py:385-402 vs 469-503):
    the reference's resize answers pin the crop-and-resize arithmetic too."""
    t = rp.polygon_mask_targets([[G['poly1']]], 5, 5, np.array([[0, 0, 5, 5]], np.float32), [0], 10)
    assert np.array_equal(t[0].astype(np.uint8), G['truth1'])
    t = rp.polygon_mask_targets([[G['poly2a'], G['poly2b']]], 3, 3, np.array([[0, 0, 3, 3]], np.float32), [0], 6)
    assert np.array_equal(t[0].astype(np.uint8), G['truth2'])


def test_closed_form_equals_the_literal_edge_walk():
    rng = np.random.default_rng(0)
    for it in range(1500):
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        k = int(rng.integers(3, 9))
        mode = it % 4
        if mode == 0:
            xy = rng.uniform(-0.5 * w, 1.5 * w, 2 * k)
        elif mode == 1:
            xy = rng.uniform(-5 * w, 6 * w, 2 * k)                  # far outside: long edges
        elif mode == 2:
            xy = np.round(rng.uniform(-2, w + 2, 2 * k))            # integer vertices: ties everywhere
        else:
            xy = np.round(rng.uniform(-2, w + 2, 2 * k) * 5) / 5 + 0.1 * (rng.integers(0, 3, 2 * k) - 1)
        if it % 7 == 0:
            xy[2:4] = xy[0:2]                                       # duplicate vertex: zero-length edge
        a = sorted(rp.fr_poly_points_literal(xy, h, w))
        b = sorted(rp.fr_poly_events(xy, h, w))
        assert a == b, (it, h, w, xy.tolist())
        assert np.array_equal(rp.rle_decode(rp.rle_from_points(a, h, w), h, w), rp.mask_from_points(b, h, w))


def test_vertex_transform_matches_the_reference_class():
    """g12b: PolygonMasks.crop_and_resize of the reference itself (pure numpy) on random polygons and float32 boxes --
    the oracle's shifted / scaled vertices are the same float64 bits (boxes thinner than one pixel included).  The
    golden was produced under numpy 2.2 (this container), whose scalar promotion (NEP 50) keeps the scale in float32:
    it pins the restatement's ``'nep50'`` variant; the ``'legacy'`` variant (NumPy 1.x, the reference's own era, what
    the device kernel follows) differs from it only by that rounding of the scale (<= 1.2e-7 relative)."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g12b_polygon_vertices.npz'))
    masks = [[g[f'obj{i}_{j}'] for j in range(int(g[f'obj{i}_parts']))] for i in range(int(g['n_obj']))]
    for size in (14, 112):
        res = rp.crop_and_resize_polygons(masks, g['boxes'], (size, size), g['inds'], scalar_promotion='nep50')
        leg = rp.crop_and_resize_polygons(masks, g['boxes'], (size, size), g['inds'])
        for i, parts in enumerate(res):
            for j, p in enumerate(parts):
                ref = g[f's{size}_roi{i}_{j}']
                assert p.dtype == np.float64 and np.array_equal(p, ref), (size, i, j)
                np.testing.assert_allclose(leg[i][j], ref, rtol=1.3e-7, atol=0)
