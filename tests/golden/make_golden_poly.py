"""Extracts the polygon rasterisation known answers the reference's own tests hold
(/root/reference/tests/test_masks.py: test_polygon_mask_rescale / _resize / _crop) into
tests/golden/g12_polygon_truth.npz.  Only DATA is taken: the polygon vertex arrays and the expected bitmaps,
which in the reference were produced by pycocotools (not installed here).  Run in the build container:
    python tests/golden/make_golden_poly.py"""
import os
import re

import numpy as np

REF = '/root/reference/tests/test_masks.py'
src = open(REF).read()


def body(name):
    m = re.search(r'def %s\(\):\n(.*?)\n\n\ndef ' % name, src, re.S)
    return m.group(1)


def arrays(text, var):
    """every ``var = np.array(...)`` literal of a test body, evaluated on its own"""
    out = []
    for m in re.finditer(r'\b%s = (np\.array\(.*?\))\n' % re.escape(var), text, re.S):
        expr = m.group(1)
        depth, end, seen = 0, None, False
        for i, ch in enumerate(expr):
            depth += ch == '('
            depth -= ch == ')'
            seen = seen or ch == '('
            if seen and depth == 0:
                end = i + 1
                break
        out.append(eval(expr[:end].replace('np.float)', 'np.float64)').replace('dtype=np.float,', 'dtype=np.float64,'),
                        {'np': np}))
    return out


resize = body('test_polygon_mask_resize')
truth1 = arrays(resize, 'truth1')[0]
truth2 = arrays(resize, 'truth2')[0]
poly1 = np.array([1, 1, 3, 1, 4, 3, 2, 4, 1, 3], dtype=np.float64)                 # raw_masks1, test_masks.py:371
poly2a, poly2b = np.array([0., 0., 1., 0., 1., 1.]), np.array([1., 1., 2., 1., 2., 2., 1., 2.])   # raw_masks2, :390-393
assert '[1, 1, 3, 1, 4, 3, 2, 4, 1, 3]' in resize and '[0., 0., 1., 0., 1., 1.]' in resize
rescale = body('test_polygon_mask_rescale')
truth_rescale = arrays(rescale, 'truth')[0]
crop = body('test_polygon_mask_crop')
truth_crop = arrays(crop, 'truth')[0]
poly_crop = np.array([1., 3., 5., 1., 5., 6., 1, 6])                                # :462
assert '[1., 3., 5., 1., 5., 6., 1, 6]' in crop
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g12_polygon_truth.npz')
np.savez_compressed(out, poly1=poly1, truth1=truth1.astype(np.uint8), poly2a=poly2a, poly2b=poly2b,
                    truth2=truth2.astype(np.uint8), truth_rescale=truth_rescale.astype(np.uint8),
                    poly_crop=poly_crop, crop_bbox=np.array([0, 0, 3, 4], dtype=np.float64), truth_crop=truth_crop.astype(np.uint8))
print('wrote', out, truth1.shape, truth2.shape, truth_rescale.shape, truth_crop.shape)


def vertex_transform_golden():
    """Second fixture: the vertex arithmetic of ``PolygonMasks.crop_and_resize`` (structures.py:469-503) from the
    REFERENCE's own class, run here (that method is pure numpy; only ``to_ndarray`` needs pycocotools).  Pins the
    dtypes of the oracle's ``crop_and_resize_polygons`` (float32 box / scale arithmetic, float64 vertices)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden_train as mgt
    st = mgt.load_train_reference()['structures']
    rng = np.random.default_rng(41)
    masks = []
    for _ in range(5):
        parts = []
        for _ in range(int(rng.integers(1, 3))):
            k = int(rng.integers(3, 8))
            parts.append(rng.uniform(0, 90, 2 * k))
        masks.append(parts)
    pm = st.PolygonMasks(masks, 80, 100)
    n = 12
    x1, y1 = rng.uniform(0, 80, n), rng.uniform(0, 60, n)
    boxes = np.stack([x1, y1, x1 + rng.uniform(0.05, 40, n), y1 + rng.uniform(0.05, 40, n)], 1).astype(np.float32)
    inds = rng.integers(0, 5, n)
    out = {'boxes': boxes, 'inds': inds, 'n_obj': np.array(len(masks))}
    for i, parts in enumerate(masks):
        out[f'obj{i}_parts'] = np.array(len(parts))
        for j, p in enumerate(parts):
            out[f'obj{i}_{j}'] = p
    for size in (14, 112):
        res = pm.crop_and_resize(boxes, (size, size), inds)
        for i, parts in enumerate(res.masks):
            for j, p in enumerate(parts):
                out[f's{size}_roi{i}_{j}'] = np.asarray(p)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g12b_polygon_vertices.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, len(out), 'arrays')


vertex_transform_golden()
