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
