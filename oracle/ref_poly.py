"""CPU oracle (TEST INFRASTRUCTURE ONLY) for polygon mask targets.

The reference turns ``PolygonMasks`` into mask targets on the host:
``PolygonMasks.crop_and_resize`` (mmdet/core/mask/structures.py:469-503) shifts and scales the polygon
vertices, ``to_ndarray`` -> ``polygon_to_bitmap`` (structures.py:544-552, 583-599) rasterises them with
pycocotools (``maskUtils.frPyObjects`` -> ``merge`` -> ``decode``).  pycocotools is a third-party dependency that is
not in the reference tree and not installed in this image (``pycocotools`` of cocoapi, unpinned in
requirements/runtime.txt): ``rle_fr_poly`` below restates the PUBLISHED algorithm of cocoapi's
``common/maskApi.c: rleFrPoly`` line by line (upsample by 5, walk every edge densely, keep the points where the
column changes, downsample, sort the boundary positions, run lengths), ``rle_decode`` its ``rleDecode``; the union
of an object's parts (``rleMerge(..., intersect=0)`` + decode) is the OR of the decoded parts.

Pinned by the reference's own known answers: the three bitmaps its tests hold for polygon rasterisation
(tests/test_masks.py:330-411 ``truth`` / ``truth1`` / ``truth2``, :449-471 the cropped ``truth``), committed as
``tests/golden/g12_polygon_truth.npz`` (``tests/golden/make_golden_poly.py`` extracts them from the test file's text).

``fr_poly_events`` is a second, closed-form derivation of the same boundary points (what the HIP kernel computes: an
edge's points are enumerated per target column instead of per upsampled step, so a polygon that reaches far outside
the box costs O(columns), not O(extent)); ``tests/test_oracle_poly.py`` checks it against the literal loop on random
polygons."""
import math

import numpy as np

SCALE = 5.0


def _trunc_int(v):
    """C's (int) of a double: truncation toward zero."""
    return int(v)


def fr_poly_points_literal(xy, h, w):
    """maskApi.c rleFrPoly up to the list of y-boundary points ``(x, y)`` (before sorting), literally."""
    k = len(xy) // 2
    x = [_trunc_int(SCALE * float(xy[2 * j]) + .5) for j in range(k)]
    y = [_trunc_int(SCALE * float(xy[2 * j + 1]) + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe = xe, xs
            ys, ye = ye, ys
        if dx >= dy:
            s = (float(ye - ys) / dx) if dx > 0 else float('nan')      # dx == dy == 0: one point, s unused below
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(_trunc_int(ys + s * t + .5) if dx > 0 else ys)
        else:
            s = float(xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(_trunc_int(xs + s * t + .5))
    pts = []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + .5) / SCALE - .5
            if math.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + .5) / SCALE - .5
            if yd < 0:
                yd = 0.0
            elif yd > h:
                yd = float(h)
            yd = math.ceil(yd)
            pts.append((int(xd), int(yd)))
    return pts


def _yd(vmin, h):
    yd = (float(vmin) + .5) / SCALE - .5
    if yd < 0:
        yd = 0.0
    elif yd > h:
        yd = float(h)
    return int(math.ceil(yd))


def fr_poly_events(xy, h, w):
    """The same boundary points in closed form, edge by edge and target column by target column.

    A point survives only where the column index changes between two consecutive upsampled points and the smaller
    of the two columns is 5n + 2 with 0 <= n <= w - 1 (``(xd + .5) / 5 - .5`` is an integer exactly then).  For an
    x-major edge consecutive points are one column apart, so the pair is (t, t + 1) with column t + xs = 5n + 2; for
    a y-major edge the column is a monotone step function of t with steps of one, and the step between 5n + 2 and
    5n + 3 is located from the edge's slope and confirmed by evaluating the reference's own rounding expression."""
    k = len(xy) // 2
    x = [_trunc_int(SCALE * float(xy[2 * j]) + .5) for j in range(k)]
    y = [_trunc_int(SCALE * float(xy[2 * j + 1]) + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    pts = []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe = xe, xs
            ys, ye = ye, ys
        if dx == 0:
            continue                                  # the column never changes on this edge
        if dx >= dy:
            s = float(ye - ys) / dx
            # columns xs .. xe = xs + dx (xs < xe after the swap); pairs (U, U + 1) with U = 5n + 2
            n_lo = max(0, -((-(xs - 2)) // 5))        # ceil((xs - 2) / 5)
            n_hi = min(w - 1, (xe - 1 - 2) // 5)      # U + 1 <= xe
            ev = []
            for n in range(n_lo, n_hi + 1):
                t = 5 * n + 2 - xs
                v0 = _trunc_int(ys + s * t + .5)
                v1 = _trunc_int(ys + s * (t + 1) + .5)
                ev.append((n, _yd(min(v0, v1), h)))
            pts.extend(reversed(ev) if flip else ev)  # emission order (irrelevant after the sort; kept for the tests)
        else:
            s = float(xe - xs) / dy
            # u(t) = (int)(xs + s t + .5), t = 0 .. dy, monotone with unit steps; rows v = t + ys
            lo, hi = min(xs, xe), max(xs, xe)
            n_lo = max(0, -((-(lo - 2)) // 5))
            n_hi = min(w - 1, (hi - 1 - 2) // 5)
            ev = []
            for n in range(n_lo, n_hi + 1):
                U = 5 * n + 2

                def u_at(t):
                    return _trunc_int(xs + s * t + .5)
                # the step between columns U and U + 1: first t with u(t) on the far side
                if s > 0:
                    t0 = int(math.floor((U + 0.5 - xs) / s))
                    t = None
                    for c in range(max(1, t0 - 2), min(dy, t0 + 3) + 1):
                        if u_at(c - 1) == U and u_at(c) == U + 1:
                            t = c
                            break
                else:
                    t0 = int(math.floor((U + 0.5 - xs) / s)) + 1
                    t = None
                    for c in range(max(1, t0 - 2), min(dy, t0 + 3) + 1):
                        if u_at(c - 1) == U + 1 and u_at(c) == U:
                            t = c
                            break
                if t is None:
                    continue
                ev.append((n, _yd(t - 1 + ys, h), t))
            ev.sort(key=lambda e: e[2])
            ev = [(n, yy) for n, yy, _ in ev]
            pts.extend(reversed(ev) if flip else ev)
    return pts


def rle_from_points(pts, h, w):
    """maskApi.c rleFrPoly, second half: boundary positions -> sorted -> run lengths (zero-length runs merged)."""
    a = sorted(int(px) * int(h) + int(py) for px, py in pts)
    a.append(h * w)
    p = 0
    for j in range(len(a)):
        t = a[j]
        a[j] -= p
        p = t
    b = [a[0]]
    j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return b


def rle_decode(counts, h, w):
    """maskApi.c rleDecode: column-major runs, starting with zeros."""
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T.copy()


def mask_from_points(pts, h, w):
    """Parity form of the same thing (what the HIP kernel does): pixel i (column-major) is set iff an odd number of
    boundary positions are <= i."""
    tog = np.zeros(h * w + 1, dtype=np.int64)
    for px, py in pts:
        tog[int(px) * h + int(py)] += 1
    par = (np.cumsum(tog)[:h * w] & 1).astype(np.uint8)
    return par.reshape(w, h).T.copy()


def polygon_to_bitmap(polygons, h, w, events=fr_poly_points_literal):
    """structures.py:583-599: union of the parts of one object."""
    out = np.zeros((h, w), dtype=np.uint8)
    for p in polygons:
        p = np.asarray(p, dtype=np.float64)
        out |= rle_decode(rle_from_points(events(p, h, w), h, w), h, w)
    return out


def crop_and_resize_polygons(masks, bboxes, out_shape, inds, scalar_promotion='legacy'):
    """structures.py:469-503 with its dtypes: ``bboxes`` float32 (``proposals_np``), the vertices float64.

    The box differences ``x2 - x1`` are float32 (two float32 scalars).  What happens next depends on the NumPy the
    reference runs under:
    * ``'legacy'`` (NumPy 1.x, the reference's own era -- mmdet 2.x pins numpy<2): a NumPy scalar combined with a
      Python scalar follows the Python scalar's default type, so ``np.maximum(x2 - x1, 1)`` and
      ``out_w / max(w, 0.1)`` are float64: scale = float64(out) / float64(float32 difference).  The product
      (polygon.hip) follows this.
    * ``'nep50'`` (NumPy >= 2): Python scalars are weak, everything stays float32 and the scale is a float32 quotient
      (6e-8 relative away from the legacy one).  Golden g12b was produced by running the reference's class in THIS
      container (numpy 2.2), so it pins this variant; tests/test_oracle_poly.py checks the restatement against it and
      the two variants against each other."""
    out_h, out_w = out_shape
    bboxes = np.asarray(bboxes, dtype=np.float32)
    res = []
    for i in range(len(bboxes)):
        x1, y1, x2, y2 = bboxes[i]
        if scalar_promotion == 'legacy':
            w = max(float(np.float32(x2 - x1)), 1.0)
            h = max(float(np.float32(y2 - y1)), 1.0)
            h_scale = np.float64(out_h) / max(h, 0.1)
            w_scale = np.float64(out_w) / max(w, 0.1)
        else:
            w = np.maximum(x2 - x1, np.float32(1))
            h = np.maximum(y2 - y1, np.float32(1))
            h_scale = np.float32(out_h) / np.maximum(h, np.float32(0.1))
            w_scale = np.float32(out_w) / np.maximum(w, np.float32(0.1))
        parts = []
        for p in masks[int(inds[i])]:
            p = np.asarray(p, dtype=np.float64).copy()
            p[0::2] -= np.float64(x1)
            p[1::2] -= np.float64(y1)
            p[0::2] *= np.float64(w_scale)
            p[1::2] *= np.float64(h_scale)
            parts.append(p)
        res.append(parts)
    return res


def polygon_mask_targets(masks, height, width, boxes, inds, size, events=fr_poly_points_literal):
    """dynamask_head.py:248-262 for ``PolygonMasks``: clip the boxes to the image, crop_and_resize, rasterise."""
    b = np.asarray(boxes, dtype=np.float32).copy()
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, width)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, height)
    polys = crop_and_resize_polygons(masks, b, (size, size), inds)
    if not polys:
        return np.zeros((0, size, size), dtype=np.float32)
    return np.stack([polygon_to_bitmap(p, size, size, events) for p in polys]).astype(np.float32)
