// Mask targets from polygon annotations on the device.
//
// Reference: DynaMaskHead.get_targets (mmdet/models/roi_heads/mask_heads/dynamask_head.py:248-262) with
// gt_masks = PolygonMasks: PolygonMasks.crop_and_resize (mmdet/core/mask/structures.py:469-503) shifts and scales the
// vertices of the assigned object's polygons into the S x S box frame, to_ndarray -> polygon_to_bitmap
// (structures.py:544-552, 583-599) rasterises every part with pycocotools (cocoapi common/maskApi.c: rleFrPoly ->
// rleMerge -> rleDecode) on the host, and the bitmaps are uploaded.  Here one workgroup does that for one positive
// RoI, following rleFrPoly's arithmetic exactly (doubles, C truncation, no fused multiply-add):
//   * vertices: (int)(5 * v + .5) of the shifted / scaled vertex;
//   * boundary points: rleFrPoly walks every edge densely on the 5x grid and keeps the points where the column changes
//     and the smaller of the two columns is 5n + 2 (the only case in which its downsampled column is an integer n);
//     a work item here is (edge, target column n): for an x-major edge the pair of consecutive points is known in
//     closed form, for a y-major edge the step of the column function between 5n + 2 and 5n + 3 is located from the
//     slope and confirmed with the reference's own rounding expression.  O(edges x S) whatever the polygon's extent
//     (a literal walk of an object 50 x the box would take 10^5 steps per edge);
//   * rleFrPoly sorts the boundary positions and alternates runs, merging zero-length runs: pixel i (column-major) is
//     set iff an odd number of positions are <= i.  Positions toggle bits of an LDS bitmap, a prefix parity fills it;
//   * parts of an object are OR-ed (rleMerge with intersect = 0).
// oracle/ref_poly.py holds the literal restatement, the closed form, and the reference's known answers.
#include "common.h"

namespace {

struct PolyArgs {
  const double* verts;      // [V][2] (x, y) of all polygons of the image, float64 as the reference's annotations
  const int* poly_start;    // [P + 1] first vertex of polygon p
  const int* inst_start;    // [G + 1] first polygon of object g
  int G;
  const float* boxes;       // [N][4] float32, already clipped to the image (dynamask_head.py:253-254)
  const long long* inds;    // [N] object of each RoI
  int N, S;
  float* out;               // [N][S][S] 0 / 1
};

__device__ __forceinline__ int poly_yd(int vmin, int h) {
#pragma clang fp contract(off)
  double yd = ((double)vmin + .5) / 5.0 - .5;
  if (yd < 0) yd = 0;
  else if (yd > (double)h) yd = (double)h;
  return (int)ceil(yd);
}

__global__ __launch_bounds__(256) void polygon_target_kernel(PolyArgs a) {
#pragma clang fp contract(off)
  extern __shared__ unsigned int pl_lds[];
  const int S = a.S, HW = S * S;
  const int words = (HW + 1 + 31) >> 5;
  unsigned int* tog = pl_lds;                 // boundary positions of the current part (bit toggles)
  unsigned int* res = pl_lds + words;         // union of the parts
  unsigned int* carry = pl_lds + 2 * words;   // parity of the toggles in the words before word w
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < words; i += 256) res[i] = 0u;
  const long long g = a.inds[n];
  const float bx1 = a.boxes[4 * n], by1 = a.boxes[4 * n + 1], bx2 = a.boxes[4 * n + 2], by2 = a.boxes[4 * n + 3];
  // structures.py:486-491: the box differences are float32; under the NumPy of the reference's era (1.x) a NumPy
  // scalar combined with a Python scalar takes the Python scalar's default type, so np.maximum(x2 - x1, 1) and
  // out_w / max(w, 0.1) are float64 (NumPy >= 2 would keep float32: oracle/ref_poly.py restates both)
  const double bw = fmax((double)(bx2 - bx1), 1.0), bh = fmax((double)(by2 - by1), 1.0);
  const double w_scale = (double)S / fmax(bw, 0.1), h_scale = (double)S / fmax(bh, 0.1);
  const double ox = (double)bx1, oy = (double)by1;
  int p0 = 0, p1 = 0;
  if (g >= 0 && g < a.G) { p0 = a.inst_start[g]; p1 = a.inst_start[g + 1]; }
  for (int p = p0; p < p1; ++p) {
    const int v0 = a.poly_start[p], k = a.poly_start[p + 1] - v0;
    __syncthreads();
    for (int i = tid; i < words; i += 256) tog[i] = 0u;
    __syncthreads();
    if (k >= 1) {
      const long long items = (long long)k * S;
      for (long long it = tid; it < items; it += 256) {
        const int j = (int)(it / S), nn = (int)(it - (long long)j * S);
        const int j1 = (j + 1 == k) ? 0 : j + 1;
        const double* va = a.verts + 2 * (size_t)(v0 + j);
        const double* vb = a.verts + 2 * (size_t)(v0 + j1);
        int xs = (int)(5.0 * ((va[0] - ox) * w_scale) + .5), ys = (int)(5.0 * ((va[1] - oy) * h_scale) + .5);
        int xe = (int)(5.0 * ((vb[0] - ox) * w_scale) + .5), ye = (int)(5.0 * ((vb[1] - oy) * h_scale) + .5);
        const int dx = abs(xe - xs), dy = abs(ys - ye);
        if (dx == 0) continue;                       // the column never changes along this edge
        const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
        if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
        const int U = 5 * nn + 2;
        int vmin;
        if (dx >= dy) {
          if (U < xs || U + 1 > xe) continue;
          const double s = (double)(ye - ys) / dx;
          const int t = U - xs;
          const int va0 = (int)(ys + s * t + .5), va1 = (int)(ys + s * (t + 1) + .5);
          vmin = min(va0, va1);
        } else {
          const int lo = min(xs, xe), hi = max(xs, xe);
          if (U < lo || U + 1 > hi) continue;
          const double s = (double)(xe - xs) / dy;
          int t0 = (int)floor(((double)U + 0.5 - (double)xs) / s);
          if (s < 0) t0 += 1;
          int found = -1;
          const int c_lo = max(1, t0 - 2), c_hi = min(dy, t0 + 3);
          for (int c = c_lo; c <= c_hi; ++c) {
            const int ua = (int)(xs + s * (c - 1) + .5), ub = (int)(xs + s * c + .5);
            if ((s > 0 && ua == U && ub == U + 1) || (s < 0 && ua == U + 1 && ub == U)) { found = c; break; }
          }
          if (found < 0) continue;
          vmin = found - 1 + ys;
        }
        const int pos = nn * S + poly_yd(vmin, S);   // <= S * S
        atomicXor(&tog[pos >> 5], 1u << (pos & 31));
      }
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int c = 0;
      for (int wdx = 0; wdx < words; ++wdx) {
        carry[wdx] = c;
        c ^= (unsigned int)(__popc(tog[wdx]) & 1);
      }
    }
    __syncthreads();
    for (int wdx = tid; wdx < words; wdx += 256) {
      // bit b of the word is inside iff carry ^ parity(toggles at bits <= b): a prefix XOR inside the word
      unsigned int x = tog[wdx];
      x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
      if (carry[wdx]) x = ~x;
      res[wdx] |= x;
    }
  }
  __syncthreads();
  float* o = a.out + (size_t)n * HW;
  for (int idx = tid; idx < HW; idx += 256) {
    const int y = idx / S, x = idx - y * S;
    const int i = x * S + y;                         // maskApi.c stores masks column-major
    o[idx] = ((res[i >> 5] >> (i & 31)) & 1u) ? 1.f : 0.f;
  }
}

}  // namespace

extern "C" int dm_polygon_mask_targets(const double* verts, const int* poly_start, const int* inst_start, int num_objects,
                                       const float* boxes, const int64_t* inds, int N, int S, float* out,
                                       dm_stream_t stream) {
  if (N < 0 || S <= 0 || S > 256 || num_objects < 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!poly_start || !inst_start || !boxes || !inds || !out) return DM_ERR_INVALID_ARG;
  PolyArgs a;
  a.verts = verts; a.poly_start = poly_start; a.inst_start = inst_start; a.G = num_objects;
  a.boxes = boxes; a.inds = (const long long*)inds; a.N = N; a.S = S; a.out = out;
  const int words = (S * S + 1 + 31) / 32;
  DM_LAUNCH(polygon_target_kernel, dim3((unsigned)N), dim3(256), (size_t)3 * words * sizeof(unsigned int), (hipStream_t)stream, a);
  return dm_check_launch();
}
