// Bbox branch post-processing (the two fully connected layers and the
// two predictors of Shared2FCBBoxHead are plain GEMMs and go to the library):
//   K20  softmax over the class logits + DeltaXYWH decode + clip + rescale
//        (BBoxHead.get_bboxes, roi_heads/bbox_heads/bbox_head.py:186-217;
//         delta2bbox, core/bbox/coder/delta_xywh_bbox_coder.py:165-204)
//   K21  NMS suppression matrix for score-sorted boxes (mmcv.ops.nms, called through
//        batched_nms by multiclass_nms, core/post_processing/bbox_nms.py:5-68) -- the
//        classic 64 x 64 tile bitmask; the greedy pass over the rows is host code
//        (dm_nms_reduce), as in the reference's extension.
#include "common.h"

namespace {

struct DecodeArgs {
  const float* rois;      // [N, roi_stride] (x1 at column roi_x0)
  int roi_stride, roi_x0;
  const float* cls_score; // [N, NC + 1] or null
  const float* bbox_pred; // [N, 4 * NB] (NB = NC, or 1 if class agnostic) or null
  int N, NC, NB;
  float mean[4], std[4];
  float max_ratio;
  float clip_w, clip_h;   // <= 0: no clipping
  float inv_sx, inv_sy;   // rescale: boxes / scale_factor (1 if none)
  float* scores;          // [N, NC + 1]
  float* bboxes;          // [N, 4 * NB]
};

__global__ __launch_bounds__(128) void bbox_decode_kernel(DecodeArgs a) {
  __shared__ float red[2];
  const int i = blockIdx.x;
  const int t = threadIdx.x;
  if (a.cls_score) {
    const float* s = a.cls_score + (size_t)i * (a.NC + 1);
    float m = -INFINITY;
    for (int c = t; c <= a.NC; c += 128) m = fmaxf(m, s[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(red[0], red[1]);
    __syncthreads();
    float sum = 0.f;
    for (int c = t; c <= a.NC; c += 128) sum += expf(s[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1];
    for (int c = t; c <= a.NC; c += 128) a.scores[(size_t)i * (a.NC + 1) + c] = expf(s[c] - m) / sum;
  }
  const float* r = a.rois + (size_t)i * a.roi_stride + a.roi_x0;
  const float rx1 = r[0], ry1 = r[1], rx2 = r[2], ry2 = r[3];
  for (int c = t; c < a.NB; c += 128) {
    float x1 = rx1, y1 = ry1, x2 = rx2, y2 = ry2;
    if (a.bbox_pred) {
      const float* d = a.bbox_pred + ((size_t)i * a.NB + c) * 4;
      const float dx = d[0] * a.std[0] + a.mean[0];
      const float dy = d[1] * a.std[1] + a.mean[1];
      float dw = d[2] * a.std[2] + a.mean[2];
      float dh = d[3] * a.std[3] + a.mean[3];
      dw = fminf(fmaxf(dw, -a.max_ratio), a.max_ratio);
      dh = fminf(fmaxf(dh, -a.max_ratio), a.max_ratio);
      const float px = (rx1 + rx2) * 0.5f, py = (ry1 + ry2) * 0.5f;
      const float pw = rx2 - rx1, ph = ry2 - ry1;
      const float gw = pw * expf(dw), gh = ph * expf(dh);
      const float gx = px + pw * dx, gy = py + ph * dy;
      x1 = gx - gw * 0.5f;
      y1 = gy - gh * 0.5f;
      x2 = gx + gw * 0.5f;
      y2 = gy + gh * 0.5f;
    }
    if (a.clip_w > 0.f) {
      x1 = fminf(fmaxf(x1, 0.f), a.clip_w);
      x2 = fminf(fmaxf(x2, 0.f), a.clip_w);
      y1 = fminf(fmaxf(y1, 0.f), a.clip_h);
      y2 = fminf(fmaxf(y2, 0.f), a.clip_h);
    }
    float* o = a.bboxes + ((size_t)i * a.NB + c) * 4;
    o[0] = x1 * a.inv_sx;
    o[1] = y1 * a.inv_sy;
    o[2] = x2 * a.inv_sx;
    o[3] = y2 * a.inv_sy;
  }
}

// mask[i][w] bit b set <=> box j = 64*w + b (j > i) overlaps box i by more than thr.
// grid = (col tiles, row tiles) of 64 boxes; only the upper triangle does work.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int M, float thr, float off,
                                                      unsigned long long* __restrict__ mask, int words) {
  __shared__ float cb[64 * 4];
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  if (blockIdx.x < blockIdx.y) return;
  const int t = threadIdx.x;
  const int ncol = min(64, M - col0);
  if (t < ncol) {
    cb[t * 4 + 0] = boxes[(size_t)(col0 + t) * 4 + 0];
    cb[t * 4 + 1] = boxes[(size_t)(col0 + t) * 4 + 1];
    cb[t * 4 + 2] = boxes[(size_t)(col0 + t) * 4 + 2];
    cb[t * 4 + 3] = boxes[(size_t)(col0 + t) * 4 + 3];
  }
  __syncthreads();
  const int i = row0 + t;
  if (i >= M) return;
  const float x1 = boxes[(size_t)i * 4 + 0], y1 = boxes[(size_t)i * 4 + 1];
  const float x2 = boxes[(size_t)i * 4 + 2], y2 = boxes[(size_t)i * 4 + 3];
  const float area = (x2 - x1 + off) * (y2 - y1 + off);
  unsigned long long bits = 0;
  const int start = (row0 == col0) ? t + 1 : 0;
  for (int j = start; j < ncol; ++j) {
    const float bx1 = cb[j * 4 + 0], by1 = cb[j * 4 + 1], bx2 = cb[j * 4 + 2], by2 = cb[j * 4 + 3];
    const float w = fmaxf(fminf(x2, bx2) - fmaxf(x1, bx1) + off, 0.f);
    const float h = fmaxf(fminf(y2, by2) - fmaxf(y1, by1) + off, 0.f);
    const float inter = w * h;
    const float barea = (bx2 - bx1 + off) * (by2 - by1 + off);
    const float iou = inter / (area + barea - inter);
    if (iou > thr) bits |= 1ull << j;
  }
  mask[(size_t)i * words + blockIdx.x] = bits;
}

}  // namespace

extern "C" int dm_bbox_decode(const float* rois, int roi_stride, int roi_x0, const float* cls_score,
                              const float* bbox_pred, int N, int num_classes, int class_agnostic, const float* means,
                              const float* stds, float wh_ratio_clip, float clip_h, float clip_w, float scale_x,
                              float scale_y, float* scores, float* bboxes, dm_stream_t stream) {
  if (N < 0 || num_classes <= 0 || roi_stride < 4 || roi_x0 < 0 || roi_x0 + 4 > roi_stride) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!rois || !bboxes || !means || !stds || (cls_score && !scores)) return DM_ERR_INVALID_ARG;
  if (!(wh_ratio_clip > 0.f) || !(scale_x > 0.f) || !(scale_y > 0.f)) return DM_ERR_INVALID_ARG;
  DecodeArgs a;
  a.rois = rois; a.roi_stride = roi_stride; a.roi_x0 = roi_x0; a.cls_score = cls_score; a.bbox_pred = bbox_pred;
  a.N = N; a.NC = num_classes; a.NB = class_agnostic ? 1 : num_classes;
  for (int k = 0; k < 4; ++k) { a.mean[k] = means[k]; a.std[k] = stds[k]; }
  a.max_ratio = fabsf(logf(wh_ratio_clip));
  a.clip_w = clip_w; a.clip_h = clip_h;
  a.inv_sx = 1.0f / scale_x; a.inv_sy = 1.0f / scale_y;
  a.scores = scores; a.bboxes = bboxes;
  DM_LAUNCH(bbox_decode_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}

extern "C" int dm_nms_mask(const float* boxes_sorted, int M, float iou_threshold, int offset, unsigned long long* mask,
                           dm_stream_t stream) {
  if (M < 0) return DM_ERR_INVALID_ARG;
  if (M == 0) return DM_OK;
  if (!boxes_sorted || !mask) return DM_ERR_INVALID_ARG;
  const int words = dm_ceil_div(M, 64);
  hipError_t e = hipMemsetAsync(mask, 0, (size_t)M * words * sizeof(unsigned long long), (hipStream_t)stream);
  if (e != hipSuccess) return DM_ERR_LAUNCH;
  DM_LAUNCH(nms_mask_kernel, dim3(words, words), dim3(64), 0, (hipStream_t)stream, boxes_sorted, M, iou_threshold,
            offset ? 1.f : 0.f, mask, words);
  return dm_check_launch();
}

// Greedy pass over the suppression matrix (host code, like the reference extension's
// CPU tail): walks the boxes in score order, keeps a box unless an earlier kept box
// suppresses it.  Returns the number kept (indices into the sorted order, ascending).
extern "C" int dm_nms_reduce(const unsigned long long* mask_host, int M, int* keep, int max_keep) {
  if (M < 0 || (M > 0 && (!mask_host || !keep))) return 0;
  const int words = (M + 63) / 64;
  unsigned long long removed[1024];
  unsigned long long* rem = removed;
  unsigned long long* heap = nullptr;
  if (words > 1024) {
    heap = new unsigned long long[words];
    rem = heap;
  }
  for (int w = 0; w < words; ++w) rem[w] = 0;
  int n = 0;
  for (int i = 0; i < M; ++i) {
    if (rem[i >> 6] & (1ull << (i & 63))) continue;
    if (max_keep >= 0 && n >= max_keep) break;
    keep[n++] = i;
    const unsigned long long* row = mask_host + (size_t)i * words;
    for (int w = i >> 6; w < words; ++w) rem[w] |= row[w];
  }
  delete[] heap;
  return n;
}
