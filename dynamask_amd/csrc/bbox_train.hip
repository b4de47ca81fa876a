// Training side of the bbox branch and of the RoI sampling that feeds the mask path
// (second half; the caller of the path in training):
//   K23  bbox_overlaps        core/bbox/iou_calculators/iou2d_calculator.py:37-131 (non-aligned)
//   K24  MaxIoUAssigner       core/bbox/assigners/max_iou_assigner.py:129-212 (assign_wrt_overlaps)
//   K25  bbox2delta           core/bbox/coder/delta_xywh_bbox_coder.py:74-116
//   K26  softmax cross entropy + top-1 accuracy, fused forward/backward
//        losses/cross_entropy_loss.py:9-38, losses/utils.py:26-52, losses/accuracy.py:4-49
//   K27  L1 loss on the positive rows' class columns, fused forward/backward
//        roi_heads/bbox_heads/bbox_head.py:159-182, losses/smooth_l1_loss.py:29-42
//   K28  sum of squares / clip coefficient of the flat gradient (clip_grad_norm_, config :274)
// All of it is small, latency-bound integer/compare work: one thread per element, reductions
// in a fixed order (the results do not depend on the launch geometry or on atomics).
// Contraction is off so that products and sums round exactly as the reference's separate
// torch ops do: IoUs are compared with thresholds and must land on the same side.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* __restrict__ b1, int n1,
                                                            const float* __restrict__ b2, int n2, int iof, float eps,
                                                            float* __restrict__ out) {
#pragma clang fp contract(off)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n1 * n2) return;
  const int i = (int)(idx / n2), j = (int)(idx % n2);
  const float ax1 = b1[4 * i], ay1 = b1[4 * i + 1], ax2 = b1[4 * i + 2], ay2 = b1[4 * i + 3];
  const float bx1 = b2[4 * j], by1 = b2[4 * j + 1], bx2 = b2[4 * j + 2], by2 = b2[4 * j + 3];
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float overlap = w * h;
  const float area1 = (ax2 - ax1) * (ay2 - ay1);
  float uni = area1;
  if (!iof) {
    const float area2 = (bx2 - bx1) * (by2 - by1);
    uni = area1 + area2 - overlap;
  }
  out[idx] = overlap / fmaxf(uni, eps);
}

// per gt row: max over the boxes and the first index reaching it
__global__ __launch_bounds__(256) void gt_row_max_kernel(const float* __restrict__ ov, int k, int n,
                                                         float* __restrict__ gt_max, int* __restrict__ gt_argmax) {
  __shared__ float sm[256];
  __shared__ int si[256];
  const int i = blockIdx.x, t = threadIdx.x;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int j = t; j < n; j += 256) {
    const float v = ov[(size_t)i * n + j];
    if (v > m) { m = v; mi = j; }
  }
  sm[t] = m; si[t] = mi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      const float o = sm[t + s];
      const int oi = si[t + s];
      if (o > sm[t] || (o == sm[t] && oi < si[t])) { sm[t] = o; si[t] = oi; }
    }
    __syncthreads();
  }
  if (t == 0) { gt_max[i] = sm[0]; gt_argmax[i] = si[0]; }
}

struct AssignArgs {
  const float* ov;          // [k, n]
  int k, n;
  float pos_thr, neg_lo, neg_hi, min_pos_iou;
  int match_low_quality, gt_max_assign_all;
  const float* gt_max;
  const int* gt_argmax;
  const int64_t* gt_labels; // [k] or null
  int64_t* gt_inds;         // [n]
  float* max_overlaps;      // [n]
  int64_t* labels;          // [n] or null
};

__global__ __launch_bounds__(256) void max_iou_assign_kernel(AssignArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.n) return;
  float m = -INFINITY;
  int am = 0;
  for (int i = 0; i < a.k; ++i) {
    const float v = a.ov[(size_t)i * a.n + j];
    if (v > m) { m = v; am = i; }
  }
  long long g = -1;                                        // 1. default
  if (m >= a.neg_lo && m < a.neg_hi) g = 0;                 // 2. negatives
  if (m >= a.pos_thr) g = am + 1;                           // 3. positives
  if (a.match_low_quality) {                                // 4. every gt keeps its best boxes; later gts overwrite
    for (int i = 0; i < a.k; ++i) {
      const float gm = a.gt_max[i];
      if (gm >= a.min_pos_iou) {
        if (a.gt_max_assign_all ? (a.ov[(size_t)i * a.n + j] == gm) : (a.gt_argmax[i] == j)) g = i + 1;
      }
    }
  }
  a.gt_inds[j] = g;
  a.max_overlaps[j] = m;
  if (a.labels) a.labels[j] = g > 0 ? a.gt_labels[g - 1] : -1;
}

struct EncodeArgs {
  const float* p;
  const float* g;
  int n;
  float mean[4], std[4];
  float* out;
};

__global__ __launch_bounds__(256) void bbox_encode_kernel(EncodeArgs a) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float* p = a.p + 4 * (size_t)i;
  const float* g = a.g + 4 * (size_t)i;
  const float px = (p[0] + p[2]) * 0.5f, py = (p[1] + p[3]) * 0.5f, pw = p[2] - p[0], ph = p[3] - p[1];
  const float gx = (g[0] + g[2]) * 0.5f, gy = (g[1] + g[3]) * 0.5f, gw = g[2] - g[0], gh = g[3] - g[1];
  float d[4] = {(gx - px) / pw, (gy - py) / ph, logf(gw / pw), logf(gh / ph)};
#pragma unroll
  for (int c = 0; c < 4; ++c) a.out[4 * (size_t)i + c] = (d[c] - a.mean[c]) / a.std[c];
}

// one wave per row: loss_i = w_i * (logsumexp(s_i) - s_i[label_i]); grad row = scale * w_i * (softmax - onehot)
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(const float* __restrict__ score,
                                                              const int64_t* __restrict__ labels,
                                                              const float* __restrict__ weight, int N, int C, float scale,
                                                              float* __restrict__ row_loss, float* __restrict__ row_correct,
                                                              float* __restrict__ grad) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float* s = score + (size_t)row * C;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = s[c];
    if (v > m) { m = v; mi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(s[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const int64_t lab = labels[row];
  const float w = weight ? weight[row] : 1.f;
  if (lane == 0) {
    row_loss[row] = w * (logf(sum) + m - s[lab]);
    row_correct[row] = (mi == (int)lab) ? 1.f : 0.f;
  }
  if (grad) {
    const float inv = 1.f / sum;
    for (int c = lane; c < C; c += 64) {
      const float p = expf(s[c] - m) * inv;
      grad[(size_t)row * C + c] = scale * w * (p - (c == (int)lab ? 1.f : 0.f));
    }
  }
}

// fixed-order sum of n floats by one workgroup: out[0] = scale_a * sum(a), out[1] = scale_b * sum(b) (b optional)
__global__ __launch_bounds__(256) void ordered_sum2_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                           float scale_a, float scale_b, float* __restrict__ out_a,
                                                           float* __restrict__ out_b) {
  __shared__ float sa[256], sb[256];
  const int t = threadIdx.x;
  float xa = 0.f, xb = 0.f;
  for (int i = t; i < n; i += 256) {
    xa += a[i];
    if (b) xb += b[i];
  }
  sa[t] = xa; sb[t] = xb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) { sa[t] += sa[t + s]; sb[t] += sb[t + s]; }
    __syncthreads();
  }
  if (t == 0) {
    out_a[0] = sa[0] * scale_a;
    if (b && out_b) out_b[0] = sb[0] * scale_b;
  }
}

// thread per row: |pred[row, label] - target| * weight over the 4 coordinates of a positive row
__global__ __launch_bounds__(256) void l1_rows_kernel(const float* __restrict__ pred, const int64_t* __restrict__ labels,
                                                      const float* __restrict__ target, const float* __restrict__ weight,
                                                      int N, int NB, int num_classes, float scale,
                                                      float* __restrict__ row_loss, float* __restrict__ grad) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  const int64_t lab = labels[row];
  float l = 0.f;
  if (lab >= 0 && lab < num_classes) {
    const int col = NB == 1 ? 0 : (int)lab;
    const size_t base = ((size_t)row * NB + col) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = pred[base + c] - target[4 * (size_t)row + c];
      const float w = weight[4 * (size_t)row + c];
      l += fabsf(d) * w;
      if (grad) grad[base + c] = scale * w * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
  }
  row_loss[row] = l;
}

// two-stage fixed-order sum of squares: partial[b] per workgroup, then one workgroup
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n,
                                                            float* __restrict__ partial) {
  __shared__ float sm[256];
  const int t = threadIdx.x;
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (long long i = lo + t; i < hi; i += 256) s += x[i] * x[i];
  sm[t] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (t < k) sm[t] += sm[t + k];
    __syncthreads();
  }
  if (t == 0) partial[blockIdx.x] = sm[0];
}

// x *= min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_, norm_type 2)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ x, long long n, const float* __restrict__ sumsq,
                                                         float max_norm) {
  const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
  if (!(coef < 1.f)) return;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= coef;
}

// max_iou_assigner.py:107-118: overlaps[:, n] = -1 for every box n whose largest IoF with an ignore region is > thr
__global__ __launch_bounds__(256) void ignore_columns_kernel(float* __restrict__ overlaps, int G, int N,
                                                             const float* __restrict__ iof, int I, int boxes_major, float thr) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float m = -INFINITY;
  for (int i = 0; i < I; ++i) m = fmaxf(m, boxes_major ? iof[(size_t)n * I + i] : iof[(size_t)i * N + n]);
  if (m > thr)
    for (int g = 0; g < G; ++g) overlaps[(size_t)g * N + n] = -1.f;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n, float factor) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= factor;
}

}  // namespace

extern "C" int dm_bbox_overlaps(const float* bboxes1, int n1, const float* bboxes2, int n2, int mode_iof, float eps,
                                float* out, dm_stream_t stream) {
  if (n1 < 0 || n2 < 0) return DM_ERR_INVALID_ARG;
  if ((long long)n1 * n2 == 0) return DM_OK;
  if (!bboxes1 || !bboxes2 || !out) return DM_ERR_INVALID_ARG;
  const long long total = (long long)n1 * n2;
  DM_LAUNCH(bbox_overlaps_kernel, dim3(dm_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, bboxes1, n1, bboxes2, n2,
            mode_iof, eps, out);
  return dm_check_launch();
}

extern "C" int dm_max_iou_assign(const float* overlaps, int num_gts, int num_bboxes, float pos_iou_thr, float neg_iou_lo,
                                 float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                                 const int64_t* gt_labels, float* scratch, int64_t* gt_inds, float* max_overlaps,
                                 int64_t* labels, dm_stream_t stream) {
  if (num_gts <= 0 || num_bboxes <= 0) return DM_ERR_INVALID_ARG;      // the empty cases are the caller's (fills)
  if (!overlaps || !scratch || !gt_inds || !max_overlaps || (labels && !gt_labels)) return DM_ERR_INVALID_ARG;
  float* gt_max = scratch;
  int* gt_argmax = reinterpret_cast<int*>(scratch + num_gts);
  DM_LAUNCH(gt_row_max_kernel, dim3(num_gts), dim3(256), 0, (hipStream_t)stream, overlaps, num_gts, num_bboxes, gt_max,
            gt_argmax);
  AssignArgs a;
  a.ov = overlaps; a.k = num_gts; a.n = num_bboxes;
  a.pos_thr = pos_iou_thr; a.neg_lo = neg_iou_lo; a.neg_hi = neg_iou_hi; a.min_pos_iou = min_pos_iou;
  a.match_low_quality = match_low_quality; a.gt_max_assign_all = gt_max_assign_all;
  a.gt_max = gt_max; a.gt_argmax = gt_argmax; a.gt_labels = gt_labels;
  a.gt_inds = gt_inds; a.max_overlaps = max_overlaps; a.labels = labels;
  DM_LAUNCH(max_iou_assign_kernel, dim3(dm_ceil_div(num_bboxes, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// K29  RoI sampling (what BaseSampler.sample + RandomSampler + SamplingResult produce,
//      core/bbox/samplers/base_sampler.py:35-101, random_sampler.py:31-75, sampling_result.py:21-49),
//      formulated as a selection by key: every candidate box carries a random key; a class
//      (positives: gt_inds > 0, negatives: gt_inds == 0) larger than its quota keeps the quota's
//      smallest keys (ties: lower box index), a smaller one is kept whole; the kept boxes leave in
//      ascending index order (the reference's `.unique()`), positives with their gt box / label /
//      flag gathered alongside.  One workgroup: M is a few thousand boxes, the work is compares.
// ---------------------------------------------------------------------------------------------
namespace {

struct SampleArgs {
  const int64_t* gt_inds; const float* bboxes; const float* gt_bboxes; const int64_t* labels;
  const float* pos_keys; const float* neg_keys;
  int M, n_prepended, num, quota_pos, keys_by_class_rank;
  double neg_pos_ub;
  int* crank; float* ekey; int* keep;            // scratch, M each
  int64_t* pos_inds; int64_t* neg_inds; int32_t* counts;
  float* pos_bboxes; float* neg_bboxes; float* pos_gt_bboxes;
  int64_t* pos_assigned_gt_inds; int64_t* pos_gt_labels; uint8_t* pos_is_gt;
};

constexpr int SMP_T = 1024;

// exclusive scan of (a, b) over the workgroup; returns the totals through ta / tb
__device__ __forceinline__ void block_scan2(int a, int b, int& ea, int& eb, int& ta, int& tb, int (*part)[2]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  if (lane == 63) { part[wave][0] = ia; part[wave][1] = ib; }
  __syncthreads();
  int oa = 0, ob = 0, sa = 0, sb = 0;
  for (int w = 0; w < SMP_T / 64; ++w) {
    if (w == wave) { oa = sa; ob = sb; }
    sa += part[w][0]; sb += part[w][1];
  }
  ea = oa + ia - a; eb = ob + ib - b; ta = sa; tb = sb;
  __syncthreads();
}

__global__ __launch_bounds__(SMP_T) void random_sample_kernel(SampleArgs a) {
  __shared__ int part[SMP_T / 64][2];
  __shared__ float tkey[SMP_T];
  __shared__ signed char tcls[SMP_T];
  const int t = threadIdx.x;
  // (1) class ranks in index order, class sizes
  int n_pos = 0, n_neg = 0;
  for (int base = 0; base < a.M; base += SMP_T) {
    const int i = base + t;
    const int64_t g = i < a.M ? a.gt_inds[i] : -1;
    int ep, en, tp, tn;
    block_scan2(g > 0, g == 0, ep, en, tp, tn, part);
    if (i < a.M) {
      const int r = g > 0 ? n_pos + ep : n_neg + en;
      a.crank[i] = r;
      float k = 0.f;
      if (g > 0) k = a.pos_keys[a.keys_by_class_rank ? r : i];
      else if (g == 0) k = a.neg_keys[a.keys_by_class_rank ? r : i];
      a.ekey[i] = k;
    }
    n_pos += tp; n_neg += tn;
  }
  // (2) quotas
  const int k_pos = min(n_pos, a.quota_pos);
  int quota_neg = a.num - k_pos;
  if (a.neg_pos_ub >= 0.0) {
    const int ub = (int)(a.neg_pos_ub * (double)max(1, k_pos));
    quota_neg = min(quota_neg, ub);
  }
  const int k_neg = min(n_neg, max(quota_neg, 0));
  __syncthreads();          // crank / ekey of every box are visible to the workgroup below
  // (3) a class over its quota keeps the boxes whose key rank is below the quota
  for (int base = 0; base < a.M; base += SMP_T) {
    const int i = base + t;
    const int64_t g = i < a.M ? a.gt_inds[i] : -1;
    const int cls = g > 0 ? 1 : (g == 0 ? 0 : -1);
    const bool contested = cls == 1 ? n_pos > k_pos : (cls == 0 ? n_neg > k_neg : false);
    const float key = i < a.M ? a.ekey[i] : 0.f;
    int below = 0;
    const bool any = __syncthreads_or(contested);
    if (any) {
      for (int jb = 0; jb < a.M; jb += SMP_T) {
        const int j = jb + t;
        if (j < a.M) {
          const int64_t gj = a.gt_inds[j];
          tkey[t] = a.ekey[j];
          tcls[t] = gj > 0 ? 1 : (gj == 0 ? 0 : -1);
        } else {
          tcls[t] = -2;
        }
        __syncthreads();
        if (contested) {
          const int lim = min(SMP_T, a.M - jb);
          for (int u = 0; u < lim; ++u) {
            const float ku = tkey[u];
            below += (tcls[u] == cls) & ((ku < key) | ((ku == key) & (jb + u < i)));
          }
        }
        __syncthreads();
      }
    }
    if (i < a.M) a.keep[i] = cls < 0 ? 0 : (contested ? below < (cls == 1 ? k_pos : k_neg) : 1);
  }
  __syncthreads();
  // (4) kept boxes leave in index order
  int o_pos = 0, o_neg = 0;
  for (int base = 0; base < a.M; base += SMP_T) {
    const int i = base + t;
    const int64_t g = i < a.M ? a.gt_inds[i] : -1;
    const int kp = i < a.M ? a.keep[i] : 0;
    int ep, en, tp, tn;
    block_scan2(kp && g > 0, kp && g == 0, ep, en, tp, tn, part);
    if (kp && g > 0) {
      const int o = o_pos + ep;
      a.pos_inds[o] = i;
      a.pos_assigned_gt_inds[o] = g - 1;
      a.pos_is_gt[o] = i < a.n_prepended;
      if (a.pos_gt_labels) a.pos_gt_labels[o] = a.labels[i];
      for (int c = 0; c < 4; ++c) {
        a.pos_bboxes[4 * o + c] = a.bboxes[4 * (size_t)i + c];
        a.pos_gt_bboxes[4 * o + c] = a.gt_bboxes[4 * (size_t)(g - 1) + c];
      }
    } else if (kp && g == 0) {
      const int o = o_neg + en;
      a.neg_inds[o] = i;
      for (int c = 0; c < 4; ++c) a.neg_bboxes[4 * o + c] = a.bboxes[4 * (size_t)i + c];
    }
    o_pos += tp; o_neg += tn;
  }
  if (t == 0) { a.counts[0] = o_pos; a.counts[1] = o_neg; a.counts[2] = n_pos; a.counts[3] = n_neg; }
}

}  // namespace

extern "C" int dm_random_sample(const int64_t* gt_inds, const float* bboxes, int M, int n_prepended,
                                const float* gt_bboxes, int num_gts, const int64_t* labels, const float* pos_keys,
                                const float* neg_keys, int keys_by_class_rank, int num, int quota_pos, double neg_pos_ub,
                                int32_t* scratch, int64_t* pos_inds, int64_t* neg_inds, int32_t* counts,
                                float* pos_bboxes, float* neg_bboxes, float* pos_gt_bboxes,
                                int64_t* pos_assigned_gt_inds, int64_t* pos_gt_labels, uint8_t* pos_is_gt,
                                dm_stream_t stream) {
  if (M <= 0 || num <= 0 || quota_pos < 0 || quota_pos > num || n_prepended < 0 || n_prepended > M || num_gts < 0)
    return DM_ERR_INVALID_ARG;
  if (!gt_inds || !bboxes || !pos_keys || !neg_keys || !scratch || !pos_inds || !neg_inds || !counts || !pos_bboxes ||
      !neg_bboxes || !pos_gt_bboxes || !pos_assigned_gt_inds || !pos_is_gt || (pos_gt_labels && !labels) ||
      (num_gts > 0 && !gt_bboxes))
    return DM_ERR_INVALID_ARG;
  SampleArgs a;
  a.gt_inds = gt_inds; a.bboxes = bboxes; a.gt_bboxes = gt_bboxes; a.labels = labels;
  a.pos_keys = pos_keys; a.neg_keys = neg_keys;
  a.M = M; a.n_prepended = n_prepended; a.num = num; a.quota_pos = quota_pos; a.keys_by_class_rank = keys_by_class_rank;
  a.neg_pos_ub = neg_pos_ub;
  a.crank = scratch; a.ekey = reinterpret_cast<float*>(scratch + M); a.keep = scratch + 2 * (size_t)M;
  a.pos_inds = pos_inds; a.neg_inds = neg_inds; a.counts = counts;
  a.pos_bboxes = pos_bboxes; a.neg_bboxes = neg_bboxes; a.pos_gt_bboxes = pos_gt_bboxes;
  a.pos_assigned_gt_inds = pos_assigned_gt_inds; a.pos_gt_labels = pos_gt_labels; a.pos_is_gt = pos_is_gt;
  DM_LAUNCH(random_sample_kernel, dim3(1), dim3(SMP_T), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}

extern "C" int dm_bbox_encode(const float* proposals, const float* gt, int n, const float* means, const float* stds,
                              float* deltas, dm_stream_t stream) {
  if (n < 0) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  if (!proposals || !gt || !means || !stds || !deltas) return DM_ERR_INVALID_ARG;
  EncodeArgs a;
  a.p = proposals; a.g = gt; a.n = n; a.out = deltas;
  for (int c = 0; c < 4; ++c) { a.mean[c] = means[c]; a.std[c] = stds[c]; }
  DM_LAUNCH(bbox_encode_kernel, dim3(dm_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}

extern "C" int dm_softmax_ce_fwd_bwd(const float* cls_score, const int64_t* labels, const float* weight, int N, int C,
                                     float scale, float* scratch, float* loss, float* correct, float* grad,
                                     dm_stream_t stream) {
  if (N <= 0 || C <= 0 || !cls_score || !labels || !scratch || !loss) return DM_ERR_INVALID_ARG;
  float* row_loss = scratch;
  float* row_correct = scratch + N;
  DM_LAUNCH(softmax_ce_rows_kernel, dim3(dm_ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, cls_score, labels, weight, N,
            C, scale, row_loss, row_correct, grad);
  DM_LAUNCH(ordered_sum2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss,
            (const float*)(correct ? row_correct : nullptr), N, scale, 100.0f / (float)N, loss, correct);
  return dm_check_launch();
}

extern "C" int dm_l1_loss_fwd_bwd(const float* bbox_pred, const int64_t* labels, const float* targets, const float* weights,
                                  int N, int num_boxes_per_row, int num_classes, float scale, float* scratch, float* loss,
                                  float* grad, dm_stream_t stream) {
  if (N <= 0 || num_boxes_per_row <= 0 || !bbox_pred || !labels || !targets || !weights || !scratch || !loss)
    return DM_ERR_INVALID_ARG;
  if (grad) {
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)N * num_boxes_per_row * 4 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return DM_ERR_LAUNCH;
  }
  DM_LAUNCH(l1_rows_kernel, dim3(dm_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, bbox_pred, labels, targets, weights,
            N, num_boxes_per_row, num_classes, scale, scratch, grad);
  DM_LAUNCH(ordered_sum2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, (const float*)nullptr, N,
            scale, 0.f, loss, (float*)nullptr);
  return dm_check_launch();
}

extern "C" long long dm_sumsq_scratch_floats(void) { return 1024; }

extern "C" int dm_sumsq(const float* x, long long count, float* scratch, float* out, dm_stream_t stream) {
  if (count < 0 || !out || !scratch || (count > 0 && !x)) return DM_ERR_INVALID_ARG;
  const int blocks = count > 0 ? (int)((count + 16383) / 16384 < 1024 ? (count + 16383) / 16384 : 1024) : 1;
  DM_LAUNCH(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, count, scratch);
  DM_LAUNCH(ordered_sum2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, (const float*)nullptr,
            blocks, 1.0f, 0.f, out, (float*)nullptr);
  return dm_check_launch();
}

extern "C" int dm_clip_scale(float* x, long long count, const float* sumsq, float max_norm, dm_stream_t stream) {
  if (count < 0 || !sumsq || !(max_norm > 0.f) || (count > 0 && !x)) return DM_ERR_INVALID_ARG;
  if (count == 0) return DM_OK;
  DM_LAUNCH(clip_scale_kernel, dim3(dm_ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, x, count, sumsq, max_norm);
  return dm_check_launch();
}

extern "C" int dm_scale(float* x, long long count, float factor, dm_stream_t stream) {
  if (count < 0 || (count > 0 && !x)) return DM_ERR_INVALID_ARG;
  if (count == 0) return DM_OK;
  DM_LAUNCH(scale_kernel, dim3(dm_ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, x, count, factor);
  return dm_check_launch();
}

extern "C" int dm_ignore_columns(float* overlaps, int G, int N, const float* iof, int I, int boxes_major, float thr,
                                 dm_stream_t stream) {
  if (G < 0 || N < 0 || I < 0) return DM_ERR_INVALID_ARG;
  if (G == 0 || N == 0 || I == 0) return DM_OK;
  if (!overlaps || !iof) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(ignore_columns_kernel, dim3(dm_ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, overlaps, G, N, iof, I,
            boxes_major, thr);
  return dm_check_launch();
}
