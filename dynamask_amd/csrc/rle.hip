// K19: COCO run-length encoding of the pasted masks on the device (SURVEY 8f rank 2:
// get_seg_masks -> encode_mask_results, mmdet/core/mask/utils.py:36-63, which calls
// pycocotools' rleEncode on a column-major copy of every [h, w] bitmap on the host).
//
// RLE walks a mask column-major (y fastest) and stores the lengths of alternating
// 0/1 runs, starting with a (possibly empty) run of zeros.  A run boundary is a
// position j with v[j] != v[j-1] (v[-1] = 0), so the encoder is a stream compaction of
// those positions:
//   pass 1  every workgroup counts the boundaries of its 4096-pixel segment,
//   pass 2  one workgroup scans the counts of all (mask, segment) pairs,
//   pass 3  the segments write their boundary positions, packed mask after mask.
// The pixel value comes from a functor: either a uint8 canvas [N, h, w] (row-major, as
// dm_paste_masks writes it) or the paste itself (grid-sample of the S x S logits into the
// box + threshold, the arithmetic of paste_masks_kernel) -- in the fused form the
// [N, h, w] canvas is never materialised and only the run boundaries (a few hundred
// int32 per mask) cross PCIe instead of h*w bytes per mask.
// The host turns boundary positions into run lengths and the printable string
// (dm_rle_string, the published rleToString of cocoapi/common/maskApi.c).
#include "common.h"

namespace {

constexpr int kSeg = 4096;        // pixels per workgroup segment (256 threads x 16)

struct CanvasSrc {
  const uint8_t* canvas;
  int img_h, img_w;
  __device__ __forceinline__ int at(int n, int y, int x) const {
    return canvas[((size_t)n * img_h + y) * img_w + x] != 0;
  }
};

struct PasteSrc {
  const float* masks;
  const float* boxes;
  int mh, mw;
  float thr;
  int apply_sigmoid;
  // same arithmetic as paste_masks_kernel (pointwise.hip); kept bit-identical on purpose
  __device__ __forceinline__ int at(int n, int py, int px) const {
    const float x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
    const float* m = masks + (size_t)n * mh * mw;
    float gx = ((float)px + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
    float gy = ((float)py + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
    if (isinf(gx)) gx = 0.f;
    if (isinf(gy)) gy = 0.f;
    const float sx = ((gx + 1.f) * (float)mw - 1.f) / 2.f;
    const float sy = ((gy + 1.f) * (float)mh - 1.f) / 2.f;
    float v = 0.f;
    if (sx > -1.f && sx < (float)mw && sy > -1.f && sy < (float)mh) {
      const float fx = floorf(sx), fy = floorf(sy);
      const int ix = (int)fx, iy = (int)fy;
      const float lx = sx - fx, ly = sy - fy;
      auto tap = [&](int yy, int xx) -> float {
        if (yy < 0 || yy >= mh || xx < 0 || xx >= mw) return 0.f;
        const float t = m[yy * mw + xx];
        return apply_sigmoid ? 1.f / (1.f + expf(-t)) : t;
      };
      v = tap(iy, ix) * (1.f - lx) * (1.f - ly) + tap(iy, ix + 1) * lx * (1.f - ly) + tap(iy + 1, ix) * (1.f - lx) * ly +
          tap(iy + 1, ix + 1) * lx * ly;
    }
    return v >= thr ? 1 : 0;
  }
};

// boundaries of the 16 column-major positions this thread owns: bit e set <=> v[j0+e] != v[j0+e-1]
template <class Src>
__device__ __forceinline__ unsigned thread_boundaries(const Src& src, int n, int img_h, long long total, long long j0) {
  unsigned bits = 0;
  if (j0 >= total) return 0;
  int x = (int)(j0 / img_h), y = (int)(j0 - (long long)x * img_h);
  int prev = 0;
  if (j0 > 0) {
    int px = x, py = y - 1;
    if (py < 0) { py = img_h - 1; --px; }
    prev = src.at(n, py, px);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    if (j0 + e < total) {
      const int v = src.at(n, y, x);
      bits |= (unsigned)(v != prev) << e;
      prev = v;
      if (++y == img_h) { y = 0; ++x; }
    }
  }
  return bits;
}

__device__ __forceinline__ int block_sum_256(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <class Src>
__global__ __launch_bounds__(256) void rle_count_kernel(Src src, int img_h, int img_w, int segs, int* __restrict__ seg_counts) {
  __shared__ int red[4];
  const int n = blockIdx.y, seg = blockIdx.x;
  const long long total = (long long)img_h * img_w;
  const long long j0 = (long long)seg * kSeg + threadIdx.x * 16;
  const unsigned bits = thread_boundaries(src, n, img_h, total, j0);
  const int s = block_sum_256(__popc(bits), red);
  if (threadIdx.x == 0) seg_counts[(size_t)n * segs + seg] = s;
}

// exclusive scan of seg_counts[0 .. M) in place (one workgroup; M = N * segs), per-mask
// run totals and packed start offsets
__global__ __launch_bounds__(1024) void rle_scan_kernel(int* __restrict__ seg_counts, int M, int N, int segs,
                                                        int* __restrict__ mask_runs, int* __restrict__ mask_start) {
  __shared__ int warp_tot[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < M; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < M ? seg_counts[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) warp_tot[wv] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += warp_tot[w];
    const int carry = carry_s;
    if (i < M) seg_counts[i] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  // per-mask totals from the scanned array
  const int grand = carry_s;
  for (int n = threadIdx.x; n < N; n += 1024) {
    const int s0 = seg_counts[(size_t)n * segs];
    const int s1 = (n + 1 < N) ? seg_counts[(size_t)(n + 1) * segs] : grand;
    mask_start[n] = s0;
    mask_runs[n] = s1 - s0;
  }
  if (threadIdx.x == 0) mask_start[N] = grand;
}

template <class Src>
__global__ __launch_bounds__(256) void rle_write_kernel(Src src, int img_h, int img_w, int segs,
                                                        const int* __restrict__ seg_offsets, int capacity,
                                                        int* __restrict__ positions) {
  __shared__ int wtot[4];
  const int n = blockIdx.y, seg = blockIdx.x;
  const long long total = (long long)img_h * img_w;
  const long long j0 = (long long)seg * kSeg + threadIdx.x * 16;
  const unsigned bits = thread_boundaries(src, n, img_h, total, j0);
  const int cnt = __popc(bits);
  // exclusive scan of cnt over the workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  int off = seg_offsets[(size_t)n * segs + seg] + inc - cnt;
  for (int w = 0; w < wv; ++w) off += wtot[w];
  unsigned b = bits;
  while (b) {
    const int e = __ffs(b) - 1;
    b &= b - 1;
    if (off < capacity) positions[off] = (int)(j0 + e);
    ++off;
  }
}

template <class Src>
int rle_launch(const Src& src, int N, int img_h, int img_w, int* seg_scratch, int* mask_runs, int* mask_start,
               int* positions, int capacity, hipStream_t st) {
  const long long total = (long long)img_h * img_w;
  if (total > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  const int segs = (int)((total + kSeg - 1) / kSeg);
  DM_LAUNCH(rle_count_kernel<Src>, dim3(segs, N), dim3(256), 0, st, src, img_h, img_w, segs, seg_scratch);
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(rle_scan_kernel, dim3(1), dim3(1024), 0, st, seg_scratch, N * segs, N, segs, mask_runs, mask_start);
  rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(rle_write_kernel<Src>, dim3(segs, N), dim3(256), 0, st, src, img_h, img_w, segs, seg_scratch, capacity, positions);
  return dm_check_launch();
}

}  // namespace

extern "C" long long dm_rle_scratch_ints(int N, int img_h, int img_w) {
  if (N < 0 || img_h <= 0 || img_w <= 0) return -1;
  const long long total = (long long)img_h * img_w;
  return (long long)N * ((total + kSeg - 1) / kSeg);
}

extern "C" int dm_rle_encode_canvas(const uint8_t* canvas, int N, int img_h, int img_w, int* seg_scratch, int* mask_runs,
                                    int* mask_start, int* positions, int capacity, dm_stream_t stream) {
  if (N < 0 || img_h <= 0 || img_w <= 0 || capacity < 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!canvas || !seg_scratch || !mask_runs || !mask_start || (!positions && capacity > 0)) return DM_ERR_INVALID_ARG;
  CanvasSrc src{canvas, img_h, img_w};
  return rle_launch(src, N, img_h, img_w, seg_scratch, mask_runs, mask_start, positions, capacity, (hipStream_t)stream);
}

extern "C" int dm_paste_rle(const float* masks, const float* boxes, int N, int mask_h, int mask_w, int img_h, int img_w,
                            float threshold, int apply_sigmoid, int* seg_scratch, int* mask_runs, int* mask_start,
                            int* positions, int capacity, dm_stream_t stream) {
  if (N < 0 || mask_h <= 0 || mask_w <= 0 || img_h <= 0 || img_w <= 0 || capacity < 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!masks || !boxes || !seg_scratch || !mask_runs || !mask_start || (!positions && capacity > 0)) return DM_ERR_INVALID_ARG;
  PasteSrc src{masks, boxes, mask_h, mask_w, threshold, apply_sigmoid};
  return rle_launch(src, N, img_h, img_w, seg_scratch, mask_runs, mask_start, positions, capacity, (hipStream_t)stream);
}

// Host side of the encoder: run boundaries -> run lengths -> COCO's printable string
// (rleToString, cocoapi/common/maskApi.c: each count, from the third on as the difference
// to the count two back, is written in 5-bit groups, bit 5 = "more", offset 48).
// Returns the string length, or -(needed) if `cap` is too small.  Pure host code.
extern "C" long long dm_rle_string(const int* positions, int runs, long long total_pixels, char* out, long long cap) {
  if (runs < 0 || total_pixels < 0 || (!positions && runs > 0) || (!out && cap > 0)) return 0;
  // counts: c[0] = p[0], c[i] = p[i] - p[i-1], c[runs] = total - p[runs-1]
  long long len = 0;
  long long prev2 = 0, prev1 = 0;      // counts i-2 and i-1
  const int m = runs + 1;
  for (int i = 0; i < m; ++i) {
    const long long lo = i == 0 ? 0 : positions[i - 1];
    const long long hi = i < runs ? positions[i] : total_pixels;
    const long long cnt = hi - lo;
    long long x = cnt;
    if (i > 2) x -= prev2;
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (len < cap) out[len] = c;
      ++len;
    }
    prev2 = prev1;
    prev1 = cnt;
  }
  return len <= cap ? len : -len;
}
