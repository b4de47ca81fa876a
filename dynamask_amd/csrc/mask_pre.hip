// K9 (MaskPre, roi_heads/base_roi_head.py:10-27): the pieces that are not
// convolutions -- train-mode BatchNorm statistics and the fused
// BN -> ReLU -> max_pool2d(3, stride 2, pad 1).  The 1x1 / 3x3 convs and the
// two FC layers run on the implicit-GEMM kernel (an FC is a 1x1 conv on a 1x1
// map).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum_(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float block_sum_bcast(float v, float* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += smem[w];   // fixed order: deterministic
  __syncthreads();
  return r;
}

// One workgroup per channel, two passes (mean, then centred second moment): no
// E[x^2]-mean^2 cancellation, fixed reduction tree -> run-to-run deterministic
// (the selector index downstream must be reproducible).
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ x, int NB, int C, int HW,
                                                        float* __restrict__ mean, float* __restrict__ var,
                                                        float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, float momentum) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const long long total = (long long)NB * HW;
  float s = 0.f;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = (int)(i / HW);
    const int p = (int)(i - (long long)n * HW);
    s += x[((size_t)n * C + c) * HW + p];
  }
  const float m = block_sum_bcast(s, red) / (float)total;
  float q = 0.f;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = (int)(i / HW);
    const int p = (int)(i - (long long)n * HW);
    const float d = x[((size_t)n * C + c) * HW + p] - m;
    q += d * d;
  }
  const float v = block_sum_bcast(q, red) / (float)total;   // biased (used for normalisation)
  if (threadIdx.x == 0) {
    mean[c] = m;
    var[c] = v;
    if (running_mean && running_var) {
      const float unbiased = total > 1 ? v * (float)total / (float)(total - 1) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ x, int NB, int C, int H, int W,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ var,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ out, int OH, int OW) {
  const size_t total = (size_t)NB * C * OH * OW;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int c = (int)((idx / ((size_t)OW * OH)) % C);
    const size_t n = idx / ((size_t)OW * OH * C);
    const float invstd = 1.0f / sqrtf(var[c] + eps);
    const float g = gamma[c], b = beta[c], m = mean[c];
    const float* p = x + (n * C + c) * (size_t)H * W;
    float best = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * oy - 1 + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * ox - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float v = fmaxf((p[y * W + xx] - m) * invstd * g + b, 0.f);
        best = fmaxf(best, v);
      }
    }
    out[idx] = best;
  }
}

}  // namespace

extern "C" int dm_bn_stats(const float* x, int NB, int C, int HW, float* mean, float* var, float* running_mean,
                           float* running_var, float momentum, dm_stream_t stream) {
  if (!x || !mean || !var || NB <= 0 || C <= 0 || HW <= 0) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(bn_stats_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, NB, C, HW, mean, var,
                     running_mean, running_var, momentum);
  return dm_check_launch();
}

extern "C" int dm_bn_relu_maxpool_fwd(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                                      const float* gamma, const float* beta, float eps, float* out,
                                      dm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)NB * C * OH * OW;
  const int blocks = (int)min((size_t)dm_ceil_div((long long)total, 256), (size_t)16384);
  DM_LAUNCH(bn_relu_maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, NB, C, H, W, mean, var,
                     gamma, beta, eps, out, OH, OW);
  return dm_check_launch();
}
