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
                                                        float* __restrict__ running_var, float momentum,
                                                        const float* __restrict__ mean_shift) {
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
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mean_shift ? m + mean_shift[c] : m);
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

// Split form (scratch given): the channel's NB*HW elements are divided over kBnSplits
// workgroups, so a 128-channel tensor fills the chip (one workgroup per channel leaves half
// of the 256 CUs idle and ran at 1.6 TB/s); partial sums are combined in a fixed order ->
// still deterministic.  Pass 0: partial sums; pass 1: partial centred second moments.
constexpr int kBnSplits = 16;

__device__ __forceinline__ float bn_partials_ordered(const float* part, int c) {
  float r = 0.f;
  for (int s = 0; s < kBnSplits; ++s) r += part[c * kBnSplits + s];
  return r;
}

template <int PASS>
__global__ __launch_bounds__(256) void bn_stats_split_kernel(const float* __restrict__ x, int NB, int C, int HW,
                                                             float* __restrict__ part_sum, float* __restrict__ part_sq) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int n0 = (int)((long long)NB * sp / kBnSplits), n1 = (int)((long long)NB * (sp + 1) / kBnSplits);
  const float m = PASS == 1 ? bn_partials_ordered(part_sum, c) / (float)((long long)NB * HW) : 0.f;
  float acc = 0.f;
  for (int n = n0; n < n1; ++n) {
    const float* p = x + ((size_t)n * C + c) * HW;
    if ((HW & 3) == 0) {
      const dm_f32x4* p4 = reinterpret_cast<const dm_f32x4*>(p);
      for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
        const dm_f32x4 v = p4[i];
        if (PASS == 0) acc += (v[0] + v[1]) + (v[2] + v[3]);
        else { const float a = v[0] - m, b = v[1] - m, cc = v[2] - m, d = v[3] - m; acc += (a * a + b * b) + (cc * cc + d * d); }
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = p[i];
        if (PASS == 0) acc += v;
        else acc += (v - m) * (v - m);
      }
    }
  }
  const float r = block_sum_bcast(acc, red);
  if (threadIdx.x == 0) (PASS == 0 ? part_sum : part_sq)[c * kBnSplits + sp] = r;
}

__global__ void bn_stats_finalize_kernel(const float* __restrict__ part_sum, const float* __restrict__ part_sq, int C,
                                         long long total, float* __restrict__ mean, float* __restrict__ var,
                                         float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                         const float* __restrict__ mean_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = bn_partials_ordered(part_sum, c) / (float)total;
  const float v = bn_partials_ordered(part_sq, c) / (float)total;
  mean[c] = m;
  var[c] = v;
  if (running_mean && running_var) {
    const float unbiased = total > 1 ? v * (float)total / (float)(total - 1) : v;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mean_shift ? m + mean_shift[c] : m);
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
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

// The same block for even W (MaskPre's 56 x 56 and 28 x 28 maps): a window's columns 2ox and 2ox + 1 come as one 8-byte
// load per row (consecutive lanes read consecutive pairs: whole lines), column 2ox - 1 as a dword the L1 already holds;
// 32-bit index arithmetic with multiply-high divisions (the kernel above divides 64-bit indices four times per output
// and ran at 1.8 TB/s).  Same expression per tap, max over the same taps: same bits.
__global__ __launch_bounds__(256) void bn_relu_maxpool_rows_kernel(const float* __restrict__ x, int pblocks, int C, int H, int W,
                                                                   const float* __restrict__ mean, const float* __restrict__ var,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float eps, float* __restrict__ out, int OH, int OW, unsigned m_ow) {
  // workgroup = 1024 outputs of one plane, four per thread with their 24 loads issued together (one output per thread
  // left 36 KB in flight per CU: 2.2 TB/s); plane, channel and the BatchNorm constants are uniform (scalar)
  const int nc = blockIdx.x / pblocks;
  const int o0 = (blockIdx.x - nc * pblocks) * 1024 + threadIdx.x;
  const int OHW = OH * OW;
  const int c = nc % C;
  const float invstd = 1.0f / sqrtf(var[c] + eps);
  const float g = gamma[c], b = beta[c], m = mean[c];
  const float* p = x + (size_t)nc * H * W;
  struct __attribute__((aligned(8))) F2 { float a, b; };
  float left[4][3];
  F2 ab[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = min(o0 + k * 256, OHW - 1);                             // (surplus lanes recompute the last output, no store)
    const int oy = (int)__umulhi((unsigned)o, m_ow), ox = o - oy * OW;      // o < 2^16: exact
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = min(max(2 * oy - 1 + dy, 0), H - 1);                   // a clamped row repeats a row of the window: same max
      const float* row = p + y * W + 2 * ox;
      left[k][dy] = row[ox > 0 ? -1 : 0];                                 // (no left neighbour: column 2ox again)
      ab[k][dy] = *reinterpret_cast<const F2*>(row);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float best = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      best = fmaxf(best, fmaxf((left[k][dy] - m) * invstd * g + b, 0.f));
      best = fmaxf(best, fmaxf((ab[k][dy].a - m) * invstd * g + b, 0.f));
      best = fmaxf(best, fmaxf((ab[k][dy].b - m) * invstd * g + b, 0.f));
    }
    const int o = o0 + k * 256;
    if (o < OHW) out[(size_t)nc * OHW + o] = best;
  }
}

}  // namespace

extern "C" long long dm_bn_scratch_floats(int C) { return C > 0 ? (long long)C * kBnSplits * 2 : -1; }

extern "C" int dm_bn_stats(const float* x, int NB, int C, int HW, float* mean, float* var, float* running_mean,
                           float* running_var, float momentum, const float* mean_shift, float* scratch, dm_stream_t stream) {
  if (!x || !mean || !var || NB <= 0 || C <= 0 || HW <= 0) return DM_ERR_INVALID_ARG;
  if (scratch && NB >= kBnSplits) {
    hipStream_t st = (hipStream_t)stream;
    float* ps = scratch;
    float* pq = scratch + (size_t)C * kBnSplits;
    DM_LAUNCH(bn_stats_split_kernel<0>, dim3(C, kBnSplits), dim3(256), 0, st, x, NB, C, HW, ps, pq);
    int rc = dm_check_launch();
    if (rc != DM_OK) return rc;
    DM_LAUNCH(bn_stats_split_kernel<1>, dim3(C, kBnSplits), dim3(256), 0, st, x, NB, C, HW, ps, pq);
    rc = dm_check_launch();
    if (rc != DM_OK) return rc;
    DM_LAUNCH(bn_stats_finalize_kernel, dim3(dm_ceil_div(C, 128)), dim3(128), 0, st, ps, pq, C, (long long)NB * HW, mean, var,
              running_mean, running_var, momentum, mean_shift);
    return dm_check_launch();
  }
  DM_LAUNCH(bn_stats_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, NB, C, HW, mean, var,
                     running_mean, running_var, momentum, mean_shift);
  return dm_check_launch();
}

extern "C" int dm_bn_relu_maxpool_fwd(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                                      const float* gamma, const float* beta, float eps, float* out,
                                      dm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)NB * C * OH * OW;
  const int pblocks = dm_ceil_div(OH * OW, 1024);
  if (!(W & 1) && W >= 4 && OH * OW < 65536 && (long long)NB * C * pblocks < 0x7fffffffLL && (((uintptr_t)x) & 7) == 0) {
    const unsigned m_ow = 0xFFFFFFFFu / (unsigned)OW + 1u;      // OW >= 2; o * OW < 2^32
    DM_LAUNCH(bn_relu_maxpool_rows_kernel, dim3((unsigned)(NB * C * pblocks)), dim3(256), 0, (hipStream_t)stream, x, pblocks, C, H, W,
              mean, var, gamma, beta, eps, out, OH, OW, m_ow);
    return dm_check_launch();
  }
  const int blocks = (int)min((size_t)dm_ceil_div((long long)total, 256), (size_t)16384);
  DM_LAUNCH(bn_relu_maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, NB, C, H, W, mean, var,
                     gamma, beta, eps, out, OH, OW);
  return dm_check_launch();
}

// Which tap each pooled output took: the plane index (y * W + x) of the window's first maximum of z = relu(bn(x)),
// found with the expression and scan order of the backward kernels below.  Diagnostic: a test compares the
// choices with the reference's max_pool2d(return_indices=True) and counts the windows (fp32 ties) that differ.
namespace {
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(const float* __restrict__ x, int NB, int C, int H, int W,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int32_t* __restrict__ arg, int OH, int OW) {
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
    int bi = -1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * oy - 1 + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * ox - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float v = fmaxf((p[y * W + xx] - m) * invstd * g + b, 0.f);
        if (v > best) {
          best = v;
          bi = y * W + xx;
        }
      }
    }
    arg[idx] = bi;
  }
}
}  // namespace

extern "C" int dm_bn_relu_maxpool_argmax(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                                         const float* gamma, const float* beta, float eps, int32_t* argmax,
                                         dm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !argmax || NB <= 0 || C <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const size_t total = (size_t)NB * C * OH * OW;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  DM_LAUNCH(maxpool_argmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, NB, C, H, W, mean, var, gamma, beta,
            eps, argmax, OH, OW);
  return dm_check_launch();
}

// ---------------------------------------------------------------------------
// MaskPre backward pieces: max_pool2d(3,2,1) o ReLU o BatchNorm(train).
namespace {

// scatter the pooled gradient to the window's first maximum of z = relu(bn(x))
// (torch's max_pool2d keeps the first maximum in scan order); g_z is zero-filled
// by the caller.  ReLU mask folded in: a zero maximum passes no gradient.
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const float* __restrict__ x, int NB, int C, int H, int W,
                                                               const float* __restrict__ mean, const float* __restrict__ var,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, const float* __restrict__ gout,
                                                               float* __restrict__ gz, int OH, int OW) {
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
    int bi = -1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * oy - 1 + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * ox - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float v = fmaxf((p[y * W + xx] - m) * invstd * g + b, 0.f);
        if (v > best) {
          best = v;
          bi = y * W + xx;
        }
      }
    }
    if (bi >= 0 && best > 0.f) atomicAdd(gz + (n * C + c) * (size_t)H * W + bi, gout[idx]);
  }
}

// The same routing with the plane staged in LDS, in gather form (round 3; round 2 scattered with LDS float atomics).
// A workgroup takes one (image, channel) plane at a time: z = relu(bn(x)) is staged once with 16-byte loads; every
// pooled output records which tap it took (arg) and its gradient (val) in LDS; then every INPUT pixel adds the
// gradients of the <= 4 windows that contain it and chose it, in a fixed order (window row, then column) -- no
// atomics, no memset, and the sum has the same bits on every run.  The plane leaves as 16-byte stores.
// Needs H*W % 4 == 0; LDS: H*W floats + OH*OW (int, float) pairs.
__global__ __launch_bounds__(256) void maxpool_relu_bwd_lds_kernel(const float* __restrict__ x, int NB, int C, int H, int W,
                                                                   const float* __restrict__ mean, const float* __restrict__ var,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float eps, const float* __restrict__ gout,
                                                                   float* __restrict__ gz, int OH, int OW) {
  extern __shared__ __attribute__((aligned(16))) float pl[];       // z plane [HW], arg [OHW], val [OHW]
  const int HW = H * W, OHW = OH * OW;
  float* zp = pl;
  int* argp = reinterpret_cast<int*>(pl + HW);
  float* valp = pl + HW + OHW;
  for (int nc = blockIdx.x; nc < NB * C; nc += gridDim.x) {
    const int c = nc % C;
    const float invstd = 1.0f / sqrtf(var[c] + eps);
    const float g = gamma[c], b = beta[c], m = mean[c];
    const dm_f32x4* x4 = reinterpret_cast<const dm_f32x4*>(x + (size_t)nc * HW);
    for (int i = threadIdx.x; i < HW / 4; i += 256) {
      const dm_f32x4 v = x4[i];
      dm_f32x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = fmaxf((v[e] - m) * invstd * g + b, 0.f);      // the expression of the kernel above
      reinterpret_cast<dm_f32x4*>(zp)[i] = z;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < OHW; o += 256) {
      const int oy = o / OW, ox = o - oy * OW;
      float best = -INFINITY;
      int bi = -1;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * oy - 1 + dy;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = 2 * ox - 1 + dx;
          if (xx < 0 || xx >= W) continue;
          const float v = zp[y * W + xx];
          if (v > best) {
            best = v;
            bi = y * W + xx;
          }
        }
      }
      argp[o] = best > 0.f ? bi : -1;          // a zero maximum passes no gradient (ReLU)
      valp[o] = gout[(size_t)nc * OHW + o];
    }
    __syncthreads();
    dm_f32x4* o4 = reinterpret_cast<dm_f32x4*>(gz + (size_t)nc * HW);
    for (int i = threadIdx.x; i < HW / 4; i += 256) {
      dm_f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = 4 * i + e;
        const int y = p / W, xx = p - y * W;
        // windows containing (y, xx): rows (y - 1 + 1) / 2 .. (y + 1) / 2, i.e. one for even y, two for odd
        const int oy0 = y >> 1, oy1 = min((y + 1) >> 1, OH - 1);
        const int ox0 = xx >> 1, ox1 = min((xx + 1) >> 1, OW - 1);
        float acc = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy)
          for (int ox = ox0; ox <= ox1; ++ox) {
            const int o = oy * OW + ox;
            if (argp[o] == p) acc += valp[o];
          }
        r[e] = acc;
      }
      o4[i] = r;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float bsum(float v, float* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum_(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += smem[w];
  __syncthreads();
  return r;
}

// BatchNorm (training) backward, one workgroup per channel:
//   g_gamma = sum gz*xhat, g_beta = sum gz,
//   g_x = gamma*invstd * (gz - g_beta/M - xhat*g_gamma/M)       (written over gz)
__global__ __launch_bounds__(1024) void bn_bwd_kernel(const float* __restrict__ x, float* __restrict__ gz, int NB, int C,
                                                      int HW, const float* __restrict__ mean, const float* __restrict__ var,
                                                      const float* __restrict__ gamma, float eps,
                                                      float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const long long total = (long long)NB * HW;
  const float invstd = 1.0f / sqrtf(var[c] + eps), m = mean[c];
  float s1 = 0.f, s2 = 0.f;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const long long n = i / HW, p = i - n * HW;
    const size_t a = ((size_t)n * C + c) * HW + p;
    const float g = gz[a];
    s1 += g;
    s2 += g * (x[a] - m) * invstd;
  }
  const float sb = bsum(s1, red);
  const float sg = bsum(s2, red);
  if (threadIdx.x == 0) {
    g_beta[c] = sb;
    g_gamma[c] = sg;
  }
  const float k = gamma[c] * invstd, invM = 1.0f / (float)total;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    const long long n = i / HW, p = i - n * HW;
    const size_t a = ((size_t)n * C + c) * HW + p;
    const float xh = (x[a] - m) * invstd;
    gz[a] = k * (gz[a] - sb * invM - xh * sg * invM);
  }
}

// Split form of bn_bwd_kernel: pass 0 partial (sum gz, sum gz*xhat) per (channel, split), pass 1
// combines them in a fixed order and rewrites its slice of gz.
template <int PASS>
__global__ __launch_bounds__(256) void bn_bwd_split_kernel(const float* __restrict__ x, float* __restrict__ gz, int NB, int C,
                                                           int HW, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, float eps, float* __restrict__ part,
                                                           float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int n0 = (int)((long long)NB * sp / kBnSplits), n1 = (int)((long long)NB * (sp + 1) / kBnSplits);
  const float invstd = 1.0f / sqrtf(var[c] + eps), m = mean[c];
  float* p1 = part;                                   // [C][kBnSplits] sum gz
  float* p2 = part + (size_t)C * kBnSplits;           // [C][kBnSplits] sum gz * xhat
  if (PASS == 0) {
    float s1 = 0.f, s2 = 0.f;
    for (int n = n0; n < n1; ++n) {
      const size_t base = ((size_t)n * C + c) * HW;
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float g = gz[base + i];
        s1 += g;
        s2 += g * (x[base + i] - m) * invstd;
      }
    }
    const float sb = bsum(s1, red);
    const float sg = bsum(s2, red);
    if (threadIdx.x == 0) {
      p1[c * kBnSplits + sp] = sb;
      p2[c * kBnSplits + sp] = sg;
    }
  } else {
    const float sb = bn_partials_ordered(p1, c), sg = bn_partials_ordered(p2, c);
    if (sp == 0 && threadIdx.x == 0) {
      g_beta[c] = sb;
      g_gamma[c] = sg;
    }
    const float k = gamma[c] * invstd, invM = 1.0f / (float)((long long)NB * HW);
    for (int n = n0; n < n1; ++n) {
      const size_t base = ((size_t)n * C + c) * HW;
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float xh = (x[base + i] - m) * invstd;
        gz[base + i] = k * (gz[base + i] - sb * invM - xh * sg * invM);
      }
    }
  }
}

// maxpool_relu_bwd_lds_kernel with bn_bwd_split_kernel<0> folded in: workgroup (channel, split) walks the planes of its
// split, writes the pooling adjoint gz and keeps the two plane sums BatchNorm's backward needs (sum gz, sum gz * xhat; x is
// re-read from the L2 for xhat), one partial pair per workgroup in the [C][kBnSplits] scratch of the split form -- combined
// by bn_bwd_split_kernel<1> in a fixed order: deterministic.  Saves the pass that read x and gz (0.82 GB at
// 256 x 128 x 56 x 56) only for those sums: 0.735 -> 0.57 ms for the block's backward.
// (Measured and dropped: a second pass that rebuilds gz instead of reading it back, so that gz is never written --
// 1.44 instead of 3.0 GB through HBM, and slower, 0.98 ms: the plane pipeline (stage, arg-max, gather: three barriers)
// is latency-bound at 2.8 TB/s, the streaming passes it would replace run at 4.6-5.3.)
__global__ __launch_bounds__(256) void maxpool_relu_bwd_sums_kernel(const float* __restrict__ x, int NB, int C, int H, int W,
                                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float eps, const float* __restrict__ gout,
                                                                    float* __restrict__ gz, int OH, int OW, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float pl[];       // z plane [HW], arg [OHW], val [OHW]
  __shared__ float red[4];
  const int HW = H * W, OHW = OH * OW;
  float* zp = pl;
  int* argp = reinterpret_cast<int*>(pl + HW);
  float* valp = pl + HW + OHW;
  const int c = blockIdx.x, sp = blockIdx.y;
  const int n0 = (int)((long long)NB * sp / kBnSplits), n1 = (int)((long long)NB * (sp + 1) / kBnSplits);
  const float invstd = 1.0f / sqrtf(var[c] + eps);
  const float g = gamma[c], b = beta[c], m = mean[c];
  float s1 = 0.f, s2 = 0.f;
  const bool even = !(H & 1) && !(W & 1) && W >= 4 && (((uintptr_t)x | (uintptr_t)gz) & 7) == 0;
  const unsigned m_bw = W >= 4 ? 0xFFFFFFFFu / (unsigned)(W >> 1) + 1u : 0u;      // blk / (W / 2) by multiply-high (W / 2 >= 2)
  // a plane's x quads and grad_out values are fetched into registers one plane ahead (maps up to 4096 / 1024 elements:
  // MaskPre's 56 x 56 and 28 x 28), so that the stage -> arg-max -> gather chain of a plane does not start with a round
  // trip to HBM
  constexpr int PIT = 4;
  const bool pre = HW / 4 <= PIT * 256 && OHW <= PIT * 256;
  dm_f32x4 nx[PIT];
  float ng[PIT];
  auto fetch = [&](int n) {
    const size_t nc = (size_t)n * C + c;
    const dm_f32x4* x4 = reinterpret_cast<const dm_f32x4*>(x + nc * HW);
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      const int i = threadIdx.x + k * 256;
      nx[k] = x4[min(i, HW / 4 - 1)];
      ng[k] = gout[nc * OHW + min(i, OHW - 1)];
    }
  };
  auto stage = [&](const dm_f32x4& v, int i) {
    dm_f32x4 z;
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = fmaxf((v[e] - m) * invstd * g + b, 0.f);      // the forward's expression (bn_relu_maxpool_kernel)
    reinterpret_cast<dm_f32x4*>(zp)[i] = z;
  };
  auto window = [&](int o, float gval) {
    const int oy = o / OW, ox = o - oy * OW;
    float best = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * oy - 1 + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * ox - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float v = zp[y * W + xx];
        if (v > best) {
          best = v;
          bi = y * W + xx;
        }
      }
    }
    argp[o] = best > 0.f ? bi : -1;          // a zero maximum passes no gradient (ReLU)
    valp[o] = gval;
  };
  if (pre && n0 < n1) fetch(n0);
  for (int n = n0; n < n1; ++n) {
    const size_t nc = (size_t)n * C + c;
    const dm_f32x4* x4 = reinterpret_cast<const dm_f32x4*>(x + nc * HW);
    if (pre) {
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < HW / 4) stage(nx[k], i);
      }
      float cg[PIT];
#pragma unroll
      for (int k = 0; k < PIT; ++k) cg[k] = ng[k];
      __syncthreads();
      fetch(min(n + 1, n1 - 1));             // (unconditional: a branch around loads makes the compiler wait for them at the join)
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int o = threadIdx.x + k * 256;
        if (o < OHW) window(o, cg[k]);
      }
    } else {
      for (int i = threadIdx.x; i < HW / 4; i += 256) stage(x4[i], i);
      __syncthreads();
      for (int o = threadIdx.x; o < OHW; o += 256) window(o, gout[nc * OHW + o]);
    }
    __syncthreads();
    if (even) {
      // 2 x 2 blocks of the map: the windows that can hold a block's four elements are (a, b), (a, b + 1), (a + 1, b),
      // (a + 1, b + 1) -- 8 LDS reads and one division per block where the element form below does up to 18 reads and a
      // division per element; a block's partial sums are added in the element form's order (window rows, then columns)
      struct __attribute__((aligned(8))) F2 { float a, b; };
      const int BW = W >> 1, nblk = (H >> 1) * BW;
      const float* xp = x + nc * HW;
      float* gp = gz + nc * HW;
      for (int blk = threadIdx.x; blk < nblk; blk += 256) {
        const int ba = (int)__umulhi((unsigned)blk, m_bw), bb = blk - ba * BW;
        const int w00 = ba * OW + bb;
        const bool right = bb + 1 < OW, down = ba + 1 < OH;
        const int w01 = right ? w00 + 1 : w00, w10 = down ? w00 + OW : w00, w11 = (right && down) ? w00 + OW + 1 : w00;
        const int p00 = (2 * ba) * W + 2 * bb, p01 = p00 + 1, p10 = p00 + W, p11 = p10 + 1;
        const int a00 = argp[w00], a01 = right ? argp[w01] : -2, a10 = down ? argp[w10] : -2, a11 = (right && down) ? argp[w11] : -2;
        const float v00 = valp[w00], v01 = valp[w01], v10 = valp[w10], v11 = valp[w11];
        float r00 = 0.f, r01 = 0.f, r10 = 0.f, r11 = 0.f;
        if (a00 == p00) r00 += v00;
        if (a00 == p01) r01 += v00;
        if (a01 == p01) r01 += v01;
        if (a00 == p10) r10 += v00;
        if (a10 == p10) r10 += v10;
        if (a00 == p11) r11 += v00;
        if (a01 == p11) r11 += v01;
        if (a10 == p11) r11 += v10;
        if (a11 == p11) r11 += v11;
        const F2 xt = *reinterpret_cast<const F2*>(xp + p00), xb = *reinterpret_cast<const F2*>(xp + p10);
        *reinterpret_cast<F2*>(gp + p00) = F2{r00, r01};
        *reinterpret_cast<F2*>(gp + p10) = F2{r10, r11};
        s1 += r00; s2 += r00 * (xt.a - m) * invstd;
        s1 += r01; s2 += r01 * (xt.b - m) * invstd;
        s1 += r10; s2 += r10 * (xb.a - m) * invstd;
        s1 += r11; s2 += r11 * (xb.b - m) * invstd;
      }
    } else {
    dm_f32x4* o4 = reinterpret_cast<dm_f32x4*>(gz + nc * HW);
    for (int i = threadIdx.x; i < HW / 4; i += 256) {
      const dm_f32x4 v = x4[i];
      dm_f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = 4 * i + e;
        const int y = p / W, xx = p - y * W;
        const int oy0 = y >> 1, oy1 = min((y + 1) >> 1, OH - 1);
        const int ox0 = xx >> 1, ox1 = min((xx + 1) >> 1, OW - 1);
        float acc = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy)
          for (int ox = ox0; ox <= ox1; ++ox) {
            const int o = oy * OW + ox;
            if (argp[o] == p) acc += valp[o];
          }
        r[e] = acc;
        s1 += acc;
        s2 += acc * (v[e] - m) * invstd;                          // bn_bwd_split_kernel<0>'s term
      }
      o4[i] = r;
    }
    }
    __syncthreads();
  }
  const float tb = bsum(s1, red);
  const float tg = bsum(s2, red);
  if (threadIdx.x == 0) {
    part[c * kBnSplits + sp] = tb;
    part[(size_t)C * kBnSplits + c * kBnSplits + sp] = tg;
  }
}

}  // namespace

extern "C" int dm_bn_relu_maxpool_bwd(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                                      const float* gamma, const float* beta, float eps, const float* grad_out,
                                      float* grad_x, float* grad_gamma, float* grad_beta, float* scratch,
                                      dm_stream_t stream) {
  if (!x || !mean || !var || !gamma || !beta || !grad_out || !grad_x || !grad_gamma || !grad_beta) return DM_ERR_INVALID_ARG;
  if (NB <= 0 || C <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)NB * C * OH * OW;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const size_t plane_lds = ((size_t)H * W + 2 * (size_t)OH * OW) * 4;
  // (a workgroup per (channel, split): needs enough channels to fill the chip -- 256 x 16 x 28 x 28 measured 0.082 vs 0.055 ms)
  if (scratch && NB >= kBnSplits && C * kBnSplits >= 4 * dm_num_cus() && (H * W) % 4 == 0 && plane_lds <= 64 * 1024) {
    DM_LAUNCH(maxpool_relu_bwd_sums_kernel, dim3(C, kBnSplits), dim3(256), plane_lds, st, x, NB, C, H, W, mean, var, gamma, beta,
              eps, grad_out, grad_x, OH, OW, scratch);
    rc = dm_check_launch();
    if (rc != DM_OK) return rc;
    DM_LAUNCH(bn_bwd_split_kernel<1>, dim3(C, kBnSplits), dim3(256), 0, st, x, grad_x, NB, C, H * W, mean, var, gamma, eps,
              scratch, grad_gamma, grad_beta);
    return dm_check_launch();
  }
  if ((H * W) % 4 == 0 && plane_lds <= 64 * 1024) {
    const int blocks = min(NB * C, 16 * dm_num_cus());
    DM_LAUNCH(maxpool_relu_bwd_lds_kernel, dim3(blocks), dim3(256), plane_lds, st, x, NB, C, H, W, mean, var, gamma, beta,
              eps, grad_out, grad_x, OH, OW);
    rc = dm_check_launch();
  } else {
    if (hipMemsetAsync(grad_x, 0, (size_t)NB * C * H * W * sizeof(float), st) != hipSuccess) return DM_ERR_LAUNCH;
    const int blocks = (int)min((size_t)dm_ceil_div((long long)total, 256), (size_t)16384);
    DM_LAUNCH(maxpool_relu_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, NB, C, H, W, mean, var, gamma, beta, eps, grad_out,
              grad_x, OH, OW);
    rc = dm_check_launch();
  }
  if (rc != DM_OK) return rc;
  if (scratch && NB >= kBnSplits) {
    DM_LAUNCH(bn_bwd_split_kernel<0>, dim3(C, kBnSplits), dim3(256), 0, st, x, grad_x, NB, C, H * W, mean, var, gamma, eps,
              scratch, grad_gamma, grad_beta);
    rc = dm_check_launch();
    if (rc != DM_OK) return rc;
    DM_LAUNCH(bn_bwd_split_kernel<1>, dim3(C, kBnSplits), dim3(256), 0, st, x, grad_x, NB, C, H * W, mean, var, gamma, eps,
              scratch, grad_gamma, grad_beta);
    return dm_check_launch();
  }
  DM_LAUNCH(bn_bwd_kernel, dim3(C), dim3(1024), 0, st, x, grad_x, NB, C, H * W, mean, var, gamma, eps, grad_gamma,
            grad_beta);
  return dm_check_launch();
}
