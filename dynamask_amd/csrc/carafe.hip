// K17: CARAFE content-aware reassembly, forward.  Fuses mmcv's
// kernel_normalizer (pixel_shuffle + softmax over the k*k kernel) with the
// reassembly: a thread owns one output pixel, builds its normalised k*k kernel
// in registers once and reuses it over the channel chunk of the workgroup.
#include "common.h"

namespace {

template <int K>
__global__ __launch_bounds__(256) void carafe_kernel(const float* __restrict__ x, const float* __restrict__ enc, int NB,
                                                     int C, int H, int W, int group, int scale, float* __restrict__ out,
                                                     int CT, int pix_blocks) {
  constexpr int KK = K * K;
  const int OH = H * scale, OW = W * scale;
  const int cpg = C / group;
  const int chunks_per_group = (cpg + CT - 1) / CT;
  int bid = blockIdx.x;
  const int pb = bid % pix_blocks;
  bid /= pix_blocks;
  const int chunk = bid % chunks_per_group;
  bid /= chunks_per_group;
  const int g = bid % group;
  const int n = bid / group;
  const int pix = pb * blockDim.x + threadIdx.x;
  if (pix >= OH * OW) return;
  const int oy = pix / OW, ox = pix - oy * OW;
  const int y = oy / scale, xx = ox / scale;
  const int sub = (oy - y * scale) * scale + (ox - xx * scale);
  // pixel_shuffle: mask[n, ch, oy, ox] = enc[n, ch*scale^2 + sub, y, x], ch = g*KK + kk
  const int s2 = scale * scale;
  const float* e = enc + (((size_t)n * (group * KK * s2)) + (size_t)(g * KK) * s2 + sub) * H * W + y * W + xx;
  float wk[KK];
  float mx = -INFINITY;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    wk[kk] = e[(size_t)kk * s2 * H * W];
    mx = fmaxf(mx, wk[kk]);
  }
  float sum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    wk[kk] = expf(wk[kk] - mx);
    sum += wk[kk];
  }
  const float inv = 1.f / sum;
  int off[KK];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int yy = y + i - K / 2, xj = xx + j - K / 2;
      const bool ok = yy >= 0 && yy < H && xj >= 0 && xj < W;
      off[i * K + j] = ok ? yy * W + xj : 0;
      wk[i * K + j] = ok ? wk[i * K + j] * inv : 0.f;
    }
  const int c0 = g * cpg + chunk * CT;
  const int c1 = min(c0 + CT, (g + 1) * cpg);
  for (int c = c0; c < c1; ++c) {
    const float* xc = x + ((size_t)n * C + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) acc += xc[off[kk]] * wk[kk];
    out[((size_t)n * C + c) * OH * OW + pix] = acc;
  }
}

}  // namespace

extern "C" int dm_carafe_fwd(const float* x, const float* enc, int NB, int C, int H, int W, int up_kernel, int group,
                             int scale, float* out, dm_stream_t stream) {
  if (!x || !enc || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || scale <= 0 || C % group != 0)
    return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  const int CT = 16;
  const int cpg = C / group;
  const int chunks = dm_ceil_div(cpg, CT);
  const int pix_blocks = dm_ceil_div((long long)H * scale * W * scale, 256);
  const dim3 grid((unsigned)(NB * group * chunks * pix_blocks));
  hipStream_t st = (hipStream_t)stream;
  if (up_kernel == 5) {
    DM_LAUNCH(carafe_kernel<5>, grid, dim3(256), 0, st, x, enc, NB, C, H, W, group, scale, out, CT, pix_blocks);
  } else if (up_kernel == 3) {
    DM_LAUNCH(carafe_kernel<3>, grid, dim3(256), 0, st, x, enc, NB, C, H, W, group, scale, out, CT, pix_blocks);
  } else {
    return DM_ERR_UNSUPPORTED;
  }
  return dm_check_launch();
}
