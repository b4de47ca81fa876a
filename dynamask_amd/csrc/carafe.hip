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

// Fast path for the mask head's shape (scale 2, H*W <= 256, e.g. 14x14 -> 28x28).
// Workgroup = one image x CT channels of one group, thread = one INPUT pixel.  The four
// output pixels of an input pixel read the same K*K taps and differ only in their
// normalised kernels, so a thread keeps its 4 x K*K weights in registers, reads every
// tap once -- from an LDS tile of the zero-padded input, channel-quad interleaved so
// that one ds_read_b128 brings 4 channels -- and feeds 16 FMAs with it.  (The first
// kernel gathered every tap of every output pixel from global memory: 420 GB/s.)
template <int K>
__global__ __launch_bounds__(256) void carafe_tile_kernel(const float* __restrict__ x, const float* __restrict__ enc, int C,
                                                          int H, int W, int group, float* __restrict__ out, int CT) {
  constexpr int KK = K * K, R = K / 2;
  extern __shared__ __attribute__((aligned(16))) float4 tile[];    // [CT/4][(H+2R)*(W+2R)]
  const int cpg = C / group;
  const int chunks = cpg / CT;
  int bid = blockIdx.x;
  const int chunk = bid % chunks;
  bid /= chunks;
  const int g = bid % group;
  const int n = bid / group;
  const int c0 = g * cpg + chunk * CT;
  const int NQ = CT / 4;
  const int PW = W + 2 * R, PH = H + 2 * R, PP = PW * PH;
  const int HW = H * W, OW = 2 * W;
  const int tid = threadIdx.x;
  // ---- stage the padded tile: slot = (quad, padded pixel); 4 channel planes -> one float4
  const float* xb = x + ((size_t)n * C + c0) * HW;
  for (int s = tid; s < NQ * PP; s += 256) {
    const int q = s / PP, p = s - q * PP;
    const int py = p / PW - R, px = p - (p / PW) * PW - R;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (py >= 0 && py < H && px >= 0 && px < W) {
      const float* src = xb + (size_t)(q * 4) * HW + py * W + px;
      v = make_float4(src[0], src[HW], src[2 * HW], src[3 * HW]);
    }
    tile[s] = v;
  }
  // ---- this thread's 4 normalised kernels (pixel_shuffle + softmax of mmcv's kernel_normalizer)
  const bool active = tid < HW;
  const int y = active ? tid / W : 0, xx = active ? tid - (tid / W) * W : 0;
  float wk[4][KK];
  {
    const float* e = enc + ((size_t)n * (group * KK * 4) + (size_t)(g * KK) * 4) * HW + y * W + xx;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      float mx = -INFINITY;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        wk[sub][kk] = active ? e[(size_t)(kk * 4 + sub) * HW] : 0.f;
        mx = fmaxf(mx, wk[sub][kk]);
      }
      float sum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        wk[sub][kk] = expf(wk[sub][kk] - mx);
        sum += wk[sub][kk];
      }
      const float inv = 1.f / sum;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) wk[sub][kk] *= inv;
    }
  }
  __syncthreads();
  if (!active) return;
  const float4* t0 = tile + y * PW + xx;     // tap (i, j) sits at (y + i) * PW + xx + j in the padded tile
  for (int q = 0; q < NQ; ++q) {
    float4 acc[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) acc[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* tq = t0 + q * PP;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float4 v = tq[i * PW + j];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const float w = wk[sub][i * K + j];
          acc[sub].x += w * v.x;
          acc[sub].y += w * v.y;
          acc[sub].z += w * v.z;
          acc[sub].w += w * v.w;
        }
      }
    // output pixel (2y + dy, 2x + dx) = sub dy*2 + dx: the two dx of a row are adjacent -> float2 stores
    float* ob = out + ((size_t)n * C + c0 + q * 4) * (4 * HW) + (size_t)(2 * y) * OW + 2 * xx;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      float* o = ob + dy * OW;
      *reinterpret_cast<float2*>(o) = make_float2(acc[dy * 2].x, acc[dy * 2 + 1].x);
      *reinterpret_cast<float2*>(o + (size_t)4 * HW) = make_float2(acc[dy * 2].y, acc[dy * 2 + 1].y);
      *reinterpret_cast<float2*>(o + (size_t)8 * HW) = make_float2(acc[dy * 2].z, acc[dy * 2 + 1].z);
      *reinterpret_cast<float2*>(o + (size_t)12 * HW) = make_float2(acc[dy * 2].w, acc[dy * 2 + 1].w);
    }
  }
}

}  // namespace

extern "C" int dm_carafe_fwd(const float* x, const float* enc, int NB, int C, int H, int W, int up_kernel, int group,
                             int scale, float* out, dm_stream_t stream) {
  if (!x || !enc || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 || scale <= 0 || C % group != 0)
    return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  {
    const int cpg = C / group;
    const int CTt = (cpg % 32 == 0) ? 32 : (cpg % 16 == 0) ? 16 : (cpg % 4 == 0) ? 4 : 0;
    if (scale == 2 && H * W <= 256 && CTt > 0 && (up_kernel == 5 || up_kernel == 3)) {
      const int R = up_kernel / 2;
      const size_t lds = (size_t)(CTt / 4) * (H + 2 * R) * (W + 2 * R) * sizeof(float4);
      if (lds <= 64 * 1024) {
        const dim3 grid((unsigned)(NB * group * (cpg / CTt)));
        if (up_kernel == 5)
          DM_LAUNCH(carafe_tile_kernel<5>, grid, dim3(256), lds, (hipStream_t)stream, x, enc, C, H, W, group, out, CTt);
        else
          DM_LAUNCH(carafe_tile_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, x, enc, C, H, W, group, out, CTt);
        return dm_check_launch();
      }
    }
  }
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
