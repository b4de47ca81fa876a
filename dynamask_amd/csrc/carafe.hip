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

// ---------------------------------------------------------------------------
// CARAFE backward for the mask head's shape (scale 2, H*W <= 256, K in {3, 5}): gradients of
//   out[n,c,2y+dy,2x+dx] = sum_{i,j} x[n,c,y+i-R,x+j-R] * m[n,g,sub,(i,j),y,x],   m = softmax_(i,j)(enc shuffled)
// with respect to x and to enc (mmcv's carafe backward + kernel_normalizer backward;
// upsample of FCNMaskHead, mask_heads/fcn_mask_head.py:84-87).  Same decomposition as the forward tile
// kernel (thread = input pixel, workgroup = image x channel chunk of one group):
//   carafe_bwd_mask_kernel   g_m[sub][(i,j)] += sum_c g_out[c,sub] * x[c, tap (i,j)]  (x tile in LDS; 4*K*K
//                            accumulators per thread; partial sums of the chunks meet by float atomics)
//   carafe_bwd_enc_kernel    softmax backward: g_enc = m * (g_m - sum m*g_m)
//   carafe_bwd_x_kernel      g_x[c,yy,xx] = sum_{i,j,sub} g_out[c, out pixel of (yy-i+R, xx-j+R)] * m[...]: gather
//                            form (no atomics): the normalised kernels of the whole image sit in LDS, a thread
//                            collects the 4*K*K weights that reach ITS input pixel once and reuses them for all
//                            channels; g_out is staged per channel quad.
template <int K>
__global__ __launch_bounds__(256) void carafe_bwd_mask_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                              int C, int H, int W, int group, float* __restrict__ gm,
                                                              int CT) {
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
  __syncthreads();
  if (tid >= HW) return;
  const int y = tid / W, xx = tid - y * W;
  float acc[4][KK];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) acc[sub][kk] = 0.f;
  const float4* t0 = tile + y * PW + xx;
  for (int q = 0; q < NQ; ++q) {
    // the 4 output pixels' gradients of the quad's 4 channels
    const float* gb = gout + ((size_t)n * C + c0 + q * 4) * (4 * HW) + (size_t)(2 * y) * OW + 2 * xx;
    float go[4][4];      // [channel][sub]
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float2 r0 = *reinterpret_cast<const float2*>(gb + (size_t)ch * 4 * HW);
      const float2 r1 = *reinterpret_cast<const float2*>(gb + (size_t)ch * 4 * HW + OW);
      go[ch][0] = r0.x; go[ch][1] = r0.y; go[ch][2] = r1.x; go[ch][3] = r1.y;
    }
    const float4* tq = t0 + q * PP;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float4 v = tq[i * PW + j];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
          acc[sub][i * K + j] += go[0][sub] * v.x + go[1][sub] * v.y + go[2][sub] * v.z + go[3][sub] * v.w;
      }
  }
  float* gmp = gm + ((size_t)n * group + g) * (4 * KK) * HW + tid;     // [n][g][sub][kk][HW]
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) atomicAdd(gmp + (size_t)(sub * KK + kk) * HW, acc[sub][kk]);
}

template <int K>
__global__ __launch_bounds__(256) void carafe_bwd_enc_kernel(const float* __restrict__ enc, const float* __restrict__ gm,
                                                             int NB, int H, int W, int group, float* __restrict__ genc) {
  constexpr int KK = K * K;
  const int HW = H * W;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // (n, g, sub, pixel)
  if (idx >= (long long)NB * group * 4 * HW) return;
  const int p = (int)(idx % HW);
  const int sub = (int)((idx / HW) % 4);
  const int g = (int)((idx / ((long long)HW * 4)) % group);
  const int n = (int)(idx / ((long long)HW * 4 * group));
  const float* e = enc + ((size_t)n * (group * KK * 4) + (size_t)(g * KK) * 4 + sub) * HW + p;   // channel (g*KK+kk)*4+sub
  const float* gmp = gm + (((size_t)n * group + g) * 4 + sub) * KK * HW + p;
  float m[KK];
  float mx = -INFINITY;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    m[kk] = e[(size_t)kk * 4 * HW];
    mx = fmaxf(mx, m[kk]);
  }
  float sum = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    m[kk] = expf(m[kk] - mx);
    sum += m[kk];
  }
  const float inv = 1.f / sum;
  float dot = 0.f;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    m[kk] *= inv;
    dot += m[kk] * gmp[(size_t)kk * HW];
  }
  float* ge = genc + ((size_t)n * (group * KK * 4) + (size_t)(g * KK) * 4 + sub) * HW + p;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) ge[(size_t)kk * 4 * HW] = m[kk] * (gmp[(size_t)kk * HW] - dot);
}

template <int K>
__global__ __launch_bounds__(256) void carafe_bwd_x_kernel(const float* __restrict__ enc, const float* __restrict__ gout,
                                                           int C, int H, int W, int group, float* __restrict__ gx, int CT) {
  constexpr int KK = K * K, R = K / 2;
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int HW = H * W, OW = 2 * W, OHW = 4 * HW;
  float* mtab = lds_f;                                              // [4*KK][HW] normalised kernels of the image
  float4* gtile = reinterpret_cast<float4*>(lds_f + 4 * KK * HW);   // [2H*2W] one channel quad of g_out, interleaved
  const int cpg = C / group;
  const int chunks = cpg / CT;
  int bid = blockIdx.x;
  const int chunk = bid % chunks;
  bid /= chunks;
  const int g = bid % group;
  const int n = bid / group;
  const int c0 = g * cpg + chunk * CT;
  const int tid = threadIdx.x;
  const bool active = tid < HW;
  const int y = active ? tid / W : 0, xx = active ? tid - (tid / W) * W : 0;
  if (active) {
    const float* e = enc + ((size_t)n * (group * KK * 4) + (size_t)(g * KK) * 4) * HW + tid;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      float m[KK];
      float mx = -INFINITY;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        m[kk] = e[(size_t)(kk * 4 + sub) * HW];
        mx = fmaxf(mx, m[kk]);
      }
      float sum = 0.f;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        m[kk] = expf(m[kk] - mx);
        sum += m[kk];
      }
      const float inv = 1.f / sum;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) mtab[(size_t)(sub * KK + kk) * HW + tid] = m[kk] * inv;
    }
  }
  __syncthreads();
  // weights that reach input pixel (y, xx): tap (i, j) of source pixel (y - i + R, xx - j + R)
  float wg[KK][4];
  int src[KK];           // offset of the source pixel's first output pixel in the 2H x 2W tile, or -1
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int sy = y - i + R, sx = xx - j + R;
      const bool ok = active && sy >= 0 && sy < H && sx >= 0 && sx < W;
      src[i * K + j] = ok ? (2 * sy) * OW + 2 * sx : 0;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
        wg[i * K + j][sub] = ok ? mtab[(size_t)(sub * KK + i * K + j) * HW + sy * W + sx] : 0.f;
    }
  for (int q = 0; q < CT / 4; ++q) {
    __syncthreads();
    const float* gb = gout + ((size_t)n * C + c0 + q * 4) * OHW;
    for (int s = tid; s < OHW; s += 256) gtile[s] = make_float4(gb[s], gb[OHW + s], gb[2 * OHW + s], gb[3 * OHW + s]);
    __syncthreads();
    if (active) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const float4* t = gtile + src[kk];
        const float4 v0 = t[0], v1 = t[1], v2 = t[OW], v3 = t[OW + 1];
        acc.x += wg[kk][0] * v0.x + wg[kk][1] * v1.x + wg[kk][2] * v2.x + wg[kk][3] * v3.x;
        acc.y += wg[kk][0] * v0.y + wg[kk][1] * v1.y + wg[kk][2] * v2.y + wg[kk][3] * v3.y;
        acc.z += wg[kk][0] * v0.z + wg[kk][1] * v1.z + wg[kk][2] * v2.z + wg[kk][3] * v3.z;
        acc.w += wg[kk][0] * v0.w + wg[kk][1] * v1.w + wg[kk][2] * v2.w + wg[kk][3] * v3.w;
      }
      float* o = gx + ((size_t)n * C + c0 + q * 4) * HW + tid;
      o[0] = acc.x;
      o[HW] = acc.y;
      o[2 * HW] = acc.z;
      o[3 * HW] = acc.w;
    }
  }
}

template <int K>
int launch_carafe_bwd(const float* x, const float* enc, const float* gout, int NB, int C, int H, int W, int group,
                      float* gx, float* genc, float* gm, hipStream_t st) {
  constexpr int KK = K * K, R = K / 2;
  const int cpg = C / group, HW = H * W;
  const int CT = (cpg % 32 == 0) ? 32 : (cpg % 16 == 0) ? 16 : 4;      // x tile of CT channels: 41 KB at 32 (18 x 18 padded)
  if (hipMemsetAsync(gm, 0, (size_t)NB * group * 4 * KK * HW * sizeof(float), st) != hipSuccess) return DM_ERR_LAUNCH;
  const size_t lds_a = (size_t)(CT / 4) * (H + 2 * R) * (W + 2 * R) * sizeof(float4);
  const size_t lds_b = (size_t)4 * KK * HW * sizeof(float) + (size_t)4 * HW * sizeof(float4);
  if (lds_a > 64 * 1024 || lds_b > 160 * 1024) return DM_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(NB * group * (cpg / CT)));
  static bool attr_set[DM_MAX_DEVICES] = {false};
  if (lds_b > 64 * 1024 &&
      dm_ensure_lds_limit(reinterpret_cast<const void*>(&carafe_bwd_x_kernel<K>), 160 * 1024, attr_set) != DM_OK)
    return DM_ERR_LAUNCH;
  DM_LAUNCH(carafe_bwd_mask_kernel<K>, grid, dim3(256), lds_a, st, x, gout, C, H, W, group, gm, CT);
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(carafe_bwd_enc_kernel<K>, dim3((unsigned)dm_ceil_div((long long)NB * group * 4 * HW, 256)), dim3(256), 0, st, enc,
            (const float*)gm, NB, H, W, group, genc);
  rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(carafe_bwd_x_kernel<K>, grid, dim3(256), lds_b, st, enc, gout, C, H, W, group, gx, CT);
  return dm_check_launch();
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

extern "C" long long dm_carafe_bwd_scratch_floats(int NB, int H, int W, int up_kernel, int group) {
  return (long long)NB * group * 4 * up_kernel * up_kernel * H * W;
}

extern "C" int dm_carafe_bwd(const float* x, const float* enc, const float* grad_out, int NB, int C, int H, int W,
                             int up_kernel, int group, int scale, float* grad_x, float* grad_enc, float* scratch,
                             dm_stream_t stream) {
  if (!x || !enc || !grad_out || !grad_x || !grad_enc || !scratch || NB < 0 || C <= 0 || H <= 0 || W <= 0 || group <= 0 ||
      C % group != 0)
    return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  // the mask head's shape only (14x14 -> 28x28): one image's normalised kernels fit LDS
  if (scale != 2 || H * W > 256 || (C / group) % 4 != 0 || W % 1 != 0) return DM_ERR_UNSUPPORTED;
  if (up_kernel == 5) return launch_carafe_bwd<5>(x, enc, grad_out, NB, C, H, W, group, grad_x, grad_enc, scratch, (hipStream_t)stream);
  if (up_kernel == 3) return launch_carafe_bwd<3>(x, enc, grad_out, NB, C, H, W, group, grad_x, grad_enc, scratch, (hipStream_t)stream);
  return DM_ERR_UNSUPPORTED;
}
