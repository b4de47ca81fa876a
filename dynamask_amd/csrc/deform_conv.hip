// K8: deformable convolution v1 (3x3, stride 1, pad 1, dilation 1, groups 1, no
// bias), forward.  Arithmetic spec: mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu
// :84-115 (bilinear) and :190-243 (im2col) of the reference tree; the reference
// materialises the [C*9, N*H*W] column matrix in HBM and calls a library GEMM
// (deform_conv_cuda.cpp:198-237).  Here the deformable im2col is the PRODUCER
// of the MFMA B operand: a workgroup owns TN flat pixels x ALL output channels,
// gathers the bilinear-sampled columns of CK input channels into LDS and feeds
// them straight to v_mfma_f32_32x32x2_f32.  The column matrix never exists in
// HBM, and because one workgroup covers every output channel each sample is
// gathered exactly once.
//
// The 4 tap offsets + 4 bilinear weights of a (kernel tap, pixel) pair depend
// only on the deformable group, so each thread keeps them in registers for its
// (tap, pixel) pairs and reuses them for all C/deform_groups channels.
#include "common.h"

namespace {

struct DcnArgs {
  const float* x;
  const float* offset;
  int NB, C, H, W, HW, Q;
  const float* wp;  // [9][C][CoutP]
  int Cout, CoutP, dg, relu;
  float* out;
  int MT;
};

template <int WGM, int WGN, int WM, int WN, int CK>
__global__ __launch_bounds__(WGM* WGN * 64) void deform_conv_kernel(DcnArgs a) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TG = NT / TN;                 // thread groups sharing a pixel column
  constexpr int MAXT = (9 + TG - 1) / TG;     // taps owned per thread
  static_assert(NT % TN == 0, "threads must tile the pixel columns");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                    // [9][CK][TM]
  float* ldsB = lds + 9 * CK * TM;      // [CK][9][TN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WGN, wave_n = wave % WGN;
  const int hi = lane >> 5, l31 = lane & 31;
  const int m_tile = blockIdx.x % a.MT;
  const int n_tile = blockIdx.x / a.MT;
  const int m0 = m_tile * TM;
  const int q0 = n_tile * TN;
  const int HW = a.HW, W = a.W, H = a.H;

  // MFMA-side columns of this lane
  int col_n[WN], col_p[WN];
  bool col_ok[WN];
#pragma unroll
  for (int wn = 0; wn < WN; ++wn) {
    int q = q0 + (wave_n * WN + wn) * 32 + l31;
    col_ok[wn] = q < a.Q;
    q = min(q, a.Q - 1);
    col_n[wn] = q / HW;
    col_p[wn] = q - col_n[wn] * HW;
  }

  // gather-side column of this thread
  const int gj = tid % TN;
  const int tg = tid / TN;
  int gq = q0 + gj;
  const bool g_ok = gq < a.Q;
  gq = min(gq, a.Q - 1);
  const int gn = gq / HW;
  const int gp = gq - gn * HW;
  const int gy = gp / W, gx = gp - gy * W;

  int o1[MAXT], o2[MAXT], o3[MAXT], o4[MAXT];
  float w1[MAXT], w2[MAXT], w3[MAXT], w4[MAXT];
  int cur_group = -1;
  const int cpg = a.C / a.dg;

  dm_f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int c0 = 0; c0 < a.C; c0 += CK) {
    const int ckv = min(CK, a.C - c0);
    const int ckp = (ckv + 1) & ~1;
    const int group = c0 / cpg;   // CK divides cpg (checked on the host)
    if (group != cur_group) {
      cur_group = group;
      const float* offp = a.offset + ((size_t)gn * a.dg + group) * 18 * HW + gp;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int tap = tg + t * TG;
        o1[t] = o2[t] = o3[t] = o4[t] = 0;
        w1[t] = w2[t] = w3[t] = w4[t] = 0.f;
        if (tap < 9 && g_ok) {
          const int ki = tap / 3, kj = tap - ki * 3;
          const float off_h = offp[(size_t)(2 * tap) * HW];
          const float off_w = offp[(size_t)(2 * tap + 1) * HW];
          const float h_im = (float)(gy - 1 + ki) + off_h;
          const float w_im = (float)(gx - 1 + kj) + off_w;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw = 1.f - lw;
            if (h_low >= 0 && w_low >= 0) { o1[t] = h_low * W + w_low; w1[t] = hh * hw; }
            if (h_low >= 0 && w_high <= W - 1) { o2[t] = h_low * W + w_high; w2[t] = hh * lw; }
            if (h_high <= H - 1 && w_low >= 0) { o3[t] = h_high * W + w_low; w3[t] = lh * hw; }
            if (h_high <= H - 1 && w_high <= W - 1) { o4[t] = h_high * W + w_high; w4[t] = lh * lw; }
          }
        }
      }
    }

    // ---- stage A (weights) --------------------------------------------------
    for (int idx = tid; idx < 9 * CK * (TM / 4); idx += NT) {
      const int row = idx / (TM / 4);
      const int c4 = idx - row * (TM / 4);
      const int tap = row / CK;
      const int ci = row - tap * CK;
      const int co = m0 + c4 * 4;
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ci < ckv && co < a.CoutP) v = *reinterpret_cast<const dm_f32x4*>(a.wp + ((size_t)(tap * a.C + c0 + ci)) * a.CoutP + co);
      *reinterpret_cast<dm_f32x4*>(ldsA + row * TM + c4 * 4) = v;
    }
    // ---- stage B: deformable im2col of CK channels --------------------------
    {
      const float* xp = a.x + ((size_t)gn * a.C + c0) * HW;
      for (int ci = 0; ci < ckp; ++ci) {
        const float* xc = xp + (size_t)ci * HW;
        const bool live = ci < ckv;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int tap = tg + t * TG;
          if (tap < 9) {
            float v = 0.f;
            if (live) v = w1[t] * xc[o1[t]] + w2[t] * xc[o2[t]] + w3[t] * xc[o3[t]] + w4[t] * xc[o4[t]];
            ldsB[(ci * 9 + tap) * TN + gj] = v;
          }
        }
      }
    }
    __syncthreads();

    // ---- MFMA ------------------------------------------------------------------
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* pa = ldsA + (tap * CK + hi) * TM + wave_m * (WM * 32) + l31;
      const float* pb = ldsB + (hi * 9 + tap) * TN + wave_n * (WN * 32) + l31;
      for (int kk = 0; kk < ckp; kk += 2) {
        float av[WM], bv[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) av[i] = pa[kk * TM + i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = pb[kk * 9 * TN + j * 32];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + (wave_m * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (co < a.Cout) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (col_ok[j]) {
            float v = acc[i][j][r];
            if (a.relu) v = fmaxf(v, 0.f);
            a.out[((size_t)col_n[j] * a.Cout + co) * HW + col_p[j]] = v;
          }
        }
      }
    }
  }
}

template <int WGM, int WGN, int WM, int WN, int CK>
int launch_dcn(DcnArgs& a, hipStream_t st) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  a.MT = dm_ceil_div(a.CoutP, TM);
  const int NTiles = dm_ceil_div(a.Q, TN);
  const size_t lds_bytes = sizeof(float) * ((size_t)9 * CK * TM + (size_t)CK * 9 * TN);
  if (lds_bytes > 64 * 1024) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH((deform_conv_kernel<WGM, WGN, WM, WN, CK>), dim3(a.MT * NTiles), dim3(NT), lds_bytes, st, a);
  return dm_check_launch();
}

}  // namespace

extern "C" int dm_deform_conv_fwd(const float* x, const float* offset, int NB, int C, int H, int W,
                                  const float* w_packed, int Cout, int deform_groups, int relu, float* out,
                                  dm_stream_t stream) {
  if (!x || !offset || !w_packed || !out) return DM_ERR_INVALID_ARG;
  if (NB < 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || deform_groups <= 0 || C % deform_groups != 0)
    return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if ((C / deform_groups) % 4 != 0) return DM_ERR_UNSUPPORTED;  // channel chunk must not straddle a deformable group
  if (NB == 0) return DM_OK;
  DcnArgs a;
  a.x = x; a.offset = offset; a.NB = NB; a.C = C; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wp = w_packed; a.Cout = Cout; a.CoutP = dm_conv_packed_cout(Cout); a.dg = deform_groups; a.relu = relu; a.out = out;
  hipStream_t st = (hipStream_t)stream;
  if (Cout > 128) return launch_dcn<4, 2, 2, 2, 4>(a, st);   // 256 couts x 128 px, 8 waves
  if (Cout > 64) return launch_dcn<2, 2, 2, 2, 4>(a, st);    // 128 x 128, 4 waves
  if (Cout > 32) return launch_dcn<1, 4, 2, 1, 4>(a, st);    // 64 x 128
  return launch_dcn<1, 4, 1, 1, 4>(a, st);                   // 32 x 128
}
