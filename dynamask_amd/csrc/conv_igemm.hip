// K5/K6: dense stride-1 "same" convolution (1x1 / 3x3) as an implicit GEMM on the
// fp32-input MFMA units of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
//
// GEMM view (per launch):   Out[co, q] = bias[co] + sum_k  Wp[k, co] * X[k, q]
//   rows   co : output channels (MFMA "A" operand = packed weights, M side)
//   cols   q  : FLAT pixel index over the whole batch, q = n*H*W + y*W + x
//               (MFMA "B" operand = the im2col of the input, N side).  Flat
//               columns mean no padding waste for 14x14 RoI maps (196 px) and one
//               kernel for RoI tensors and whole FPN maps alike.
//   k         : (tap, input channel); the channel-concatenation of up to 4
//               source tensors is walked in the K loop (torch.cat never
//               materialises).
// D layout of the 32x32 MFMA puts the pixel on the lane, so every accumulator
// register stores 32 consecutive pixels of one output channel: 128-B coalesced
// NCHW stores.
//
// LDS per workgroup: A chunk [taps][CK][TM] + B chunk [CK][plane].
//   3x3: B holds, per input channel, the image rows the tile's pixels need (+1
//        halo row/column of zeros each side, per image segment), so a tap is a
//        constant LDS offset from the lane's base address.
//   1x1: B is [CK][TN] straight.
// fp32 MFMA issues one 32x32x2 per 64 cycles per SIMD and needs only 512 B of
// operands for it, so LDS bandwidth is ~25 % used: the kernel is MFMA-bound by
// construction; 2-3 workgroups per CU overlap staging with compute.
#include "common.h"

namespace {

struct ConvArgs {
  const float* src[DM_MAX_SOURCES];
  int src_c[DM_MAX_SOURCES];
  long long src_bs[DM_MAX_SOURCES];  // batch stride of each source, in floats
  int num_srcs;
  int NB, H, W, HW, Q;
  const float* wp;
  const float* bias;
  int Cin, Cout, CoutP;
  int relu;
  float* out;
  int out_ch_total, out_ch_offset;
  int Wp, plane;  // 3x3: W+2, Rmax*Wp ; 1x1: unused, TN
  int MT;         // number of cout tiles
  int shuffle;    // deconv 2x2/s2 epilogue: packed cout = phase*shuffle + co, stored at (2y+dy, 2x+dx)
};

template <int KS, int WGM, int WGN, int WM, int WN, int CK>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_igemm_kernel(ConvArgs a) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TAPS = KS * KS;
  constexpr int MAXPOS = 4;  // 3x3: staged plane positions per thread

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                       // [TAPS][CK][TM]
  float* ldsB = lds + TAPS * CK * TM;      // [CK][plane]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WGN;
  const int wave_n = wave % WGN;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  const int m_tile = blockIdx.x % a.MT;
  const int n_tile = blockIdx.x / a.MT;
  const int m0 = m_tile * TM;
  const int q0 = n_tile * TN;
  const int HW = a.HW, W = a.W, H = a.H;
  const int plane = a.plane;

  // ---- tile geometry (uniform) -------------------------------------------
  const int qlast = min(q0 + TN, a.Q) - 1;
  const int n0 = q0 / HW;
  const int n1 = qlast / HW;
  const int y00 = (q0 - n0 * HW) / W;
  const int y1l = (qlast - n1 * HW) / W;
  const int rows0 = (n1 > n0) ? (H - y00) : (y1l - y00 + 1);
  const int Wp = a.Wp;

  // ---- per-lane column bookkeeping ----------------------------------------
  int lane_base[WN];   // LDS float offset of tap (0,0) of this lane's pixel
  int col_n[WN], col_p[WN];
  bool col_ok[WN];
#pragma unroll
  for (int wn = 0; wn < WN; ++wn) {
    const int j = (wave_n * WN + wn) * 32 + l31;
    int q = q0 + j;
    col_ok[wn] = q < a.Q;
    q = min(q, a.Q - 1);
    const int n = q / HW;
    const int p = q - n * HW;
    col_n[wn] = n;
    col_p[wn] = p;
    if (KS == 3) {
      const int y = p / W;
      const int x = p - y * W;
      const int seg = n - n0;
      const int r = (seg == 0) ? (y - y00 + 1) : (rows0 + 2) + (seg - 1) * (H + 2) + (y + 1);
      lane_base[wn] = (r - 1) * Wp + x + hi * plane;
    } else {
      lane_base[wn] = j + hi * plane;
    }
  }

  // ---- 3x3: which plane positions this thread stages (fixed for the tile) --
  int st_pix[MAXPOS], st_n[MAXPOS];
  if (KS == 3) {
    const int Rused = (n1 == n0) ? (rows0 + 2) : (rows0 + 2) + (n1 - n0 - 1) * (H + 2) + (y1l + 3);
#pragma unroll
    for (int k = 0; k < MAXPOS; ++k) {
      const int pos = tid + k * NT;
      st_n[k] = -1;
      st_pix[k] = 0;
      if (pos < plane) {
        const int r = pos / Wp;
        const int c = pos - r * Wp;
        int seg, y;
        if (r < rows0 + 2) {
          seg = 0;
          y = y00 - 1 + r;
        } else {
          const int rr = r - (rows0 + 2);
          seg = 1 + rr / (H + 2);
          y = rr % (H + 2) - 1;
        }
        const int n = n0 + seg;
        if (r < Rused && n < a.NB && y >= 0 && y < H && c >= 1 && c <= W) {
          st_n[k] = n;
          st_pix[k] = y * W + (c - 1);
        }
      }
    }
  }

  dm_f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int cbase = 0;
  for (int s = 0; s < a.num_srcs; ++s) {
    const float* __restrict__ sp = a.src[s];
    const int Cs = a.src_c[s];
    const size_t bs = (size_t)a.src_bs[s];
    for (int c0 = 0; c0 < Cs; c0 += CK) {
      const int ckv = min(CK, Cs - c0);
      const int ckp = (ckv + 1) & ~1;

      // ---- stage A: packed weights rows (tap, ci) x TM couts ---------------
      for (int idx = tid; idx < TAPS * CK * (TM / 4); idx += NT) {
        const int row = idx / (TM / 4);
        const int c4 = idx - row * (TM / 4);
        const int tap = row / CK;
        const int ci = row - tap * CK;
        const int co = m0 + c4 * 4;
        dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ci < ckv && co < a.CoutP) {
          v = *reinterpret_cast<const dm_f32x4*>(a.wp + ((size_t)(tap * a.Cin + cbase + c0 + ci)) * a.CoutP + co);
        }
        *reinterpret_cast<dm_f32x4*>(ldsA + row * TM + c4 * 4) = v;
      }
      // ---- stage B ---------------------------------------------------------
      if (KS == 3) {
#pragma unroll
        for (int k = 0; k < MAXPOS; ++k) {
          const int pos = tid + k * NT;
          if (pos < plane) {
            const bool ok = st_n[k] >= 0;
            const float* gp = sp + (size_t)max(st_n[k], 0) * bs + (size_t)c0 * HW + st_pix[k];
            for (int ci = 0; ci < ckp; ++ci) {
              float v = 0.f;
              if (ok && ci < ckv) v = gp[(size_t)ci * HW];
              ldsB[ci * plane + pos] = v;
            }
          }
        }
      } else {
        constexpr int SUB = NT / TN > 0 ? NT / TN : 1;
        const int j = tid % TN;
        const int sub = tid / TN;
        int q = q0 + j;
        const bool ok = q < a.Q;
        q = min(q, a.Q - 1);
        const int n = q / HW;
        const int p = q - n * HW;
        const float* gp = sp + (size_t)n * bs + (size_t)c0 * HW + p;
        for (int ci = sub; ci < ckp; ci += SUB) {
          float v = 0.f;
          if (ok && ci < ckv) v = gp[(size_t)ci * HW];
          ldsB[ci * plane + j] = v;
        }
      }
      __syncthreads();

      // ---- MFMA over this chunk ---------------------------------------------
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int tapoff = (KS == 3) ? ((tap / 3) * Wp + (tap % 3)) : 0;
        const float* pa = ldsA + (tap * CK + hi) * TM + wave_m * (WM * 32) + l31;
        for (int kk = 0; kk < ckp; kk += 2) {
          float av[WM], bv[WN];
#pragma unroll
          for (int i = 0; i < WM; ++i) av[i] = pa[kk * TM + i * 32];
#pragma unroll
          for (int j = 0; j < WN; ++j) bv[j] = ldsB[kk * plane + lane_base[j] + tapoff];
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    cbase += Cs;
  }

  // ---- epilogue: bias + ReLU, 32 consecutive pixels per register ------------
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + (wave_m * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (co < a.Cout) {
        if (a.shuffle == 0) {
          const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (col_ok[j]) {
              float v = acc[i][j][r] + b;
              if (a.relu) v = fmaxf(v, 0.f);
              a.out[((size_t)col_n[j] * a.out_ch_total + a.out_ch_offset + co) * HW + col_p[j]] = v;
            }
          }
        } else {
          const int phase = co / a.shuffle;
          const int oc = co - phase * a.shuffle;
          const int dy = phase >> 1, dx = phase & 1;
          const float b = a.bias ? a.bias[oc] : 0.f;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (col_ok[j]) {
              float v = acc[i][j][r] + b;
              if (a.relu) v = fmaxf(v, 0.f);
              const int y = col_p[j] / W, x = col_p[j] - y * W;
              a.out[(((size_t)col_n[j] * a.shuffle + oc) * (2 * H) + 2 * y + dy) * (2 * W) + 2 * x + dx] = v;
            }
          }
        }
      }
    }
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int kk, int flip,
                                   float* __restrict__ wp, int rows_out, int cols_in, int colsP) {
  // forward:  wp[(tap*Cin + ci)*CoutP + co] = w[co][ci][tap]
  // flipped:  wp[(tap*Cout + co)*CinP + ci] = w[co][ci][kk-1-tap]
  const int total = kk * rows_out * colsP;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int col = idx % colsP;
    const int row = (idx / colsP) % rows_out;
    const int tap = idx / (colsP * rows_out);
    float v = 0.f;
    if (col < cols_in) {
      if (!flip) {
        v = w[((size_t)col * Cin + row) * kk + tap];           // col = co, row = ci
      } else {
        v = w[((size_t)row * Cin + col) * kk + (kk - 1 - tap)];  // row = co, col = ci
      }
    }
    wp[idx] = v;
  }
}

template <int KS, int WGM, int WGN, int WM, int WN, int CK>
int launch_conv(ConvArgs& a, hipStream_t st) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  a.MT = dm_ceil_div(a.CoutP, TM);
  const int NTiles = dm_ceil_div(a.Q, TN);
  if (KS == 3) {
    a.Wp = a.W + 2;
    const int nsegmax = dm_ceil_div(TN - 1, a.HW) + 1;
    const int rmax = dm_ceil_div(TN - 1, a.W) + 1 + 2 * nsegmax;
    a.plane = rmax * a.Wp;
    if (a.plane > NT * 4) return DM_ERR_UNSUPPORTED;  // MAXPOS positions per thread
  } else {
    a.Wp = 0;
    a.plane = TN;
  }
  const size_t lds_bytes = sizeof(float) * ((size_t)KS * KS * CK * TM + (size_t)CK * a.plane);
  if (lds_bytes > 64 * 1024) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH((conv_igemm_kernel<KS, WGM, WGN, WM, WN, CK>), dim3(a.MT * NTiles), dim3(NT), lds_bytes, st, a);
  return dm_check_launch();
}

}  // namespace

extern "C" int dm_conv_packed_cout(int Cout) { return (Cout + 31) / 32 * 32; }

extern "C" int dm_conv_pack_weight(const float* w_oihw, int Cout, int Cin, int ksize, int transpose_flip,
                                   float* w_packed, dm_stream_t stream) {
  if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return DM_ERR_INVALID_ARG;
  const int kk = ksize * ksize;
  const int rows_out = transpose_flip ? Cout : Cin;
  const int cols_in = transpose_flip ? Cin : Cout;
  const int colsP = dm_conv_packed_cout(cols_in);
  const int total = kk * rows_out * colsP;
  const int blocks = min(dm_ceil_div(total, 256), 2048);
  DM_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kk,
                     transpose_flip, w_packed, rows_out, cols_in, colsP);
  return dm_check_launch();
}

extern "C" int dm_conv2d_fwd(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                             int num_srcs, int NB, int H, int W,
                             const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                             int out_ch_total, int out_ch_offset, dm_stream_t stream) {
  if (!srcs || !src_channels || num_srcs < 1 || num_srcs > DM_MAX_SOURCES || !w_packed || !out) return DM_ERR_INVALID_ARG;
  if (NB < 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return DM_ERR_INVALID_ARG;
  if (out_ch_offset < 0 || out_ch_offset + Cout > out_ch_total) return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  ConvArgs a;
  a.Cin = 0;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) {
    a.src[s] = s < num_srcs ? srcs[s] : nullptr;
    a.src_c[s] = s < num_srcs ? src_channels[s] : 0;
    a.src_bs[s] = 0;
    if (s < num_srcs) {
      if (!srcs[s] || src_channels[s] <= 0) return DM_ERR_INVALID_ARG;
      a.Cin += src_channels[s];
      a.src_bs[s] = src_batch_strides ? src_batch_strides[s] : (long long)src_channels[s] * H * W;
      if (a.src_bs[s] < (long long)src_channels[s] * H * W) return DM_ERR_INVALID_ARG;
    }
  }
  a.num_srcs = num_srcs;
  a.NB = NB; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wp = w_packed; a.bias = bias; a.Cout = Cout; a.CoutP = dm_conv_packed_cout(Cout);
  a.relu = relu; a.out = out; a.out_ch_total = out_ch_total; a.out_ch_offset = out_ch_offset;
  a.shuffle = 0;
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 3) {
    if (Cout > 64) return launch_conv<3, 2, 2, 2, 2, 8>(a, st);
    if (Cout > 32) return launch_conv<3, 1, 4, 2, 1, 8>(a, st);
    return launch_conv<3, 1, 4, 1, 1, 8>(a, st);
  }
  if (Cout > 64) return launch_conv<1, 2, 2, 2, 2, 32>(a, st);
  if (Cout > 32) return launch_conv<1, 1, 4, 2, 1, 32>(a, st);
  return launch_conv<1, 1, 4, 1, 1, 32>(a, st);
}

// ---------------------------------------------------------------------------
// K16: ConvTranspose2d(k=2, s=2) = four independent 1x1 GEMMs (one per output
// phase).  Run as ONE 1x1 implicit GEMM with 4*Cout packed output channels and
// a pixel-shuffling epilogue.
namespace {
__global__ void pack_deconv_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wp, int colsP) {
  // wp[ci][phase*Cout + co] = w[ci][co][dy][dx], phase = dy*2+dx
  const int total = Cin * colsP;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int col = idx % colsP;
    const int ci = idx / colsP;
    float v = 0.f;
    if (col < 4 * Cout) {
      const int phase = col / Cout, co = col - phase * Cout;
      v = w[((size_t)ci * Cout + co) * 4 + phase];
    }
    wp[idx] = v;
  }
}
}  // namespace

extern "C" int dm_deconv_pack_weight(const float* w_iohw, int Cin, int Cout, float* w_packed, dm_stream_t stream) {
  if (!w_iohw || !w_packed || Cin <= 0 || Cout <= 0) return DM_ERR_INVALID_ARG;
  const int colsP = dm_conv_packed_cout(4 * Cout);
  const int total = Cin * colsP;
  DM_LAUNCH(pack_deconv_kernel, dim3(min(dm_ceil_div(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream,
                     w_iohw, Cin, Cout, w_packed, colsP);
  return dm_check_launch();
}

extern "C" int dm_deconv2x2_fwd(const float* x, int NB, int C, int H, int W, const float* w_packed, const float* bias,
                                int Cout, int relu, float* out, dm_stream_t stream) {
  if (!x || !w_packed || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0) return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  ConvArgs a;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) { a.src[s] = nullptr; a.src_c[s] = 0; a.src_bs[s] = 0; }
  a.src[0] = x; a.src_c[0] = C; a.src_bs[0] = (long long)C * H * W; a.num_srcs = 1; a.Cin = C;
  a.NB = NB; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wp = w_packed; a.bias = bias; a.Cout = 4 * Cout; a.CoutP = dm_conv_packed_cout(4 * Cout);
  a.relu = relu; a.out = out; a.out_ch_total = 0; a.out_ch_offset = 0; a.shuffle = Cout;
  return launch_conv<1, 2, 2, 2, 2, 32>(a, (hipStream_t)stream);
}
