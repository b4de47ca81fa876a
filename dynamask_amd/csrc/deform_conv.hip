// K8: deformable convolution v1 (3x3, stride 1, pad 1, dilation 1, groups 1, no
// bias), forward.  Arithmetic spec: mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu
// :84-115 (bilinear) and :190-243 (im2col) of the reference tree; the reference
// materialises the [C*9, N*H*W] column matrix in HBM and calls a library GEMM
// (deform_conv_cuda.cpp:198-237).  Here the deformable im2col is the PRODUCER
// of the MFMA B operand: a workgroup owns TN flat pixels x (up to 256) output
// channels, gathers the bilinear-sampled columns of 8 input channels and feeds
// them to v_mfma_f32_32x32x2_f32 through LDS.  The column matrix never exists
// in HBM, and since one workgroup covers every output channel (C <= 256) each
// sample is gathered exactly once.
//
// Same software pipeline and quad K-ordering as conv_igemm.hip: the gather of
// chunk c+1 (4 loads per sample, issued into registers) overlaps the 144 MFMAs
// per wave of chunk c; LDS images  A: [tap][quad][TM][4]  B: [quad][tap][TN][4].
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
  const float* wp;  // [9][KQ][CoutP][4]
  int Cout, CoutP, KQ, dg, relu;
  int q_begin = 0;   // first flat pixel of this launch (Q is its end)
  float* out;
  int MT;
  // split-K (round 4, the <= 100-RoI inference calls): blockIdx.y = split, which walks channels [split * kchan, + kchan)
  // and stores bare sums to ws + split * ws_stride ([NB][Cout][HW]); dcn_splitk_reduce_kernel adds the splits in order
  float* ws = nullptr;
  long long ws_stride = 0, ws_floats = 0;
  int ksplit = 1, kchan = 0;
  // the 1x1 convolution behind the DCN, chained in the epilogue of the band kernel (round 6, dm_deform_conv_tout_fwd):
  // out2[m] = relu(b2[m] + sum_k w2[m][k] * relu(dcn[k])), w2t = [Cout][M2P] (transposed, rows of 32-padded couts);
  // store_out = 0: the DCN output itself is not written
  const float* w2t = nullptr;
  const float* b2 = nullptr;
  float* out2 = nullptr;
  int M2 = 0, out2_ct = 0, store_out = 1;
};

// One bilinear sample from its two row pairs.  Spelled as an explicit fma chain so that every kernel
// variant rounds the same way (left to the compiler, the two gathers contracted differently and
// results differed in the last bit depending on which variant a launch size selected).
__device__ __forceinline__ float dcn_bilinear(float wt0, float wt1, float wb0, float wb1, float ta, float tb, float ba, float bb) {
  return __builtin_fmaf(wb1, bb, __builtin_fmaf(wb0, ba, __builtin_fmaf(wt1, tb, wt0 * ta)));
}

template <int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(WGM* WGN * 64) void deform_conv_kernel(DcnArgs a) {
  constexpr int CK = 8;
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TG = NT / TN;                 // thread groups sharing a pixel column
  constexpr int MAXT = (9 + TG - 1) / TG;     // taps owned per thread
  constexpr int A_F4 = 9 * 2 * TM;
  constexpr int A_PER_T = (A_F4 + NT - 1) / NT;
  static_assert(NT % TN == 0, "threads must tile the pixel columns");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  // split-K: this workgroup's channel range and destination (see DcnArgs)
  const int kc_begin = (a.ksplit > 1) ? (int)blockIdx.y * a.kchan : 0;
  const int kc_end = (a.ksplit > 1) ? min(a.C, kc_begin + a.kchan) : a.C;
  float* const e_out = (a.ksplit > 1) ? a.ws + (size_t)blockIdx.y * a.ws_stride : a.out;
  const int e_relu = (a.ksplit > 1) ? 0 : a.relu;
  dm_f32x4* ldsA = reinterpret_cast<dm_f32x4*>(lds);           // [9][2][TM]
  dm_f32x4* ldsB = reinterpret_cast<dm_f32x4*>(lds) + A_F4;    // [2][9][TN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WGN, wave_n = wave % WGN;
  const int hi = lane >> 5, l31 = lane & 31;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own
  // L2), so the MT cout tiles of one pixel tile -- which read the same input -- are placed
  // 8 apart in launch order: same XCD, back to back.  The ragged tail keeps the plain order.
  int m_tile, n_tile;
  {
    const int b = blockIdx.x, grp = 8 * a.MT;
    const int full = (gridDim.x / grp) * grp;
    if (b < full) {
      const int g = b / grp, r = b - g * grp;
      m_tile = r / 8;
      n_tile = g * 8 + (r & 7);
    } else {
      const int r = b - full;
      m_tile = r % a.MT;
      n_tile = full / a.MT + r / a.MT;
    }
  }
  const int m0 = m_tile * TM;
  const int q0 = a.q_begin + n_tile * TN;
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

  // Per owned (tap, pixel): the 2x2 bilinear footprint as two 8-byte row pairs.
  // pair base column cb = clamp(w_low, 0, W-2); a tap that falls outside the
  // image keeps weight 0 (spec: deform_conv_cuda_kernel.cu:84-115), so the pair
  // loads stay in bounds and the products equal the reference's w1..w4 * v1..v4.
  int ot[MAXT], ob[MAXT];
  float wt0[MAXT], wt1[MAXT], wb0[MAXT], wb1[MAXT];
  int cur_group = -1;
  const int cpg = a.C / a.dg;

  auto load_params = [&](int group) {
    const float* offp = a.offset + ((size_t)gn * a.dg + group) * 18 * HW + gp;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int tap = tg + t * TG;
      ot[t] = ob[t] = 0;
      wt0[t] = wt1[t] = wb0[t] = wb1[t] = 0.f;
      if (tap < 9 && g_ok) {
        const int ki = tap / 3, kj = tap - ki * 3;
        const float off_h = offp[(size_t)(2 * tap) * HW];
        const float off_w = offp[(size_t)(2 * tap + 1) * HW];
        const float h_im = (float)(gy - 1 + ki) + off_h;
        const float w_im = (float)(gx - 1 + kj) + off_w;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float wr_t = (h_low >= 0) ? hh : 0.f;
          const float wr_b = (h_low + 1 <= H - 1) ? lh : 0.f;
          const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_low + 1, 0), H - 1);
          const int cb = min(max(w_low, 0), W - 2);
          const float wc0 = (cb == w_low ? hw : 0.f) + (cb == w_low + 1 ? lw : 0.f);
          const float wc1 = (cb + 1 == w_low ? hw : 0.f) + (cb + 1 == w_low + 1 ? lw : 0.f);
          ot[t] = rt * W + cb;
          ob[t] = rbm * W + cb;
          wt0[t] = wr_t * wc0; wt1[t] = wr_t * wc1;
          wb0[t] = wr_b * wc0; wb1[t] = wr_b * wc1;
        }
      }
    }
  };

  dm_f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct __attribute__((packed, aligned(4))) F2 { float a, b; };
  dm_f32x4 ra[A_PER_T];       // next chunk's weights
  dm_f32x4 raw[MAXT * 4];     // raw row pairs of ONE channel quad of the next chunk (in flight)
  dm_f32x4 rb0[MAXT];         // combined quad 0 of the next chunk

  auto prefetch_a = [&](int c0) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < A_F4) {
        const int m = idx % TM;
        const int tq = idx / TM;
        const int qd = tq & 1;
        const int tap = tq >> 1;
        if (m0 + m < a.CoutP)
          v = *reinterpret_cast<const dm_f32x4*>(a.wp + (((size_t)tap * a.KQ + (c0 >> 2) + qd) * a.CoutP + m0 + m) * 4);
      }
      ra[i] = v;
    }
  };
  auto issue_quad = [&](int c0, int qd) {   // 2 x 8-byte loads per sample, nothing waits on them here
    const float* xp = a.x + ((size_t)gn * a.C + c0 + qd * 4) * HW;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* xc = xp + (size_t)e * HW;
        const F2 top = *reinterpret_cast<const F2*>(xc + ot[t]);
        const F2 bot = *reinterpret_cast<const F2*>(xc + ob[t]);
        dm_f32x4 v = {top.a, top.b, bot.a, bot.b};
        raw[t * 4 + e] = v;
      }
    }
  };
  auto combine = [&](int t) {
    dm_f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const dm_f32x4 l = raw[t * 4 + e];
      v[e] = dcn_bilinear(wt0[t], wt1[t], wb0[t], wb1[t], l[0], l[1], l[2], l[3]);
    }
    return v;
  };
  auto commit = [&]() {   // A registers + both B quads -> LDS (quad 1 is combined here)
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      if (idx < A_F4) ldsA[idx] = ra[i];
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int tap = tg + t * TG;
      if (tap < 9) {
        ldsB[(0 * 9 + tap) * TN + gj] = rb0[t];
        ldsB[(1 * 9 + tap) * TN + gj] = combine(t);
      }
    }
  };
  // operand fragments double-buffered in registers (see conv_igemm.hip)
  auto load_frag = [&](int tap, dm_f32x4* av, dm_f32x4* bv) {
#pragma unroll
    for (int i = 0; i < WM; ++i) av[i] = ldsA[(tap * 2 + hi) * TM + (wave_m * WM + i) * 32 + l31];
#pragma unroll
    for (int j = 0; j < WN; ++j) bv[j] = ldsB[(hi * 9 + tap) * TN + (wave_n * WN + j) * 32 + l31];
  };
  auto mfma_taps = [&](int t0, int t1) {
    dm_f32x4 av[2][WM], bv[2][WN];
    load_frag(t0, av[0], bv[0]);
#pragma unroll
    for (int tap = t0; tap < t1; ++tap) {
      const int cur = (tap - t0) & 1;
      if (tap + 1 < t1) load_frag(tap + 1, av[cur ^ 1], bv[cur ^ 1]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][e], bv[cur][j][e], acc[i][j], 0, 0, 0);
    }
  };

  // prologue: chunk 0 entirely in registers
  cur_group = kc_begin / cpg;
  load_params(cur_group);
  prefetch_a(kc_begin);
  issue_quad(kc_begin, 0);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) rb0[t] = combine(t);
  issue_quad(kc_begin, 1);

  for (int c0 = kc_begin; c0 < kc_end; c0 += CK) {
    commit();
    __syncthreads();
    const int cn = c0 + CK;
    const bool more = cn < kc_end;
    if (more) {
      const int group = cn / cpg;    // CK divides cpg (checked on the host)
      if (group != cur_group) {
        cur_group = group;
        load_params(group);
      }
      prefetch_a(cn);
      issue_quad(cn, 0);             // in flight under taps 0..3
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_taps(0, 4);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) rb0[t] = combine(t);
      issue_quad(cn, 1);             // in flight under taps 4..8, combined at commit()
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_taps(4, 9);
    __syncthreads();
  }

  // epilogue with hoisted addresses (one base per pixel column, one per-lane channel offset; see conv_igemm.hip)
  {
    float* pj[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) pj[j] = e_out + (size_t)col_n[j] * a.Cout * HW + col_p[j];
    const int co_lane = m0 + wave_m * WM * 32 + 4 * hi;
    const size_t off_lane = (size_t)co_lane * HW;
    const bool relu = e_relu != 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = i * 32 + (r & 3) + 8 * (r >> 2);
        if (co_lane + k < a.Cout) {
          const size_t o = off_lane + (size_t)k * HW;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (col_ok[j]) {
              float v = acc[i][j][r];
              if (relu) v = fmaxf(v, 0.f);
              pj[j][o] = v;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Variant for small maps (128 <= H*W <= 256, i.e. the 14x14 stage): 128 couts x 128 pixels per
// workgroup like the plain conv kernel.  What kept the kernel above at 128 x 64 is the register
// cost of gathering from global memory (the raw row pairs of the next chunk live in VGPRs under
// the MFMAs).  Here the 8 channels of the (at most two) images a pixel tile touches are
// staged in LDS per chunk and the bilinear gather reads LDS:
// short latency, so a tap's B operand is produced just in time, three taps ahead of its MFMAs.
// The staged layout is [image][channel quad][pixel] float4 (the conv kernel's B layout): a sample's corner is ONE
// 16-byte read for the four channels of the thread's quad -- four reads per tap.  (Round 2 kept whole planes and read
// every corner pair of every channel with a ds_read2_b32: 8 reads and ~20 register moves per tap; tools/micro/mfma_mix.hip
// prices 8 such reads next to 16 MFMAs at a third of the MFMAs' own time.)
//
// The K loop is a sequence of steps of 3 taps (3 steps per chunk of 8 channels).  Step s does the
// MFMAs of its 3 taps from ring slot s & 1 of the A (weights) and B (gathered pixels) images while
// it fills slot (s + 1) & 1 for the next step: B by the gather, A from registers loaded one step
// earlier.  The channel quads are double buffered by chunk; those of chunk c + 1 are loaded
// in step 0 of chunk c, stored in step 1, and first read by the gather of step 2.  One barrier per
// step, no phase without MFMAs.
__global__ __launch_bounds__(256, 2) void deform_conv_lds_kernel(DcnArgs a) {
  constexpr int TM = 128, TN = 128, NT = 256;
  constexpr int AS_F4 = 3 * 2 * TM;               // one A slot: [tap of the step][quad][cout] float4
  constexpr int BS_F4 = 3 * 2 * TN;               // one B slot: [tap of the step][quad][pixel] float4
  constexpr int A_PER_T = AS_F4 / NT;             // 3
  constexpr int X_PER_T = 4;                      // float4 per thread for 2 images x 8 planes (H*W <= 256)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // split-K: this workgroup's channel range and destination (see DcnArgs)
  const int kc_begin = (a.ksplit > 1) ? (int)blockIdx.y * a.kchan : 0;
  const int kc_end = (a.ksplit > 1) ? min(a.C, kc_begin + a.kchan) : a.C;
  float* const e_out = (a.ksplit > 1) ? a.ws + (size_t)blockIdx.y * a.ws_stride : a.out;
  const int e_relu = (a.ksplit > 1) ? 0 : a.relu;
  dm_f32x4* ldsA = reinterpret_cast<dm_f32x4*>(lds);                 // [2][AS_F4]
  dm_f32x4* ldsB = ldsA + 2 * AS_F4;                                 // [2][BS_F4]
  dm_f32x4* ldsX = ldsB + 2 * BS_F4;                                 // [2 buffers][2 image slots][2 quads][H*W] float4

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  int m_tile, n_tile;
  {
    const int b = blockIdx.x, grp = 8 * a.MT;
    const int full = (gridDim.x / grp) * grp;
    if (b < full) {
      const int g = b / grp, r = b - g * grp;
      m_tile = r / 8;
      n_tile = g * 8 + (r & 7);
    } else {
      const int r = b - full;
      m_tile = r % a.MT;
      n_tile = full / a.MT + r / a.MT;
    }
  }
  const int m0 = m_tile * TM, q0 = a.q_begin + n_tile * TN;
  const int HW = a.HW, W = a.W, H = a.H;
  const int n0 = q0 / HW;                          // first image of the tile; the tile touches n0 and n0 + 1 at most
  const int xbuf = 2 * 2 * HW;                     // float4 per staging buffer

  int col_n[2], col_p[2];
  bool col_ok[2];
#pragma unroll
  for (int wn = 0; wn < 2; ++wn) {
    int q = q0 + (wave_n * 2 + wn) * 32 + l31;
    col_ok[wn] = q < a.Q;
    q = min(q, a.Q - 1);
    col_n[wn] = q / HW;
    col_p[wn] = q - col_n[wn] * HW;
  }

  // gather role: pixel gj of the tile, channel quad gh of the chunk
  const int gj = tid & 127, gh = tid >> 7;
  int gq = q0 + gj;
  const bool g_ok = gq < a.Q;
  gq = min(gq, a.Q - 1);
  const int gn = gq / HW, gp = gq - gn * HW;
  const int gy = gp / W, gx = gp - gy * W;
  const int xbase = ((gn - n0) * 2 + gh) * HW;               // this thread's quad inside a staging buffer

  int otb[9];                                      // top | bottom << 16: offsets inside a plane (< 256)
  float wt0[9], wt1[9], wb0[9], wb1[9];
  const int cpg = a.C / a.dg;
  auto load_params = [&](int group) {
    const float* offp = a.offset + ((size_t)gn * a.dg + group) * 18 * HW + gp;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      otb[tap] = 0;
      wt0[tap] = wt1[tap] = wb0[tap] = wb1[tap] = 0.f;
      if (g_ok) {
        const int ki = tap / 3, kj = tap - ki * 3;
        const float h_im = (float)(gy - 1 + ki) + offp[(size_t)(2 * tap) * HW];
        const float w_im = (float)(gx - 1 + kj) + offp[(size_t)(2 * tap + 1) * HW];
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float wr_t = (h_low >= 0) ? hh : 0.f;
          const float wr_b = (h_low + 1 <= H - 1) ? lh : 0.f;
          const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_low + 1, 0), H - 1);
          const int cb = min(max(w_low, 0), W - 2);
          const float wc0 = (cb == w_low ? hw : 0.f) + (cb == w_low + 1 ? lw : 0.f);
          const float wc1 = (cb + 1 == w_low ? hw : 0.f) + (cb + 1 == w_low + 1 ? lw : 0.f);
          otb[tap] = (rt * W + cb) | ((rbm * W + cb) << 16);
          wt0[tap] = wr_t * wc0; wt1[tap] = wr_t * wc1;
          wb0[tap] = wr_b * wc0; wb1[tap] = wr_b * wc1;
        }
      }
    }
  };

  dm_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  dm_f32x4 ra[A_PER_T], rx[X_PER_T];
  // staging slot i of this thread: (image slot, quad, pixel) = its place in a staging buffer; x_off = float offset of
  // the quad's first channel at that pixel, relative to (image n0, channel c0).  Slots past the staging buffer or past
  // the last image load an address that exists (slot 0 / the last image) and are not stored / never gathered from.
  int x_off[X_PER_T];
#pragma unroll
  for (int i = 0; i < X_PER_T; ++i) {
    const int idx = tid + i * NT;
    const int img = idx / (2 * HW), r = idx - img * 2 * HW;
    const int quad = r / HW, px = r - quad * HW;
    const int ii = (idx < 4 * HW) ? img : 0, qq = (idx < 4 * HW) ? quad : 0, pp = (idx < 4 * HW) ? px : 0;
    x_off[i] = ((min(n0 + ii, a.NB - 1) - n0) * a.C + qq * 4) * HW + pp;
  }
  // weights of taps 3g .. 3g+2, channels c0 .. c0+7
  auto load_a = [&](int c0, int g) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      const int m = idx % TM, tq = idx / TM;       // tq = (tap of the step) * 2 + quad
      // (rows past CoutP: the last row again -- their products land in accumulator rows the epilogue never stores;
      // an unconditional load keeps exec-mask branches out of the loop)
      const int mm = min(m0 + m, a.CoutP - 1);
      ra[i] = *reinterpret_cast<const dm_f32x4*>(a.wp + (((size_t)(3 * g + (tq >> 1)) * a.KQ + (c0 >> 2) + (tq & 1)) * a.CoutP + mm) * 4);
    }
  };
  auto store_a = [&](int slot) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) ldsA[slot * AS_F4 + tid + i * NT] = ra[i];
  };
  auto load_x = [&](int c0) {                     // lanes = consecutive pixels: coalesced dword loads, as the conv kernel's B staging
    const float* xc = a.x + ((size_t)n0 * a.C + c0) * HW;
#pragma unroll
    for (int i = 0; i < X_PER_T; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rx[i][e] = xc[x_off[i] + e * HW];
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < X_PER_T; ++i) {
      const int idx = tid + i * NT;
      if (idx < 4 * HW) ldsX[buf * xbuf + idx] = rx[i];
    }
  };
  auto gather3 = [&](int t0, int slot, int buf) {   // taps t0 .. t0+2 of the chunk in staging buffer buf -> B slot
    const dm_f32x4* xq = ldsX + buf * xbuf + xbase;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int tap = t0 + u;
      const dm_f32x4* pt = xq + (otb[tap] & 0xffff);
      const dm_f32x4* pb = xq + (otb[tap] >> 16);
      const dm_f32x4 tl = pt[0], tr = pt[1], bl = pb[0], br = pb[1];
      dm_f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dcn_bilinear(wt0[tap], wt1[tap], wb0[tap], wb1[tap], tl[e], tr[e], bl[e], br[e]);
      ldsB[slot * BS_F4 + (u * 2 + gh) * TN + gj] = v;
    }
  };
  // One step: the MFMAs of 3 taps from ring slot ``slot``; if ``fill``, the gather of taps t0 .. t0+2 (staging buffer
  // ``gbuf``) into the other slot, one tap inside each MFMA cluster: the four corner reads are issued with the
  // cluster's operand reads, the fma chain and the 16-byte write sit in the middle of the cluster -- their LDS latency
  // passes under MFMAs of the same wave instead of in front of them (round 2 gathered all three taps, then ran the 48 MFMAs).
  auto step = [&](int slot, bool fill, int t0, int gbuf) {
    const dm_f32x4* xq = ldsX + gbuf * xbuf + xbase;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      dm_f32x4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = ldsA[slot * AS_F4 + (u * 2 + hi) * TM + (wave_m * 2 + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = ldsB[slot * BS_F4 + (u * 2 + hi) * TN + (wave_n * 2 + j) * 32 + l31];
      const int tap = t0 + u;
      dm_f32x4 tl, tr, bl, br;
      if (fill) {
        const dm_f32x4* pt = xq + (otb[tap] & 0xffff);
        const dm_f32x4* pb = xq + (otb[tap] >> 16);
        tl = pt[0]; tr = pt[1]; bl = pb[0]; br = pb[1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (fill) {
        dm_f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dcn_bilinear(wt0[tap], wt1[tap], wb0[tap], wb1[tap], tl[e], tr[e], bl[e], br[e]);
        ldsB[(slot ^ 1) * BS_F4 + (u * 2 + gh) * TN + gj] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 2; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: planes of chunk 0, weights and gathered pixels of step 0, loads of step 1 in flight
  int cur_group = kc_begin / cpg;
  load_params(cur_group);
  load_x(kc_begin);
  load_a(kc_begin, 0);
  store_x(0);
  store_a(0);
  load_a(kc_begin, 1);
  __syncthreads();
  gather3(0, 0, 0);
  __syncthreads();
  int slot = 0;                                   // ring slot of the current step
  for (int c0 = kc_begin, buf = 0; c0 < kc_end; c0 += 8, buf ^= 1) {
    const int cn = c0 + 8;
    const bool more = cn < kc_end;
    // step 0: MFMAs of taps 0..2; fills taps 3..5
    if (more) load_x(cn);
    store_a(slot ^ 1);
    load_a(c0, 2);
    step(slot, true, 3, buf);
    __syncthreads();
    slot ^= 1;
    // step 1: MFMAs of taps 3..5; fills taps 6..8; the quads of the next chunk go to the other buffer
    store_a(slot ^ 1);
    if (more) {
      load_a(cn, 0);
      store_x(buf ^ 1);
    }
    step(slot, true, 6, buf);
    __syncthreads();
    slot ^= 1;
    // step 2: MFMAs of taps 6..8; fills taps 0..2 of the next chunk
    if (more) {
      const int group = cn / cpg;
      if (group != cur_group) {
        cur_group = group;
        load_params(group);
      }
      store_a(slot ^ 1);
      load_a(cn, 1);
    }
    step(slot, more, 0, buf ^ 1);
    __syncthreads();
    slot ^= 1;
  }

  {
    float* pj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) pj[j] = e_out + (size_t)col_n[j] * a.Cout * HW + col_p[j];
    const int co_lane = m0 + wave_m * 64 + 4 * hi;
    const size_t off_lane = (size_t)co_lane * HW;
    const bool relu = e_relu != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = i * 32 + (r & 3) + 8 * (r >> 2);
        if (co_lane + k < a.Cout) {
          const size_t o = off_lane + (size_t)k * HW;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (col_ok[j]) {
              float v = acc[i][j][r];
              if (relu) v = fmaxf(v, 0.f);
              pj[j][o] = v;
            }
        }
      }
  }
}

// ---------------------------------------------------------------------------
// Variant for the large maps (28 x 28, 56 x 56): the bilinear gather runs IN the wave that consumes it.
// What holds the kernel at the top to a third of the MFMA rate at 64 channels is the path of a sample:
// two 8-byte global loads (texture path, L2 hits), a combine, a 16-byte LDS write, and an LDS read back by
// every cout wave -- with 64 couts a sample feeds 128 flop.  Here a workgroup owns 128 pixels of ONE image
// and all couts; a wave owns 32 of those pixels and ALL cout tiles, so the B operand of an MFMA
// (lane = pixel, lane half = channel quad) is exactly what the lane itself can gather: four 16-byte LDS reads per tap from a
// band of the chunk's 8 channels, staged as [quad][pixel] float4 (BR rows around the tile's rows; a 128-pixel tile spans <= 4 rows of
// a 56-wide map, offsets of +-6 rows stay inside), one fma chain, and the value goes straight into WM
// MFMAs.  No B image in LDS, no texture-path gathers; a sample outside the band (rare: large offsets) is
// loaded from global memory by the lanes concerned.  Per chunk: A (weights) and the band are loaded into
// registers under the MFMAs of the previous chunk and stored to LDS between two barriers.
// Same products, same K order (chunk of 8 -> tap -> element -> quad) and the same fma chain as the kernels
// above: results do not depend on which variant a shape takes.
constexpr int DCN_NEAR_ROWS = 5;

template <int WM, int WGM, int XR>
__global__ __launch_bounds__(256, 2) void deform_conv_band_kernel(DcnArgs a, int tiles_per_img, int BR) {
  // WGM waves share a pixel column (each with WM of the cout tiles: 128 couts = 2 x 2 -- four tiles in one wave spill)
  constexpr int TM = WM * WGM * 32, TN = (4 / WGM) * 32, NT = 256;
  constexpr int A_F4 = 9 * 2 * TM;                       // float4 of one chunk's weights: [tap][quad][cout]
  constexpr int A_PER_T = (A_F4 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // split-K: this workgroup's channel range and destination (see DcnArgs)
  const int kc_begin = (a.ksplit > 1) ? (int)blockIdx.y * a.kchan : 0;
  const int kc_end = (a.ksplit > 1) ? min(a.C, kc_begin + a.kchan) : a.C;
  float* const e_out = (a.ksplit > 1) ? a.ws + (size_t)blockIdx.y * a.ws_stride : a.out;
  const int e_relu = (a.ksplit > 1) ? 0 : a.relu;
  dm_f32x4* ldsA = reinterpret_cast<dm_f32x4*>(lds);
  dm_f32x4* ldsX = reinterpret_cast<dm_f32x4*>(lds) + A_F4;      // [2 quads][BR * W] float4: the 4 channels of a quad at a band pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wave_m = wave % WGM, wave_n = wave / WGM;
  const int HW = a.HW, W = a.W, H = a.H;
  const int PS = BR * W, PS4 = PS >> 2;                  // W % 4 == 0 (host)
  const int n = blockIdx.x / tiles_per_img, tile = blockIdx.x - n * tiles_per_img;
  const int p0 = tile * TN;
  const int p_end = min(p0 + TN, HW);
  const int y_first = p0 / W, y_last = (p_end - 1) / W;
  const int band_y0 = min(max(y_first - (BR - (y_last - y_first + 1)) / 2, 0), H - BR);     // H >= BR (host)
  int gp = p0 + wave_n * 32 + l31;
  const bool g_ok = gp < HW;
  gp = min(gp, HW - 1);
  const int gy = gp / W, gx = gp - gy * W;

  int otb[9];                          // top | bottom << 16.  In band: offsets inside a band plane; else: inside the image plane (H * W < 65536, host)
  float wt0[9], wt1[9], wb0[9], wb1[9];
  unsigned oob = 0, any_oob = 0;       // taps of this lane / of any lane of the wave that leave the band
  const int cpg = a.C / a.dg;
  auto load_params = [&](int group) {
    const float* offp = a.offset + ((size_t)n * a.dg + group) * 18 * HW + gp;
    oob = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      otb[tap] = 0;
      wt0[tap] = wt1[tap] = wb0[tap] = wb1[tap] = 0.f;
      if (g_ok) {
        const int ki = tap / 3, kj = tap - ki * 3;
        const float h_im = (float)(gy - 1 + ki) + offp[(size_t)(2 * tap) * HW];
        const float w_im = (float)(gx - 1 + kj) + offp[(size_t)(2 * tap + 1) * HW];
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float wr_t = (h_low >= 0) ? hh : 0.f;
          const float wr_b = (h_low + 1 <= H - 1) ? lh : 0.f;
          const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_low + 1, 0), H - 1);
          const int cb = min(max(w_low, 0), W - 2);
          const float wc0 = (cb == w_low ? hw : 0.f) + (cb == w_low + 1 ? lw : 0.f);
          const float wc1 = (cb + 1 == w_low ? hw : 0.f) + (cb + 1 == w_low + 1 ? lw : 0.f);
          if (rt >= gy - DCN_NEAR_ROWS && rbm <= gy + DCN_NEAR_ROWS) {     // inside the band of every tiling (host checks the pad)
            otb[tap] = ((rt - band_y0) * W + cb) | (((rbm - band_y0) * W + cb) << 16);
            wt0[tap] = wr_t * wc0; wt1[tap] = wr_t * wc1;
            wb0[tap] = wr_b * wc0; wb1[tap] = wr_b * wc1;
          } else {
            oob |= 1u << tap;          // weight 0 here; the sample is added by the slow pass of every chunk
          }
        }
      }
    }
    any_oob = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      if (__ballot((oob >> tap) & 1u) != 0ull) any_oob |= 1u << tap;
  };

  dm_f32x16 acc[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  dm_f32x4 ra[A_PER_T], rx[XR];        // XR * 256 float4 cover the 8 band planes (host)
  // weights: slot offsets fixed per thread (rows past CoutP and slots past the chunk repeat a row that exists: their products
  // land in accumulator rows that are never stored / their LDS slots are never written) -- no exec-mask branch per load
  unsigned a_off[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int idx = min(tid + i * NT, A_F4 - 1);
    const int m = idx % TM, tq = idx / TM;               // tq = tap * 2 + quad
    a_off[i] = (unsigned)((((size_t)(tq >> 1) * a.KQ + (tq & 1)) * a.CoutP + min(m, a.CoutP - 1)) * 4);
  }
  auto load_a = [&](int c0) {
    const float* wb = a.wp + (size_t)(c0 >> 2) * a.CoutP * 4;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) ra[i] = *reinterpret_cast<const dm_f32x4*>(wb + a_off[i]);
  };
  // staging slot i of this thread = (quad, band pixel) = idx / PS, idx % PS (two quads: one compare); the loads of a slot
  // are the quad's four channels at that pixel.  Slots past the band load slot 0 and are not stored.
  auto load_x = [&](int c0) {          // lanes = consecutive band pixels: coalesced dword loads, four channels per slot
    const float* xb = a.x + ((size_t)n * a.C + c0) * HW + (size_t)band_y0 * W;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int idx = tid + i * NT;
      const int off = (idx < PS) ? idx : (idx < 2 * PS ? 4 * HW + idx - PS : 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) rx[i][e] = xb[off + e * HW];
    }
  };
  auto store_ax = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      if (idx < A_F4) ldsA[idx] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int idx = tid + i * NT;
      if (idx < 2 * PS) ldsX[idx] = rx[i];
    }
  };
  struct __attribute__((packed, aligned(4))) F2 { float a, b; };      // (the slow pass's global row pairs)

  int cur_group = kc_begin / cpg;
  load_params(cur_group);
  load_a(kc_begin);
  load_x(kc_begin);
  for (int c0 = kc_begin; c0 < kc_end; c0 += 8) {
    __syncthreads();                   // the previous chunk's operand reads are done
    store_ax();
    __syncthreads();
    const int cn = c0 + 8;
    if (cn < kc_end) {
      load_a(cn);
      load_x(cn);
    }
    const unsigned any_oob_s = __builtin_amdgcn_readfirstlane(any_oob);
    const dm_f32x4* pq = ldsX + hi * PS;                                    // this lane half's channel quad
    const float* gl = a.x + ((size_t)n * a.C + c0 + 4 * hi) * HW;          // the same channels in global memory
    // Nine taps: the four corners of a tap are four 16-byte reads for the quad's four channels (round 2 kept channel
    // planes: two ds_read2_b32 per sample, 72 reads per chunk and wave against 36).  Software-pipelined by one tap: the
    // four values of a tap are formed first, then the next tap's corners and weight fragments are issued and run under
    // this tap's 4 * WM MFMAs.
    // (the weight fragments are double-buffered where the registers allow: with four cout tiles per wave a second set
    // spills, and they are read behind the tap's MFMAs instead -- their latency passes under the next tap's fma chains)
    constexpr int AB = (WM < 4) ? 2 : 1;
    dm_f32x4 av[AB][WM];
    dm_f32x4 tl, tr, bl, br;
    auto issue_x = [&](int tap) {
      const dm_f32x4* pt = pq + (otb[tap] & 0xffff);
      const dm_f32x4* pb = pq + (otb[tap] >> 16);
      tl = pt[0]; tr = pt[1]; bl = pb[0]; br = pb[1];
    };
    auto issue_a = [&](int tap) {
#pragma unroll
      for (int i = 0; i < WM; ++i) av[tap % AB][i] = ldsA[(tap * 2 + hi) * TM + (wave_m * WM + i) * 32 + l31];
    };
    issue_x(0);
    issue_a(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dcn_bilinear(wt0[tap], wt1[tap], wb0[tap], wb1[tap], tl[e], tr[e], bl[e], br[e]);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 9) {
        issue_x(tap + 1);
        if (AB == 2) issue_a(tap + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tap % AB][i][e], v[e], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (AB == 1 && tap + 1 < 9) issue_a(tap + 1);
    }
    if (any_oob_s) {
      // Slow pass: taps that leave the band for some lane of this wave.  The lanes concerned recompute the tap
      // from the offsets and gather from global memory; the others contribute 0.  (Kept out of the loop above:
      // with the two sources in one place the compiler selects between the POINTERS and every sample becomes a
      // flat load, and a branch per tap keeps it from overlapping one tap's LDS reads with another's MFMAs.)
      const float* offp = a.offset + ((size_t)n * a.dg + cur_group) * 18 * HW + gp;
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        if (!((any_oob_s >> tap) & 1u)) continue;
        const bool mine = (oob >> tap) & 1u;
        float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
        int g_t = 0, g_b = 0;
        if (mine) {
          const int ki = tap / 3, kj = tap - ki * 3;
          const float h_im = (float)(gy - 1 + ki) + offp[(size_t)(2 * tap) * HW];
          const float w_im = (float)(gx - 1 + kj) + offp[(size_t)(2 * tap + 1) * HW];
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float wr_t = (h_low >= 0) ? hh : 0.f;
          const float wr_b = (h_low + 1 <= H - 1) ? lh : 0.f;
          const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_low + 1, 0), H - 1);
          const int cb = min(max(w_low, 0), W - 2);
          const float wc0 = (cb == w_low ? hw : 0.f) + (cb == w_low + 1 ? lw : 0.f);
          const float wc1 = (cb + 1 == w_low ? hw : 0.f) + (cb + 1 == w_low + 1 ? lw : 0.f);
          g_t = rt * W + cb;
          g_b = rbm * W + cb;
          w00 = wr_t * wc0; w01 = wr_t * wc1;
          w10 = wr_b * wc0; w11 = wr_b * wc1;
        }
        dm_f32x4 av[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) av[i] = ldsA[(tap * 2 + hi) * TM + (wave_m * WM + i) * 32 + l31];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* gc = gl + (size_t)e * HW;
          const F2 top = *reinterpret_cast<const F2*>(gc + g_t);
          const F2 bot = *reinterpret_cast<const F2*>(gc + g_b);
          const float g = dcn_bilinear(w00, w01, w10, w11, top.a, top.b, bot.a, bot.b);
#pragma unroll
          for (int i = 0; i < WM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], g, acc[i], 0, 0, 0);
        }
      }
    }
    if (cn < kc_end) {
      const int group = cn / cpg;      // 8 divides cpg (host)
      if (group != cur_group) {
        cur_group = group;
        load_params(group);
      }
    }
  }
  // The 1x1 convolution + ReLU that follows the DCN in an SFM stage (fuse_transform_out, dynamask_head.py:117-121), chained
  // here while the DCN's 32 pixels x all couts are still in this wave's accumulators: the MFMA D layout IS a B operand --
  // register r of tile i holds dcn[k = i*32 + (r&3) + 8*(r>>2) + 4*hi][pixel = lane & 31], i.e. for the k pair (kb, kb + 4)
  // lanes 0-31 / 32-63 carry exactly B[k][n] of a 32x32x2 step -- so the second GEMM needs no LDS round trip for its
  // input, only its weights (staged transposed: lane = cout).  Walking i, r upward adds the products in the order
  // (0,4),(1,5),(2,6),(3,7),(8,12),... -- the order the 1x1 kernel of conv_igemm.hip uses (quad pairs of a 16-channel
  // chunk): the result has the bits of the two launches it replaces.  The DCN output is then never written (inference).
  if (WGM == 1 && a.w2t) {
    constexpr int MT2 = (WM + 1) / 2;          // cout tiles of the 1x1 (host: ceil(M2 / 32) <= MT2)
    constexpr int M2P = MT2 * 32;
    __syncthreads();                           // every wave is done with the last chunk's operands in LDS
    {
      const dm_f32x4* src = reinterpret_cast<const dm_f32x4*>(a.w2t);
      dm_f32x4* dst = reinterpret_cast<dm_f32x4*>(lds);
      for (int idx = tid; idx < a.Cout * (M2P / 4); idx += NT) dst[idx] = src[idx];
    }
    __syncthreads();
    const float* w2t = lds;
    dm_f32x16 acc2[MT2];
#pragma unroll
    for (int t = 0; t < MT2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kb = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float b = fmaxf(acc[i][r], 0.f);
#pragma unroll
        for (int t = 0; t < MT2; ++t)
          acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2t[kb * M2P + t * 32 + l31], b, acc2[t], 0, 0, 0);
      }
    const int p = p0 + wave_n * 32 + l31;
    if (p < HW) {
      float* po2 = a.out2 + (size_t)n * a.out2_ct * HW + p;
#pragma unroll
      for (int t = 0; t < MT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (co < a.M2) po2[(size_t)co * HW] = fmaxf(acc2[t][r] + a.b2[co], 0.f);
        }
    }
    if (!a.store_out) return;
  }
  // epilogue: lane = pixel column, registers walk the couts
  {
    const int p = p0 + wave_n * 32 + l31;
    if (p < HW) {
      float* po = e_out + (size_t)n * a.Cout * HW + p;
      const bool relu = e_relu != 0;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (wave_m * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (co < a.Cout) {
            float v = acc[i][r];
            if (relu) v = fmaxf(v, 0.f);
            po[(size_t)co * HW] = v;
          }
        }
    }
  }
}

// out = relu?(sum over the splits, in index order) of a split-K launch (ws: [splits][NB * Cout * HW])
__global__ __launch_bounds__(256) void dcn_splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long stride,
                                                               long long total, int relu, float* __restrict__ out) {
  // (total, stride: multiples of 4 floats, ws / out 16-byte aligned -- checked by dcn_finish_split, which otherwise passes v4 = 0
  // through the sign of ``splits``)
  const bool v4 = splits > 0;
  splits = v4 ? splits : -splits;
  if (v4) {
    const float4* ws4 = reinterpret_cast<const float4*>(ws);
    float4* out4 = reinterpret_cast<float4*>(out);
    const long long total4 = total >> 2, stride4 = stride >> 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long long)gridDim.x * blockDim.x) {
      float4 v = ws4[e];
      for (int s_ = 1; s_ < splits; ++s_) {
        const float4 w = ws4[(size_t)s_ * stride4 + e];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      out4[e] = v;
    }
    return;
  }
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float v = ws[e];
    for (int s_ = 1; s_ < splits; ++s_) v += ws[(size_t)s_ * stride + e];
    out[e] = relu ? fmaxf(v, 0.f) : v;
  }
}

// Split-K for a launch of ``wgs`` workgroups on ``slots`` slots (caller-owned workspace of ws_floats floats): a workgroup
// walks its chunks of 8 channels one after the other at ~3.5 us each whatever the RoI count (tools/small_n.py: dcn14 105-115 us
// for 32 chunks at 8-16 RoIs); S splits save (1 - 1/S) of that and cost a second launch that reads S and writes one copy of
// the output.  Sets a.ksplit / kchan / ws_stride; a.ksplit stays 1 when it does not pay.
static void dcn_choose_split(DcnArgs& a, long long wgs, long long slots, long long ws_floats, bool lds_tiles = false) {
  a.ksplit = 1;
  if (!a.ws || a.q_begin != 0 || wgs <= 0) return;
  const int chunks = a.C / 8;
  const long long per = (long long)a.NB * a.Cout * a.HW;
  int S = 1;
  if (a.NB <= 24) {
    // a handful of RoIs: the cost model (13 % / 3 % on the full head at 8 / 16 RoIs)
    long long Smax = min(min(8LL, slots / wgs), (long long)chunks / 2);
    if (per > 0) Smax = min(Smax, ws_floats / per);
    const double chain_us = chunks * 3.5, out_mb = (double)per * 4e-6;
    double best = 8.0;
    for (int c = 2; c <= Smax; ++c) {
      const double g = chain_us * (1.0 - 1.0 / c) - (5.0 + (c + 1) * out_mb * 0.25);
      if (g > best) { best = g; S = c; }
    }
  } else if (lds_tiles && wgs < slots && chunks >= 12 && per > 0 && ws_floats / per >= 3) {
    // 25 .. 128 RoIs of 14 x 14 on the 128 x 128 LDS kernel (round 5, tools/small_n.py with forced splits): the launch is
    // less than one round of workgroups, each walking 32 chunks alone on its CU -- three splits: 32 / 50 / 64 / 100 RoIs
    // 0.179 / 0.198 / 0.200 / 0.332 -> 0.157 / 0.138 / 0.197 / 0.248 ms (two: 0.177 / 0.181 / 0.183 / 0.279).  The band
    // kernels of the 28 x 28 / 56 x 56 stages gain <= 8 % or lose (56 x 56: 0.283 -> 0.346 at 100 RoIs): not split.
    S = 3;
  }
  if (S >= 2) {
    const int per_split = dm_ceil_div(chunks, S);
    a.kchan = per_split * 8;
    a.ksplit = dm_ceil_div(chunks, per_split);
    a.ws_stride = per;
  }
}

static int dcn_finish_split(const DcnArgs& a, int relu, float* out, hipStream_t st) {
  if (a.ksplit <= 1) return dm_check_launch();
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  const long long total = a.ws_stride;
  const bool v4 = (total & 3) == 0 && ((((uintptr_t)a.ws) | ((uintptr_t)out)) & 15) == 0;
  DM_LAUNCH(dcn_splitk_reduce_kernel, dim3((unsigned)min((long long)4096, ((v4 ? total / 4 : total) + 255) / 256)), dim3(256), 0, st, a.ws,
            v4 ? a.ksplit : -a.ksplit, a.ws_stride, total, relu, out);
  return dm_check_launch();
}

template <int WM, int WGM, int XR>
int launch_dcn_band(DcnArgs& a, hipStream_t st) {
  const int BR = 16;
  if (8 * BR * a.W / 4 > XR * 256 || a.H < BR) return DM_ERR_UNSUPPORTED;      // the staging registers must cover the band planes
  const int tiles = dm_ceil_div(a.HW, (4 / WGM) * 32);
  const size_t lds_bytes = 16 * (size_t)(9 * 2 * WM * WGM * 32) + 4 * (size_t)8 * BR * a.W;
  if (a.w2t) {
    // the chained 1x1: this wave layout only (a wave holds every cout of its pixels), its weights must fit the kernel's LDS
    if (WGM != 1 || dm_ceil_div(a.M2, 32) > (WM + 1) / 2 || (size_t)a.Cout * ((WM + 1) / 2) * 32 * 4 > lds_bytes || a.Cout != WM * 32)
      return DM_ERR_UNSUPPORTED;
    a.ws = nullptr;                            // (no split-K: the second GEMM needs the complete channel sums)
  }
  dcn_choose_split(a, (long long)a.NB * tiles, 2LL * dm_num_cus(), a.ws_floats);
  DM_LAUNCH((deform_conv_band_kernel<WM, WGM, XR>), dim3((unsigned)(a.NB * tiles), (unsigned)a.ksplit), dim3(256), lds_bytes, st, a, tiles, BR);
  return dcn_finish_split(a, a.relu, a.out, st);
}

template <int WGM, int WGN, int WM, int WN>
int launch_dcn(DcnArgs& a, hipStream_t st) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  a.MT = dm_ceil_div(a.CoutP, TM);
  const int NTiles = dm_ceil_div(a.Q - a.q_begin, TN);
  const size_t lds_bytes = 16 * ((size_t)9 * 2 * TM + (size_t)2 * 9 * TN);
  static bool attr_set[DM_MAX_DEVICES] = {false};
  if (lds_bytes > 64 * 1024 &&
      dm_ensure_lds_limit(reinterpret_cast<const void*>(&deform_conv_kernel<WGM, WGN, WM, WN>), (int)lds_bytes, attr_set) != DM_OK)
    return DM_ERR_LAUNCH;
  dcn_choose_split(a, (long long)a.MT * NTiles, (NT == 256 ? 2LL : 1LL) * dm_num_cus(), a.ws_floats);
  DM_LAUNCH((deform_conv_kernel<WGM, WGN, WM, WN>), dim3(a.MT * NTiles, a.ksplit), dim3(NT), lds_bytes, st, a);
  return dcn_finish_split(a, a.relu, a.out, st);
}

}  // namespace

struct DcnTout {      // the chained 1x1 (nullptr members: plain DCN)
  const float* w2t = nullptr;
  const float* b2 = nullptr;
  float* out2 = nullptr;
  int M2 = 0, out2_ct = 0;
};
static int deform_conv_fwd_impl(const float* x, const float* offset, int NB, int C, int H, int W,
                                const float* w_packed, int Cout, int deform_groups, int relu, float* out, float* ws,
                                long long ws_floats, dm_stream_t stream, const DcnTout* tout = nullptr);

extern "C" int dm_deform_conv_fwd(const float* x, const float* offset, int NB, int C, int H, int W,
                                  const float* w_packed, int Cout, int deform_groups, int relu, float* out,
                                  dm_stream_t stream) {
  return deform_conv_fwd_impl(x, offset, NB, C, H, W, w_packed, Cout, deform_groups, relu, out, nullptr, 0, stream);
}

// (ABI 21) dm_deform_conv_fwd with a caller-owned workspace: a launch that leaves most of the chip idle (the <= 100-RoI
// inference calls) splits its channel loop over up to eight workgroups per tile; a second kernel adds the splits in index
// order (+ ReLU).  Same bits every run; they differ from dm_deform_conv_fwd's by the association of the channel sums.
extern "C" long long dm_deform_conv_splitk_floats(int NB, int C, int H, int W, int Cout) {
  if (NB <= 0 || C < 32 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const long long wgs = (long long)dm_ceil_div(Cout, 128) * dm_ceil_div((long long)NB * H * W, 128);
  if (wgs * 2 > 2LL * dm_num_cus() && !(H * W <= 256 && wgs < 2LL * dm_num_cus())) return 0;      // (14 x 14: up to one round)
  return 8LL * NB * Cout * H * W;
}

extern "C" int dm_deform_conv_fwd_ws(const float* x, const float* offset, int NB, int C, int H, int W,
                                     const float* w_packed, int Cout, int deform_groups, int relu, float* out,
                                     float* workspace, long long workspace_floats, dm_stream_t stream) {
  return deform_conv_fwd_impl(x, offset, NB, C, H, W, w_packed, Cout, deform_groups, relu, out, workspace, workspace_floats, stream);
}

// Does this shape take the band kernel in the wave layout that can chain the 1x1 (a wave = 32 pixels x all couts)?
static bool dcn_tout_shape_ok(int NB, int C, int H, int W, int Cout, int M2) {
  if (NB <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout != C || M2 <= 0) return false;
  const int HW = H * W, CoutP = dm_conv_packed_cout(Cout);
  const int max_rows = (W - 1 + 128 + W - 1) / W;
  if (!(H >= 16 && (W & 3) == 0 && 8 * 16 * W / 4 <= 7 * 256 && (16 - max_rows) / 2 >= DCN_NEAR_ROWS && HW > 256 && HW < 65536 &&
        (CoutP == 64 || CoutP == 128) && Cout > 32 && Cout == CoutP))
    return false;
  const bool narrow = 8 * 16 * W / 4 <= 4 * 256;
  const bool few = (long long)NB * dm_ceil_div(HW, 128) * 4 < (long long)dm_num_cus() * 3;
  if (few) return false;                                   // (one cout tile per wave there: 2-4 x the workgroups, worth more)
  if (CoutP == 128 && !narrow) return false;               // <2, 2, 7>: two waves share a pixel column
  return dm_ceil_div(M2, 32) <= (CoutP / 32 + 1) / 2;
}

extern "C" int dm_deform_conv_tout_supported(int NB, int C, int H, int W, int Cout, int M2) {
  return dcn_tout_shape_ok(NB, C, H, W, Cout, M2) ? 1 : 0;
}

// (ABI 27) DCN 3x3 + ReLU + the 1x1 convolution + bias + ReLU behind it (SFMStage.fuse_conv[1] -> fuse_transform_out,
// mmdet/models/roi_heads/mask_heads/dynamask_head.py:117-121) in ONE launch: out2[:, :M2] of a tensor with out2_ch_total
// channels; w2t = the 1x1 weight transposed to [Cout][32-padded M2] (zeros in the padding); out_dcn: NULL (inference: the
// DCN output is never written) or [NB, Cout, H, W] to keep relu(DCN) as well.  DM_ERR_UNSUPPORTED when
// dm_deform_conv_tout_supported() says 0 -- the caller then launches the two operators.  Same bits as those two launches.
extern "C" int dm_deform_conv_tout_fwd(const float* x, const float* offset, int NB, int C, int H, int W, const float* w_packed,
                                       int Cout, int deform_groups, const float* w2t, const float* b2, int M2, float* out2,
                                       int out2_ch_total, float* out_dcn, dm_stream_t stream) {
  if (!w2t || !b2 || !out2 || M2 <= 0 || M2 > out2_ch_total) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  if (!dcn_tout_shape_ok(NB, C, H, W, Cout, M2)) return DM_ERR_UNSUPPORTED;
  DcnTout t;
  t.w2t = w2t; t.b2 = b2; t.out2 = out2; t.M2 = M2; t.out2_ct = out2_ch_total;
  // (out == NULL is the "do not store" request; the implementation needs a non-null pointer to get past its checks)
  return deform_conv_fwd_impl(x, offset, NB, C, H, W, w_packed, Cout, deform_groups, 1, out_dcn ? out_dcn : out2, nullptr, 0, stream,
                              &t) ;
}

static int deform_conv_fwd_impl(const float* x, const float* offset, int NB, int C, int H, int W,
                                const float* w_packed, int Cout, int deform_groups, int relu, float* out, float* ws,
                                long long ws_floats, dm_stream_t stream, const DcnTout* tout) {
  if (!x || !offset || !w_packed || !out) return DM_ERR_INVALID_ARG;
  if (NB < 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || deform_groups <= 0 || C % deform_groups != 0)
    return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if ((C / deform_groups) % 8 != 0) return DM_ERR_UNSUPPORTED;  // channel chunk must not straddle a deformable group
  if (W < 2) return DM_ERR_UNSUPPORTED;                         // the gather loads row pairs
  if (NB == 0) return DM_OK;
  DcnArgs a;
  a.x = x; a.offset = offset; a.NB = NB; a.C = C; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wp = w_packed; a.Cout = Cout; a.CoutP = dm_conv_packed_cout(Cout); a.KQ = (C + 7) / 8 * 2; a.dg = deform_groups;
  a.relu = relu & 1; a.out = out;
  a.ws = (ws && ws_floats > 0) ? ws : nullptr;
  a.ws_floats = a.ws ? ws_floats : 0;
  if (tout) {
    a.w2t = tout->w2t; a.b2 = tout->b2; a.out2 = tout->out2; a.M2 = tout->M2; a.out2_ct = tout->out2_ct;
    a.store_out = (out != tout->out2) ? 1 : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  // 8-wave workgroups: 4 threads share a pixel column, so a thread owns <= 3 taps
  // (a 256-cout tile would gather each sample once but needs > 256 VGPRs: it spills)
  // 4-wave workgroups of 128 x 64: two of them share a CU (the 8-wave 128 x 128 tile needs
  // > 128 VGPRs and runs alone: 1.81 ms vs 1.59 ms at 256 channels)
  // Launches that leave most of the chip idle (a handful of RoIs: real inference) are bound by the time
  // of one workgroup; 64 x 64 tiles give four times the workgroups of the 128 x 128 LDS kernel
  // (14 x 14, 8 RoIs: 0.208 -> 0.108 ms, 32 RoIs: 0.213 -> 0.168 ms; from 64 RoIs on the big tiles win).
  // Same products in the same order in every variant: results do not depend on the choice.
  {
    // large maps: gather in the consuming wave from an LDS band of 16 rows (the 8 band planes of a chunk must fit the
    // 7 float4 per thread of the widest staging build: W <= 56).
    // Every launch size of an eligible shape takes this kernel (rows must not depend on the batch).
    const int max_rows = (W - 1 + 128 + W - 1) / W;                         // rows a 128-pixel tile can span
    if (H >= 16 && (W & 3) == 0 && 8 * 16 * W / 4 <= 7 * 256 && (16 - max_rows) / 2 >= DCN_NEAR_ROWS && a.HW > 256 &&
        a.HW < 65536 && (a.CoutP == 64 || a.CoutP == 128) && Cout > 32) {
      const bool narrow = 8 * 16 * W / 4 <= 4 * 256;                        // band planes fit 4 float4 per thread
      // a handful of RoIs (real inference): the launch is bound by the time of one workgroup -- one cout tile per
      // wave, the waves sharing a pixel column (2-4 x the workgroups, 1/2-1/4 of the K-loop time each; same bits)
      bool few = (long long)NB * dm_ceil_div(a.HW, 128) * 4 < (long long)dm_num_cus() * 3;
      // ... unless the caller brought a workspace: a workgroup's time is its number of chunks times a load round trip
      // (~3.5 us) whatever it computes per chunk, so the full layout (every cout of 128 pixels: one gather per sample, four
      // times the MFMAs per round trip) with the K loop split (dcn_choose_split's cost model, <= 24 RoIs) beats it:
      // 16 / 24 detections 0.5845 / 0.733 -> 0.578 / 0.712 ms (profiles/r06_infer_notes.txt (6))
      if (few && a.ws && NB <= 24 && !tout) few = false;
      if (a.CoutP == 64) {
        if (few) return narrow ? launch_dcn_band<1, 2, 4>(a, st) : launch_dcn_band<1, 2, 7>(a, st);
        return narrow ? launch_dcn_band<2, 1, 4>(a, st) : launch_dcn_band<2, 1, 7>(a, st);
      }
      if (few) return narrow ? launch_dcn_band<1, 4, 4>(a, st) : launch_dcn_band<1, 4, 7>(a, st);
      return narrow ? launch_dcn_band<4, 1, 4>(a, st) : launch_dcn_band<2, 2, 7>(a, st);
    }
  }
  if (tout) return DM_ERR_UNSUPPORTED;                      // (dcn_tout_shape_ok mirrors the conditions above)
  // A handful of RoIs: 64 x 64 tiles (four times the workgroups of the 128 x 128 LDS kernel) -- unless the caller brought a
  // workspace and the map takes the LDS kernel: then that kernel with its K loop split (dcn_choose_split: up to 8 ways at
  // <= 24 RoIs, 3 above) is the faster way to fill the chip (round 6: the 16- / 32-detection inference calls 0.652 / 0.949 ->
  // 0.626 / 0.918 ms; the 14 x 14 DCN was their largest launch, 90 us at 16 RoIs).
  const bool lds_split = a.ws && Cout > 64 && a.HW >= 128 && a.HW <= 256 && (a.HW & 3) == 0;
  if (!lds_split && Cout > 64 &&
      (long long)dm_ceil_div(a.CoutP, 128) * dm_ceil_div(a.Q, 128) * 20 <= (long long)dm_num_cus() * 9)
    return launch_dcn<2, 2, 1, 1>(a, st);
  if (Cout > 64 && a.HW >= 128 && a.HW <= 256 && (a.HW & 3) == 0 && (C / deform_groups) % 8 == 0) {
    // small maps (14x14): planes in LDS, 128 x 128 tile
    a.MT = dm_ceil_div(a.CoutP, 128);
    const int NTiles = dm_ceil_div(a.Q, 128);
    // A ring + B ring (2 x 3 taps x 2 quads x 128 float4 each) + 2 plane buffers of 2 images x 8 channels
    const size_t lds_bytes = 16 * ((size_t)2 * 2 * 3 * 2 * 128) + (size_t)4 * 2 * 2 * 8 * a.HW;
    static bool attr_lds[DM_MAX_DEVICES] = {false};      // once per device, for the largest map this path takes (H*W = 256)
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&deform_conv_lds_kernel), 16 * 2 * 2 * 3 * 2 * 128 + 4 * 2 * 2 * 8 * 256,
                            attr_lds) != DM_OK)
      return DM_ERR_LAUNCH;
    // rounds (see conv_igemm.hip): the LDS kernel runs two workgroups per CU; the pixels of a nearly empty
    // last round go to a second launch with 64 x 64 tiles (same bits): 512 RoIs 1.24 -> 1.19 ms.  Only
    // for small remainders (the 64 x 64 build is slower per pixel: at 0.5 of a round the split loses), and
    // not when the caller overlaps launches on a second stream (flag bit 3: measured 259 vs 251 img/s).
    static const int tail_mode = getenv("DM_DCN_TAIL") ? atoi(getenv("DM_DCN_TAIL")) : 1;
    const int slots = 2 * dm_num_cus();
    const int full_rounds = (a.MT * NTiles) / slots;
    const int rem = a.MT * NTiles - full_rounds * slots;
    if (tail_mode && !(relu & 8) && full_rounds >= 1 && rem > 0 && rem * 20 <= 3 * slots) {
      const int n_main = full_rounds * slots / a.MT;
      const int Q = a.Q;
      a.Q = n_main * 128;
      DM_LAUNCH(deform_conv_lds_kernel, dim3(a.MT * n_main), dim3(256), lds_bytes, st, a);
      int rc = dm_check_launch();
      if (rc != DM_OK) return rc;
      a.q_begin = a.Q;
      a.Q = Q;
      return launch_dcn<2, 2, 1, 1>(a, st);
    }
    dcn_choose_split(a, (long long)a.MT * NTiles, slots, a.ws_floats, true);
    DM_LAUNCH(deform_conv_lds_kernel, dim3(a.MT * NTiles, a.ksplit), dim3(256), lds_bytes, st, a);
    return dcn_finish_split(a, a.relu, a.out, st);
  }
  if (Cout > 64 && (long long)dm_ceil_div(a.CoutP, 128) * dm_ceil_div(a.Q, 64) * 20 <= (long long)dm_num_cus() * 9)
    return launch_dcn<2, 2, 1, 1>(a, st);                 // same rule for the 128 x 64 tiles (28 x 28, 8 RoIs: 0.081 -> 0.056 ms)
  if (Cout > 64) return launch_dcn<2, 2, 2, 1>(a, st);    // 128 couts x 64 px
  if (Cout > 32) return launch_dcn<2, 2, 1, 1>(a, st);    // 64 x 64 (2.57 -> 2.28 ms at 64 channels, 56x56)
  return launch_dcn<1, 4, 1, 1>(a, st);                   // 32 x 128 (4 waves)
}
