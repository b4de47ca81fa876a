// K22: fully connected layer  out[n][m] = sum_k x[n][k] * w[m][k] + bias[m]  (nn.Linear
// layout on both sides: K is the contiguous axis of x AND of w), optional ReLU.
// Used by the bbox head (Shared2FCBBoxHead, roi_heads/bbox_heads/convfc_bbox_head.py:138-186:
// 12544 -> 1024 -> 1024 -> {81, 320}).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).  x rows are the MFMA A operand
// and w rows the B operand, so a D tile has the output feature m on the lane: rows of
// out[n][.] are written as 128-byte runs.  Both operands are staged with 16-byte loads along
// K into the quad images of the conv kernels ([k/4][row][4]: one ds_read_b128 per operand
// feeds 4 MFMA k-steps).  Workgroup = 128 x 128 outputs, 4 waves x (2 x 2) tiles, K chunks of
// 32 with register prefetch.  The layers here have few output tiles (N = 1000, M = 1024:
// 64) and a long K, so K is split over workgroups.  The split is a function of K ALONE (segments of
// 256 or 1024, see fc_seg): every split writes its partial tile to a scratch slab and a second pass adds the slabs in
// a fixed order, then bias and ReLU.  A row of the output therefore has the same bits whatever the number
// of rows in the launch and from run to run (round 1 added partials with float atomics and chose the
// split from N: the selector logits of an RoI depended on how many RoIs shared the launch).
#include "common.h"

namespace {

struct FcArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* out;
  float* part;      // [splits][N][M] partial sums (splits > 1)
  int N, K, M;
  int relu, splits, chunks_per_split, MT, NT;
};

__global__ __launch_bounds__(256) void fc_gemm_kernel(FcArgs a) {
  constexpr int T = 128, KT = 32, NQ = KT / 4;
  __shared__ dm_f32x4 ldsA[NQ * T];
  __shared__ dm_f32x4 ldsB[NQ * T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_r = wave >> 1, wave_c = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  const int m_tile = bid % a.MT;
  bid /= a.MT;
  const int n_tile = bid % a.NT;
  const int split = bid / a.NT;
  const int n0 = n_tile * T, m0 = m_tile * T;

  dm_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: item = (row, quad); 4 items of A and of B per thread and chunk
  const int c_begin = split * a.chunks_per_split;
  const int c_end = min(c_begin + a.chunks_per_split, (a.K + KT - 1) / KT);
  dm_f32x4 ra[4], rb[4];
  auto fetch = [&](int ch) {
    const int k0 = ch * KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, qd = idx & 7;
      const int k = k0 + 4 * qd;
      dm_f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (k < a.K) {          // K % 4 == 0 (checked on the host): a quad is all in or all out
        if (n0 + row < a.N) va = *reinterpret_cast<const dm_f32x4*>(a.x + (size_t)(n0 + row) * a.K + k);
        if (m0 + row < a.M) vb = *reinterpret_cast<const dm_f32x4*>(a.w + (size_t)(m0 + row) * a.K + k);
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (int ch = c_begin; ch < c_end; ++ch) {
    __syncthreads();      // the previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, qd = idx & 7;
      ldsA[qd * T + row] = ra[i];
      ldsB[qd * T + row] = rb[i];
    }
    __syncthreads();
    if (ch + 1 < c_end) fetch(ch + 1);
#pragma unroll
    for (int t = 0; t < NQ / 2; ++t) {
      dm_f32x4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = ldsA[(2 * t + hi) * T + (wave_r * 2 + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = ldsB[(2 * t + hi) * T + (wave_c * 2 + j) * 32 + l31];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
    }
  }
  // D[row = n][col = m]: lane & 31 = m.  Addresses and the two bias values of the lane are hoisted.
  float* pj[2];
  float bj[2];
  bool m_ok[2];
  const bool single = a.splits == 1, relu = a.relu != 0;
  float* dst = single ? a.out : a.part + (size_t)split * a.N * a.M;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (wave_c * 2 + j) * 32 + l31;
    m_ok[j] = m < a.M;
    pj[j] = dst + m;
    bj[j] = (single && a.bias && m_ok[j]) ? a.bias[m] : 0.f;
  }
  const int n_lane = n0 + wave_r * 64 + 4 * hi;
  const size_t off_lane = (size_t)n_lane * a.M;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = i * 32 + (r & 3) + 8 * (r >> 2);
      if (n_lane + k < a.N) {
        const size_t o = off_lane + (size_t)k * a.M;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (m_ok[j]) {
            float v = acc[i][j][r] + bj[j];
            if (single && relu) v = fmaxf(v, 0.f);
            pj[j][o] = v;
          }
        }
      }
    }
}

// out = relu?(bias + part[0] + part[1] + ... ) in that order
__global__ __launch_bounds__(256) void fc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, int splits,
                                                        size_t NM, int M, int relu, float* __restrict__ out) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < NM; i += (size_t)gridDim.x * blockDim.x * 4) {
    // M % 4 == 0 is not guaranteed (81 classes): fall back to scalars at the row tails
    if (i + 3 < NM && (M & 3) == 0) {
      dm_f32x4 v = *reinterpret_cast<const dm_f32x4*>(part + i);
      for (int s = 1; s < splits; ++s) {
        const dm_f32x4 p = *reinterpret_cast<const dm_f32x4*>(part + (size_t)s * NM + i);
        v += p;
      }
      if (bias) {
        const int m = (int)(i % M);
        v[0] += bias[m]; v[1] += bias[m + 1]; v[2] += bias[m + 2]; v[3] += bias[m + 3];
      }
      if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *reinterpret_cast<dm_f32x4*>(out + i) = v;
    } else {
      for (size_t e = i; e < NM && e < i + 4; ++e) {
        float v = part[e];
        for (int s = 1; s < splits; ++s) v += part[(size_t)s * NM + e];
        if (bias) v += bias[e % M];
        out[e] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

}  // namespace

// K elements per split: a function of K alone, so that the order of the partial sums does not depend on the
// launch.  Short K (the 1024 -> 1024 / 320 / 81 layers of the bbox head: 8 .. 64 output tiles) is cut finer:
// with one workgroup per tile those layers were a chain of 32 dependent staging round trips (0.105 ms);
// long K (12544) coarser, or the partial slabs would cost more traffic than the operands.
static inline int fc_seg(int K) {
  return K <= 4096 ? 256 : 1024;
}

extern "C" long long dm_fc_scratch_floats(int N, int K, int M) {
  const int splits = (K + fc_seg(K) - 1) / fc_seg(K);
  return splits > 1 ? (long long)splits * N * M : 0;
}

extern "C" int dm_fc_fwd(const float* x, const float* w, const float* bias, int N, int K, int M, int relu, float* out,
                         float* scratch, dm_stream_t stream) {
  if (N < 0 || K <= 0 || M <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!x || !w || !out) return DM_ERR_INVALID_ARG;
  if (K % 4 != 0) return DM_ERR_UNSUPPORTED;      // 16-byte staging loads along K
  hipStream_t st = (hipStream_t)stream;
  FcArgs a;
  a.x = x; a.w = w; a.bias = bias; a.out = out; a.part = scratch; a.N = N; a.K = K; a.M = M; a.relu = relu;
  a.MT = dm_ceil_div(M, 128);
  a.NT = dm_ceil_div(N, 128);
  const int tiles = a.MT * a.NT;
  a.chunks_per_split = fc_seg(K) / 32;
  a.splits = dm_ceil_div(K, fc_seg(K));
  if (a.splits > 1 && !scratch) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(fc_gemm_kernel, dim3((unsigned)(tiles * a.splits)), dim3(256), 0, st, a);
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  if (a.splits > 1) {
    const size_t NM = (size_t)N * M;
    DM_LAUNCH(fc_reduce_kernel, dim3((unsigned)min((size_t)2048, (NM / 4 + 255) / 256 + 1)), dim3(256), 0, st,
              (const float*)scratch, bias, a.splits, NM, M, relu, out);
    rc = dm_check_launch();
  }
  return rc;
}
