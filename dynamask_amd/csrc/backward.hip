// Backward kernels of the mask-head path (training step, BASELINE configs[2..3]).
//
//   conv weight gradient  : MFMA GEMM  dW[co][j] += sum_q dY[co,q] * Xcol[j,q], j=(ci,tap),
//                           split over the pixel dimension, fp32 atomics into dW
//   conv data gradient    : the forward implicit GEMM with transposed/rotated
//                           weights (dm_conv_pack_weight(transpose_flip=1)) -- no kernel here
//   bias gradient         : per-channel reduction
//   ReLU / sigmoid masks  : elementwise on the gradient
//   upsample, point sample, class logits: adjoint scatter / gathered dot products
//   DCNv1                 : deformable im2col (column matrix materialised for the
//                           backward only) + col2im / coordinate gradient, spec
//                           mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:117-188,279-436;
//                           the two GEMMs of deform_conv_cuda.cpp:262-486 run on the
//                           implicit-GEMM conv kernels (1x1 over the column matrix).
#include "common.h"
#include <climits>

#define DM_FIX_SCALE 68719476736.0 /* 2^36: 64-bit fixed-point LDS accumulators */

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ----------------------------------------------------------------- elementwise
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ g, const float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) g[i] = 0.f;
}

// 16 bytes per lane (both pointers 16-byte aligned, n % 4 == 0): the mask pass is pure streaming
__global__ __launch_bounds__(256) void relu_bwd4_kernel(dm_f32x4* __restrict__ g, const dm_f32x4* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const dm_f32x4 m = y[i];
    dm_f32x4 v = g[i];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (!(m[e] > 0.f)) v[e] = 0.f;
    g[i] = v;
  }
}

// g_logit[n,p] (+)= (ga[n,p] + gb[n,p]) * s * (1 - s),  s = sig[n, ch, p] taken from a
// channel of a wider tensor; ga / gb are channel slices too (gb optional).
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ sig, long long sig_bs,
                                                          const float* __restrict__ ga, long long ga_bs,
                                                          const float* __restrict__ gb, long long gb_bs, int N, int HW,
                                                          float* __restrict__ g_logit, int accumulate) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW;
    const size_t p = i - n * HW;
    const float s = sig[n * sig_bs + p];
    float g = ga[n * ga_bs + p];
    if (gb) g += gb[n * gb_bs + p];
    const float v = g * s * (1.f - s);
    g_logit[i] = accumulate ? g_logit[i] + v : v;
  }
}

// per-channel sum over batch and pixels (bias gradient): grid = (C, splits), one
// atomic per workgroup into out (zero-filled by the launcher unless accumulating)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g, long long bs, int NB, int C, int HW,
                                                          float* __restrict__ out, int fx) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  const long long total = (long long)NB * HW;
  if ((HW & 3) == 0 && (bs & 3) == 0 && ((uintptr_t)g & 15) == 0) {
    const int HWq = HW >> 2;
    const long long total4 = (long long)NB * HWq;
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.y * blockDim.x) {
      const long long n = i / HWq;
      const long long p = i - n * HWq;
      const dm_f32x4 v = *reinterpret_cast<const dm_f32x4*>(g + n * bs + (long long)c * HW + p * 4);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.y * blockDim.x) {
      const long long n = i / HW;
      const long long p = i - n * HW;
      s += g[n * bs + (long long)c * HW + p];
    }
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dm_acc_add(out, c, red[0] + red[1] + red[2] + red[3], fx != 0);
}

// ----------------------------------------------------------------- conv weight gradient
struct WgradArgs {
  const float* dy;
  long long dy_bs;
  const float* x;
  long long x_bs;
  int Cout, Cs, NB, H, W, HW, Q;
  float* dw;
  int ldw, coloff;
  int JT, MT, chunks_per_split, nsplit;
  int fx;            // dw (and db) are 64-bit fixed-point accumulators (dm_conv2d_wgrad_fx)
  float* db;         // optional bias gradient db[co] += sum_q dy[co][q]: the row sums of the A operand, taken from the
                     // values the column-tile-0 workgroups stage anyway (round 2 read dy a second time: dm_channel_sum)
  // slab mode (dm_conv2d_wgrad_slab): every (split, K-wave) writes its partial tile to its own slab instead of adding into
  // dw with atomics, wgrad_slab_reduce_kernel adds the slabs in index order -- the reference's weight gradient is a
  // deterministic addmm_ (deform_conv_cuda.cpp:460-465), and 2 M float atomics onto the 4 K addresses of a 64 x 64 tile cost
  // 40 us.  slab == nullptr: atomics.
  float* scratch = nullptr;    // host side: the caller's scratch (dm_conv2d_wgrad_slab) the launcher carves the slabs from
  long long scratch_floats = 0;
  float* slab = nullptr;       // [slabs][slab_rows][slab_ld]
  float* slab_db = nullptr;    // [splits][slab_rows]
  long long slab_stride = 0;   // floats per slab
  int slab_ld = 0, slab_rows = 0;
};

// dW[co][j] = sum_q dy[co][q] * xshift[j][q]  (j = (ci, tap), q = flat pixel): a GEMM
// whose K dimension is the pixel axis.  Four waves, each owning a 64 x 64 accumulator
// (2 x 2 MFMA tiles); WGM x WGN x WGK arranges them over the output tile and -- for the
// small-output layers (64 couts x 64 columns) -- over the K chunk, so narrow weight
// matrices do not pad to 128 x 128.  Split-K over workgroups; partial sums land with
// float atomics.  The next chunk's global loads are issued before the MFMA loop.
// TAIL: 32 < Cout <= 36 (the DCN offset convs): rows 0..31 on one 32x32x2 tile per column tile,
// rows 32..35 on v_mfma_f32_4x4x1 (see conv_igemm.hip) instead of a second, 87 % empty tile row.
template <int KS, int WGM, int WGN, int WGK, bool TAIL = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  static_assert(WGM * WGN * WGK == 4, "four waves");
  static_assert(!TAIL || WGM == 1, "tail rows need a single cout tile");
  constexpr int TM = 64 * WGM, TN = 64 * WGN, KT = 32, LD = KT + 1, TAPS = KS * KS;
  constexpr int RA = TM / 8, RB = TN / 8, KW = KT / WGK;
  __shared__ float ldsA[TM * LD];
  __shared__ float ldsB[TN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_k = wave % WGK, wave_n = (wave / WGK) % WGN, wave_m = wave / (WGK * WGN);
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  const int m_tile = bid % a.MT;
  bid /= a.MT;
  const int j_tile = bid % a.JT;
  const int split = bid / a.JT;
  const int m0 = m_tile * TM, j0 = j_tile * TN;
  const int HW = a.HW, W = a.W, H = a.H;
  const int Jtot = a.Cs * TAPS;

  dm_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  dm_f32x4 acct[2];               // TAIL: couts 32..35 x this lane's column (lanes >= 32: the odd k)
  acct[0] = acct[1] = dm_f32x4{0.f, 0.f, 0.f, 0.f};

  const int kq = tid & 31;        // this thread's pixel column inside a chunk
  const int r0 = tid >> 5;        // rows r0 + 8*i
  // decode the B rows this thread stages once: j -> (ci, tap), kept as plane offset + shift
  int b_off[RB];                  // ci*HW + dy*W + dx, or INT_MIN past the last column
  unsigned b_sh[RB];              // (dy+1) | (dx+1) << 2
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int jg = j0 + r0 + 8 * i;
    const int ci = jg / TAPS, tap = jg - ci * TAPS;
    const int dy = (KS == 3) ? tap / 3 - 1 : 0, dx = (KS == 3) ? tap % 3 - 1 : 0;
    b_off[i] = (jg < Jtot) ? ci * HW + dy * W + dx : INT_MIN;
    b_sh[i] = (unsigned)(dy + 1) | ((unsigned)(dx + 1) << 2);
  }
  const int c_begin = split * a.chunks_per_split;
  const int c_end = min(c_begin + a.chunks_per_split, (a.Q + KT - 1) / KT);
  const bool want_bias = a.db != nullptr && j_tile == 0;
  float brow = 0.f;               // bias gradient: thread t < TM sums row t of every staged A tile
  float va[RA], vb[RB];
  // 1x1, chunk inside the matrix: nothing is predicated -- rows past the matrix repeat its last row (their products land in
  // rows / columns of the tile that are never stored), addresses are a per-thread base plus per-row offsets kept in
  // registers.  (The guarded fetch below spends an exec-mask branch or a select per dword: ~250 branches in the K loop of the
  // 128 x 128 build, the same disease conv_igemm's 1x1 staging had.)
  int a_roff[RA], b_roff[(KS == 1) ? RB : 1];
#pragma unroll
  for (int i = 0; i < RA; ++i) a_roff[i] = min(m0 + r0 + 8 * i, a.Cout - 1) * HW;
  if (KS == 1) {
#pragma unroll
    for (int i = 0; i < RB; ++i) b_roff[i] = min(j0 + r0 + 8 * i, Jtot - 1) * HW;
  }
  auto fetch = [&](int ch) {
    const int q = ch * KT + kq;
    if (KS == 1 && !TAIL && ch * KT + KT <= a.Q) {
      const int n = q / HW;
      const int p = q - n * HW;
      const float* dyp = a.dy + (size_t)n * a.dy_bs + p;
      const float* xp = a.x + (size_t)n * a.x_bs + p;
#pragma unroll
      for (int i = 0; i < RA; ++i) va[i] = dyp[a_roff[i]];
#pragma unroll
      for (int i = 0; i < RB; ++i) vb[i] = xp[b_roff[i]];
      return;
    }
    const bool qok = q < a.Q;
    const int qq = min(q, a.Q - 1);
    const int n = qq / HW;
    const int p = qq - n * HW;
    const int y = p / W, x = p - y * W;
    const float* dyp = a.dy + (size_t)n * a.dy_bs + p;
    const float* xp = a.x + (size_t)n * a.x_bs + p;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int co = m0 + r0 + 8 * i;
      va[i] = (qok && co < a.Cout) ? dyp[(size_t)co * HW] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      bool ok = qok && b_off[i] != INT_MIN;
      if (KS == 3) {
        const int yy = y + (int)(b_sh[i] & 3u) - 1, xx = x + (int)(b_sh[i] >> 2) - 1;
        ok = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
      }
      vb[i] = ok ? xp[b_off[i]] : 0.f;
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (int ch = c_begin; ch < c_end; ++ch) {
    __syncthreads();   // previous chunk's MFMA reads are done
#pragma unroll
    for (int i = 0; i < RA; ++i) ldsA[(r0 + 8 * i) * LD + kq] = va[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) ldsB[(r0 + 8 * i) * LD + kq] = vb[i];
    __syncthreads();
    if (ch + 1 < c_end) fetch(ch + 1);
    if (want_bias && tid < TM) {
      // pixels past Q and rows past Cout were staged as zeros; LD is odd: the 64 rows of a wave hit 64 banks
      float sacc = 0.f;
#pragma unroll 8
      for (int kk = 0; kk < KT; ++kk) sacc += ldsA[tid * LD + kk];
      brow += sacc;
    }
#pragma unroll
    for (int k2 = 0; k2 < KW; k2 += 2) {
      const int kk = wave_k * KW + k2;
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        av[i] = (TAIL && i == 1) ? ldsA[(32 + (lane & 3)) * LD + kk + hi] : ldsA[((wave_m * 2 + i) * 32 + l31) * LD + kk + hi];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = ldsB[((wave_n * 2 + j) * 32 + l31) * LD + kk + hi];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[j], acc[0][j], 0, 0, 0);
        if (TAIL) acct[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[1], bv[j], acct[j], 0, 0, 0);
        else acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[j], acc[1][j], 0, 0, 0);
      }
    }
  }
  {
    // hoisted addresses: one base per column, one per-lane row offset (see conv_igemm.hip's epilogue)
    size_t cj[2];
    bool j_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int jg = j0 + (wave_n * 2 + j) * 32 + l31;
      j_ok[j] = jg < Jtot;
      cj[j] = (size_t)a.coloff + jg;
    }
    const int co_lane = m0 + wave_m * 64 + 4 * hi;
    const size_t off_lane = (size_t)co_lane * a.ldw;
    if (a.slab) {
      // the slab is padded to whole tiles: no bounds checks, 32 consecutive floats per store instruction
      float* sp = a.slab + (size_t)(split * WGK + wave_k) * a.slab_stride;
#pragma unroll
      for (int i = 0; i < (TAIL ? 1 : 2); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int j = 0; j < 2; ++j) sp[(size_t)(co_lane + k) * a.slab_ld + j0 + (wave_n * 2 + j) * 32 + l31] = acc[i][j][r];
        }
    } else if (!a.fx) {
#pragma unroll
      for (int i = 0; i < (TAIL ? 1 : 2); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = i * 32 + (r & 3) + 8 * (r >> 2);
          if (co_lane + k < a.Cout) {
            const size_t o = off_lane + (size_t)k * a.ldw;
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j_ok[j]) atomicAdd(a.dw + cj[j] + o, acc[i][j][r]);
          }
        }
    } else {
      unsigned long long* dwx = reinterpret_cast<unsigned long long*>(a.dw);
#pragma unroll
      for (int i = 0; i < (TAIL ? 1 : 2); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = i * 32 + (r & 3) + 8 * (r >> 2);
          if (co_lane + k < a.Cout) {
            const size_t o = off_lane + (size_t)k * a.ldw;
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j_ok[j]) atomicAdd(dwx + cj[j] + o, dm_to_fx(acc[i][j][r]));
          }
        }
    }
  }
  if (want_bias && tid < TM) {
    if (a.slab) a.slab_db[(size_t)split * a.slab_rows + m0 + tid] = brow;
    else if (m0 + tid < a.Cout) dm_acc_add(a.db, m0 + tid, brow, a.fx != 0);
  }
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int jg = j0 + (wave_n * 2 + j) * 32 + l31;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acct[j][i] + __shfl_xor(acct[j][i], 32, 64);      // the two k parities
        const int co = m0 + 32 + i;
        if (hi != 0) continue;
        if (a.slab) a.slab[(size_t)(split * WGK + wave_k) * a.slab_stride + (size_t)co * a.slab_ld + jg] = v;
        else if (co < a.Cout && jg < Jtot) dm_acc_add(a.dw, (size_t)co * a.ldw + a.coloff + jg, v, a.fx != 0);
      }
    }
  }
}

// dw[co][coloff + j] += sum over the slabs (and db[co] += sum over the splits' bias rows), in a fixed order: a workgroup
// owns 256 / P consecutive QUADS of outputs (four neighbouring columns of a row: one 16-byte load per slab; ld and the
// slab stride are multiples of 4 floats), partition p of P adds the slabs p, p + P, p + 2P ... in that order (independent
// loads, unrolled), and the partitions' sums are added in order p = 0 .. P-1.  (One thread per output walking up to 2048
// slabs one after the other took 0.4 ms for a 64 x 64 tile; one thread per output and dword loads -- rounds 4 / 5 --
// 49 us on average for the 33 MB of slabs a launch leaves, 0.7 TB/s out of the Infinity Cache.)
template <int P>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slab, int slabs, long long stride, int ld,
                                                                int Cout, int Jtot, float* __restrict__ dw, int ldw, int coloff,
                                                                const float* __restrict__ slab_db, int splits, int rows,
                                                                float* __restrict__ db) {
  constexpr int OUT = 256 / P;
  __shared__ float4 part[P][OUT];                        // 4 KB: also the bias workgroups' [8][32] floats
  const int o = threadIdx.x % OUT, pidx = threadIdx.x / OUT;
  const int Q4 = (Jtot + 3) >> 2;                       // quads per row
  const long long total = (long long)Cout * Q4;
  const long long nblk = (total + OUT - 1) / OUT;
  if ((long long)blockIdx.x < nblk) {
    const long long idx = (long long)blockIdx.x * OUT + o;
    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    int co = 0, j = 0;
    if (idx < total) {
      co = (int)(idx / Q4); j = 4 * (int)(idx - (long long)co * Q4);
      const float* p = slab + (size_t)co * ld + j;
#pragma unroll 8
      for (int k = pidx; k < slabs; k += P) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * stride);
        sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
      }
    }
    if (P > 1) {
      part[pidx][o] = sacc;
      __syncthreads();
      if (pidx == 0) {
#pragma unroll
        for (int q = 1; q < P; ++q) {
          const float4 v = part[q][o];
          sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
        }
      }
    }
    if (pidx == 0 && idx < total) {
      float* d = dw + (size_t)co * ldw + coloff + j;
      d[0] += sacc.x;
      if (j + 1 < Jtot) d[1] += sacc.y;
      if (j + 2 < Jtot) d[2] += sacc.z;
      if (j + 3 < Jtot) d[3] += sacc.w;
    }
  } else if (db) {
    // the workgroups behind the outputs: bias rows, 32 channels each, the splits dealt over 8 partitions (k = q, q + 8 ...)
    // whose sums are added in order q = 0 .. 7.  (Round 4 / early round 5: ONE workgroup, a thread per channel walking
    // the splits one dependent load after the other -- 105-120 us for 170 splits, the whole cost of the reduce launches
    // that carried a bias: 0.75 ms of the training step's 20.7 ms of kernel time.)
    float* pb = reinterpret_cast<float*>(part);                 // [8][32]
    const int c = ((int)(blockIdx.x - nblk)) * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
    float sacc = 0.f;
    if (c < Cout) {
#pragma unroll 8
      for (int k = q; k < splits; k += 8) sacc += slab_db[(size_t)k * rows + c];
    }
    pb[q * 32 + (threadIdx.x & 31)] = sacc;
    __syncthreads();
    if (q == 0 && c < Cout) {
#pragma unroll
      for (int r = 1; r < 8; ++r) sacc += pb[r * 32 + threadIdx.x];
      db[c] += sacc;
    }
  }
}

// Narrow 3x3 weight gradients (Cout <= 36: the DCN offset convs, MaskPre's 128 -> 16 conv).  The general kernel above
// stages the nine shifted copies of x as separate B rows -- nine bounds-checked dword loads per x element, and the
// address arithmetic of that fetch costs as much issue time as the MFMAs it feeds (13-49 TFLOP/s on these shapes).
// Here a workgroup owns 32 input channels and walks (image, row band) units: the band of x (R + 2 rows, one shared
// zero column between rows, zero rows outside the image) and the band of dy sit in LDS, and the B operand of tap
// (ky, kx) is read straight from the x band at a per-lane base (lane = input channel, odd plane stride: no bank
// conflicts) plus a wave-uniform offset -- no transposed staging, no bounds checks in the K loop.  N index = tap-major
// (nine 32-column tiles, one per tap), M = one 32-row tile (+ rows 32..35 on v_mfma_f32_4x4x1, TAIL), K = pixel pairs
// of the band, dealt round-robin to the four waves (every wave holds all nine accumulator tiles); the waves' sums are
// added through LDS and leave with one float atomic per element and workgroup.
struct WgradNarrowArgs {
  const float* dy;
  long long dy_bs;
  const float* x;
  long long x_bs;
  int Cout, Cs, NB, H, W, HW;
  float* dw;
  int ldw, coloff;
  int R, Wp, PL, LDA, bands, units, units_per_split, groups;
  unsigned magic_w2, magic_rr, magic_band;      // ceil(2^32 / d) for d = W/2, R + 2, R * W/2
  int fx;
  float* db;         // optional bias gradient (see WgradArgs), summed from the dy band in LDS by the group-0 workgroups
  float* slab = nullptr;       // slab mode (see WgradArgs): [splits][Cout][slab_ld], slab_db [splits][slab_rows]
  float* slab_db = nullptr;
  long long slab_stride = 0;
  int slab_ld = 0, slab_rows = 0;
};

template <bool TAIL>
__global__ __launch_bounds__(256, 2) void conv_wgrad3_narrow_kernel(WgradNarrowArgs a) {
  constexpr int MR = TAIL ? 36 : 32, CG = 32, SB = TAIL ? 6 : 8;     // SB: what fits next to the accumulators without spilling
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int group = blockIdx.x % a.groups, split = blockIdx.x / a.groups;
  const int ci0 = group * CG;
  const int W = a.W, H = a.H, Wp = a.Wp, PL = a.PL, LDA = a.LDA, R = a.R, W2 = a.W >> 1, RR = a.R + 2;
  float* xb = smem;                 // [CG][PL]: element (row rr in -1..R, column c in -1..W) at (rr + 1) * Wp + c + 1
  float* dyb = smem + CG * PL;      // [MR][LDA]
  // zero once: the shared pad column, channels past Cs, cout rows past Cout are never written again
  for (int i = tid; i < CG * PL + MR * LDA; i += 256) smem[i] = 0.f;

  dm_f32x16 acc[9];
  dm_f32x4 acct[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    acct[t] = dm_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int u0 = split * a.units_per_split, u1 = min(a.units, u0 + a.units_per_split);
  const int band_items = R * W2;
  float bias_acc0 = 0.f, bias_acc1 = 0.f;      // this thread's share of row tid >> 3 (and of row 32 + (tid >> 3), tid < 32)
  for (int u = u0; u < u1; ++u) {
    const int n = u / a.bands, band = u - n * a.bands;
    const int r0 = band * R, rows = min(R, H - r0);
    __syncthreads();                // the previous unit's operand reads (first pass: the zero fill) are done
    // staging in batches of SB independent 8-byte loads per lane (issued together, then written to LDS)
    const float* xn = a.x + (size_t)n * a.x_bs + (size_t)ci0 * a.HW;
    const int xtotal = CG * RR * W2;
    for (int base = tid; base < xtotal; base += 256 * SB) {
      float2 v[SB];
      int dst[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int i = base + j * 256;
        const int t = (int)__umulhi((unsigned)i, a.magic_w2), c2 = i - t * W2;
        const int ch = (int)__umulhi((unsigned)t, a.magic_rr), rr = t - ch * RR;
        const int gy = r0 - 1 + rr;
        const bool live = i < xtotal && rr < rows + 2;
        dst[j] = live ? ch * PL + rr * Wp + 2 * c2 + 1 : -1;
        v[j] = make_float2(0.f, 0.f);
        if (live && ci0 + ch < a.Cs && gy >= 0 && gy < H)
          v[j] = *reinterpret_cast<const float2*>(xn + (size_t)ch * a.HW + gy * W + 2 * c2);
      }
#pragma unroll
      for (int j = 0; j < SB; ++j)
        if (dst[j] >= 0) {
          xb[dst[j]] = v[j].x;
          xb[dst[j] + 1] = v[j].y;
        }
    }
    const float* dyn = a.dy + (size_t)n * a.dy_bs + (size_t)r0 * W;
    const int live_pairs = rows * W2;
    const int dtotal = a.Cout * band_items;
    for (int base = tid; base < dtotal; base += 256 * SB) {
      float2 v[SB];
      int dst[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int i = base + j * 256;
        const int co = (int)__umulhi((unsigned)i, a.magic_band), e = i - co * band_items;
        const bool live = i < dtotal && e < live_pairs;
        dst[j] = live ? co * LDA + 2 * e : -1;
        v[j] = make_float2(0.f, 0.f);
        if (live) v[j] = *reinterpret_cast<const float2*>(dyn + (size_t)co * a.HW + 2 * e);
      }
#pragma unroll
      for (int j = 0; j < SB; ++j)
        if (dst[j] >= 0) {
          dyb[dst[j]] = v[j].x;
          dyb[dst[j] + 1] = v[j].y;
        }
    }
    const int live = live_pairs;
    __syncthreads();
    if (a.db != nullptr && group == 0) {
      // 8 threads per cout row (rows 32..35 in a second round), each a strided share of the band
      for (int co = tid >> 3; co < a.Cout; co += 32) {
        const float* rowp = dyb + co * LDA;
        float sacc = 0.f;
        for (int e = tid & 7; e < 2 * live; e += 8) sacc += rowp[e];
        if (co < 32) bias_acc0 += sacc; else bias_acc1 += sacc;
      }
    }
    const float* ap = dyb + l31 * LDA + hi;
    const float* at = dyb + (32 + (lane & 3)) * LDA + hi;
    const float* bp = xb + l31 * PL + hi;
    int r = 0, c2 = wave;           // W2 >= 4 is checked by the launcher: one wrap per step at most
    for (int kp = wave; kp < live; kp += 4) {
      const float a0 = ap[2 * kp];
      const float a1 = TAIL ? at[2 * kp] : 0.f;
      const float* b0 = bp + r * Wp + 2 * c2;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float b = b0[ky * Wp + kx];
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[ky * 3 + kx], 0, 0, 0);
          if (TAIL) acct[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b, acct[ky * 3 + kx], 0, 0, 0);
        }
      c2 += 4;
      if (c2 >= W2) { c2 -= W2; ++r; }
    }
  }
  if (a.db != nullptr && group == 0) {
    float v0 = bias_acc0, v1 = bias_acc1;
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
      v0 += __shfl_xor(v0, d, 64);
      v1 += __shfl_xor(v1, d, 64);
    }
    const int co = tid >> 3;
    if ((tid & 7) == 0) {
      if (a.slab) {
        if (co < a.Cout) a.slab_db[(size_t)split * a.slab_rows + co] = v0;
        if (32 + co < a.Cout) a.slab_db[(size_t)split * a.slab_rows + 32 + co] = v1;
      } else {
        if (co < a.Cout) dm_acc_add(a.db, co, v0, a.fx != 0);
        if (32 + co < a.Cout) dm_acc_add(a.db, 32 + co, v1, a.fx != 0);
      }
    }
  }
  // the four waves' partial sums -> red[tap][row][channel] (wave after wave: 4 short phases), then one atomic each
  float* red = smem;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* d = red + (t * MR + (r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31;
          *d = (w == 0) ? acc[t][r] : *d + acc[t][r];
        }
        if (TAIL) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = acct[t][i] + __shfl_xor(acct[t][i], 32, 64);      // the two pixel parities
            float* d = red + (t * MR + 32 + i) * 32 + l31;
            if (hi == 0) *d = (w == 0) ? v : *d + v;
          }
        }
      }
    }
  }
  __syncthreads();
  const int ncols = min(CG, a.Cs - ci0) * 9;
  for (int i = tid; i < a.Cout * CG * 9; i += 256) {
    const int co = i / (CG * 9), jj = i - co * (CG * 9);
    if (jj >= ncols) continue;
    const int cil = jj / 9, t = jj - cil * 9;
    const float v = red[(t * MR + co) * 32 + cil];
    if (a.slab) a.slab[(size_t)split * a.slab_stride + (size_t)co * a.slab_ld + (size_t)ci0 * 9 + jj] = v;
    else dm_acc_add(a.dw, (size_t)co * a.ldw + a.coloff + (size_t)ci0 * 9 + jj, v, a.fx != 0);
  }
}

static unsigned dm_magic(unsigned d) { return (unsigned)((0x100000000ULL + d - 1) / d); }

// returns DM_OK after launching, or 1 when the shape is not this kernel's
static int wgrad_target_wgs() {
  // split-K until the launch is one round of workgroups (two per CU): every further split adds an epilogue
  // (2048 workgroups: 256->256 1x1 at 14x14 0.140 ms, 512: 0.091 ms; the 2304-row DCN GEMMs 0.63 -> 0.60 ms)
  static const int wgs_env = getenv("DM_WGRAD_WGS") ? atoi(getenv("DM_WGRAD_WGS")) : 0;      // tuning knob (read once)
  return wgs_env > 0 ? wgs_env : 2 * dm_num_cus();
}

static void launch_slab_reduce(const WgradArgs& a, int slabs, int splits, int Jtot, hipStream_t st) {
  const long long total = (long long)a.Cout * ((Jtot + 3) / 4);                      // quads of outputs
  const int P = slabs <= 8 ? 1 : (slabs <= 64 ? 4 : (slabs <= 256 ? 16 : 64));      // 680 slabs of a 64 x 64 tile: 64 partitions
  const long long nblk = (total + 256 / P - 1) / (256 / P) + (a.db ? (a.Cout + 31) / 32 : 0);       // + the bias rows' workgroups
  const float* sdb = a.db ? a.slab_db : nullptr;
  if (P == 1)
    DM_LAUNCH((wgrad_slab_reduce_kernel<1>), dim3((unsigned)nblk), dim3(256), 0, st, a.slab, slabs, a.slab_stride, a.slab_ld, a.Cout, Jtot,
              a.dw, a.ldw, a.coloff, sdb, splits, a.slab_rows, a.db);
  else if (P == 4)
    DM_LAUNCH((wgrad_slab_reduce_kernel<4>), dim3((unsigned)nblk), dim3(256), 0, st, a.slab, slabs, a.slab_stride, a.slab_ld, a.Cout, Jtot,
              a.dw, a.ldw, a.coloff, sdb, splits, a.slab_rows, a.db);
  else if (P == 16)
    DM_LAUNCH((wgrad_slab_reduce_kernel<16>), dim3((unsigned)nblk), dim3(256), 0, st, a.slab, slabs, a.slab_stride, a.slab_ld, a.Cout, Jtot,
              a.dw, a.ldw, a.coloff, sdb, splits, a.slab_rows, a.db);
  else
    DM_LAUNCH((wgrad_slab_reduce_kernel<64>), dim3((unsigned)nblk), dim3(256), 0, st, a.slab, slabs, a.slab_stride, a.slab_ld, a.Cout, Jtot,
              a.dw, a.ldw, a.coloff, sdb, splits, a.slab_rows, a.db);
}

static int launch_wgrad3_narrow(const WgradArgs& g, hipStream_t st) {
  if (g.Cout > 36 || (g.W & 1) || g.W < 8 || g.H < 1) return 1;
  if (((uintptr_t)g.dy | (uintptr_t)g.x) & 7u) return 1;
  if ((g.dy_bs | g.x_bs) & 1LL) return 1;
  const bool tail = g.Cout > 32;
  const int MR = tail ? 36 : 32, Wp = g.W + 1;
  const size_t budget = 78 * 1024;
  auto bytes_for = [&](int R, int* PL, int* LDA) {
    *PL = ((R + 2) * Wp + 1) | 1;
    *LDA = (R * g.W) | 1;
    return sizeof(float) * ((size_t)32 * *PL + (size_t)MR * *LDA);
  };
  int PL = 0, LDA = 0, Rmax = 0;
  for (int R = min(g.H, 64); R >= 1; --R)
    if (bytes_for(R, &PL, &LDA) <= budget) { Rmax = R; break; }
  if (Rmax < 1) return 1;
  const int nb = dm_ceil_div(g.H, Rmax);
  const int R = dm_ceil_div(g.H, nb);
  size_t bytes = bytes_for(R, &PL, &LDA);
  bytes = max(bytes, sizeof(float) * (size_t)9 * MR * 32);                   // the epilogue's reduction buffer
  if ((long long)32 * (R + 2) * (g.W / 2) >= 65536 || (long long)g.Cout * R * (g.W / 2) >= 65536) return 1;   // magic-division range
  WgradNarrowArgs a;
  a.dy = g.dy; a.dy_bs = g.dy_bs; a.x = g.x; a.x_bs = g.x_bs; a.Cout = g.Cout; a.Cs = g.Cs; a.NB = g.NB; a.H = g.H; a.W = g.W;
  a.HW = g.HW; a.dw = g.dw; a.ldw = g.ldw; a.coloff = g.coloff; a.fx = g.fx; a.db = g.db;
  a.R = R; a.Wp = Wp; a.PL = PL; a.LDA = LDA; a.bands = nb; a.units = g.NB * nb; a.groups = dm_ceil_div(g.Cs, 32);
  a.magic_w2 = dm_magic(g.W / 2); a.magic_rr = dm_magic(R + 2); a.magic_band = dm_magic(R * (g.W / 2));
  const int target = wgrad_target_wgs();
  const int nsplit = max(1, min(a.units, target / a.groups));
  a.units_per_split = dm_ceil_div(a.units, nsplit);
  const int splits = dm_ceil_div(a.units, a.units_per_split);
  WgradArgs red = g;
  if (g.scratch && !g.fx && (((uintptr_t)g.scratch) & 15) == 0) {      // (the reduce reads the slabs by 16-byte loads)
    const int ld = a.groups * 32 * 9;
    const long long stride = (long long)g.Cout * ld, need = (long long)splits * stride + (long long)splits * MR;
    if (need <= g.scratch_floats) {
      a.slab = g.scratch; a.slab_db = g.scratch + (long long)splits * stride;
      a.slab_stride = stride; a.slab_ld = ld; a.slab_rows = MR;
      red.slab = a.slab; red.slab_db = a.slab_db; red.slab_stride = stride; red.slab_ld = ld; red.slab_rows = MR;
    }
  }
  static bool attr_t[DM_MAX_DEVICES] = {false}, attr_n[DM_MAX_DEVICES] = {false};
  if (tail) {
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&conv_wgrad3_narrow_kernel<true>), 80 * 1024, attr_t) != DM_OK) return DM_ERR_LAUNCH;
    DM_LAUNCH((conv_wgrad3_narrow_kernel<true>), dim3((unsigned)(a.groups * splits)), dim3(256), bytes, st, a);
  } else {
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&conv_wgrad3_narrow_kernel<false>), 80 * 1024, attr_n) != DM_OK) return DM_ERR_LAUNCH;
    DM_LAUNCH((conv_wgrad3_narrow_kernel<false>), dim3((unsigned)(a.groups * splits)), dim3(256), bytes, st, a);
  }
  if (a.slab) launch_slab_reduce(red, splits, splits, g.Cs * 9, st);
  return DM_OK;
}

template <int KS>
void launch_wgrad(WgradArgs a, hipStream_t st) {
  const int J = a.Cs * KS * KS;
  const int TM = a.Cout <= 64 ? 64 : 128;
  // candidate column-tile widths; cost = padded columns x (1 + 32/TN) (A re-load share)
  const int cands[3] = {256, 128, 64};
  int TN = 64;
  double best = 1e30;
  for (int c = 0; c < 3; ++c) {
    const int tn = cands[c];
    if (TM == 128 && tn == 256) continue;
    const double cost = (double)dm_ceil_div(J, tn) * tn * (1.0 + 32.0 / tn);
    if (cost < best) { best = cost; TN = tn; }
  }
  a.MT = dm_ceil_div(a.Cout, TM);
  a.JT = dm_ceil_div(J, TN);
  const int chunks = dm_ceil_div(a.Q, 32);
  const int target_wgs = wgrad_target_wgs();
  int nsplit = max(1, min(chunks, target_wgs / max(1, a.MT * a.JT)));
  a.chunks_per_split = dm_ceil_div(chunks, nsplit);
  a.nsplit = dm_ceil_div(chunks, a.chunks_per_split);
  const dim3 grid((unsigned)(a.MT * a.JT * a.nsplit));
  const int WGK = (TM == 128 && TN == 128) ? 1 : (TM == 128) ? 2 : (TN == 256) ? 1 : (TN == 128) ? 2 : 4;
  if (a.scratch && !a.fx && (((uintptr_t)a.scratch) & 15) == 0) {      // (the reduce reads the slabs by 16-byte loads)
    const int rows = a.MT * TM, ld = a.JT * TN;
    const long long stride = (long long)rows * ld, need = (long long)a.nsplit * WGK * stride + (long long)a.nsplit * rows;
    if (need <= a.scratch_floats) {
      a.slab = a.scratch; a.slab_db = a.scratch + (long long)a.nsplit * WGK * stride;
      a.slab_stride = stride; a.slab_ld = ld; a.slab_rows = rows;
    }
  }
  if (TM == 128 && TN == 128) DM_LAUNCH((conv_wgrad_kernel<KS, 2, 2, 1>), grid, dim3(256), 0, st, a);
  else if (TM == 128) DM_LAUNCH((conv_wgrad_kernel<KS, 2, 1, 2>), grid, dim3(256), 0, st, a);
  else if (TN == 256 && a.Cout > 32 && a.Cout <= 36) DM_LAUNCH((conv_wgrad_kernel<KS, 1, 4, 1, true>), grid, dim3(256), 0, st, a);
  else if (TN == 128 && a.Cout > 32 && a.Cout <= 36) DM_LAUNCH((conv_wgrad_kernel<KS, 1, 2, 2, true>), grid, dim3(256), 0, st, a);
  else if (TN == 256) DM_LAUNCH((conv_wgrad_kernel<KS, 1, 4, 1>), grid, dim3(256), 0, st, a);
  else if (TN == 128) DM_LAUNCH((conv_wgrad_kernel<KS, 1, 2, 2>), grid, dim3(256), 0, st, a);
  else DM_LAUNCH((conv_wgrad_kernel<KS, 1, 1, 4>), grid, dim3(256), 0, st, a);
  if (a.slab) launch_slab_reduce(a, a.nsplit * WGK, a.nsplit, J, st);
}

// ----------------------------------------------------------------- K11 backward
// adjoint of the x2 bilinear upsample (output-driven scatter, 4 atomics per element);
// optional ReLU mask from the forward output.
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ yout,
                                                             int NC, int H, int W, int ac, float* __restrict__ gin) {
  const int OH = 2 * H, OW = 2 * W;
  const size_t total = (size_t)NC * OH * OW;
  const float rh = ac ? (OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f) : 0.5f;
  const float rw = ac ? (OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f) : 0.5f;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float g = gout[idx];
    if (yout && !(yout[idx] > 0.f)) g = 0.f;
    if (g == 0.f) continue;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const size_t nc = idx / ((size_t)OW * OH);
    float sy, sx;
    if (ac) {
      sy = rh * (float)oy;
      sx = rw * (float)ox;
    } else {
      sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);
      sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
    }
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    float* p = gin + nc * H * W;
    atomicAdd(p + y0 * W + x0, g * hy * hx);
    atomicAdd(p + y0 * W + x1, g * hy * lx);
    atomicAdd(p + y1 * W + x0, g * ly * hx);
    atomicAdd(p + y1 * W + x1, g * ly * lx);
  }
}

// align_corners=False, scale 2: closed-form adjoint as a GATHER (no atomics).  Output row
// oy = 2k samples src = k - 0.25 (rows k-1: 0.25, k: 0.75; k = 0 clamps: row 0 gets 1),
// oy = 2k+1 samples src = k + 0.25 (rows k: 0.75, k+1: 0.25; at the last row both taps
// fold into row k).  So input row y gathers from output rows 2y-1, 2y, 2y+1, 2y+2.
__device__ __forceinline__ void up2_adjoint_weights(int y, int H, float w[4]) {
  w[0] = (y >= 1) ? 0.25f : 0.f;                 // oy = 2y-1
  w[1] = (y == 0) ? 1.0f : 0.75f;                // oy = 2y
  w[2] = (y == H - 1) ? 1.0f : 0.75f;            // oy = 2y+1
  w[3] = (y <= H - 2) ? 0.25f : 0.f;             // oy = 2y+2
}

__global__ __launch_bounds__(256) void upsample2x_bwd_gather_kernel(const float* __restrict__ gout,
                                                                    const float* __restrict__ yout, int NC, int H, int W,
                                                                    float* __restrict__ gin) {
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)NC * H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const long long nc = idx / ((long long)W * H);
    float wy[4], wx[4];
    up2_adjoint_weights(y, H, wy);
    up2_adjoint_weights(x, W, wx);
    const float* g = gout + nc * OH * OW;
    const float* m = yout ? yout + nc * OH * OW : nullptr;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = 2 * y - 1 + a;
      if (wy[a] == 0.f) continue;
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ox = 2 * x - 1 + b;
        if (wx[b] == 0.f) continue;
        float v = g[oy * OW + ox];
        if (m && !(m[oy * OW + ox] > 0.f)) v = 0.f;
        row += wx[b] * v;
      }
      acc += wy[a] * row;
    }
    gin[idx] = acc;
  }
}

// The same adjoint with the (ReLU-masked) output-gradient plane staged in LDS by 16-byte loads: the gather above
// issues 32 dword loads per input pixel (16 gradients + 16 mask values, half of each sector unused) and ran at
// 1.5 TB/s; here every global byte is read once, 16 bytes per lane.  One plane per workgroup pass; needs
// (2H * 2W) % 4 == 0 and a plane of at most 16 K floats (64 KB).
__global__ __launch_bounds__(256) void upsample2x_bwd_lds_kernel(const float* __restrict__ gout, const float* __restrict__ yout,
                                                                 int NC, int H, int W, float* __restrict__ gin) {
  extern __shared__ __attribute__((aligned(16))) float plane[];     // [2H][2W] masked gradients
  const int OH = 2 * H, OW = 2 * W, OHW = OH * OW, HW = H * W;
  for (int nc = blockIdx.x; nc < NC; nc += gridDim.x) {
    const dm_f32x4* g4 = reinterpret_cast<const dm_f32x4*>(gout + (size_t)nc * OHW);
    const dm_f32x4* m4 = yout ? reinterpret_cast<const dm_f32x4*>(yout + (size_t)nc * OHW) : nullptr;
    for (int i = threadIdx.x; i < OHW / 4; i += 256) {
      dm_f32x4 v = g4[i];
      if (m4) {
        const dm_f32x4 m = m4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!(m[e] > 0.f)) v[e] = 0.f;
      }
      reinterpret_cast<dm_f32x4*>(plane)[i] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < HW; idx += 256) {
      const int y = idx / W, x = idx - y * W;
      float wy[4], wx[4];
      up2_adjoint_weights(y, H, wy);
      up2_adjoint_weights(x, W, wx);
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int oy = min(max(2 * y - 1 + a, 0), OH - 1);          // clamped rows / columns carry weight 0
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int ox = min(max(2 * x - 1 + b, 0), OW - 1);
          row += wx[b] * plane[oy * OW + ox];
        }
        acc += wy[a] * row;
      }
      gin[(size_t)nc * HW + idx] = acc;
    }
    __syncthreads();
  }
}

// align_corners=True adjoint as an LDS-staged gather.  The scatter form (upsample2x_bwd_kernel) sends four global
// atomics per output pixel into a plane a quarter of the size, i.e. 16 colliding atomics per address: 150 us for
// the 12.8 MB of the final 56 -> 112 logits.  Here the gradient plane is staged once (16-byte loads) and every
// input pixel gathers the output rows / columns whose forward footprint (y0, y1) contains it, with the
// forward's own weights: the source coordinate is rh * oy with rh = (H-1)/(2H-1) just under 1/2, so the
// candidates of input row y are the integers of the open interval ((y-1)/rh, (y+1)/rh), at most 7 of them.
template <int NT>
__global__ __launch_bounds__(NT) void upsample2x_bwd_ac_lds_kernel(const float* __restrict__ gout,
                                                                   const float* __restrict__ yout, int NC, int H, int W,
                                                                   float* __restrict__ gin) {
  // Two passes over the staged plane, as the sums are written: rows[oy][x] = sum over the candidate columns ox of
  // wx(ox, x) * plane[oy][ox], then gin[y][x] = sum over the candidate rows oy of wy(oy, y) * rows[oy][x].  (One pass
  // recomputed every row sum for each of the ~7 output rows that use it: 49 weighted candidates per input pixel, 41 us for
  // the 256 planes of the 112 -> 56 case on one workgroup of 256 threads per CU.)  Same terms in the same order: same bits.
  extern __shared__ __attribute__((aligned(16))) float plane[];     // [2H][2W] masked gradients, then rows [2H][W]
  const int OH = 2 * H, OW = 2 * W, OHW = OH * OW, HW = H * W;
  float* rows = plane + OHW;
  const float rh = (float)(H - 1) / (float)(OH - 1), rw = (float)(W - 1) / (float)(OW - 1);
  const float ih = (float)(OH - 1) / (float)(H - 1), iw = (float)(OW - 1) / (float)(W - 1);
  for (int nc = blockIdx.x; nc < NC; nc += gridDim.x) {
    const dm_f32x4* g4 = reinterpret_cast<const dm_f32x4*>(gout + (size_t)nc * OHW);
    const dm_f32x4* m4 = yout ? reinterpret_cast<const dm_f32x4*>(yout + (size_t)nc * OHW) : nullptr;
    for (int i = threadIdx.x; i < OHW / 4; i += NT) {
      dm_f32x4 v = g4[i];
      if (m4) {
        const dm_f32x4 m = m4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!(m[e] > 0.f)) v[e] = 0.f;
      }
      reinterpret_cast<dm_f32x4*>(plane)[i] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < OH * W; idx += NT) {
      const int oy = idx / W, x = idx - oy * W;
      // one candidate more on either side than the interval needs: rounding of the bounds cannot lose a column
      const int ox_lo = max((int)((float)(x - 1) * iw) - 1, 0), ox_hi = min((int)((float)(x + 1) * iw) + 2, OW - 1);
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const float sx = rw * (float)ox;
        const int x0 = (int)sx;
        const int x1 = x0 + ((x0 < W - 1) ? 1 : 0);
        const float lx = sx - (float)x0;
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        row += wx * plane[oy * OW + ox];
      }
      rows[idx] = row;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < HW; idx += NT) {
      const int y = idx / W, x = idx - y * W;
      const int oy_lo = max((int)((float)(y - 1) * ih) - 1, 0), oy_hi = min((int)((float)(y + 1) * ih) + 2, OH - 1);
      float acc = 0.f;
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const float sy = rh * (float)oy;
        const int y0 = (int)sy;
        const int y1 = y0 + ((y0 < H - 1) ? 1 : 0);
        const float ly = sy - (float)y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        acc += wy * rows[oy * W + x];
      }
      gin[(size_t)nc * HW + idx] = acc;
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------- K4 backward
// Adjoint of the point sample.  A small RoI maps its S x S lattice onto a handful of
// feature pixels, so thousands of samples of one workgroup hit the same addresses:
// when the RoI's footprint fits an LDS tile the scatter-adds are first reduced there
// (64-bit fixed-point ds atomics, see dcn_col2im_lds_kernel) and flushed with ONE global
// atomic per touched pixel; large RoIs (sparse lattice, little contention) go straight
// to global atomics.  grid = (channel chunks of CT, N).
__device__ __forceinline__ void ps_coord(float lo, float hi, int i, int S, int size, float scale, float& s) {
  const float g0 = ((float)(2 * i + 1)) / (float)S - 1.0f;
  float p = (g0 + 1.0f) / 2.0f;
  p = p * (hi - lo) + lo;
  p = p / (float)size * scale;
  const float g = p * 2.0f - 1.0f;
  s = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
}

template <int CT>
__global__ __launch_bounds__(256) void point_sample_bwd_kernel(const float* __restrict__ gout, int B, int C, int H, int W,
                                                               const float* __restrict__ rois, int N, int S, float scale,
                                                               float* __restrict__ gfeat, int lds_elems, int fixed) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];   // [CT][TH][TW] fixed point
  __shared__ int bad[CT];            // a NaN / Inf gradient cannot be represented in fixed point: it poisons its plane
  if (threadIdx.x < CT) bad[threadIdx.x] = 0;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * CT;
  const float* r = rois + (size_t)n * 5;
  const int b = (int)r[0];
  if (b < 0 || b >= B) return;
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  // footprint of the lattice on the feature map (coordinates are monotonic in the lattice index)
  float sxa, sxb, sya, syb;
  ps_coord(x1, x2, 0, S, W, scale, sxa);
  ps_coord(x1, x2, S - 1, S, W, scale, sxb);
  ps_coord(y1, y2, 0, S, H, scale, sya);
  ps_coord(y1, y2, S - 1, S, H, scale, syb);
  const float sxmin = fminf(sxa, sxb), sxmax = fmaxf(sxa, sxb), symin = fminf(sya, syb), symax = fmaxf(sya, syb);
  if (sxmax < -1.f || sxmin > (float)W || symax < -1.f || symin > (float)H) return;   // every tap void
  const int fx0 = max((int)floorf(fmaxf(sxmin, -1.f)), 0), fx1 = min((int)floorf(fminf(sxmax, (float)W)) + 1, W - 1);
  const int fy0 = max((int)floorf(fmaxf(symin, -1.f)), 0), fy1 = min((int)floorf(fminf(symax, (float)H)) + 1, H - 1);
  const int TW = fx1 - fx0 + 1, TH = fy1 - fy0 + 1;
  // Channels per pass.  A footprint too large for CT accumulator planes may still fit CT/2 or one: the workgroup then
  // walks its channels in passes (the sample geometry is recomputed, every gradient is still read once) and keeps
  // reducing on chip.  Round 2 sent such RoIs -- 156 to 312 image pixels wide on a stride-4 map, where neighbouring
  // samples still share pixels -- straight to the memory-side atomics, four per sample: 10.4 us per RoI against 1.3.
  int cpp = CT;
  if (TW > 0 && TH > 0 && TW * TH * CT > lds_elems) {
    if (CT >= 4 && TW * TH * 2 <= lds_elems) cpp = 2;
    else if (TW * TH <= lds_elems) cpp = 1;
  }
  const bool use_lds = TW > 0 && TH > 0 && TW * TH * cpp <= lds_elems;
  if (!use_lds) cpp = CT;
  const size_t plane = (size_t)H * W;
  const int nch_all = min(CT, C - c0);
  for (int cs = 0; cs < nch_all; cs += cpp) {
  const int nch = min(cpp, nch_all - cs);
  if (use_lds) {
    if (cs > 0) __syncthreads();                 // the previous pass has flushed its planes
    if (threadIdx.x < CT) bad[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < nch * TH * TW; i += blockDim.x) tile[i] = 0ull;
    __syncthreads();
  }
  // fixed: gfeat is a 64-bit fixed-point map (dm_point_sample_bwd_fx); cell indices are the same, cells twice as wide
  float* gf = gfeat + ((size_t)b * C + c0 + cs) * plane * (fixed ? 2 : 1);
  for (int pos = threadIdx.x; pos < S * S; pos += blockDim.x) {
    const int iy = pos / S, ix = pos - iy * S;
    float sx, sy;
    ps_coord(x1, x2, ix, S, W, scale, sx);
    ps_coord(y1, y2, iy, S, H, scale, sy);
    const float fx = floorf(sx), fy = floorf(sy);
    if (fx < -1.f || fx > (float)W || fy < -1.f || fy > (float)H) continue;
    const int x0 = (int)fx, y0 = (int)fy, x1i = x0 + 1, y1i = y0 + 1;
    const float lx = sx - fx, ly = sy - fy;
    const float w_nw = (1.f - lx) * (1.f - ly), w_ne = lx * (1.f - ly), w_sw = (1.f - lx) * ly, w_se = lx * ly;
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1i >= 0 && x1i < W, oky0 = y0 >= 0 && y0 < H, oky1 = y1i >= 0 && y1i < H;
    const float* go = gout + ((size_t)n * C + c0 + cs) * S * S + pos;
    float gv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) gv[c] = (c < nch) ? go[(size_t)c * S * S] : 0.f;
    if (use_lds) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (!isfinite(gv[c])) { bad[c] = 1; gv[c] = 0.f; }
      const int tx0 = x0 - fx0, ty0 = y0 - fy0;     // in range by construction of the footprint
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c >= nch) break;
        unsigned long long* t = tile + (size_t)c * TH * TW;
        const double g = (double)gv[c] * DM_FIX_SCALE;
        if (okx0 && oky0) atomicAdd(t + ty0 * TW + tx0, (unsigned long long)__double2ll_rn(g * (double)w_nw));
        if (okx1 && oky0) atomicAdd(t + ty0 * TW + tx0 + 1, (unsigned long long)__double2ll_rn(g * (double)w_ne));
        if (okx0 && oky1) atomicAdd(t + (ty0 + 1) * TW + tx0, (unsigned long long)__double2ll_rn(g * (double)w_sw));
        if (okx1 && oky1) atomicAdd(t + (ty0 + 1) * TW + tx0 + 1, (unsigned long long)__double2ll_rn(g * (double)w_se));
      }
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c >= nch) break;
        const size_t cb = (size_t)c * plane;
        if (okx0 && oky0) dm_acc_add(gf, cb + y0 * W + x0, gv[c] * w_nw, fixed != 0);
        if (okx1 && oky0) dm_acc_add(gf, cb + y0 * W + x1i, gv[c] * w_ne, fixed != 0);
        if (okx0 && oky1) dm_acc_add(gf, cb + y1i * W + x0, gv[c] * w_sw, fixed != 0);
        if (okx1 && oky1) dm_acc_add(gf, cb + y1i * W + x1i, gv[c] * w_se, fixed != 0);
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nch * TH * TW; i += blockDim.x) {
      const long long q = (long long)tile[i];
      const int c = i / (TH * TW);
      if (q != 0 || bad[c]) {
        const int rem = i - c * TH * TW;
        const int ty = rem / TW, tx = rem - ty * TW;
        const size_t cell = (size_t)c * plane + (size_t)(fy0 + ty) * W + fx0 + tx;
        if (fixed) {      // the LDS sum is already exact in the same fixed-point format: hand it over as it is
          atomicAdd(reinterpret_cast<unsigned long long*>(gf) + cell, bad[c] ? (unsigned long long)(1LL << 62) : (unsigned long long)q);
        } else {
          const float v = bad[c] ? __builtin_nanf("") : (float)((double)q * (1.0 / DM_FIX_SCALE));
          atomicAdd(gf + cell, v);
        }
      }
    }
  }
  }      // channel passes
}

// ----------------------------------------------------------------- K7 backward
// gx[n,c,p] (+)= wi[lab][c]*gi[n,p] + wd[lab][c]*gd[n,p];
// gWi[lab][c] += sum_p gi*x ; gbi[lab] += sum_p gi  (same for the detail branch).
// grid = (C, N): one workgroup per (RoI, channel).
// Small maps (H*W <= 1024, a multiple of 4): a WAVE owns a channel plane (16-byte accesses, its sums by wave shuffles: no LDS, no
// barrier) and walks CPW channels; a workgroup = 4 waves.  The kernel below gives every (channel, RoI) plane a 256-thread
// workgroup of its own -- 65536 workgroups of 196 elements at 14 x 14, each ending in a barrier and two global atomics.
template <int CPW, int IT>
__global__ __launch_bounds__(256) void class_logits_bwd_wave_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                                    const float* __restrict__ wi, const float* __restrict__ wd,
                                                                    int num_classes, const int64_t* __restrict__ labels,
                                                                    const float* __restrict__ gi, const float* __restrict__ gd,
                                                                    float* __restrict__ gx, int accumulate,
                                                                    float* __restrict__ gwi, float* __restrict__ gbi,
                                                                    float* __restrict__ gwd, float* __restrict__ gbd, int fx,
                                                                    float* __restrict__ part) {
  // IT = quads per lane and plane (HW / 4 <= 64 * IT).  The RoI's two logit-gradient planes are read once per wave, and a
  // plane's x (and gx, when accumulating) quads are all requested before the first is used: written as a loop over the
  // quads, a 28 x 28 plane was four dependent round trips per channel (1.7 TB/s where the 14 x 14 case ran at 4.4).
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int lab = (int)labels[n];
  lab = min(max(lab, 0), num_classes - 1);
  const int HWq = HW >> 2;
  const dm_f32x4* gip = reinterpret_cast<const dm_f32x4*>(gi + (size_t)n * HW);
  const dm_f32x4* gdp = reinterpret_cast<const dm_f32x4*>(gd + (size_t)n * HW);
  dm_f32x4 g1[IT], g2[IT];
  float ti = 0.f, td = 0.f;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int p = lane + it * 64;
    const bool ok = p < HWq;
    g1[it] = gip[ok ? p : 0];
    g2[it] = gdp[ok ? p : 0];
    if (!ok) g1[it] = g2[it] = dm_f32x4{0.f, 0.f, 0.f, 0.f};      // (zero gradients: surplus lanes add nothing to any sum)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ti += g1[it][e];
      td += g2[it][e];
    }
  }
  const int cbase = (blockIdx.x * 4 + wave) * CPW;
  for (int k = 0; k < CPW; ++k) {
    const int c = cbase + k;
    if (c >= C) break;
    const float a = wi[(size_t)lab * C + c], b = wd[(size_t)lab * C + c];
    const dm_f32x4* xp = reinterpret_cast<const dm_f32x4*>(x + ((size_t)n * C + c) * HW);
    dm_f32x4* gxp = reinterpret_cast<dm_f32x4*>(gx + ((size_t)n * C + c) * HW);
    dm_f32x4 xv[IT], gv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int p = min(lane + it * 64, HWq - 1);
      xv[it] = xp[p];
      if (accumulate) gv[it] = gxp[p];
    }
    float si = 0.f, sd = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int p = lane + it * 64;
      dm_f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = a * g1[it][e] + b * g2[it][e];
        si += g1[it][e] * xv[it][e];
        sd += g2[it][e] * xv[it][e];
      }
      if (accumulate) v += gv[it];
      if (p < HWq) gxp[p] = v;
    }
    si = wsum(si);
    sd = wsum(sd);
    if (lane == 0) {
      if (part) {            // slab mode: the RoI's partial sums, added per class in RoI order by class_logits_bwd_reduce_kernel
        part[((size_t)n * C + c) * 2] = si;
        part[((size_t)n * C + c) * 2 + 1] = sd;
      } else {
        dm_acc_add(gwi, (size_t)lab * C + c, si, fx != 0);
        dm_acc_add(gwd, (size_t)lab * C + c, sd, fx != 0);
      }
    }
    if (c == 0) {   // bias gradient once per RoI
      const float tis = wsum(ti), tds = wsum(td);
      if (lane == 0) {
        if (part) {
          part[(size_t)N * C * 2 + (size_t)n * 2] = tis;
          part[(size_t)N * C * 2 + (size_t)n * 2 + 1] = tds;
        } else {
          dm_acc_add(gbi, lab, tis, fx != 0);
          dm_acc_add(gbd, lab, tds, fx != 0);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void class_logits_bwd_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                               const float* __restrict__ wi, const float* __restrict__ wd,
                                                               int num_classes, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ gi, const float* __restrict__ gd,
                                                               float* __restrict__ gx, int accumulate,
                                                               float* __restrict__ gwi, float* __restrict__ gbi,
                                                               float* __restrict__ gwd, float* __restrict__ gbd, int fx,
                                                               float* __restrict__ part) {
  __shared__ float red[8];
  const int c = blockIdx.x, n = blockIdx.y;
  int lab = (int)labels[n];
  lab = min(max(lab, 0), num_classes - 1);
  const float a = wi[(size_t)lab * C + c], b = wd[(size_t)lab * C + c];
  const float* xp = x + ((size_t)n * C + c) * HW;
  float* gxp = gx + ((size_t)n * C + c) * HW;
  const float* gip = gi + (size_t)n * HW;
  const float* gdp = gd + (size_t)n * HW;
  float si = 0.f, sd = 0.f, ti = 0.f, td = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float g1 = gip[p], g2 = gdp[p], xv = xp[p];
    const float v = a * g1 + b * g2;
    gxp[p] = accumulate ? gxp[p] + v : v;
    si += g1 * xv;
    sd += g2 * xv;
    ti += g1;
    td += g2;
  }
  si = wsum(si);
  sd = wsum(sd);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[w] = si;
    red[4 + w] = sd;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (part) {
      part[((size_t)n * C + c) * 2] = red[0] + red[1] + red[2] + red[3];
      part[((size_t)n * C + c) * 2 + 1] = red[4] + red[5] + red[6] + red[7];
    } else {
      dm_acc_add(gwi, (size_t)lab * C + c, red[0] + red[1] + red[2] + red[3], fx != 0);
      dm_acc_add(gwd, (size_t)lab * C + c, red[4] + red[5] + red[6] + red[7], fx != 0);
    }
  }
  if (c == 0) {   // bias gradient once per RoI
    __syncthreads();
    ti = wsum(ti);
    td = wsum(td);
    if ((threadIdx.x & 63) == 0) {
      red[w] = ti;
      red[4 + w] = td;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (part) {
        part[(size_t)N * C * 2 + (size_t)n * 2] = red[0] + red[1] + red[2] + red[3];
        part[(size_t)N * C * 2 + (size_t)n * 2 + 1] = red[4] + red[5] + red[6] + red[7];
      } else {
        dm_acc_add(gbi, lab, red[0] + red[1] + red[2] + red[3], fx != 0);
        dm_acc_add(gbd, lab, red[4] + red[5] + red[6] + red[7], fx != 0);
      }
    }
  }
}

// Slab mode of the class-gathered logits' parameter gradients.  The kernels above add every RoI's (channel) sums into the
// row of its class with atomics: the RoIs of an image share a handful of classes, and 256 RoIs of ONE class serialise 256
// float atomics on each of the row's addresses (measured, 256 x 256 x 14 x 14: 27 us with 80 random classes, 130 us with
// one).  Here they write their sums to part[n][c][2] (+ part_b[n][2]) and this kernel adds, for class blockIdx.y, the RoIs
// of that class in RoI order: no contention, and a fixed order of additions (the reference's is autograd's index_put_
// accumulate: unordered).  One uncontended atomic per touched address lands the sum, so that calls on two streams may share
// the gradient buffers.
__global__ __launch_bounds__(256) void class_logits_bwd_reduce_kernel(const float* __restrict__ part, const int64_t* __restrict__ labels,
                                                                      int N, int C, int num_classes, float* __restrict__ gwi,
                                                                      float* __restrict__ gbi, float* __restrict__ gwd,
                                                                      float* __restrict__ gbd) {
  // workgroup = (16 channels, class l): 16 partitions of the RoIs (n = partition, partition + 16, ...), each added in RoI order
  // with unconditional loads (a RoI of another class adds 0: a branch per RoI made this loop a chain of dependent round
  // trips, 80 us for 256 RoIs of one class), the partitions then in partition order
  __shared__ float red[16][16][4];
  __shared__ unsigned char hit[1024];
  const int l = blockIdx.y;
  const int co = threadIdx.x & 15, pidx = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + co;
  const int cc = min(c, C - 1);
  const float* pb = part + (size_t)N * C * 2;
  float si = 0.f, sd = 0.f, ti = 0.f, td = 0.f;
  int any = 0;
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int tile = min(1024, N - n0);
    int mine = 0;
    for (int i = threadIdx.x; i < tile; i += 256) {
      int lab = (int)labels[n0 + i];
      lab = min(max(lab, 0), num_classes - 1);
      hit[i] = lab == l;
      mine |= lab == l;
    }
    const int here = __syncthreads_or(mine);
    any |= here;
    if (here) {
#pragma unroll 4
      for (int i = pidx; i < tile; i += 16) {
        const size_t n = (size_t)(n0 + i);
        const float a = part[(n * C + cc) * 2], b = part[(n * C + cc) * 2 + 1];
        const float ba = pb[n * 2], bb = pb[n * 2 + 1];
        const bool h = hit[i] != 0;
        si += h ? a : 0.f;
        sd += h ? b : 0.f;
        ti += h ? ba : 0.f;
        td += h ? bb : 0.f;
      }
    }
    __syncthreads();
  }
  if (!any) return;
  red[pidx][co][0] = si;
  red[pidx][co][1] = sd;
  red[pidx][co][2] = ti;
  red[pidx][co][3] = td;
  __syncthreads();
  if (pidx != 0) return;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] += red[q][co][k];
  if (c < C) {
    atomicAdd(gwi + (size_t)l * C + c, r[0]);
    atomicAdd(gwd + (size_t)l * C + c, r[1]);
  }
  if (c == 0) {
    atomicAdd(gbi + l, r[2]);
    atomicAdd(gbd + l, r[3]);
  }
}

// ----------------------------------------------------------------- DCN backward pieces
struct DcnSample {
  bool valid;
  int h_low, w_low;
  float h_im, w_im;
};

__device__ __forceinline__ DcnSample dcn_sample(const float* offp, int tap, int HW, int y, int x, int H, int W) {
  DcnSample s;
  const int ki = tap / 3, kj = tap - ki * 3;
  s.h_im = (float)(y - 1 + ki) + offp[(size_t)(2 * tap) * HW];
  s.w_im = (float)(x - 1 + kj) + offp[(size_t)(2 * tap + 1) * HW];
  s.valid = s.h_im > -1.f && s.w_im > -1.f && s.h_im < (float)H && s.w_im < (float)W;
  s.h_low = (int)floorf(s.h_im);
  s.w_low = (int)floorf(s.w_im);
  return s;
}

// col[n][(tap*C + ci)][p] = bilinear sample (deformable im2col, tap-major rows)
__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            int NB, int C, int H, int W, int dg, float* __restrict__ col,
                                                            int CT) {
  const int HW = H * W;
  const int pblocks = (HW + 255) / 256;
  const int chunks = (C / dg + CT - 1) / CT;
  int bid = blockIdx.x;
  const int pb = bid % pblocks; bid /= pblocks;
  const int chunk = bid % chunks; bid /= chunks;
  const int tap = bid % 9; bid /= 9;
  const int g = bid % dg;
  const int n = bid / dg;
  const int p = pb * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, xx = p - y * W;
  const float* offp = offset + ((size_t)n * dg + g) * 18 * HW + p;
  const DcnSample s = dcn_sample(offp, tap, HW, y, xx, H, W);
  const int cpg = C / dg;
  const int c0 = g * cpg + chunk * CT, c1 = min(c0 + CT, (g + 1) * cpg);
  // the 2x2 footprint as two 8-byte row pairs (as the forward kernel, deform_conv.hip): pair base column
  // cb = clamp(w_low, 0, W-2); taps outside the image keep weight 0, so the loads stay in bounds and the
  // four products are the reference's w1..w4 * v1..v4
  struct __attribute__((packed, aligned(4))) F2 { float a, b; };
  float wt0 = 0.f, wt1 = 0.f, wb0 = 0.f, wb1 = 0.f;
  int ot = 0, ob = 0;
  if (s.valid) {
    const float lh = s.h_im - (float)s.h_low, lw = s.w_im - (float)s.w_low, hh = 1.f - lh, hw = 1.f - lw;
    const float wr_t = (s.h_low >= 0) ? hh : 0.f;
    const float wr_b = (s.h_low + 1 <= H - 1) ? lh : 0.f;
    const int rt = min(max(s.h_low, 0), H - 1), rbm = min(max(s.h_low + 1, 0), H - 1);
    const int cb = min(max(s.w_low, 0), W - 2);
    const float wc0 = (cb == s.w_low ? hw : 0.f) + (cb == s.w_low + 1 ? lw : 0.f);
    const float wc1 = (cb + 1 == s.w_low ? hw : 0.f) + (cb + 1 == s.w_low + 1 ? lw : 0.f);
    ot = rt * W + cb;
    ob = rbm * W + cb;
    wt0 = wr_t * wc0; wt1 = wr_t * wc1;
    wb0 = wr_b * wc0; wb1 = wr_b * wc1;
  }
  for (int c = c0; c < c1; ++c) {
    const float* xc = x + ((size_t)n * C + c) * HW;
    const F2 top = *reinterpret_cast<const F2*>(xc + ot);
    const F2 bot = *reinterpret_cast<const F2*>(xc + ob);
    col[((size_t)n * 9 * C + (size_t)tap * C + c) * HW + p] = wt0 * top.a + wt1 * top.b + wb0 * bot.a + wb1 * bot.b;
  }
}

// Second generation of the deformable im2col (round 2).  The kernel above issues, per column-matrix element,
// two 8-byte gathers and one 4-byte store: 3 vector-memory instructions per 4 bytes of output, and it ran at
// 2 TB/s of stores.  Here a workgroup owns (image, deformable group, CT channels): the CT planes are staged in
// LDS with 16-byte loads, a thread handles 4 consecutive pixels of one tap -- their 2x2 footprints are read from
// LDS -- and writes ONE 16-byte store per channel.  Same bilinear expression as above (bit-identical rows).
template <int CT>
__global__ __launch_bounds__(256) void deform_im2col_lds_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                                int C, int H, int W, int dg, float* __restrict__ col) {
  // Third generation (end of round 3): the CT planes are staged as float4 per pixel and channel quad ([CT / 4][HW] quads, the
  // layout of the DCN forward kernels), so that a sample's corner is ONE 16-byte LDS read for four channels -- 16 reads per
  // four pixels and quad where the planar layout took 64.
  extern __shared__ __attribute__((aligned(16))) float planes[];      // [CT / 4][HW] float4
  static_assert(CT % 4 == 0, "channel quads");
  constexpr int NQ = CT / 4;
  const int HW = H * W, HWq = HW >> 2;
  const int cpg = C / dg;
  const int chunks = cpg / CT;
  int bid = blockIdx.x;
  const int chunk = bid % chunks; bid /= chunks;
  const int g = bid % dg;
  const int n = bid / dg;
  const int c0 = g * cpg + chunk * CT;
  const int tid = threadIdx.x;
  dm_f32x4* pq = reinterpret_cast<dm_f32x4*>(planes);
  {
    // a thread loads 4 consecutive pixels of the 4 planes of a quad (16-byte loads) and writes 4 interleaved float4
    const float* src = x + ((size_t)n * C + c0) * HW;
    for (int i = tid; i < NQ * HWq; i += 256) {
      const int q = i / HWq, p4 = i - q * HWq;
      const dm_f32x4 a = reinterpret_cast<const dm_f32x4*>(src + (size_t)(4 * q) * HW)[p4];
      const dm_f32x4 b = reinterpret_cast<const dm_f32x4*>(src + (size_t)(4 * q + 1) * HW)[p4];
      const dm_f32x4 c = reinterpret_cast<const dm_f32x4*>(src + (size_t)(4 * q + 2) * HW)[p4];
      const dm_f32x4 d = reinterpret_cast<const dm_f32x4*>(src + (size_t)(4 * q + 3) * HW)[p4];
      dm_f32x4* dst = pq + (size_t)q * HW + 4 * p4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = dm_f32x4{a[e], b[e], c[e], d[e]};
    }
  }
  __syncthreads();
  const float* offb = offset + ((size_t)n * dg + g) * 18 * HW;
  for (int it = tid; it < 9 * HWq; it += 256) {
    const int tap = it / HWq;
    const int p0 = (it - tap * HWq) * 4;
    const int ki = tap / 3, kj = tap - ki * 3;
    const dm_f32x4 oh = *reinterpret_cast<const dm_f32x4*>(offb + (size_t)(2 * tap) * HW + p0);
    const dm_f32x4 ow = *reinterpret_cast<const dm_f32x4*>(offb + (size_t)(2 * tap + 1) * HW + p0);
    int ot[4], ob[4];
    float wt0[4], wt1[4], wb0[4], wb1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = p0 + e;
      const int y = p / W, xx = p - y * W;
      const float h_im = (float)(y - 1 + ki) + oh[e];
      const float w_im = (float)(xx - 1 + kj) + ow[e];
      ot[e] = ob[e] = 0;
      wt0[e] = wt1[e] = wb0[e] = wb1[e] = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
        const float wr_t = (h_low >= 0) ? hh : 0.f;
        const float wr_b = (h_low + 1 <= H - 1) ? lh : 0.f;
        const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_low + 1, 0), H - 1);
        const int cb = min(max(w_low, 0), W - 2);
        const float wc0 = (cb == w_low ? hw : 0.f) + (cb == w_low + 1 ? lw : 0.f);
        const float wc1 = (cb + 1 == w_low ? hw : 0.f) + (cb + 1 == w_low + 1 ? lw : 0.f);
        ot[e] = rt * W + cb;
        ob[e] = rbm * W + cb;
        wt0[e] = wr_t * wc0; wt1[e] = wr_t * wc1;
        wb0[e] = wr_b * wc0; wb1[e] = wr_b * wc1;
      }
    }
    float* dst = col + ((size_t)n * 9 * C + (size_t)tap * C + c0) * HW + p0;
#pragma unroll 2
    for (int q = 0; q < NQ; ++q) {
      const dm_f32x4* pl = pq + (size_t)q * HW;
      dm_f32x4 v[4];      // [channel of the quad][pixel]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const dm_f32x4 ta = pl[ot[e]], tb = pl[ot[e] + 1], ba = pl[ob[e]], bb = pl[ob[e] + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k][e] = wt0[e] * ta[k] + wt1[e] * tb[k] + wb0[e] * ba[k] + wb1[e] * bb[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<dm_f32x4*>(dst + (size_t)(4 * q + k) * HW) = v[k];
    }
  }
}

// colgrad[n][(tap*C + ci)][p] -> goffset (coordinate gradient, summed over the channels
// of the deformable group by the owning thread: no atomics).  Thread = (n, group, tap, pixel).
__global__ __launch_bounds__(256) void dcn_coord_grad_kernel(const float* __restrict__ colgrad, const float* __restrict__ x,
                                                             const float* __restrict__ offset, int NB, int C, int H, int W,
                                                             int dg, float* __restrict__ goffset) {
  const int HW = H * W;
  const int pblocks = (HW + 255) / 256;
  int bid = blockIdx.x;
  const int pb = bid % pblocks; bid /= pblocks;
  const int tap = bid % 9; bid /= 9;
  const int g = bid % dg;
  const int n = bid / dg;
  const int p = pb * 256 + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, xx = p - y * W;
  const float* offp = offset + ((size_t)n * dg + g) * 18 * HW + p;
  const DcnSample s = dcn_sample(offp, tap, HW, y, xx, H, W);
  float* goff = goffset + ((size_t)n * dg + g) * 18 * HW + p;
  if (!s.valid) {
    goff[(size_t)(2 * tap) * HW] = 0.f;
    goff[(size_t)(2 * tap + 1) * HW] = 0.f;
    return;
  }
  const int h_low = s.h_low, w_low = s.w_low, h_high = h_low + 1, w_high = w_low + 1;
  const float lh = s.h_im - (float)h_low, lw = s.w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
  // corner values through two 8-byte row-pair loads (pair base column cb = clamp(w_low, 0, W-2)); a
  // corner outside the image contributes 0: its selector weights are 0
  struct __attribute__((packed, aligned(4))) F2 { float a, b; };
  const bool rt_ok = h_low >= 0, rb_ok = h_high <= H - 1;
  const int rt = min(max(h_low, 0), H - 1), rbm = min(max(h_high, 0), H - 1);
  const int cb = min(max(w_low, 0), W - 2);
  // value at column w_low = sa0 * pair.a + sa1 * pair.b, at column w_high = sb0 * pair.a + sb1 * pair.b
  const float sa0 = (w_low == cb) ? 1.f : 0.f, sa1 = (w_low == cb + 1) ? 1.f : 0.f;
  const float sb0 = (w_high == cb) ? 1.f : 0.f, sb1 = (w_high == cb + 1) ? 1.f : 0.f;
  const float mt = rt_ok ? 1.f : 0.f, mb = rb_ok ? 1.f : 0.f;
  const int ot = rt * W + cb, ob = rbm * W + cb;
  const int cpg = C / dg;
  float acc_h = 0.f, acc_w = 0.f;
  const float* cgp = colgrad + ((size_t)n * 9 * C + (size_t)tap * C + (size_t)g * cpg) * HW + p;
  const float* xg = x + ((size_t)n * C + (size_t)g * cpg) * HW;
#pragma unroll 4
  for (int c = 0; c < cpg; ++c) {
    const float cg = cgp[(size_t)c * HW];
    const float* xc = xg + (size_t)c * HW;
    const F2 top = *reinterpret_cast<const F2*>(xc + ot);
    const F2 bot = *reinterpret_cast<const F2*>(xc + ob);
    const float x1 = mt * (sa0 * top.a + sa1 * top.b), x2 = mt * (sb0 * top.a + sb1 * top.b);
    const float x3 = mb * (sa0 * bot.a + sa1 * bot.b), x4 = mb * (sb0 * bot.a + sb1 * bot.b);
    // d val / d h_im and d val / d w_im  (get_coordinate_weight, :145-188)
    acc_h += cg * (-hw * x1 - lw * x2 + hw * x3 + lw * x4);
    acc_w += cg * (-hh * x1 + hh * x2 - lh * x3 + lh * x4);
  }
  goff[(size_t)(2 * tap) * HW] = acc_h;
  goff[(size_t)(2 * tap + 1) * HW] = acc_w;
}

// Coordinate gradient, second generation (round 3).  The kernel above takes the four corner values of every
// (channel, tap, pixel) through two scattered 8-byte global loads -- twice as many texture-path gathers as it has column
// gradients to read, and it ran at half the rate its 1.85 GB stream allows (0.77 ms at 56 x 56).  Here a workgroup owns
// a band of output rows of one (image, deformable group) and walks the channels in chunks of 8 whose rows around the
// band are staged in LDS as [quad][pixel] float4 (the DCN forward's layout): a corner of four channels is ONE 16-byte
// LDS read.  The staged band carries a zero column either side and zero rows beyond the image, so the corners of a
// sample are always the four neighbours at (h_low, w_low) and an outside corner reads 0 -- the selectors and masks of
// the first kernel are in the data, and an item keeps three registers (offset, lh, lw).  The column gradients arrive as
// coalesced dword loads; sums run over the channels in index order: no atomics.  A sample whose rows leave the staged
// band (an offset beyond DCN_COORD_HALO rows) takes the global loads of the first kernel.
constexpr int DCN_COORD_HALO = 5;
template <int IT, int XS, int NT>
__global__ __launch_bounds__(NT, 4) void dcn_coord_grad_lds_kernel(const float* __restrict__ colgrad, const float* __restrict__ x,
                                                                 const float* __restrict__ offset, int NB, int C, int H, int W,
                                                                 int dg, float* __restrict__ goffset, int bands, int BRows,
                                                                 int XR) {
  extern __shared__ __attribute__((aligned(16))) dm_f32x4 xs[];      // [2 buffers][2 quads][(XR + 2) * (W + 2)]
  const int HW = H * W, cpg = C / dg, Wp = W + 2, XP = (XR + 2) * Wp;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int band = bid % bands; bid /= bands;
  const int g = bid % dg;
  const int n = bid / dg;
  const int y_first = band * BRows;
  const int band_px = min(BRows, H - y_first) * W;
  const int items = 9 * band_px;
  const int band_y0 = min(max(y_first - (XR - BRows) / 2, 0), H - XR);      // first staged image row (host: XR <= H)

  // ---- geometry of this thread's items
  int ot[IT], ipt[IT];                   // top-left corner inside the staged band; pixel | tap << 20 (-1: no item)
  float lhv[IT], lwv[IT];
  float acc_h[IT], acc_w[IT];
  unsigned oob = 0;                      // items whose rows leave the staged band
  const float* offb = offset + ((size_t)n * dg + g) * 18 * HW;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int it = tid + k * NT;
    ot[k] = 0; ipt[k] = -1;
    lhv[k] = lwv[k] = 0.f;
    acc_h[k] = acc_w[k] = 0.f;
    if (it < items) {
      const int tap = it / band_px, pl = it - tap * band_px;
      const int p = y_first * W + pl;
      const int y = p / W, xx = p - y * W;
      ipt[k] = p | (tap << 20);
      const DcnSample s = dcn_sample(offb + p, tap, HW, y, xx, H, W);
      if (s.valid) {                      // h_low in [-1, H-1], w_low in [-1, W-1]
        lhv[k] = s.h_im - (float)s.h_low;
        lwv[k] = s.w_im - (float)s.w_low;
        if (s.h_low >= band_y0 - 1 && s.h_low + 1 <= band_y0 + XR) ot[k] = (s.h_low - (band_y0 - 1)) * Wp + s.w_low + 1;
        else oob |= 1u << k;
      } else {
        ipt[k] |= 1 << 30;               // void sample: the gradient is 0
      }
    }
  }

  // ---- channel chunks: stage the band rows of 8 channels, then every item adds its 8 channels
  int x_off[XS];                          // float offset of staging slot i relative to (n, first channel of the chunk); -1: a zero
#pragma unroll
  for (int i = 0; i < XS; ++i) {
    const int idx = tid + i * NT;
    x_off[i] = -1;
    if (idx < 2 * XP) {
      const int quad = idx / XP, ppx = idx - quad * XP;
      const int r = ppx / Wp, c = ppx - r * Wp;
      const int yy = band_y0 - 1 + r, xc = c - 1;
      if (yy >= 0 && yy < H && xc >= 0 && xc < W) x_off[i] = quad * 4 * HW + yy * W + xc;
    }
  }
  dm_f32x4 rx[XS];
  const float* xg = x + ((size_t)n * C + (size_t)g * cpg) * HW;          // this group's planes
  auto load_x = [&](int c0) {
    const float* xc = xg + (size_t)c0 * HW;
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (x_off[i] >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = xc[x_off[i] + e * HW];
      }
      rx[i] = v;
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int idx = tid + i * NT;
      if (idx < 2 * XP) xs[buf * 2 * XP + idx] = rx[i];
    }
  };
  struct __attribute__((packed, aligned(4))) F2 { float a, b; };
  load_x(0);
  store_x(0);
  __syncthreads();
  const float* cgb = colgrad + ((size_t)n * 9 * C + (size_t)g * cpg) * HW;
  for (int c0 = 0, buf = 0; c0 < cpg; c0 += 8, buf ^= 1) {
    const bool more = c0 + 8 < cpg;
    if (more) load_x(c0 + 8);
    const dm_f32x4* xq = xs + buf * 2 * XP;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      if (ipt[k] < 0 || (ipt[k] >> 30) || ((oob >> k) & 1u)) continue;
      const int p = ipt[k] & 0xfffff, tap = ipt[k] >> 20;
      const float* cgp = cgb + ((size_t)tap * C + c0) * HW + p;
      float cg[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) cg[c] = cgp[(size_t)c * HW];
      const float lh = lhv[k], lw = lwv[k], hh = 1.f - lh, hw = 1.f - lw;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const dm_f32x4* pt = xq + q * XP + ot[k];
        const dm_f32x4 x1 = pt[0], x2 = pt[1], x3 = pt[Wp], x4 = pt[Wp + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // d val / d h = -hw x1 - lw x2 + hw x3 + lw x4, d val / d w = -hh x1 + hh x2 - lh x3 + lh x4
          // (get_coordinate_weight, deform_conv_cuda_kernel.cu:145-188)
          const float gh = __builtin_fmaf(lw, x4[e], __builtin_fmaf(hw, x3[e], __builtin_fmaf(-lw, x2[e], -hw * x1[e])));
          const float gw = __builtin_fmaf(lh, x4[e], __builtin_fmaf(-lh, x3[e], __builtin_fmaf(hh, x2[e], -hh * x1[e])));
          acc_h[k] = __builtin_fmaf(cg[q * 4 + e], gh, acc_h[k]);
          acc_w[k] = __builtin_fmaf(cg[q * 4 + e], gw, acc_w[k]);
        }
      }
    }
    if (more) store_x(buf ^ 1);          // the other buffer was last read in the previous iteration, behind its barrier
    __syncthreads();
  }
  if (oob) {
    // far samples: corners from global memory, outside ones masked (the first kernel's arithmetic), all channels
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      if (!((oob >> k) & 1u)) continue;
      const int p = ipt[k] & 0xfffff, tap = ipt[k] >> 20;
      const int y = p / W, xx = p - y * W;
      const DcnSample s = dcn_sample(offb + p, tap, HW, y, xx, H, W);
      const int h_high = s.h_low + 1, w_high = s.w_low + 1;
      const int rt = min(max(s.h_low, 0), H - 1), rbm = min(max(h_high, 0), H - 1);
      const int cb = min(max(s.w_low, 0), W - 2);
      const float sa0 = (s.w_low == cb) ? 1.f : 0.f, sa1 = (s.w_low == cb + 1) ? 1.f : 0.f;
      const float sb0 = (w_high == cb) ? 1.f : 0.f, sb1 = (w_high == cb + 1) ? 1.f : 0.f;
      const float mt = (s.h_low >= 0) ? 1.f : 0.f, mb = (h_high <= H - 1) ? 1.f : 0.f;
      const float lh = lhv[k], lw = lwv[k], hh = 1.f - lh, hw = 1.f - lw;
      float ah = 0.f, aw = 0.f;
      for (int c = 0; c < cpg; ++c) {
        const float cgv = cgb[((size_t)tap * C + c) * HW + p];
        const F2 top = *reinterpret_cast<const F2*>(xg + (size_t)c * HW + rt * W + cb);
        const F2 bot = *reinterpret_cast<const F2*>(xg + (size_t)c * HW + rbm * W + cb);
        const float x1 = mt * (sa0 * top.a + sa1 * top.b), x2 = mt * (sb0 * top.a + sb1 * top.b);
        const float x3 = mb * (sa0 * bot.a + sa1 * bot.b), x4 = mb * (sb0 * bot.a + sb1 * bot.b);
        const float gh = __builtin_fmaf(lw, x4, __builtin_fmaf(hw, x3, __builtin_fmaf(-lw, x2, -hw * x1)));
        const float gw = __builtin_fmaf(lh, x4, __builtin_fmaf(-lh, x3, __builtin_fmaf(hh, x2, -hh * x1)));
        ah = __builtin_fmaf(cgv, gh, ah);
        aw = __builtin_fmaf(cgv, gw, aw);
      }
      acc_h[k] = ah;
      acc_w[k] = aw;
    }
  }
  float* goff = goffset + ((size_t)n * dg + g) * 18 * HW;
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    if (ipt[k] < 0) continue;
    const int p = ipt[k] & 0xfffff, tap = (ipt[k] >> 20) & 15;
    goff[(size_t)(2 * tap) * HW + p] = acc_h[k];
    goff[(size_t)(2 * tap + 1) * HW + p] = acc_w[k];
  }
}

// Deformable col2im (data gradient).  Every contribution to gx[n, c] comes from the
// colgrad rows (tap, c) of the same image, so a workgroup owns (n, CT channels),
// accumulates the 9 x HW x 4 scatter-adds in an LDS copy of the planes and writes each
// plane once: no global atomics (the reference's col2im kernel,
// deform_conv_cuda_kernel.cu:279-335, uses atomicAdd to HBM).
// The LDS accumulators are 64-bit FIXED POINT (2^-36 resolution, |sum| < 1.3e8):
// measured on gfx950 (tools/micro/lds_atomics.hip) ds_add_f32 runs at 0.38 lane-atomics
// per clock per CU, ds_add_u64 at 6.4 (17x) -- and integer sums are order-independent,
// so this gradient is bitwise reproducible run to run.
template <int CT, int NTH = 256>
__global__ __launch_bounds__(NTH) void dcn_col2im_lds_kernel(const float* __restrict__ colgrad,
                                                             const float* __restrict__ offset, int NB, int C, int H, int W,
                                                             int dg, float* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];   // [CT][HW] fixed point
  __shared__ int bad[CT];            // non-finite column gradients poison their plane (see the header's contract)
  if (threadIdx.x < CT) bad[threadIdx.x] = 0;
  const int HW = H * W;
  const int chunks = C / CT;
  const int n = blockIdx.x / chunks;
  const int c0 = (blockIdx.x - n * chunks) * CT;
  const int g = c0 / (C / dg);
  for (int i = threadIdx.x; i < CT * HW; i += blockDim.x) lds[i] = 0ull;
  __syncthreads();
  const float* offb = offset + ((size_t)n * dg + g) * 18 * HW;
  const float* cgb = colgrad + ((size_t)n * 9 * C + c0) * HW;
  auto scatter = [&](int tap, int p, float oh, float ow, const float (&cgv)[CT]) {
    const int y = p / W, xx = p - y * W;
    const int ki = tap / 3, kj = tap - ki * 3;
    const float h_im = (float)(y - 1 + ki) + oh;
    const float w_im = (float)(xx - 1 + kj) + ow;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im), h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const bool v1 = h_low >= 0 && w_low >= 0, v2 = h_low >= 0 && w_high <= W - 1;
    const bool v3 = h_high <= H - 1 && w_low >= 0, v4 = h_high <= H - 1 && w_high <= W - 1;
    const int o1 = h_low * W + w_low, o2 = h_low * W + w_high, o3 = h_high * W + w_low, o4 = h_high * W + w_high;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    const float a1 = w1 * 16.f, a2 = w2 * 16.f, a3 = w3 * 16.f, a4 = w4 * 16.f;          // (dm_fix36_mul's two scalings of a weight)
    const float b1 = w1 * 68719476736.f, b2 = w2 * 68719476736.f, b3 = w3 * 68719476736.f, b4 = w4 * 68719476736.f;
    // (corners outside the map are branched around: adding zero to a clamped cell instead -- no exec-mask sequences -- measured
    // slower, 0.93 -> 0.98 ms: the atomics it adds cost more than the branches it removes)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      unsigned long long* pl = lds + c * HW;
      if (v1) atomicAdd(pl + o1, dm_fix36_mul(cgv[c], a1, b1));
      if (v2) atomicAdd(pl + o2, dm_fix36_mul(cgv[c], a2, b2));
      if (v3) atomicAdd(pl + o3, dm_fix36_mul(cgv[c], a3, b3));
      if (v4) atomicAdd(pl + o4, dm_fix36_mul(cgv[c], a4, b4));
    }
  };
  if ((HW & 3) == 0) {
    // Items = (tap, 4 consecutive pixels): the column gradients of an item come as ONE 16-byte load per channel
    // (round 1 loaded them dword by dword: 1.4-1.5 TB/s of the 4-5 the stream can reach), U items per thread and
    // trip so that 2*U offset loads and CT*U gradient loads are in flight before the first LDS atomic.
    constexpr int U = (CT >= 8) ? 1 : 2;
    const int HWq = HW >> 2;
    for (int it0 = threadIdx.x; it0 < 9 * HWq; it0 += U * blockDim.x) {
      dm_f32x4 oh[U], ow[U], cg[U][CT];
      int tapv[U], pv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = min(it0 + u * (int)blockDim.x, 9 * HWq - 1);
        tapv[u] = it / HWq;
        pv[u] = (it - tapv[u] * HWq) * 4;
        oh[u] = *reinterpret_cast<const dm_f32x4*>(offb + (size_t)(2 * tapv[u]) * HW + pv[u]);
        ow[u] = *reinterpret_cast<const dm_f32x4*>(offb + (size_t)(2 * tapv[u] + 1) * HW + pv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* cgp = cgb + (size_t)tapv[u] * C * HW + pv[u];
#pragma unroll
        for (int c = 0; c < CT; ++c) cg[u][c] = *reinterpret_cast<const dm_f32x4*>(cgp + (size_t)c * HW);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (it0 + u * (int)blockDim.x >= 9 * HWq) continue;
        // The four pixels of an item lie side by side: where two neighbours sample one row and adjacent
        // columns (w_low differs by one -- every interior pixel while the offsets vary by less than a pixel), the
        // right-hand cells of the left pixel ARE the left-hand cells of the right one.  The two fp32 products are added in
        // registers (ONE more fp32 rounding: tl += ct) and the sum is cut to the 2^-36 grid and sent as one atomic: 10
        // instead of 16 per channel for a fully linked item.  So a merged cell is another association of the same
        // products than two atomics would give (last bits differ from the scatter() path of planes with HW % 4 != 0) --
        // still a function of the data alone, the same bits every run and whatever the arrival order (an
        // earlier comment claimed integer adds).  The kernel is bound by the LDS atomics (ds_add_u64, ~18 cycles per
        // wave instruction), not by anything this adds.
        const int tap = tapv[u], ki = tap / 3, kj = tap - ki * 3;
        const int y = pv[u] / W, x0 = pv[u] - y * W;
        int o1[4];
        unsigned vm[4];                   // bit i: corner i + 1 is inside the map (0: void sample)
        float wa[4][4];                   // 2^4 x the bilinear weights (dm_fix36_abs16's scaling)
        int hl[4], wl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ye = y, xe = x0 + e;                        // (HW % 4 == 0 does not make W % 4 == 0: an item may wrap)
          if (xe >= W) { xe -= W; ++ye; }
          const float h_im = (float)(ye - 1 + ki) + oh[u][e];
          const float w_im = (float)(xe - 1 + kj) + ow[u][e];
          const bool ok = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
          hl[e] = h_low; wl[e] = w_low;
          o1[e] = h_low * W + w_low;
          const bool v1 = h_low >= 0 && w_low >= 0, v2 = h_low >= 0 && w_low + 1 <= W - 1;
          const bool v3 = h_low + 1 <= H - 1 && w_low >= 0, v4 = h_low + 1 <= H - 1 && w_low + 1 <= W - 1;
          vm[e] = ok ? ((unsigned)v1 | (unsigned)v2 << 1 | (unsigned)v3 << 2 | (unsigned)v4 << 3) : 0u;
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          wa[e][0] = w1 * 16.f; wa[e][1] = w2 * 16.f; wa[e][2] = w3 * 16.f; wa[e][3] = w4 * 16.f;
        }
        bool link[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) link[e] = vm[e] != 0u && vm[e + 1] != 0u && hl[e] == hl[e + 1] && wl[e + 1] == wl[e] + 1;
        // (A branch-free body for waves whose 64 items are all interior, fully linked and finite was tried: with 4.5 rows
        // of the plane per wave nearly every wave holds a border item, and as a per-lane branch it runs beside the
        // generic body instead of replacing it: 0.808 -> 0.827 ms.)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          unsigned long long* pl = lds + c * HW;
          float ct = 0.f, cb = 0.f;                         // the previous pixel's right-hand products (x 2^4), if linked
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float gv = cg[u][c][e];
            if (!isfinite(gv)) { bad[c] = 1; gv = 0.f; }
            float tl = gv * wa[e][0], bl = gv * wa[e][2];
            if (e > 0 && link[e - 1]) { tl += ct; bl += cb; }
            // (corners outside the map are branched around: adding zero to a clamped cell instead -- no exec-mask
            // sequences -- measured slower, 0.93 -> 0.98 ms: the atomics it adds cost more than the branches it removes)
            if (vm[e] & 1u) dm_fix36_accumulate(pl + o1[e], tl);
            if (vm[e] & 4u) dm_fix36_accumulate(pl + o1[e] + W, bl);
            const float tr = gv * wa[e][1], br = gv * wa[e][3];
            if (e < 3 && link[e]) {
              ct = tr; cb = br;
            } else {
              if (vm[e] & 2u) dm_fix36_accumulate(pl + o1[e] + 1, tr);
              if (vm[e] & 8u) dm_fix36_accumulate(pl + o1[e] + W + 1, br);
            }
          }
        }
      }
    }
  } else {
    for (int it = threadIdx.x; it < 9 * HW; it += blockDim.x) {
      const int tap = it / HW, p = it - tap * HW;
      float cgv[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        cgv[c] = cgb[((size_t)tap * C + c) * HW + p];
        if (!isfinite(cgv[c])) { bad[c] = 1; cgv[c] = 0.f; }
      }
      scatter(tap, p, offb[(size_t)(2 * tap) * HW + p], offb[(size_t)(2 * tap + 1) * HW + p], cgv);
    }
  }
  __syncthreads();
  float* dst = gx + ((size_t)n * C + c0) * HW;
  for (int i = threadIdx.x; i < CT * HW; i += blockDim.x)
    dst[i] = bad[i / HW] ? __builtin_nanf("") : (float)((double)(long long)lds[i] * (1.0 / DM_FIX_SCALE));
}

// W[co][ci][tap]  <->  Wt[(tap*C + ci)][co]  (the two DCN GEMMs run as 1x1 convs over
// the tap-major column matrix)
__global__ void dcn_weight_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int C,
                                          int to_colmajor, int accumulate) {
  const int total = Cout * C * 9;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    if (to_colmajor) {        // dst[(tap*C+ci)*Cout + co] = src[(co*C+ci)*9 + tap]
      const int co = idx % Cout;
      const int j = idx / Cout;
      const int tap = j / C, ci = j - tap * C;
      dst[idx] = src[((size_t)co * C + ci) * 9 + tap];
    } else {                  // dst[(co*C+ci)*9 + tap] (+)= src[co*(9C) + tap*C + ci]
      const int tap = idx % 9;
      const int ci = (idx / 9) % C;
      const int co = idx / (9 * C);
      const float v = src[(size_t)co * 9 * C + tap * C + ci];
      dst[idx] = accumulate ? dst[idx] + v : v;
    }
  }
}

int grid_for(size_t n) { return (int)min((size_t)dm_ceil_div((long long)n, 256), (size_t)16384); }

}  // namespace

extern "C" int dm_relu_bwd(float* grad, const float* out, long long count, dm_stream_t stream) {
  if (!grad || !out || count < 0) return DM_ERR_INVALID_ARG;
  if (count == 0) return DM_OK;
  if (count % 4 == 0 && ((uintptr_t)grad & 15) == 0 && ((uintptr_t)out & 15) == 0) {
    DM_LAUNCH(relu_bwd4_kernel, dim3(grid_for((size_t)count / 4)), dim3(256), 0, (hipStream_t)stream,
              reinterpret_cast<dm_f32x4*>(grad), reinterpret_cast<const dm_f32x4*>(out), (size_t)count / 4);
    return dm_check_launch();
  }
  DM_LAUNCH(relu_bwd_kernel, dim3(grid_for((size_t)count)), dim3(256), 0, (hipStream_t)stream, grad, out, (size_t)count);
  return dm_check_launch();
}

extern "C" int dm_sigmoid_bwd(const float* sig, long long sig_bs, const float* ga, long long ga_bs, const float* gb,
                              long long gb_bs, int N, int HW, float* g_logit, int accumulate, dm_stream_t stream) {
  if (!sig || !ga || !g_logit || N < 0 || HW <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  DM_LAUNCH(sigmoid_bwd_kernel, dim3(grid_for((size_t)N * HW)), dim3(256), 0, (hipStream_t)stream, sig, sig_bs, ga, ga_bs,
            gb, gb_bs, N, HW, g_logit, accumulate);
  return dm_check_launch();
}

extern "C" int dm_channel_sum(const float* g, long long batch_stride, int NB, int C, int HW, float* out, int accumulate,
                              dm_stream_t stream) {
  if (!g || !out || NB <= 0 || C <= 0 || HW <= 0) return DM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st) != hipSuccess) return DM_ERR_LAUNCH;
  const long long total = (long long)NB * HW;
  int splits = (int)min((long long)max(1, 2048 / C), (total + 1023) / 1024);
  splits = max(splits, 1);
  DM_LAUNCH(channel_sum_kernel, dim3(C, splits), dim3(256), 0, st, g, batch_stride, NB, C, HW, out, 0);
  return dm_check_launch();
}

extern "C" int dm_channel_sum_fx(const float* g, long long batch_stride, int NB, int C, int HW, long long* out_fx,
                                 dm_stream_t stream) {
  if (!g || !out_fx || NB <= 0 || C <= 0 || HW <= 0) return DM_ERR_INVALID_ARG;
  const long long total = (long long)NB * HW;
  int splits = (int)min((long long)max(1, 2048 / C), (total + 1023) / 1024);
  splits = max(splits, 1);
  DM_LAUNCH(channel_sum_kernel, dim3(C, splits), dim3(256), 0, (hipStream_t)stream, g, batch_stride, NB, C, HW,
            reinterpret_cast<float*>(out_fx), 1);
  return dm_check_launch();
}

namespace {
__global__ __launch_bounds__(256) void fx_to_float_kernel(long long* __restrict__ fx, long long n, float* __restrict__ out,
                                                          int accumulate, int clear) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long q = fx[i];
  const bool poisoned = q >= (1LL << 61) || q <= -(1LL << 61);
  const float v = poisoned ? __builtin_nanf("") : (float)((double)q * (1.0 / DM_FX_ONE));
  out[i] = accumulate ? out[i] + v : v;
  if (clear) fx[i] = 0;
}
// Gather form of the point-sample adjoint.  The scatter kernel above is bound by its LDS atomics: four per sample and channel,
// and the samples of a 56 x 56 lattice over a 30 x 30-pixel footprint hit the same cells (0.46 ms at 256 x 64 x 56 x 56 for
// 0.24 GB of traffic; cheaper arithmetic and float LDS atomics were both tried -- DESIGN.md).  Here a thread owns a CELL of
// the footprint: the samples whose bilinear hat covers column gx are those with floor(sx) = gx - 1 (weight lx) or gx (weight
// 1 - lx), and since the sample coordinate is monotonic in the lattice index each of the two sets is a run of consecutive
// indices -- found once per workgroup (first / last index and count per map column and row, in LDS; a count that disagrees with
// the run length means rounding broke the monotonicity of a degenerate RoI: that RoI takes the per-sample path).  The CT
// gradient planes are staged in LDS and only read; a cell's sum is formed in a fixed order (rows, then columns) and lands
// with ONE atomic per cell and channel.
template <int CT>
__global__ __launch_bounds__(256) void point_sample_bwd_gather_kernel(const float* __restrict__ gout, int B, int C, int H, int W,
                                                                      const float* __restrict__ rois, int N, int S, float scale,
                                                                      float* __restrict__ gfeat, int fixed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int broken;
  const int SS = S * S;
  float* gpl = sm;                                    // [CT][S * S] gradients
  float* lxs = gpl + CT * SS;                         // [S] fractional parts
  float* lys = lxs + S;
  int* x0s = reinterpret_cast<int*>(lys + S);         // [S] floor(sx), clipped to [-2, W + 1]
  int* y0s = x0s + S;
  int* cfirst = y0s + S;                              // [W + 4] per value v + 2: first index, last index + 1, count
  int* clast = cfirst + (W + 4);
  int* ccnt = clast + (W + 4);
  int* rfirst = ccnt + (W + 4);                       // [H + 4]
  int* rlast = rfirst + (H + 4);
  int* rcnt = rlast + (H + 4);
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * CT;
  const float* r = rois + (size_t)n * 5;
  const int b = (int)r[0];
  if (b < 0 || b >= B) return;
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  float sxa, sxb, sya, syb;
  ps_coord(x1, x2, 0, S, W, scale, sxa);
  ps_coord(x1, x2, S - 1, S, W, scale, sxb);
  ps_coord(y1, y2, 0, S, H, scale, sya);
  ps_coord(y1, y2, S - 1, S, H, scale, syb);
  const float sxmin = fminf(sxa, sxb), sxmax = fmaxf(sxa, sxb), symin = fminf(sya, syb), symax = fmaxf(sya, syb);
  if (sxmax < -1.f || sxmin > (float)W || symax < -1.f || symin > (float)H) return;   // every tap void
  const int fx0 = max((int)floorf(fmaxf(sxmin, -1.f)), 0), fx1 = min((int)floorf(fminf(sxmax, (float)W)) + 1, W - 1);
  const int fy0 = max((int)floorf(fmaxf(symin, -1.f)), 0), fy1 = min((int)floorf(fminf(symax, (float)H)) + 1, H - 1);
  const int TW = fx1 - fx0 + 1, TH = fy1 - fy0 + 1;
  if (TW <= 0 || TH <= 0) return;
  const int nch = min(CT, C - c0);
  const int tid = threadIdx.x;
  if (tid == 0) broken = 0;
  for (int i = tid; i < W + 4; i += 256) { cfirst[i] = 0x7fffffff; clast[i] = 0; ccnt[i] = 0; }
  for (int i = tid; i < H + 4; i += 256) { rfirst[i] = 0x7fffffff; rlast[i] = 0; rcnt[i] = 0; }
  __syncthreads();
  if (tid < 2 * S) {
    const bool ax = tid < S;
    const int i = ax ? tid : tid - S;
    float sc;
    if (ax) ps_coord(x1, x2, i, S, W, scale, sc);
    else ps_coord(y1, y2, i, S, H, scale, sc);
    const int size = ax ? W : H;
    const float f = floorf(sc);
    const int v = (f < -1.f) ? -2 : (f > (float)size ? size + 1 : (int)f);      // out-of-range samples: sentinel values
    (ax ? lxs : lys)[i] = sc - f;
    (ax ? x0s : y0s)[i] = v;
    atomicMin((ax ? cfirst : rfirst) + v + 2, i);
    atomicMax((ax ? clast : rlast) + v + 2, i + 1);
    atomicAdd((ax ? ccnt : rcnt) + v + 2, 1);
  }
  // the gradient planes
  {
    const float* go = gout + ((size_t)n * C + c0) * SS;
    for (int i = tid; i < nch * SS; i += 256) gpl[i] = go[i];
  }
  __syncthreads();
  // a value whose indices are not one run: not monotonic
  for (int i = tid; i < W + 4; i += 256)
    if (ccnt[i] > 0 && clast[i] - cfirst[i] != ccnt[i]) broken = 1;
  for (int i = tid; i < H + 4; i += 256)
    if (rcnt[i] > 0 && rlast[i] - rfirst[i] != rcnt[i]) broken = 1;
  __syncthreads();
  const size_t plane = (size_t)H * W;
  float* gf = gfeat + ((size_t)b * C + c0) * plane * (fixed ? 2 : 1);
  if (broken) {
    // per-sample path (as the scatter kernel's direct form)
    for (int pos = tid; pos < SS; pos += 256) {
      const int iy = pos / S, ix = pos - iy * S;
      const int x0 = x0s[ix], y0 = y0s[iy];
      if (x0 < -1 || x0 > W || y0 < -1 || y0 > H) continue;
      const int x1i = x0 + 1, y1i = y0 + 1;
      const float lx = lxs[ix], ly = lys[iy];
      const float w_nw = (1.f - lx) * (1.f - ly), w_ne = lx * (1.f - ly), w_sw = (1.f - lx) * ly, w_se = lx * ly;
      const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1i >= 0 && x1i < W, oky0 = y0 >= 0 && y0 < H, oky1 = y1i >= 0 && y1i < H;
      for (int c = 0; c < nch; ++c) {
        const float g = gpl[c * SS + pos];
        const size_t cb = (size_t)c * plane;
        if (okx0 && oky0) dm_acc_add(gf, cb + y0 * W + x0, g * w_nw, fixed != 0);
        if (okx1 && oky0) dm_acc_add(gf, cb + y0 * W + x1i, g * w_ne, fixed != 0);
        if (okx0 && oky1) dm_acc_add(gf, cb + y1i * W + x0, g * w_sw, fixed != 0);
        if (okx1 && oky1) dm_acc_add(gf, cb + y1i * W + x1i, g * w_se, fixed != 0);
      }
    }
    return;
  }
  const unsigned m_tw = TW > 1 ? 0xFFFFFFFFu / (unsigned)TW + 1u : 0u;
  for (int cell = tid; cell < TH * TW; cell += 256) {
    const int ty = TW > 1 ? (int)__umulhi((unsigned)cell, m_tw) : cell;      // cell * TW < 2^32 (H * W of the map < 2^31)
    const int tx = cell - ty * TW;
    const int gx = fx0 + tx, gy = fy0 + ty;
    // the two runs per axis: value gx - 1 (the sample's right / lower tap lands here) and value gx (its own cell)
    const int ca0 = cfirst[gx + 1], ca1 = clast[gx + 1], cb0 = cfirst[gx + 2], cb1 = clast[gx + 2];
    const int ra0 = rfirst[gy + 1], ra1 = rlast[gy + 1], rb0 = rfirst[gy + 2], rb1 = rlast[gy + 2];
    if ((ca1 <= ca0 && cb1 <= cb0) || (ra1 <= ra0 && rb1 <= rb0)) continue;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int i0 = rs == 0 ? ra0 : rb0, i1 = rs == 0 ? ra1 : rb1;
      for (int iy = i0; iy < i1; ++iy) {
        const float ly = lys[iy];
        const float wy = rs == 0 ? ly : 1.f - ly;
        const float* grow = gpl + iy * S;
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
          const int j0 = cs == 0 ? ca0 : cb0, j1 = cs == 0 ? ca1 : cb1;
          for (int ix = j0; ix < j1; ++ix) {
            const float lx = lxs[ix];
            const float w = (cs == 0 ? lx : 1.f - lx) * wy;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] += grow[c * SS + ix] * w;      // (planes past nch hold stale values: not flushed)
          }
        }
      }
    }
    const size_t o = (size_t)gy * W + gx;
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < nch && acc[c] != 0.f) dm_acc_add(gf, (size_t)c * plane + o, acc[c], fixed != 0);
  }
}


}  // namespace

extern "C" int dm_fx_to_float(long long* fx, long long n, float* out, int accumulate, int clear, dm_stream_t stream) {
  if (n < 0 || (n > 0 && (!fx || !out))) return DM_ERR_INVALID_ARG;
  if (n == 0) return DM_OK;
  DM_LAUNCH(fx_to_float_kernel, dim3((unsigned)dm_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, fx, n, out, accumulate,
            clear);
  return dm_check_launch();
}

static int conv2d_wgrad_impl(const float* dy, long long dy_batch_stride, int Cout, const float* x,
                             long long x_batch_stride, int Cs, int NB, int H, int W, int ksize, float* dw, int ldw,
                             int col_offset, float* db, int fx, dm_stream_t stream, float* scratch = nullptr,
                             long long scratch_floats = 0) {
  if (!dy || !x || !dw || Cout <= 0 || Cs <= 0 || NB <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3))
    return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  WgradArgs a;
  a.dy = dy; a.dy_bs = dy_batch_stride; a.x = x; a.x_bs = x_batch_stride; a.Cout = Cout; a.Cs = Cs;
  a.NB = NB; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W; a.dw = dw; a.ldw = ldw; a.coloff = col_offset;
  a.fx = fx; a.db = db; a.scratch = scratch; a.scratch_floats = scratch_floats;
  if (ksize == 3) {
    const int rc = launch_wgrad3_narrow(a, (hipStream_t)stream);
    if (rc < 0) return rc;
    if (rc == 1) launch_wgrad<3>(a, (hipStream_t)stream);
  } else {
    launch_wgrad<1>(a, (hipStream_t)stream);
  }
  return dm_check_launch();
}

extern "C" int dm_conv2d_wgrad(const float* dy, long long dy_batch_stride, int Cout, const float* x,
                               long long x_batch_stride, int Cs, int NB, int H, int W, int ksize, float* dw, int ldw,
                               int col_offset, float* db, dm_stream_t stream) {
  return conv2d_wgrad_impl(dy, dy_batch_stride, Cout, x, x_batch_stride, Cs, NB, H, W, ksize, dw, ldw, col_offset, db, 0,
                           stream);
}

extern "C" long long dm_conv2d_wgrad_scratch_floats(void) {
  // an upper bound for every shape: slabs x tile <= target workgroups x (K-waves x tile) = target x 16384, + the bias rows
  return (long long)wgrad_target_wgs() * (16384 + 256);
}

extern "C" int dm_conv2d_wgrad_slab(const float* dy, long long dy_batch_stride, int Cout, const float* x, long long x_batch_stride,
                                    int Cs, int NB, int H, int W, int ksize, float* dw, int ldw, int col_offset, float* db,
                                    float* scratch, long long scratch_floats, dm_stream_t stream) {
  if (!scratch || scratch_floats <= 0) return DM_ERR_INVALID_ARG;
  return conv2d_wgrad_impl(dy, dy_batch_stride, Cout, x, x_batch_stride, Cs, NB, H, W, ksize, dw, ldw, col_offset, db, 0, stream,
                           scratch, scratch_floats);
}

extern "C" int dm_conv2d_wgrad_fx(const float* dy, long long dy_batch_stride, int Cout, const float* x,
                                  long long x_batch_stride, int Cs, int NB, int H, int W, int ksize, long long* dw_fx,
                                  int ldw, int col_offset, long long* db_fx, dm_stream_t stream) {
  return conv2d_wgrad_impl(dy, dy_batch_stride, Cout, x, x_batch_stride, Cs, NB, H, W, ksize,
                           reinterpret_cast<float*>(dw_fx), ldw, col_offset, reinterpret_cast<float*>(db_fx), 1, stream);
}

extern "C" int dm_upsample2x_bilinear_bwd(const float* grad_out, const float* fwd_out_for_relu, int NC, int H, int W,
                                          int align_corners, float* grad_in, dm_stream_t stream) {
  if (!grad_out || !grad_in || NC < 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NC == 0) return DM_OK;
  if (!align_corners && H >= 2 && W >= 2 && (4 * H * W) % 4 == 0 && (size_t)16 * H * W <= 64 * 1024) {
    // LDS-staged gather: overwrites grad_in, every byte of grad_out (and of the ReLU mask) read once
    const int blocks = min(NC, 8 * dm_num_cus());
    DM_LAUNCH(upsample2x_bwd_lds_kernel, dim3(blocks), dim3(256), (size_t)16 * H * W, (hipStream_t)stream, grad_out,
              fwd_out_for_relu, NC, H, W, grad_in);
    return dm_check_launch();
  }
  if (!align_corners && H >= 2 && W >= 2) {
    // gather form overwrites grad_in (no zero-fill needed, no atomics)
    DM_LAUNCH(upsample2x_bwd_gather_kernel, dim3(grid_for((size_t)NC * H * W)), dim3(256), 0, (hipStream_t)stream, grad_out,
              fwd_out_for_relu, NC, H, W, grad_in);
    return dm_check_launch();
  }
  if (align_corners && H >= 2 && W >= 2 && (size_t)24 * H * W <= 96 * 1024) {
    // planes [2H][2W] + row sums [2H][W] (56 x 56 inputs: 75 KB).  Few planes (the 256 x 1 x 112 x 112 logit gradients): 1024
    // threads per workgroup, so that a CU with one plane still has 16 waves
    static bool attr_a[DM_MAX_DEVICES] = {false}, attr_b[DM_MAX_DEVICES] = {false};
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&upsample2x_bwd_ac_lds_kernel<1024>), 96 * 1024, attr_a) != DM_OK ||
        dm_ensure_lds_limit(reinterpret_cast<const void*>(&upsample2x_bwd_ac_lds_kernel<256>), 96 * 1024, attr_b) != DM_OK)
      return DM_ERR_LAUNCH;
    const int blocks = min(NC, 8 * dm_num_cus());
    if (NC <= 2 * dm_num_cus())
      DM_LAUNCH((upsample2x_bwd_ac_lds_kernel<1024>), dim3(blocks), dim3(1024), (size_t)24 * H * W, (hipStream_t)stream, grad_out,
                fwd_out_for_relu, NC, H, W, grad_in);
    else
      DM_LAUNCH((upsample2x_bwd_ac_lds_kernel<256>), dim3(blocks), dim3(256), (size_t)24 * H * W, (hipStream_t)stream, grad_out,
                fwd_out_for_relu, NC, H, W, grad_in);
    return dm_check_launch();
  }
  // scatter form (1-pixel inputs, planes beyond 64 KB with align_corners): accumulates, so clear the output first
  if (hipMemsetAsync(grad_in, 0, (size_t)NC * H * W * sizeof(float), (hipStream_t)stream) != hipSuccess) return DM_ERR_LAUNCH;
  DM_LAUNCH(upsample2x_bwd_kernel, dim3(grid_for((size_t)NC * 4 * H * W)), dim3(256), 0, (hipStream_t)stream, grad_out,
            fwd_out_for_relu, NC, H, W, align_corners, grad_in);
  return dm_check_launch();
}

static int point_sample_bwd_impl(const float* grad_out, int B, int C, int H, int W, const float* rois, int N, int S,
                                 float spatial_scale, float* grad_feat, int fx, dm_stream_t stream) {
  if (!grad_out || !rois || !grad_feat || B <= 0 || C <= 0 || H <= 0 || W <= 0 || N < 0 || S <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  constexpr int CT = 4;
  const size_t glds = sizeof(float) * ((size_t)CT * S * S + 4 * (size_t)S + 3 * ((size_t)W + 4) + 3 * ((size_t)H + 4));
  if (glds <= 64 * 1024 && S <= 128 && (long long)H * W * W < (1LL << 32)) {      // (2 S threads build the tables; cell * TW < 2^32)
    DM_LAUNCH(point_sample_bwd_gather_kernel<CT>, dim3((unsigned)dm_ceil_div(C, CT), (unsigned)N), dim3(256), glds,
              (hipStream_t)stream, grad_out, B, C, H, W, rois, N, S, spatial_scale, grad_feat, fx);
    return dm_check_launch();
  }
  const int lds_elems = 6144;                    // 48 KB of 64-bit accumulators
  DM_LAUNCH(point_sample_bwd_kernel<CT>, dim3((unsigned)dm_ceil_div(C, CT), (unsigned)N), dim3(256),
            (size_t)lds_elems * sizeof(unsigned long long), (hipStream_t)stream, grad_out, B, C, H, W, rois, N, S,
            spatial_scale, grad_feat, lds_elems, fx);
  return dm_check_launch();
}

extern "C" int dm_point_sample_bwd(const float* grad_out, int B, int C, int H, int W, const float* rois, int N, int S,
                                   float spatial_scale, float* grad_feat, dm_stream_t stream) {
  return point_sample_bwd_impl(grad_out, B, C, H, W, rois, N, S, spatial_scale, grad_feat, 0, stream);
}

extern "C" int dm_point_sample_bwd_fx(const float* grad_out, int B, int C, int H, int W, const float* rois, int N, int S,
                                      float spatial_scale, long long* grad_feat_fx, dm_stream_t stream) {
  return point_sample_bwd_impl(grad_out, B, C, H, W, rois, N, S, spatial_scale, reinterpret_cast<float*>(grad_feat_fx), 1,
                               stream);
}

static int class_logits_bwd_impl(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                                 int num_classes, const int64_t* labels, const float* grad_inst, const float* grad_det,
                                 float* grad_x, int accumulate_x, float* grad_w_inst, float* grad_b_inst,
                                 float* grad_w_det, float* grad_b_det, int fx, dm_stream_t stream, float* part = nullptr) {
  if (!x || !w_inst || !w_det || !labels || !grad_inst || !grad_det || !grad_x || !grad_w_inst || !grad_b_inst ||
      !grad_w_det || !grad_b_det)
    return DM_ERR_INVALID_ARG;
  if (N < 0 || C <= 0 || HW <= 0 || num_classes <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if ((HW & 3) == 0 && HW <= 1024 && ((((uintptr_t)x | (uintptr_t)grad_x | (uintptr_t)grad_inst | (uintptr_t)grad_det) & 15) == 0)) {
    constexpr int CPW = 4;
    if (HW <= 256)
      DM_LAUNCH((class_logits_bwd_wave_kernel<CPW, 1>), dim3((unsigned)dm_ceil_div(C, 4 * CPW), N), dim3(256), 0, (hipStream_t)stream, x, N, C, HW,
                w_inst, w_det, num_classes, labels, grad_inst, grad_det, grad_x, accumulate_x, grad_w_inst, grad_b_inst, grad_w_det,
                grad_b_det, fx, part);
    else
      DM_LAUNCH((class_logits_bwd_wave_kernel<CPW, 4>), dim3((unsigned)dm_ceil_div(C, 4 * CPW), N), dim3(256), 0, (hipStream_t)stream, x, N, C, HW,
                w_inst, w_det, num_classes, labels, grad_inst, grad_det, grad_x, accumulate_x, grad_w_inst, grad_b_inst, grad_w_det,
                grad_b_det, fx, part);
  } else {
    DM_LAUNCH(class_logits_bwd_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, w_inst, w_det, num_classes,
              labels, grad_inst, grad_det, grad_x, accumulate_x, grad_w_inst, grad_b_inst, grad_w_det, grad_b_det, fx, part);
  }
  int rc = dm_check_launch();
  if (rc != DM_OK || !part) return rc;
  DM_LAUNCH(class_logits_bwd_reduce_kernel, dim3((unsigned)dm_ceil_div(C, 16), (unsigned)num_classes), dim3(256), 0, (hipStream_t)stream,
            part, labels, N, C, num_classes, grad_w_inst, grad_b_inst, grad_w_det, grad_b_det);
  return dm_check_launch();
}

extern "C" int dm_class_logits_bwd(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                                   int num_classes, const int64_t* labels, const float* grad_inst, const float* grad_det,
                                   float* grad_x, int accumulate_x, float* grad_w_inst, float* grad_b_inst,
                                   float* grad_w_det, float* grad_b_det, dm_stream_t stream) {
  return class_logits_bwd_impl(x, N, C, HW, w_inst, w_det, num_classes, labels, grad_inst, grad_det, grad_x, accumulate_x,
                               grad_w_inst, grad_b_inst, grad_w_det, grad_b_det, 0, stream);
}

extern "C" long long dm_class_logits_bwd_scratch_floats(int N, int C) {
  return (N < 0 || C <= 0) ? -1 : (long long)N * C * 2 + (long long)N * 2;
}

extern "C" int dm_class_logits_bwd_slab(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                                        int num_classes, const int64_t* labels, const float* grad_inst, const float* grad_det,
                                        float* grad_x, int accumulate_x, float* grad_w_inst, float* grad_b_inst,
                                        float* grad_w_det, float* grad_b_det, float* scratch, long long scratch_floats,
                                        dm_stream_t stream) {
  if (!scratch || scratch_floats < dm_class_logits_bwd_scratch_floats(N, C) || num_classes > 65535) return DM_ERR_INVALID_ARG;
  return class_logits_bwd_impl(x, N, C, HW, w_inst, w_det, num_classes, labels, grad_inst, grad_det, grad_x, accumulate_x,
                               grad_w_inst, grad_b_inst, grad_w_det, grad_b_det, 0, stream, scratch);
}

extern "C" int dm_class_logits_bwd_fx(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                                      int num_classes, const int64_t* labels, const float* grad_inst,
                                      const float* grad_det, float* grad_x, int accumulate_x, long long* grad_w_inst_fx,
                                      long long* grad_b_inst_fx, long long* grad_w_det_fx, long long* grad_b_det_fx,
                                      dm_stream_t stream) {
  return class_logits_bwd_impl(x, N, C, HW, w_inst, w_det, num_classes, labels, grad_inst, grad_det, grad_x, accumulate_x,
                               reinterpret_cast<float*>(grad_w_inst_fx), reinterpret_cast<float*>(grad_b_inst_fx),
                               reinterpret_cast<float*>(grad_w_det_fx), reinterpret_cast<float*>(grad_b_det_fx), 1, stream);
}

extern "C" int dm_deform_im2col(const float* x, const float* offset, int NB, int C, int H, int W, int deform_groups,
                                float* col, dm_stream_t stream) {
  if (!x || !offset || !col || NB < 0 || C <= 0 || H <= 0 || W <= 0 || deform_groups <= 0 || C % deform_groups)
    return DM_ERR_INVALID_ARG;
  if (W < 2) return DM_ERR_UNSUPPORTED;      // the gather loads row pairs (as dm_deform_conv_fwd)
  if (NB == 0) return DM_OK;
  {
    // LDS-plane build: H*W a multiple of 4 and CT | C/deform_groups planes that fit 64 KB
    const int HW = H * W, cpg = C / deform_groups;
    if (HW % 4 == 0) {
      const dim3 block(256);
#define DM_IM2COL(CTV)                                                                                              \
  if (cpg % CTV == 0 && (size_t)CTV * HW * 4 <= 64 * 1024) {                                                        \
    DM_LAUNCH(deform_im2col_lds_kernel<CTV>, dim3((unsigned)(NB * deform_groups * (cpg / CTV))), block,              \
              (size_t)CTV * HW * 4, (hipStream_t)stream, x, offset, C, H, W, deform_groups, col);                   \
    return dm_check_launch();                                                                                       \
  }
      DM_IM2COL(32)
      DM_IM2COL(16)
      DM_IM2COL(8)
      DM_IM2COL(4)
#undef DM_IM2COL
    }
  }
  const int CT = 32;
  const int pblocks = dm_ceil_div(H * W, 256), chunks = dm_ceil_div(C / deform_groups, CT);
  DM_LAUNCH(deform_im2col_kernel, dim3((unsigned)(NB * deform_groups * 9 * chunks * pblocks)), dim3(256), 0,
            (hipStream_t)stream, x, offset, NB, C, H, W, deform_groups, col, CT);
  return dm_check_launch();
}

static int dcn_bwd_args_ok(const float* colgrad, const float* offset, int NB, int C, int H, int W, int deform_groups) {
  return colgrad && offset && NB >= 0 && C > 0 && H > 0 && W > 0 && deform_groups > 0 && C % deform_groups == 0;
}

extern "C" int dm_deform_coord_grad(const float* colgrad, const float* x, const float* offset, int NB, int C, int H, int W,
                                    int deform_groups, float* grad_offset, dm_stream_t stream) {
  if (!dcn_bwd_args_ok(colgrad, offset, NB, C, H, W, deform_groups) || !x || !grad_offset) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  {
    // LDS-staged build: 8 | channels per group, W >= 2, H * W < 65536; a band of BRows rows (<= 224 pixels: 8 items per
    // thread) with DCN_COORD_HALO rows staged either side
    const int cpg = C / deform_groups;
    const int BRows = max(1, min(H, 224 / W));
    const int XR = min(H, BRows + 2 * DCN_COORD_HALO);
    const int bands = dm_ceil_div(H, BRows);
    const int slots = 2 * (XR + 2) * (W + 2);
    const size_t lds = (size_t)2 * slots * 16;
    if (cpg % 8 == 0 && W >= 2 && W <= 224 && H * W < (1 << 20) && 9 * BRows * W <= 2048 && slots <= 2048 &&
        lds <= 64 * 1024 && (long long)NB * deform_groups * bands <= 0x7fffffffLL) {
      const dim3 grid((unsigned)(NB * deform_groups * bands));
      if (slots <= 2 * 512)
        DM_LAUNCH((dcn_coord_grad_lds_kernel<4, 2, 512>), grid, dim3(512), lds, (hipStream_t)stream, colgrad, x, offset, NB, C, H, W,
                  deform_groups, grad_offset, bands, BRows, XR);
      else
        DM_LAUNCH((dcn_coord_grad_lds_kernel<2, 2, 1024>), grid, dim3(1024), lds, (hipStream_t)stream, colgrad, x, offset, NB, C, H, W,
                  deform_groups, grad_offset, bands, BRows, XR);
      return dm_check_launch();
    }
  }
  const int pblocks = dm_ceil_div(H * W, 256);
  DM_LAUNCH(dcn_coord_grad_kernel, dim3((unsigned)(NB * deform_groups * 9 * pblocks)), dim3(256), 0, (hipStream_t)stream,
            colgrad, x, offset, NB, C, H, W, deform_groups, grad_offset);
  return dm_check_launch();
}

extern "C" int dm_deform_col2im(const float* colgrad, const float* offset, int NB, int C, int H, int W, int deform_groups,
                                float* grad_x, dm_stream_t stream) {
  if (!dcn_bwd_args_ok(colgrad, offset, NB, C, H, W, deform_groups) || !grad_x) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  hipStream_t st = (hipStream_t)stream;
  // channels per workgroup: LDS planes of CT x HW 64-bit accumulators, CT | C/deform_groups
  const int cpg = C / deform_groups;
  const size_t plane_b = (size_t)H * W * sizeof(unsigned long long);
  if (cpg % 8 == 0 && 8 * plane_b <= 64 * 1024) {
    DM_LAUNCH(dcn_col2im_lds_kernel<8>, dim3((unsigned)(NB * (C / 8))), dim3(256), 8 * plane_b, st, colgrad, offset, NB, C, H,
              W, deform_groups, grad_x);
  } else if (cpg % 4 == 0 && 4 * plane_b <= 64 * 1024) {
    DM_LAUNCH(dcn_col2im_lds_kernel<4>, dim3((unsigned)(NB * (C / 4))), dim3(256), 4 * plane_b, st, colgrad, offset, NB, C, H,
              W, deform_groups, grad_x);
  } else if (cpg % 2 == 0 && 2 * plane_b <= 64 * 1024) {
    // (56 x 56: two 25 KB planes, three workgroups per CU -- with 8 waves each 1.09 -> 1.02 ms at 256 RoIs x 64 channels; four
    // planes in 100 KB: 1.85 ms with 4 waves, 1.17 with 8; one plane: 1.29 -- a round-4 probe, docs/HISTORY.md)
    DM_LAUNCH((dcn_col2im_lds_kernel<2, 512>), dim3((unsigned)(NB * (C / 2))), dim3(512), 2 * plane_b, st, colgrad, offset, NB, C, H,
              W, deform_groups, grad_x);
  } else if (plane_b <= 64 * 1024) {
    DM_LAUNCH(dcn_col2im_lds_kernel<1>, dim3((unsigned)(NB * C)), dim3(256), plane_b, st, colgrad, offset, NB, C, H, W,
              deform_groups, grad_x);
  } else {
    return DM_ERR_UNSUPPORTED;
  }
  return dm_check_launch();
}

extern "C" int dm_deform_col2im_coord(const float* colgrad, const float* x, const float* offset, int NB, int C, int H,
                                      int W, int deform_groups, float* grad_x, float* grad_offset, dm_stream_t stream) {
  const int rc = dm_deform_coord_grad(colgrad, x, offset, NB, C, H, W, deform_groups, grad_offset, stream);
  if (rc != DM_OK) return rc;
  return dm_deform_col2im(colgrad, offset, NB, C, H, W, deform_groups, grad_x, stream);
}

extern "C" int dm_dcn_weight_permute(const float* src, float* dst, int Cout, int C, int to_colmajor, int accumulate,
                                     dm_stream_t stream) {
  if (!src || !dst || Cout <= 0 || C <= 0) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(dcn_weight_permute_kernel, dim3(grid_for((size_t)Cout * C * 9)), dim3(256), 0, (hipStream_t)stream, src, dst,
            Cout, C, to_colmajor, accumulate);
  return dm_check_launch();
}
