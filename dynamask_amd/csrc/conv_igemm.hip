// K5/K6/K16: dense stride-1 "same" convolution (1x1 / 3x3) and deconv 2x2/s2 as an
// implicit GEMM on the fp32-input MFMA units of gfx950 (v_mfma_f32_32x32x2_f32:
// exact fp32 fma chain, 1e-4 parity with the fp32 reference by construction).
//
// GEMM view (per launch):   Out[co, q] = bias[co] + sum_k  Wq[k, co] * X[k, q]
//   rows   co : output channels (MFMA "A" operand = packed weights, M side)
//   cols   q  : FLAT pixel index over the whole batch, q = n*H*W + y*W + x
//               (MFMA "B" operand = the im2col of the input, N side).  Flat
//               columns: no padding waste for 14x14 RoI maps (196 px), one kernel
//               for RoI tensors and whole FPN maps alike.
//   k         : (tap, input channel); the channel concatenation of up to 4
//               source tensors is walked in the K loop (torch.cat never
//               materialises).
// The 32x32 MFMA D layout puts the pixel on the lane, so every accumulator
// register stores 32 consecutive pixels of one output channel: 128-B coalesced
// NCHW stores.
//
// K ordering is ours to choose, so input channels are handled in QUADS: packed
// weights are [tap][channel quad][cout][4] and the LDS images are
//   A: [tap][quad][TM couts][4]      B: [quad][plane position][4]
// so ONE ds_read_b128 per operand feeds FOUR k-steps: MFMA j of a quad pair
// uses k = j (lanes 0-31, even quad) and k = 4+j (lanes 32-63, odd quad).
// Consecutive lanes read consecutive 16-B slots: conflict-free.
//
// Pipeline per workgroup (256 threads, 4 waves, 2x2 32x32 blocks per wave at
// the 128x128 tile): the NEXT chunk's global loads are issued into registers
// before the MFMAs of the current chunk and written to LDS after them
// (issue-early / write-late), so L2 latency hides under 144 MFMAs per wave.
#include <type_traits>

#include "common.h"

namespace {

struct ConvArgs {
  const float* src[DM_MAX_SOURCES];
  int src_c[DM_MAX_SOURCES];
  long long src_bs[DM_MAX_SOURCES];  // batch stride of each source, in floats
  int num_srcs;
  int NB, H, W, HW, Q;
  const float* wq;   // [taps][KQ][CoutP][4]
  const float* bias;
  int KQ, Cout, CoutP;
  int relu;   // flags: bit0 = ReLU, bit1 = accumulate into out
  float* out;
  int out_ch_total, out_ch_offset;
  int Wp, plane;  // 3x3: W+2, Rmax*Wp ; 1x1: unused, TN
  int MT;         // number of cout tiles
  int q_begin = 0;  // first flat pixel of this launch (Q is its end): a launch may cover a pixel range
  int shuffle;    // deconv 2x2/s2 epilogue: packed cout = phase*shuffle + co, stored at (2y+dy, 2x+dx)
  const float* mask = nullptr;   // same layout as out: outputs whose mask value is not > 0 are stored as 0 (a ReLU adjoint)
  int off32 = 0;     // every source spans < 4 GB: a pixel's offset inside a source fits 32 bits (the 1x1 builds' branch-free staging)
  // split-K (round 4; launches of few workgroups -- the <= 100-RoI inference calls): blockIdx.y = split, which walks
  // kchunks chunks of the K loop from chunk split * kchunks and stores its bare sums to ws + split * ws_stride
  // ([NB][Cout][HW]); conv_splitk_reduce_kernel adds the splits in index order and applies bias / accumulate / ReLU / mask
  float* ws = nullptr;
  long long ws_floats = 0, ws_stride = 0;
  int ksplit = 1, kchunks = 0;
  int want_split = 0;        // host side: the launcher's choice of splits for this launch (0: launch_conv_mp decides)
};

// CK input channels per chunk (multiple of 8); MAXPOS = plane positions per thread (3x3)
// TAIL (0 or 4): output channels TM .. TM+3 are computed with v_mfma_f32_4x4x1 (16 blocks of
// 4 x 4: the 4 lanes of a block hold the 4 tail couts as A and 4 pixels as B, so a lane
// accumulates the 4 tail couts of ITS pixel) from the same B fragments -- the 36-cout offset
// convs of the DCN layers then cost 32 + 4 rows of MFMA work instead of 64.
// The 3x3 128 x 128 build with one plane position per thread is held to 168 VGPRs (5 dwords of
// scratch): three workgroups per CU instead of two hide each other's barriers better
// (14 x 14, 3 full rounds: 0.916 -> 0.876 ms; 752 RoIs: 135 TFLOP/s).  The 1x1 builds with 32-channel
// chunks spill 35-55 dwords at 168 and lose a third of their rate; with 16-channel chunks (half the
// prefetch registers) they fit without scratch, and three per CU beats the longer chunk: the DCN
// column-gradient GEMMs 1.56 -> 1.20, 0.94 -> 0.78, 0.75 -> 0.67 ms, fusion 130->64 @56^2 0.56 -> 0.45 ms.
// The kernel body takes its place in the grid as arguments (bx of gx workgroups, split by): conv_igemm_kernel passes
// blockIdx / gridDim, conv_igemm_group_kernel (several independent convolutions in one launch) a range of its grid.
template <int KS, int WGM, int WGN, int WM, int WN, int CK, int MAXPOS, int TAIL = 0>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int bx, const int gx, const int by) {
  static_assert(TAIL == 0 || (TAIL == 4 && WGM == 1), "tail rows need a single cout tile");
  constexpr int TM = WGM * WM * 32;
  constexpr int TMA = TM + TAIL;              // rows of the LDS A image
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TAPS = KS * KS;
  constexpr int NQ = CK / 4;                      // channel quads per chunk (what a thread prefetches of a pixel)
  constexpr int NWC = NQ;                         // 16-byte words per chunk and (cout | pixel) in LDS and in the packed weights
  constexpr int A_F4 = TAPS * NWC * TMA;          // float4 slots of the A chunk
  constexpr int A_PER_T = (A_F4 + NT - 1) / NT;
  constexpr int B1_PER_T = (NQ * TN + NT - 1) / NT;   // 1x1: float4 slots per thread
  constexpr int BREG = (KS == 3) ? MAXPOS * NQ : B1_PER_T;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  dm_f32x4* ldsA = reinterpret_cast<dm_f32x4*>(lds);                 // [TAPS][NQ][TM]
  dm_f32x4* ldsB = reinterpret_cast<dm_f32x4*>(lds) + A_F4;          // [NWC][plane]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WGN;
  const int wave_n = wave % WGN;
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own
  // L2), so the MT cout tiles of one pixel tile -- which read the same input -- are placed
  // 8 apart in launch order: same XCD, back to back.  The ragged tail keeps the plain order.
  int m_tile, n_tile;
  {
    const int b = bx, grp = 8 * a.MT;
    const int full = (gx / grp) * grp;
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
  const int plane = a.plane;
  int k_skip = 0, k_left = 0x7fffffff;         // split-K: chunks to pass over, chunks to walk
  // the epilogue's destination and modes (locals: writing to the argument struct would copy all of it to scratch)
  float* e_out = a.out;
  const float* e_bias = a.bias;
  const float* e_mask = a.mask;
  int e_relu = a.relu, e_oct = a.out_ch_total, e_oco = a.out_ch_offset;
  if (a.ksplit > 1) {
    const int sp = by;
    e_out = a.ws + (size_t)sp * a.ws_stride;
    e_bias = nullptr; e_mask = nullptr; e_relu = 0;
    e_oct = a.Cout; e_oco = 0;
    k_skip = sp * a.kchunks; k_left = a.kchunks;
  }

  // ---- tile geometry (uniform) -------------------------------------------
  const int qlast = min(q0 + TN, a.Q) - 1;
  const int n0 = q0 / HW;
  const int n1 = qlast / HW;
  const int y00 = (q0 - n0 * HW) / W;
  const int y1l = (qlast - n1 * HW) / W;
  const int rows0 = (n1 > n0) ? (H - y00) : (y1l - y00 + 1);
  const int Wp = a.Wp;

  // ---- per-lane column bookkeeping ----------------------------------------
  int lane_base[WN];   // LDS float4 index of tap (0,0) of this lane's pixel (+ odd-quad offset)
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

  // ---- B staging assignment (fixed for the tile) ---------------------------
  //  3x3: thread -> up to MAXPOS plane positions (all CK channels of each)
  //  1x1: thread -> column j = tid % TN, quads (tid / TN) + i * (NT / TN)
  int st_pix[(KS == 3) ? MAXPOS : 1], st_n[(KS == 3) ? MAXPOS : 1];
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
  } else {
    int q = q0 + (tid % TN);
    const bool ok = q < a.Q;
    q = min(q, a.Q - 1);
    const int n = q / HW;
    st_pix[0] = q - n * HW;
    st_n[0] = ok ? n : -1;
  }

  dm_f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  dm_f32x4 acct[WN];      // tail couts TM .. TM+3 of this lane's pixel (lanes >= 32: the other K half)
#pragma unroll
  for (int j = 0; j < WN; ++j) acct[j] = dm_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- chunk iterator over (source, channel offset) ------------------------
  int cs = 0, cc0 = 0, ckq = 0;   // current source, channel offset in it, global quad index of the chunk
  int curC = a.src_c[0];          // channels of the current source (kept in a register: a.src_c[cs] is a scalar load from the
                                  // argument block, and its s_waitcnt lgkmcnt(0) in the K loop also waits for the LDS)
  auto chunk_valid = [&]() { return cs < a.num_srcs && k_left > 0; };
  auto chunk_advance = [&]() {
    const int ckv = min(CK, curC - cc0);
    ckq += ((ckv + 7) / 8) * 2;
    cc0 += CK;
    if (cc0 >= curC) {
      cs++;
      cc0 = 0;
      curC = cs < a.num_srcs ? a.src_c[cs] : 0;
    }
  };

  dm_f32x4 ra[A_PER_T];
  dm_f32x4 rb[BREG];
  unsigned a_off[A_PER_T];          // float offset of A slot i inside a chunk's weights (fixed per thread)
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int idx = tid + i * NT;
    const int m = idx % TMA, tq = idx / TMA;
    a_off[i] = (unsigned)((((size_t)(tq / NWC) * a.KQ + tq % NWC) * a.CoutP + min(m0 + m, a.CoutP - 1)) * 4);      // (rows past CoutP: the last row)
  }
  constexpr int NPOS = (KS == 3) ? MAXPOS : 1;
  size_t b_off[NPOS];               // float offset of this thread's pixel(s) inside the current source
  int b_src = -1;
  unsigned b_off32 = 0;             // 1x1 fast path: the same offset in 32 bits
  const float* b_srcp = nullptr;    // 1x1 fast path: base of the current source

  // issue the global loads of the chunk at (cs, cc0, ckq) into registers
  auto prefetch = [&]() {
    const int ckv = min(CK, curC - cc0);
    const int nq = ((ckv + 7) / 8) * 2;    // quads of this chunk present in the packed weights
    // the chunk's part of an address is uniform (ckq, cc0); the thread's part is fixed for the K loop
    // (per source for B) and kept in a register: no 64-bit multiplies per load next to the MFMAs
    const float* abase = a.wq + (size_t)ckq * a.CoutP * 4;
    if (KS == 1 && A_F4 % NT == 0 && NQ * TN % NT == 0 && a.off32 && ckv == CK) {
      // Full chunk of a 1x1 build: nothing is predicated.  (The general path below guards every dword -- pixel inside the
      // matrix, channel inside the source, quad inside the chunk -- with an exec-mask branch and a 64-bit vector address:
      // ~150 instructions per 16 MFMAs in the 64-cout build, 4.8 VALU + 5.6 SALU per MFMA on the 576 -> 64 GEMM
      // (profiles/r03_sq_pmc.txt) against 1.65 + 0.97 in the 3x3 build.)  A pixel past the matrix stages the pixel the
      // thread was clamped to -- its output column is never stored; rows past CoutP repeat the last row.  Addresses are a
      // uniform row base (SALU) plus the thread's 32-bit pixel offset.
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) ra[i] = *reinterpret_cast<const dm_f32x4*>(abase + a_off[i]);
      if (b_src != cs) {
        b_off32 = (unsigned)((size_t)max(st_n[0], 0) * (size_t)a.src_bs[cs] + st_pix[0]);
        b_srcp = a.src[cs];
        b_src = cs;
        b_off[0] = b_off32;
      }
      const float* rp = b_srcp + (size_t)cc0 * HW;
      const int qw = (TN % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / TN) : tid / TN;      // (uniform per wave from 64 columns on)
#pragma unroll
      for (int i = 0; i < B1_PER_T; ++i) {
        const float* rq = rp + (size_t)((qw + i * (NT / TN)) * 4) * HW;
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[i][e] = rq[(size_t)e * HW + b_off32];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < A_F4) {
        const int m = idx % TMA;
        const int qd = (idx / TMA) % NWC;
        if (qd < nq && m0 + m < a.CoutP) v = *reinterpret_cast<const dm_f32x4*>(abase + a_off[i]);
      }
      ra[i] = v;
    }
    if (b_src != cs) {
      const size_t bs = (size_t)a.src_bs[cs];
#pragma unroll
      for (int k = 0; k < NPOS; ++k) b_off[k] = (size_t)max(st_n[k], 0) * bs + st_pix[k];
      b_src = cs;
    }
    const float* sp = a.src[cs] + (size_t)cc0 * HW;
    if (KS == 3 && ckv == CK) {
      // full chunk: ONE guard per plane position (its pixel exists or it is padding) instead of one per dword
#pragma unroll
      for (int k = 0; k < MAXPOS; ++k) {
        const float* gp = sp + b_off[k];
        if (st_n[k] >= 0) {
#pragma unroll
          for (int qd = 0; qd < NQ; ++qd)
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[k * NQ + qd][e] = gp[(size_t)(qd * 4 + e) * HW];
        } else {
#pragma unroll
          for (int qd = 0; qd < NQ; ++qd) rb[k * NQ + qd] = dm_f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    } else if (KS == 3) {
#pragma unroll
      for (int k = 0; k < MAXPOS; ++k) {
        const bool ok = st_n[k] >= 0;
        const float* gp = sp + b_off[k];
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
          dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ci = qd * 4 + e;
            if (ok && ci < ckv) v[e] = gp[(size_t)ci * HW];
          }
          rb[k * NQ + qd] = v;
        }
      }
    } else {
      const bool ok = st_n[0] >= 0;
      const float* gp = sp + b_off[0];
#pragma unroll
      for (int i = 0; i < B1_PER_T; ++i) {
        const int qd = tid / TN + i * (NT / TN);
        dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ci = qd * 4 + e;
          if (ok && qd < NQ && ci < ckv) v[e] = gp[(size_t)ci * HW];
        }
        rb[i] = v;
      }
    }
  };

  auto commit = [&]() {   // registers -> LDS
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int idx = tid + i * NT;
      if (idx < A_F4) ldsA[idx] = ra[i];
    }
    if (KS == 3) {
#pragma unroll
      for (int k = 0; k < MAXPOS; ++k) {
        const int pos = tid + k * NT;
        if (pos < plane) {
#pragma unroll
          for (int qd = 0; qd < NQ; ++qd) ldsB[qd * plane + pos] = rb[k * NQ + qd];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < B1_PER_T; ++i) {
        const int qd = tid / TN + i * (NT / TN);
        if (qd < NQ) ldsB[qd * plane + (tid % TN)] = rb[i];
      }
    }
  };

  for (; k_skip > 0 && cs < a.num_srcs; --k_skip) chunk_advance();      // (split-K: this split's first chunk)
  if (chunk_valid()) prefetch();
  while (chunk_valid()) {
    const int ckv_cur = min(CK, curC - cc0);
    const int ngroups = (ckv_cur + 7) / 8;   // quad pairs holding live channels
    commit();
    __syncthreads();
    chunk_advance();
    --k_left;
    if (chunk_valid()) prefetch();

    // ---- MFMA over the committed chunk ---------------------------------------
    // Operand fragments are double-buffered in registers: the ds_reads of step s+1 are
    // issued before the 4 x WM x WN MFMAs of step s, so LDS latency never stalls the
    // matrix pipe (the compiler alone re-uses one register set and issues them late).
    {
      constexpr int NG = NQ / 2;
      constexpr int STEPS = TAPS * NG;
      auto load_frag = [&](int st, dm_f32x4* av, dm_f32x4* bv) {
        const int tap = st / NG, g = st - tap * NG;
        const int tapoff = (KS == 3) ? ((tap / 3) * Wp + (tap % 3)) : 0;
#pragma unroll
        for (int i = 0; i < WM; ++i) av[i] = ldsA[(tap * NQ + 2 * g + hi) * TMA + (wave_m * WM + i) * 32 + l31];
        if (TAIL) av[WM] = ldsA[(tap * NQ + 2 * g + hi) * TMA + TM + (lane & 3)];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = ldsB[(2 * g) * plane + lane_base[j] + tapoff];
      };
      dm_f32x4 av[2][WM + (TAIL ? 1 : 0)], bv[2][WN];
      load_frag(0, av[0], bv[0]);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int cur = st & 1;
        // quad pairs beyond the chunk's live channels hold zeros (1x1, ragged sources): skip them
        const bool live = (NG == 1) || ((st % NG) < ngroups);
        if (st + 1 < STEPS) load_frag(st + 1, av[cur ^ 1], bv[cur ^ 1]);
        if (live) {
          // (s_setprio 1 around this cluster: +4 % in tools/micro/mfma_mix.hip, -3.5 % on the headline -- with three workgroups
          // per CU the prioritised waves starve the ones staging the next chunk)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i][e], bv[cur][j][e], acc[i][j], 0, 0, 0);
          if (TAIL) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int j = 0; j < WN; ++j)
                acct[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[cur][WM][e], bv[cur][j][e], acct[j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias + ReLU, 32 consecutive pixels per register ------------
  // Address arithmetic is hoisted: one 64-bit base per pixel column, one per-lane channel offset, the
  // rest is uniform (SALU); the store modes are uniform too and pick one of three straight-line
  // variants.  (Written naively -- the full index expression per store, mode tests inside -- the
  // epilogue was 3700 VALU instructions per wave, as many as the 32 chunks of the main loop together,
  // and VALU work is not free next to MFMAs: tools/micro/mfma_mix.hip.)
  if (a.shuffle == 0) {
    float* pj[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j)
      pj[j] = e_out + ((size_t)col_n[j] * e_oct + e_oco) * HW + col_p[j];
    const int co_lane = m0 + wave_m * WM * 32 + 4 * hi;
    const size_t off_lane = (size_t)co_lane * HW;
    const bool relu = e_relu & 1;
    // all bias values first: a load inside the store loop puts an s_waitcnt vmcnt(0) -- which also
    // waits for the stores issued so far -- in front of every output row
    float bias_r[WM][16];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[i][r] = 0.f;
    if (e_bias) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[i][r] = e_bias[min(co_lane + i * 32 + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
      // land them all here: otherwise the compiler waits for each value at its use, with counts that
      // include the stores issued meanwhile
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bias_r[i][r]));
    }
    const ptrdiff_t mask_off = e_mask ? e_mask - e_out : 0;
    // full: the workgroup's tile lies inside the output (every cout row and every pixel column exists) -- uniform, and then
    // no store is predicated: the row and column tests cost an exec-mask sequence each, 600 scalar instructions per wave in
    // front of the 64 stores of a K = 64 GEMM (SQ counters: 4.8 SALU per MFMA on 64 -> 576 @56^2)
    // (1x1 builds only: in the 3x3 kernel the second copy of the epilogue cost 0.5 % of the headline, its K = 2304 loop hides the tests)
    const bool full = KS == 1 && m0 + TM <= a.Cout && q0 + TN <= a.Q;
    auto store_all = [&](auto acc_mode, auto nt_mode, auto mask_mode, auto full_mode) {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = i * 32 + (r & 3) + 8 * (r >> 2);       // compile-time part of the channel
          if (decltype(full_mode)::value || co_lane + k < a.Cout) {
            const float b = bias_r[i][r];
            const size_t o = off_lane + (size_t)k * HW;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              if (decltype(full_mode)::value || col_ok[j]) {
                float* op = pj[j] + o;
                float v = acc[i][j][r] + b;
                if (decltype(acc_mode)::value) v += *op;          // accumulate into the destination (gradient sums)
                if (relu) v = fmaxf(v, 0.f);
                if (decltype(mask_mode)::value) v = (op[mask_off] > 0.f) ? v : 0.f;   // the producer's ReLU, looking backward
                if (decltype(nt_mode)::value) __builtin_nontemporal_store(v, op);   // output larger than the Infinity Cache: stream it
                else *op = v;
              }
            }
          }
        }
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (KS == 1 && full) {
      if (e_mask) {
        if (e_relu & 2) store_all(T{}, F{}, T{}, T{});
        else store_all(F{}, F{}, T{}, T{});
      } else if (e_relu & 2) store_all(T{}, F{}, F{}, T{});
      else if (e_relu & 4) store_all(F{}, T{}, F{}, T{});
      else store_all(F{}, F{}, F{}, T{});
    } else if (e_mask) {
      if (e_relu & 2) store_all(T{}, F{}, T{}, F{});
      else store_all(F{}, F{}, T{}, F{});
    } else if (e_relu & 2) store_all(T{}, F{}, F{}, F{});
    else if (e_relu & 4) store_all(F{}, T{}, F{}, F{});
    else store_all(F{}, F{}, F{}, F{});
  } else {
    // deconv 2x2/s2: packed cout = phase * shuffle + oc, stored at (2y+dy, 2x+dx)
    int sy[WN], sx[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      sy[j] = col_p[j] / W;
      sx[j] = col_p[j] - sy[j] * W;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + (wave_m * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co < a.Cout) {
          const int phase = co / a.shuffle;
          const int oc = co - phase * a.shuffle;
          const int dy = phase >> 1, dx = phase & 1;
          const float b = e_bias ? e_bias[oc] : 0.f;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            if (col_ok[j]) {
              float v = acc[i][j][r] + b;
              if (e_relu & 1) v = fmaxf(v, 0.f);
              e_out[(((size_t)col_n[j] * a.shuffle + oc) * (2 * H) + 2 * sy[j] + dy) * (2 * W) + 2 * sx[j] + dx] = v;
            }
          }
        }
      }
    }
  }
  if (TAIL) {
    // the two lane halves hold the two K halves of the same pixels
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acct[j][i] + __shfl_xor(acct[j][i], 32, 64);
        const int co = m0 + TM + i;
        if (hi == 0 && co < a.Cout && col_ok[j]) {
          if (e_bias) v += e_bias[co];
          float* op = e_out + ((size_t)col_n[j] * e_oct + e_oco + co) * HW + col_p[j];
          if (e_relu & 2) v += *op;
          if (e_relu & 1) v = fmaxf(v, 0.f);
          if (e_mask && !(e_mask[op - e_out] > 0.f)) v = 0.f;
          *op = v;
        }
      }
    }
  }
}

template <int KS, int WGM, int WGN, int WM, int WN, int CK, int MAXPOS, int TAIL = 0>
__global__ __launch_bounds__(WGM* WGN * 64, (KS == 3 && WGM == 2 && WGN == 2 && WM == 2 && WN == 2 && MAXPOS == 1) || (KS == 1 && CK == 16) ? 3 : 1) void conv_igemm_kernel(ConvArgs a) {
  conv_igemm_body<KS, WGM, WGN, WM, WN, CK, MAXPOS, TAIL>(a, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// Up to three INDEPENDENT convolutions in one launch (round 6: the three FPN-wide semantic_transform_in 1x1 convolutions of
// the SFM stages, dynamask_head.py:104 -- P4 256->256, P3 256->128, P2 256->64: 66 / 132 / 525 workgroups that each walk
// a K = 256 loop, one launch after the other in front of everything else): workgroups [end[i-1], end[i]) run problem i
// with the body above, i.e. the same code on the same operands as a launch of its own.
struct ConvGroup {
  ConvArgs a[3];
  int end[3];
};
template <int KS, int WGM, int WGN, int WM, int WN, int CK, int MAXPOS>
__global__ __launch_bounds__(WGM* WGN * 64, (KS == 1 && CK == 16) ? 3 : 1) void conv_igemm_group_kernel(ConvGroup g) {
  const int b = (int)blockIdx.x;
  if (b < g.end[0]) conv_igemm_body<KS, WGM, WGN, WM, WN, CK, MAXPOS, 0>(g.a[0], b, g.end[0], 0);
  else if (b < g.end[1]) conv_igemm_body<KS, WGM, WGN, WM, WN, CK, MAXPOS, 0>(g.a[1], b - g.end[0], g.end[1] - g.end[0], 0);
  else conv_igemm_body<KS, WGM, WGN, WM, WN, CK, MAXPOS, 0>(g.a[2], b - g.end[1], g.end[2] - g.end[1], 0);
}

// Packed weight layout [tap][KQ][CoutP][4]: KQ quads = sum over sources of
// roundup(Cs, 8)/4, channels of a source padded with zero rows to a multiple of 8.
struct PackArgs {
  const float* w;
  float* wq;
  int Cout, Cin, kk, flip;   // dims of the OIHW tensor
  int rows, cols, colsP;     // rows = reduction channels, cols = produced channels
  int nsrc;
  int src_c[DM_MAX_SOURCES];
  int KQ;
  int deconv;                // w is [Cin][Cout][2][2]; packed col = phase*Cout + co
};

__device__ __forceinline__ void pack_weight_body(const PackArgs& p, int ld, int c0) {
  // ld / c0: row length and first column of the [Cout][ld][k][k] tensor the packed [Cout][Cin] window sits in
  // (ld = Cin, c0 = 0: the whole tensor; a window = the input-channel slice of one concat source)
  const long long total = (long long)p.kk * p.KQ * p.colsP * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3);
    const int col = (int)((idx >> 2) % p.colsP);
    const int kq = (int)((idx >> 2) / p.colsP % p.KQ);
    const int tap = (int)((idx >> 2) / ((long long)p.colsP * p.KQ));
    // padded channel index -> (source, local channel) -> dense reduction row
    int pc = kq * 4 + e, row = -1, base = 0;
    for (int s = 0; s < p.nsrc; ++s) {
      const int padded = (p.src_c[s] + 7) / 8 * 8;
      if (pc < padded) {
        if (pc < p.src_c[s]) row = base + pc;
        break;
      }
      pc -= padded;
      base += p.src_c[s];
    }
    float v = 0.f;
    if (row >= 0 && col < p.cols) {
      if (p.deconv) {
        const int phase = col / p.Cout, co = col - phase * p.Cout;
        v = p.w[((size_t)row * p.Cout + co) * 4 + phase];
      } else if (!p.flip) {
        v = p.w[((size_t)col * ld + c0 + row) * p.kk + tap];               // col = co, row = ci
      } else {
        v = p.w[((size_t)row * ld + c0 + col) * p.kk + (p.kk - 1 - tap)];    // row = co, col = ci
      }
    }
    p.wq[idx] = v;
  }
}

__global__ void pack_weight_kernel(PackArgs p) { pack_weight_body(p, p.Cin, 0); }

// Every pack of a training step in ONE launch (blockIdx.y = job): ~45 weight tensors change with every optimizer
// step, and a launch per tensor cost the host 1.3 ms of the 3.7 ms it needs to issue a forward pass.
__global__ void pack_weight_batch_kernel(const dm_pack_job* __restrict__ jobs) {
  const dm_pack_job j = jobs[blockIdx.y];
  PackArgs p;
  p.w = j.w; p.wq = j.w_packed; p.Cout = j.Cout; p.Cin = j.Cin; p.kk = j.ksize * j.ksize; p.flip = (j.transpose_flip & 1) ? 1 : 0;
  p.rows = p.flip ? j.Cout : j.Cin;
  p.cols = p.flip ? j.Cin : j.Cout;
  p.colsP = (p.cols + 31) / 32 * 32;
  p.nsrc = j.num_srcs;
  p.KQ = 0;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) {
    p.src_c[s] = s < j.num_srcs ? j.src_channels[s] : 0;
    if (s < j.num_srcs) p.KQ += (j.src_channels[s] + 7) / 8 * 2;
  }
  p.deconv = 0;
  pack_weight_body(p, j.ld, j.c0);
}

int packed_quads(int nsrc, const int* src_c) {
  int kq = 0;
  for (int s = 0; s < nsrc; ++s) kq += (src_c[s] + 7) / 8 * 2;
  return kq;
}

// out = epilogue(sum over the splits, in index order): the conv kernel's own epilogue arithmetic (bias, accumulate, ReLU,
// mask -- in that order) on the bare sums of a split-K launch.  ws: [splits][NB][Cout][HW]; out / mask: [NB][out_ch_total][HW].
// (round 5: four outputs per thread and 32-bit index arithmetic where HW % 4 == 0 and everything is 16-byte aligned --
// one thread per output with two 64-bit divisions took 27 us for the 2.5 M outputs of a 50-RoI convolution, 1.5 TB/s.)
template <bool V4>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long stride, int NB,
                                                                int Cout, int HW, const float* __restrict__ bias, int flags,
                                                                float* __restrict__ out, int out_ch_total, int out_ch_offset,
                                                                const float* __restrict__ mask) {
  if (V4) {
    // total = NB * Cout * HW < 2^31 (launcher); a quad never straddles a channel plane (HW % 4 == 0)
    const unsigned total4 = (unsigned)(((long long)NB * Cout * HW) >> 2), hw4 = (unsigned)HW >> 2;
    const float4* ws4 = reinterpret_cast<const float4*>(ws);
    const size_t stride4 = (size_t)(stride >> 2);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += gridDim.x * blockDim.x) {
      const unsigned nc = e / hw4, p4 = e - nc * hw4;
      const unsigned n = nc / (unsigned)Cout, co = nc - n * (unsigned)Cout;
      float4 v = ws4[e];
      for (int s = 1; s < splits; ++s) {
        const float4 w = ws4[(size_t)s * stride4 + e];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      if (bias) {
        const float bv = bias[co];
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      }
      const size_t o4 = ((size_t)n * out_ch_total + out_ch_offset + co) * hw4 + p4;
      float4* op = reinterpret_cast<float4*>(out) + o4;
      if (flags & 2) {
        const float4 w = *op;
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      if (flags & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (mask) {
        const float4 m = reinterpret_cast<const float4*>(mask)[o4];
        v.x = (m.x > 0.f) ? v.x : 0.f; v.y = (m.y > 0.f) ? v.y : 0.f; v.z = (m.z > 0.f) ? v.z : 0.f; v.w = (m.w > 0.f) ? v.w : 0.f;
      }
      *op = v;
    }
    return;
  }
  const long long total = (long long)NB * Cout * HW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long nc = e / HW;
    const int p = (int)(e - nc * HW);
    const int n = (int)(nc / Cout), co = (int)(nc - (long long)n * Cout);
    float v = ws[e];
    for (int s = 1; s < splits; ++s) v += ws[(size_t)s * stride + e];
    if (bias) v += bias[co];
    const size_t o = ((size_t)n * out_ch_total + out_ch_offset + co) * HW + p;
    if (flags & 2) v += out[o];
    if (flags & 1) v = fmaxf(v, 0.f);
    if (mask) v = (mask[o] > 0.f) ? v : 0.f;
    out[o] = v;
  }
}

// How many K splits pay for a launch that leaves most of the chip idle (1: none).  A launch of one round walks its chunks
// one after the other at ~2.1 us (3x3) / ~1.5 us (1x1) per chunk whatever the RoI count (tools/small_n.py: conv14 64-71 us
// for 32 chunks, fuse14 50 us for 33); S splits save (1 - 1/S) of that and cost a second launch (~5 us) that reads S and
// writes one copy of the output at ~4 TB/s; at least 8 us of net gain.
static int conv_split_choice(int chunks, bool k3, int Smax, long long out_floats) {
  const double chain_us = chunks * (k3 ? 2.1 : 1.5), out_mb = (double)out_floats * 4e-6;
  double best = 8.0;
  int S = 1;
  for (int c = 2; c <= Smax; ++c) {
    const double g = chain_us * (1.0 - 1.0 / c) - (5.0 + (c + 1) * out_mb * 0.25);
    if (g > best) { best = g; S = c; }
  }
  return S;
}

template <int KS, int WGM, int WGN, int WM, int WN, int CK, int MAXPOS, int TAIL = 0>
int launch_conv_mp(ConvArgs& a, hipStream_t st) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  const int NTiles = dm_ceil_div(a.Q - a.q_begin, TN);
  constexpr int NWC = CK / 4;
  const size_t lds_bytes = 16 * ((size_t)KS * KS * NWC * (TM + TAIL) + (size_t)NWC * a.plane);
  if (lds_bytes > 64 * 1024) return DM_ERR_UNSUPPORTED;
  // ---- split-K for launches that leave most of the chip idle (a caller-provided workspace, one launch per call)
  a.ksplit = 1;
  if (a.ws && a.shuffle == 0 && a.q_begin == 0) {
    constexpr int CKS = CK;
    int chunks = 0;
    for (int s_ = 0; s_ < a.num_srcs; ++s_) chunks += dm_ceil_div(a.src_c[s_], CKS);
    const int wgs = a.MT * NTiles, cus = dm_num_cus();
    const long long per = (long long)a.NB * a.Cout * a.HW;
    int Smax = a.want_split > 0 ? min(a.want_split, chunks / 4) : min(min(8, (3 * cus) / max(wgs, 1)), chunks / 4);
    if (per > 0) Smax = (int)min((long long)Smax, a.ws_floats / per);
    // worth it?  A launch of one round walks its chunks one after the other at ~2.1 us (3x3) / ~1.5 us (1x1) per chunk
    // whatever the RoI count (tools/small_n.py: conv14 64-71 us for 32 chunks, fuse14 50 us for 33); S splits save
    // (1 - 1/S) of that and cost a second launch (~5 us) that reads S and writes one copy of the output at ~4 TB/s.
    // (Without this test the 1x1 convolutions of the 28 x 28 / 56 x 56 stages split too: 17.7 reduce launches of 15.7 us
    // each per 100-detection call, more than the splits saved.)
    const int S = conv_split_choice(chunks, KS == 3, Smax, per);
    if (S >= 2) {
      a.kchunks = dm_ceil_div(chunks, S);
      a.ksplit = dm_ceil_div(chunks, a.kchunks);
      a.ws_stride = per;
    }
  }
  if (a.ksplit > 1) {
    const float* bias = a.bias;
    const float* mask = a.mask;
    float* out = a.out;
    const int flags = a.relu, oct = a.out_ch_total, oco = a.out_ch_offset;
    DM_LAUNCH((conv_igemm_kernel<KS, WGM, WGN, WM, WN, CK, MAXPOS, TAIL>), dim3(a.MT * NTiles, a.ksplit), dim3(NT), lds_bytes, st, a);
    int rc = dm_check_launch();
    if (rc != DM_OK) return rc;
    const long long total = a.ws_stride;
    const bool v4 = (a.HW & 3) == 0 && (total & 3) == 0 && total < 0x7fffffffLL &&
                    ((((uintptr_t)a.ws) | ((uintptr_t)out) | ((uintptr_t)mask)) & 15) == 0;
    if (v4)
      DM_LAUNCH(conv_splitk_reduce_kernel<true>, dim3((unsigned)min((long long)4096, (total / 4 + 255) / 256)), dim3(256), 0, st, a.ws,
                a.ksplit, a.ws_stride, a.NB, a.Cout, a.HW, bias, flags, out, oct, oco, mask);
    else
      DM_LAUNCH(conv_splitk_reduce_kernel<false>, dim3((unsigned)min((long long)4096, (total + 255) / 256)), dim3(256), 0, st, a.ws,
                a.ksplit, a.ws_stride, a.NB, a.Cout, a.HW, bias, flags, out, oct, oco, mask);
    return dm_check_launch();
  }
  DM_LAUNCH((conv_igemm_kernel<KS, WGM, WGN, WM, WN, CK, MAXPOS, TAIL>), dim3(a.MT * NTiles), dim3(NT), lds_bytes, st, a);
  return dm_check_launch();
}

template <int KS, int WGM, int WGN, int WM, int WN, int CK, int TAIL = 0>
int launch_conv(ConvArgs& a, hipStream_t st) {
  constexpr int TM = WGM * WM * 32;
  constexpr int TN = WGN * WN * 32;
  constexpr int NT = WGM * WGN * 64;
  a.MT = TAIL ? 1 : dm_ceil_div(a.CoutP, TM);
  if (KS == 3) {
    a.Wp = a.W + 2;
    const int nsegmax = dm_ceil_div(TN - 1, a.HW) + 1;
    const int rmax = dm_ceil_div(TN - 1, a.W) + 1 + 2 * nsegmax;
    a.plane = rmax * a.Wp;
    if (a.plane <= NT) return launch_conv_mp<KS, WGM, WGN, WM, WN, CK, 1, TAIL>(a, st);
    if (a.plane <= 2 * NT) return launch_conv_mp<KS, WGM, WGN, WM, WN, CK, 2, TAIL>(a, st);
    if (a.plane <= 4 * NT) return launch_conv_mp<KS, WGM, WGN, WM, WN, CK, 4, TAIL>(a, st);
    return DM_ERR_UNSUPPORTED;
  }
  a.Wp = 0;
  a.plane = TN;
  return launch_conv_mp<KS, WGM, WGN, WM, WN, CK, 1, TAIL>(a, st);
}

int run_pack(PackArgs& p, hipStream_t st) {
  const long long total = (long long)p.kk * p.KQ * p.colsP * 4;
  const int blocks = (int)min((long long)dm_ceil_div(total, 256), 4096LL);
  DM_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, p);
  return dm_check_launch();
}

}  // namespace

extern "C" int dm_conv_packed_cout(int Cout) { return (Cout + 31) / 32 * 32; }

extern "C" long long dm_conv_packed_floats(int Cout, int ksize, int num_srcs, const int* src_channels) {
  if (Cout <= 0 || ksize <= 0 || num_srcs < 1 || num_srcs > DM_MAX_SOURCES || !src_channels) return -1;
  return (long long)ksize * ksize * packed_quads(num_srcs, src_channels) * dm_conv_packed_cout(Cout) * 4;
}

extern "C" int dm_conv_pack_weight(const float* w_oihw, int Cout, int Cin, int ksize, int transpose_flip,
                                   int num_srcs, const int* src_channels, float* w_packed, dm_stream_t stream) {
  if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return DM_ERR_INVALID_ARG;
  if (num_srcs < 1 || num_srcs > DM_MAX_SOURCES || !src_channels) return DM_ERR_INVALID_ARG;
  PackArgs p;
  p.w = w_oihw; p.wq = w_packed; p.Cout = Cout; p.Cin = Cin; p.kk = ksize * ksize; p.flip = transpose_flip ? 1 : 0;
  p.rows = transpose_flip ? Cout : Cin;
  p.cols = transpose_flip ? Cin : Cout;
  p.colsP = dm_conv_packed_cout(p.cols);
  p.nsrc = num_srcs;
  int sum = 0;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) {
    p.src_c[s] = s < num_srcs ? src_channels[s] : 0;
    if (s < num_srcs && src_channels[s] <= 0) return DM_ERR_INVALID_ARG;
    sum += p.src_c[s];
  }
  if (sum != p.rows) return DM_ERR_INVALID_ARG;
  p.KQ = packed_quads(num_srcs, src_channels);
  p.deconv = 0;
  return run_pack(p, (hipStream_t)stream);
}

extern "C" int dm_conv_pack_weight_batch(const dm_pack_job* jobs_device, int num_jobs, dm_stream_t stream) {
  if (num_jobs < 0 || (num_jobs > 0 && !jobs_device)) return DM_ERR_INVALID_ARG;
  if (num_jobs == 0) return DM_OK;
  if (num_jobs > 65535) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(pack_weight_batch_kernel, dim3(64, (unsigned)num_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  return dm_check_launch();
}

static int conv2d_launch(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                         int num_srcs, int NB, int H, int W,
                         const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                         int out_ch_total, int out_ch_offset, const float* mask, dm_stream_t stream, float* ws = nullptr,
                         long long ws_floats = 0);

extern "C" int dm_conv2d_fwd(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                             int num_srcs, int NB, int H, int W,
                             const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                             int out_ch_total, int out_ch_offset, dm_stream_t stream) {
  return conv2d_launch(srcs, src_channels, src_batch_strides, num_srcs, NB, H, W, w_packed, bias, Cout, ksize, relu, out,
                       out_ch_total, out_ch_offset, nullptr, stream);
}

// (ABI 20) dm_conv2d_fwd with a caller-owned workspace: launches of few workgroups split their K loop over up to eight
// workgroups per tile (bare sums to the workspace, added in split order by a second kernel: the same bits every run; they
// differ from the unsplit sum in rounding only).  dm_conv2d_splitk_floats: the workspace that lets this shape split as far
// as it wants to (0: the launch would not split).
extern "C" long long dm_conv2d_splitk_floats(int NB, int H, int W, int Cout, int ksize) {
  if (NB <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const long long px = (long long)NB * H * W;
  // tiles of the smallest configuration the launcher would pick: 128 couts x 32 pixels (3x3, Cout > 64), else >= 64 x 128
  const long long wgs = (long long)dm_ceil_div(Cout, 128) * dm_ceil_div(px, 128);
  if (ksize == 3 && Cout > 64) {
    // 128 x 128 tiles, two or three workgroups per CU: split while the tiles do not fill half a round of slots
    const long long slots = 3LL * dm_num_cus();
    if (wgs * 2 > slots) return 0;
    return min(8LL, slots / wgs) * NB * Cout * H * W;
  }
  if (wgs * 2 > 3LL * dm_num_cus()) return 0;
  return 8LL * NB * Cout * H * W;
}

extern "C" int dm_conv2d_fwd_ws(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                                int num_srcs, int NB, int H, int W,
                                const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                                int out_ch_total, int out_ch_offset, float* workspace, long long workspace_floats,
                                dm_stream_t stream) {
  return conv2d_launch(srcs, src_channels, src_batch_strides, num_srcs, NB, H, W, w_packed, bias, Cout, ksize, relu, out,
                       out_ch_total, out_ch_offset, nullptr, stream, workspace, workspace_floats);
}

extern "C" int dm_conv2d_fwd_masked(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                                    int num_srcs, int NB, int H, int W,
                                    const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                                    int out_ch_total, int out_ch_offset, const float* mask, dm_stream_t stream) {
  if (!mask) return DM_ERR_INVALID_ARG;
  return conv2d_launch(srcs, src_channels, src_batch_strides, num_srcs, NB, H, W, w_packed, bias, Cout, ksize, relu, out,
                       out_ch_total, out_ch_offset, mask, stream);
}

static int conv2d_launch(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                         int num_srcs, int NB, int H, int W,
                         const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                         int out_ch_total, int out_ch_offset, const float* mask, dm_stream_t stream, float* ws,
                         long long ws_floats) {
  if (!srcs || !src_channels || num_srcs < 1 || num_srcs > DM_MAX_SOURCES || !w_packed || !out) return DM_ERR_INVALID_ARG;
  if (NB < 0 || H <= 0 || W <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return DM_ERR_INVALID_ARG;
  if (out_ch_offset < 0 || out_ch_offset + Cout > out_ch_total) return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  ConvArgs a;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) {
    a.src[s] = s < num_srcs ? srcs[s] : nullptr;
    a.src_c[s] = s < num_srcs ? src_channels[s] : 0;
    a.src_bs[s] = 0;
    if (s < num_srcs) {
      if (!srcs[s] || src_channels[s] <= 0) return DM_ERR_INVALID_ARG;
      a.src_bs[s] = src_batch_strides ? src_batch_strides[s] : (long long)src_channels[s] * H * W;
      if (a.src_bs[s] < (long long)src_channels[s] * H * W) return DM_ERR_INVALID_ARG;
    }
  }
  a.num_srcs = num_srcs;
  if (relu & ~15) return DM_ERR_INVALID_ARG;      // (bits 4, 5 selected the bf16-split layouts of ABI 18-21: removed)
  a.KQ = packed_quads(num_srcs, src_channels);
  a.NB = NB; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wq = w_packed; a.bias = bias; a.Cout = Cout; a.CoutP = dm_conv_packed_cout(Cout);
  a.relu = relu & 3; a.out = out; a.out_ch_total = out_ch_total; a.out_ch_offset = out_ch_offset;
  a.shuffle = 0;
  a.q_begin = 0;
  a.mask = mask;
  a.ws = (ws && ws_floats > 0) ? ws : nullptr;
  a.ws_floats = ws_floats;
  a.off32 = 1;
  for (int s = 0; s < num_srcs; ++s)
    if ((long long)NB * a.src_bs[s] * 4 >= (1LL << 32)) a.off32 = 0;
  // outputs that cannot stay in the 256 MB Infinity Cache next to their consumer's other traffic are
  // written with nontemporal stores (measured: -10 % on the 1.85 GB column-gradient GEMM, neutral
  // below); accumulating launches read the destination and keep the default policy
  if (!(relu & 2) && (long long)NB * Cout * H * W * 4 > (192LL << 20)) a.relu |= 4;
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 3) {
    if (Cout > 64) {
      // 128 x 128 tiles run two or three to a CU: a launch is a sequence of rounds of 512 / 768 workgroups,
      // and a last round with few workgroups takes as long as a lone workgroup (measured at two per CU:
      // 501 RoIs of 14 x 14 = 1536 workgroups 0.916 ms, 502 RoIs 1.059 ms).  The pixels of an underfull last round go to a
      // second launch with 128 x 32 tiles: four times the workgroups, a quarter of the time each.
      // The split depends on the launch shape only, and both variants add an output's products
      // in the same order (chunk, tap, channel pair), so results do not depend on it.  A caller that
      // keeps a second stream busy (flag bit 3) has that stream's workgroups fill the last round;
      // the extra dependent launch then only delays its own stream (measured: 250 vs 241 img/s).
      // Worth it while the last round is at most ~0.6 full (measured at 0.01 .. 0.99); DM_CONV_TAIL=0
      // turns it off (A/B measurements).
      static const int tail_mode = getenv("DM_CONV_TAIL") ? atoi(getenv("DM_CONV_TAIL")) : 1;
      const int MT = dm_ceil_div(a.CoutP, 128), NTiles = dm_ceil_div(a.Q, 128);
      // workgroups per CU: 3 when the staged plane needs one position per thread (the 168-VGPR build
      // of the kernel; 14 x 14 maps), else 2
      const int rmax128 = dm_ceil_div(127, a.W) + 1 + 2 * (dm_ceil_div(127, a.HW) + 1);
      const int slots = (rmax128 * (a.W + 2) <= 256 ? 3 : 2) * dm_num_cus();
      const int full_rounds = (MT * NTiles) / slots;
      const int rem = MT * NTiles - full_rounds * slots;
      // Launches that do not fill the chip: up to one 128 x 128 tile per CU costs a lone workgroup's
      // 0.18 ms however few there are, up to two 0.31 ms; 128 x 32 tiles cost 0.04 + 0.046 ms per
      // tile-per-CU (16 RoIs of 14 x 14: 0.175 -> 0.077 ms, 100 RoIs: 0.322 -> 0.275 ms).
      {
        const int wgs = MT * NTiles, cus = dm_num_cus();
        // with a workspace: the 128 x 128 tiles (the efficient build) and as many K splits as fill ONE round of slots
        // (100 RoIs of 14 x 14: 306 tiles x 2 splits on 768 slots; the 128 x 32 tiles without a split: 0.233 ms)
        if (a.ws && wgs < slots) {
          const long long per = (long long)NB * Cout * H * W;
          // (the split is DECIDED here, with launch_conv_mp's own cost model, before the tile build is chosen --
          // a launch the model then declined to split used to run the 128 x 128 tiles unsplit instead of the 128 x 32 ones)
          int chunks = 0;
          for (int s_ = 0; s_ < num_srcs; ++s_) chunks += dm_ceil_div(src_channels[s_], 8);
          int Smax = (int)min((long long)min(8, slots / wgs), per > 0 ? a.ws_floats / per : 0LL);
          Smax = min(Smax, chunks / 4);
          if (conv_split_choice(chunks, true, Smax, per) >= 2) {
            a.want_split = Smax;
            return launch_conv<3, 2, 2, 2, 2, 8>(a, st);
          }
        }
        if (tail_mode && (wgs * 10 <= cus * 7 || (wgs > cus && wgs * 20 <= cus * 29))) return launch_conv<3, 4, 1, 1, 1, 8>(a, st);
      }
      if (tail_mode && !(relu & 8) && full_rounds >= 1 && rem > 0 && rem * 5 <= 3 * slots) {
        const int n_main = full_rounds * slots / MT;
        const int Q = a.Q;
        a.Q = n_main * 128;
        int rc = launch_conv<3, 2, 2, 2, 2, 8>(a, st);
        if (rc != DM_OK) return rc;
        a.q_begin = a.Q;
        a.Q = Q;
        return launch_conv<3, 4, 1, 1, 1, 8>(a, st);
      }
      return launch_conv<3, 2, 2, 2, 2, 8>(a, st);
    }
    if (Cout > 32 && Cout <= 36 && !(relu & 2)) return launch_conv<3, 1, 4, 1, 1, 8, 4>(a, st);   // DCN offset convs: 32 + 4
    if (Cout > 32) return launch_conv<3, 1, 4, 2, 1, 8>(a, st);
    return launch_conv<3, 1, 4, 1, 1, 8>(a, st);
  }
  if (Cout > 64) {
    // A launch of few 128 x 128 tiles (the <= 100-RoI inference calls: 25-307 workgroups for 256 CUs) is a chain of K
    // chunks per workgroup, and next to another stream's 3x3 launch -- the two RoI chains of a 100-detection call -- each
    // chunk's 32 MFMAs per wave queue behind the neighbour's.  Up to 1.25 tiles per CU the launch takes 128 couts x 32
    // pixels instead: four times the workgroups, 8 MFMAs per wave and chunk.  Same products in the same order (a
    // 32-channel chunk walks its quad pairs in the order two 16-channel chunks do), so rows do not depend on the launch
    // size (tests/test_ops_gpu.py).  Measured (profiles/r06_infer_notes.txt (5)): 100 / 64 / 32 / 16 detections
    // 1.941 / 1.407 / 0.901 / 0.608 -> 1.879 / 1.360 / 0.884 / 0.584 ms; thresholds 100 / 160 / 320 / 640 / 1300 tiles:
    // 1.923 / 1.896 / 1.879-1.891 / 1.883 / 1.887 at 100 detections; 128 x 64 tiles for the next 320-1300 and 64 x 64 tiles
    // for the 64-cout build: inside the noise, not kept.  DM_CONV1_SMALL_WGS: the threshold in tiles (0: off), for A/B runs.
    static const int small_env = getenv("DM_CONV1_SMALL_WGS") ? atoi(getenv("DM_CONV1_SMALL_WGS")) : -1;
    const int small_wgs = small_env >= 0 ? small_env : (5 * dm_num_cus()) / 4;
    if (!(relu & 2) && !mask && dm_ceil_div(a.CoutP, 128) * dm_ceil_div(a.Q, 128) <= small_wgs)
      return launch_conv<1, 4, 1, 1, 1, 32>(a, st);
    return launch_conv<1, 2, 2, 2, 2, 16>(a, st);
  }
  if (Cout > 32) {
    // 64 couts x 128 px as 4 waves of 32 x 64: 0.437 -> 0.379 ms on 576 -> 64 @56^2 x 128 RoIs, 0.098 -> 0.089 ms on
    // 64 -> 64 x 256 (the other tilings tried: docs/HISTORY.md, round 3); one k order per output.
    return launch_conv<1, 2, 2, 1, 2, 16>(a, st);
  }
  return launch_conv<1, 1, 4, 1, 1, 32>(a, st);
}

// ---------------------------------------------------------------------------
// K16: ConvTranspose2d(k=2, s=2) = four independent 1x1 GEMMs (one per output
// phase).  Run as ONE 1x1 implicit GEMM with 4*Cout packed output channels and
// a pixel-shuffling epilogue.
extern "C" int dm_deconv_pack_weight(const float* w_iohw, int Cin, int Cout, float* w_packed, dm_stream_t stream) {
  if (!w_iohw || !w_packed || Cin <= 0 || Cout <= 0) return DM_ERR_INVALID_ARG;
  PackArgs p;
  p.w = w_iohw; p.wq = w_packed; p.Cout = Cout; p.Cin = Cin; p.kk = 1; p.flip = 0;
  p.rows = Cin; p.cols = 4 * Cout; p.colsP = dm_conv_packed_cout(4 * Cout);
  p.nsrc = 1;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) p.src_c[s] = 0;
  p.src_c[0] = Cin;
  p.KQ = packed_quads(1, p.src_c);
  p.deconv = 1;
  return run_pack(p, (hipStream_t)stream);
}

extern "C" int dm_deconv2x2_fwd(const float* x, int NB, int C, int H, int W, const float* w_packed, const float* bias,
                                int Cout, int relu, float* out, dm_stream_t stream) {
  if (!x || !w_packed || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0) return DM_ERR_INVALID_ARG;
  if ((long long)NB * H * W > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  ConvArgs a;
  for (int s = 0; s < DM_MAX_SOURCES; ++s) { a.src[s] = nullptr; a.src_c[s] = 0; a.src_bs[s] = 0; }
  a.src[0] = x; a.src_c[0] = C; a.src_bs[0] = (long long)C * H * W; a.num_srcs = 1;
  a.KQ = packed_quads(1, a.src_c);
  a.NB = NB; a.H = H; a.W = W; a.HW = H * W; a.Q = NB * H * W;
  a.wq = w_packed; a.bias = bias; a.Cout = 4 * Cout; a.CoutP = dm_conv_packed_cout(4 * Cout);
  a.relu = relu; a.out = out; a.out_ch_total = 0; a.out_ch_offset = 0; a.shuffle = Cout; a.q_begin = 0;
  return launch_conv<1, 2, 2, 2, 2, 16>(a, (hipStream_t)stream);
}

// (ABI 26) `count` (1..3) independent single-source 1x1 convolutions (+ bias, + ReLU) in ONE launch: x[i] [NB, Cin[i], H[i], W[i]]
// -> out[i] [NB, Cout[i], H[i], W[i]], weights packed by dm_conv_pack_weight (ksize 1, one source).  Every problem runs the
// 64-cout x 128-pixel build of the kernel, in its own range of the grid: the results are those of dm_conv2d_fwd launches
// that take that build (32 < Cout <= 64) and differ from the 128-cout build's (Cout > 64) in nothing -- both walk K in
// chunks of 16 channels in the same order.
extern "C" int dm_conv1x1_group_fwd(int count, const float* const* x, const int* Cin, const int* H, const int* W, int NB,
                                    const float* const* w_packed, const float* const* bias, const int* Cout, int relu,
                                    float* const* out, dm_stream_t stream) {
  if (count < 1 || count > 3 || !x || !Cin || !H || !W || !w_packed || !bias || !Cout || !out || NB < 0) return DM_ERR_INVALID_ARG;
  if (relu & ~1) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  ConvGroup g;
  int total = 0;
  constexpr int TM = 64, TN = 128;
  for (int i = 0; i < 3; ++i) {
    const int j = i < count ? i : count - 1;          // unused slots repeat the last problem (never selected: end[] stops)
    if (!x[j] || !w_packed[j] || !out[j] || Cin[j] <= 0 || H[j] <= 0 || W[j] <= 0 || Cout[j] <= 0) return DM_ERR_INVALID_ARG;
    if ((long long)NB * H[j] * W[j] > 0x7fffffffLL) return DM_ERR_INVALID_ARG;
    ConvArgs& a = g.a[i];
    for (int s = 0; s < DM_MAX_SOURCES; ++s) { a.src[s] = nullptr; a.src_c[s] = 0; a.src_bs[s] = 0; }
    a.src[0] = x[j]; a.src_c[0] = Cin[j]; a.src_bs[0] = (long long)Cin[j] * H[j] * W[j]; a.num_srcs = 1;
    a.KQ = packed_quads(1, a.src_c);
    a.NB = NB; a.H = H[j]; a.W = W[j]; a.HW = H[j] * W[j]; a.Q = NB * H[j] * W[j];
    a.wq = w_packed[j]; a.bias = bias[j]; a.Cout = Cout[j]; a.CoutP = dm_conv_packed_cout(Cout[j]);
    a.relu = relu & 1; a.out = out[j]; a.out_ch_total = Cout[j]; a.out_ch_offset = 0;
    a.shuffle = 0; a.q_begin = 0; a.mask = nullptr; a.ws = nullptr; a.ksplit = 1;
    a.off32 = ((long long)NB * a.src_bs[0] * 4 < (1LL << 32)) ? 1 : 0;
    a.MT = dm_ceil_div(a.CoutP, TM);
    a.Wp = 0; a.plane = TN;
    if (i < count) total += a.MT * dm_ceil_div(a.Q, TN);
    g.end[i] = total;
  }
  constexpr int CK = 16, NWC = CK / 4;
  const size_t lds_bytes = 16 * ((size_t)NWC * TM + (size_t)NWC * TN);
  DM_LAUNCH((conv_igemm_group_kernel<1, 2, 2, 1, 2, 16, 1>), dim3((unsigned)total), dim3(256), lds_bytes, (hipStream_t)stream, g);
  return dm_check_launch();
}
