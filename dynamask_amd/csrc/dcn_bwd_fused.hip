// DCNv1 3x3 data gradient in ONE kernel: (grad_x, grad_offset) from grad_out without the column-gradient matrix.
//
// The reference's backward (mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-374) is columns = W^T . gradOut (a GEMM whose
// [9C x HW] result goes to memory), then deformable_col2im_coord and deformable_col2im over it
// (deform_conv_cuda_kernel.cu:279-465).  Rounds 1-3 mirrored that structure: at 56 x 56 the matrix is 1.85 GB per 256
// RoIs, written once at 2.4 TB/s and read twice.  Here the column gradient of (tap, pixel, 16 channels) exists only as
// the accumulator registers of one MFMA tile:
//
//   workgroup = (RoI n, deformable group g), ten waves.  It walks the group's channels in blocks of 16 and, per block,
//   the plane top to bottom in bands of R rows.  Wave w owns the tap pair tp = w % 5 (taps 2tp, 2tp + 1; the pair 4
//   holds tap 8 alone) and every second 32-pixel tile of the band: A = the pair's 32 x Cout slice of W^T, held in
//   registers for the whole pass (row m = tap parity (m >> 2) & 1, channel (m & 3) + 4 (m >> 3): the MFMA's output
//   layout then gives lane l the 16 channels of ONE (tap 2tp + (l >> 5), pixel l & 31) -- no lane duplicates another's
//   sample geometry), B = grad_out of the band, staged in LDS once for all ten waves.
//   The lane then does, from its registers,
//     * the coordinate gradient: corners of four channels per 16-byte LDS read from the staged rows of x
//       ([quad][row][column] float4, zero columns either side, zero rows beyond the image: outside corners read 0),
//       summed over the 16 channels; the blocks' sums are added in block order (a store, then float atomics from the
//       same lane behind workgroup barriers: one order, the same bits every run);
//     * col2im: 64-bit fixed-point LDS atomics (common.h) into a ring of R + 2h rows of the block's 16 planes; rows
//       that left the ring are final and go to grad_x once.
//   A sample whose corner rows lie outside the ring (an offset beyond h rows) takes global loads / float atomics on
//   grad_x, which is zero-filled first; from then on the workgroup adds its rows to memory instead of storing them.
//   (Only with such samples does the sum order of grad_x depend on timing; DM_DETERMINISTIC callers keep the
//   three-kernel path.)
//
// The lanes run one instruction stream (a switched-off corner adds 0 to a cell of its own pixel: the MFMAs of the NEXT unit
// are issued between this unit's channel steps and need the exec mask full, their B values read one step ahead); the next
// band's grad_out and x rows are only touched into L2 while a band computes and staged behind the barrier.
// Measured: at parity with the three-kernel path -- both are bound by the LDS scatter -- hence opt-in
// (DM_DCN_FUSED=1).
#include <type_traits>

#include "common.h"

namespace {

struct DcnBwdArgs {
  const float* x;
  const float* offset;
  const float* gout;
  const float* wpk;
  float* gx;
  float* goff;
  int NB, C, H, W, HW, dg, cpg, nblk;
  int R, hr, RR, T, tiles, TS, bands;
  int off_xq, off_gb, off_flags;      // byte offsets inside dynamic LDS (acc first)
  float inv_w;
};

constexpr int kFusedThreads = 640;   // ten waves: (tap pair, tile parity)

// -DDM_DCN_STAMPS: workgroup 0 records s_memtime at the phase boundaries of every band of its first pass (tools/dcn_stamps.py)
#ifdef DM_DCN_STAMPS
__device__ unsigned long long dcn_stamps[64 * 10 * 8];
#define DCN_STAMP(band, k)                                                                        \
  do {                                                                                            \
    if (blockIdx.x == 0 && blk == 0 && lane == 0 && (band) < 64)                                  \
      dcn_stamps[((band) * 10 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime();                  \
  } while (0)
#else
#define DCN_STAMP(band, k) \
  do {                     \
  } while (0)
#endif

// wpk[(((g * nblk + blk) * 5 + tp) * KH + s) * 64 + lane] = W[co = 2 s + (lane >> 5)][ci][tap]
__global__ __launch_bounds__(256) void dcn_bwd_pack_kernel(const float* __restrict__ w, int Cout, int C, int dg,
                                                           float* __restrict__ out) {
  const int cpg = C / dg, nblk = cpg / 16, KH = Cout / 2;
  const long long total = (long long)dg * nblk * 5 * KH * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int s = (int)(r % KH); r /= KH;
    const int tp = (int)(r % 5); r /= 5;
    const int blk = (int)(r % nblk);
    const int g = (int)(r / nblk);
    const int m = lane & 31;
    const int tap = 2 * tp + ((m >> 2) & 1);
    const int ci = g * cpg + blk * 16 + (m & 3) + 4 * (m >> 3);
    const int co = 2 * s + (lane >> 5);
    out[i] = tap < 9 ? w[((size_t)co * C + ci) * 9 + tap] : 0.f;
  }
}

template <int KH>
__global__ __launch_bounds__(kFusedThreads) void dcn_bwd_data_fused_kernel(DcnBwdArgs a) {
  constexpr int K = 2 * KH, MPS = KH / 16;          // MFMAs of the next unit issued per channel step of this one
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);          // [16][RR][W + 2]
  dm_f32x4* xq = reinterpret_cast<dm_f32x4*>(smem + a.off_xq);                   // [4][RR][W + 2]
  float* gb = reinterpret_cast<float*>(smem + a.off_gb);                         // [K][TS]
  int* bad = reinterpret_cast<int*>(smem + a.off_flags);                         // [16], then far_seen
  int* far_seen = bad + 16;
  const int W = a.W, H = a.H, HW = a.HW, Wp = W + 2, RR = a.RR, R = a.R, hr = a.hr, T = a.T, TS = a.TS;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // a scalar: tap pair and tile parity live in SGPRs
  const int tp = wave % 5, par = (wave / 5) & 1;
  const int tap = 2 * tp + hi;
  const int tapc = min(tap, 8);
  const int ki = tapc / 3, kj = tapc - ki * 3;
  // consecutive workgroups land on different XCDs: give each XCD a contiguous run of (n, g) so both groups of a RoI
  // (same grad_out) share an L2
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  }
  const int g = wg % a.dg, n = wg / a.dg;
  const float* gout_n = a.gout + (size_t)n * K * HW;
  const float* offb = a.offset + ((size_t)n * a.dg + g) * 18 * HW;
  float* goffb = a.goff + ((size_t)n * a.dg + g) * 18 * HW;
  const int T4 = T >> 2;

  if (tid == 0) *far_seen = 0;

  for (int blk = 0; blk < a.nblk; ++blk) {
    const int cbase = g * a.cpg + blk * 16;
    const float* xc = a.x + ((size_t)n * a.C + cbase) * HW;
    float* gxc = a.gx + ((size_t)n * a.C + cbase) * HW;
    // ---- this pass's A operand: the tap pair's rows of W^T for the block's 16 channels, in registers for the pass
    float av[KH];
    const float* wp = a.wpk + ((size_t)((g * a.nblk + blk) * 5 + tp) * KH) * 64 + lane;
    auto load_a = [&]() {
#pragma unroll
      for (int s = 0; s < KH; ++s) av[s] = wp[(size_t)s * 64];
    };
    __syncthreads();                                 // the previous pass is done with LDS
    for (int i = tid; i < 16 * RR * Wp; i += kFusedThreads) acc[i] = 0ull;
    if (tid < 16) bad[tid] = 0;
    // the first ring (rows -hr .. R + hr - 1; zero columns either side, zero rows outside the image) and band 0
    for (int j = tid; j < 4 * RR * Wp; j += kFusedThreads) {
      const int q = j / (RR * Wp), rem = j - q * (RR * Wp);
      const int rr = rem / Wp, c = rem - rr * Wp;
      const int y = rr - hr, xc0 = c - 1;
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (y >= 0 && y < H && xc0 >= 0 && xc0 < W) {
        const float* p = xc + (size_t)(4 * q) * HW + y * W + xc0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p[(size_t)e * HW];
      }
      xq[(q * RR + (y & (RR - 1))) * Wp + c] = v;
    }
    for (int j = tid; j < K * (TS >> 2); j += kFusedThreads) {
      const int k = j / (TS >> 2), i4 = j - k * (TS >> 2);
      dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};                               // columns T .. TS - 1 stay zero
      if (4 * i4 < T && 4 * i4 < HW) v = *reinterpret_cast<const dm_f32x4*>(gout_n + (size_t)k * HW + 4 * i4);
      *reinterpret_cast<dm_f32x4*>(gb + k * TS + 4 * i4) = v;
    }
    load_a();
    __syncthreads();

    // the next band's grad_out rows and the x rows entering the ring are only TOUCHED while this band computes (one
    // dword per 128-byte line, summed into a value nothing depends on): they are then in L2 when all ten waves stage them behind the barrier.
    // (Holding them in registers meanwhile -- 36 float4 on two extra waves, or 8 on all ten -- spills beside the A
    // operand and the two accumulator tiles.)
    auto flush_rows = [&](int ring_y0, bool more) {
      // rows that leave the ring are final; the last band flushes everything that is left
      const int r_begin = max(ring_y0, 0), r_end = more ? min(ring_y0 + R, H) : H;
      const int nrow = r_end - r_begin;
      const bool rmw = *far_seen != 0;
      for (int idx = tid; idx < 16 * nrow * W; idx += kFusedThreads) {
        const int c = idx / (nrow * W), rem = idx - c * (nrow * W);
        const int rr = rem / W, col = rem - rr * W;
        const int y = r_begin + rr;
        unsigned long long* cell = acc + (c * RR + (y & (RR - 1))) * Wp + col + 1;
        float v = (float)((double)(long long)*cell * (1.0 / 68719476736.0));
        *cell = 0ull;
        if (bad[c]) v = __builtin_nanf("");
        float* dst = gxc + (size_t)c * HW + y * W + col;
        if (rmw) atomicAdd(dst, v);                  // far samples may already have added to this (zero-filled) cell
        else *dst = v;
      }
    };
    for (int band = 0; band < a.bands; ++band) {
      const int y0 = band * R, p0 = y0 * W;
      const bool more = band + 1 < a.bands;
      const int ring_y0 = y0 - hr;
      DCN_STAMP(band, 0);
      float touched = 0.f;
      auto touch_next = [&]() {
        if (!more) return;
        const int pn = p0 + T, lines = (T + 31) >> 5;                   // 32 floats per line
        for (int j = tid; j < K * lines; j += kFusedThreads) {
          const int k = j / lines, l = j - k * lines;
          if (pn + 32 * l < HW) touched += gout_n[(size_t)k * HW + pn + 32 * l];
        }
        const int xl = (R * W + 31) >> 5, r0 = y0 + R + hr;
        if (r0 < H)
          for (int j = tid; j < 16 * xl; j += kFusedThreads) {
            const int c = j / xl, l = j - c * xl;
            if (r0 * W + 32 * l < HW) touched += xc[(size_t)c * HW + r0 * W + 32 * l];
          }
      };
      // ---- consumer waves: units (tap pair, 32-pixel tile); the next unit's MFMAs ride between this unit's channels
        auto offsets_of = [&](int tt, float& oh, float& ow) {
          const int i = 32 * tt + l31;
          const int p = min(p0 + min(i, T - 1), HW - 1);
          oh = offb[(size_t)(2 * tapc) * HW + p];
          ow = offb[(size_t)(2 * tapc + 1) * HW + p];
        };
        auto column_gradient = [&](int tt) {
          dm_f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
          const float* bp = gb + hi * TS + 32 * tt + l31;
#pragma unroll
          for (int s = 0; s < KH; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bp[2 * s * TS], c, 0, 0, 0);
          return c;
        };
        // (sharing an odd last tile between the pair's two waves -- channels 0-7 / 8-15, 3.5 units each instead of 4 and 3
        // in front of the barrier -- was tried: three more instantiations of this body, 95 spilled registers, 2.37 ms
        // instead of 1.94)
        auto unit = [&](auto next_tag, int t, dm_f32x16 cg, float oh, float ow, dm_f32x16& cgn) {
          constexpr bool NEXT = decltype(next_tag)::value;
          constexpr int MODE = 0;
          const int tn = t + 2;
          const int i = 32 * t + l31, p = p0 + i;
          const bool live = tap < 9 && i < T && p < HW;
          const int yr = (int)(((float)i + 0.5f) * a.inv_w);
          const int y = y0 + yr, xx = i - yr * W;
          const float h_im = (float)(y - 1 + ki) + oh, w_im = (float)(xx - 1 + kj) + ow;
          const bool valid = live && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
          const float hf = floorf(h_im), wf = floorf(w_im);
          const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
          // (a switched-off lane adds its zeros at its own pixel: 32 lanes on ONE cell serialise the LDS atomic)
          const int h_low = valid ? (int)hf : y, w_low = valid ? (int)wf : xx;
          const bool near = valid && h_low >= ring_y0 && h_low + 1 < ring_y0 + RR;
          const bool v1 = h_low >= 0 && w_low >= 0, v2 = h_low >= 0 && w_low + 1 <= W - 1;
          const bool v3 = h_low + 1 <= H - 1 && w_low >= 0, v4 = h_low + 1 <= H - 1 && w_low + 1 <= W - 1;
          // 2^4 x the bilinear weights; 0 switches a corner off (outside the map, a far / void / dead sample): its
          // atomic then adds 0 to a cell inside the ring -- every lane runs the same instructions, which the MFMAs
          // between them need (exec stays full)
          const float s16 = near ? 16.f : 0.f;
          const float a1 = v1 ? hh * hw * s16 : 0.f, a2 = v2 ? hh * lw * s16 : 0.f;
          const float a3 = v3 ? lh * hw * s16 : 0.f, a4 = v4 ? lh * lw * s16 : 0.f;
          const int s0 = h_low & (RR - 1), s1 = (h_low + 1) & (RR - 1);
          const int bt = s0 * Wp + w_low + 1, bb = s1 * Wp + w_low + 1;     // w_low in [-1, W - 1]: columns 0 .. W + 1
          const dm_f32x4* xt = xq + bt;
          const dm_f32x4* xb = xq + bb;
          unsigned long long* pt = acc + bt;
          unsigned long long* pb = acc + bb;
          const float* bp = gb + hi * TS + 32 * tn + l31;
          if (NEXT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cgn[r] = 0.f;
          }
          float acc_h = 0.f, acc_w = 0.f;
          dm_f32x4 x1, x2, x3, x4;
          // B values of the next unit's MFMAs are read one channel step ahead, in front of this step's atomics: the LDS
          // returns in order, and an MFMA that waits for a read queued behind four ds_add_u64 stalls ~200 cycles a step
          float bq[MPS];
          if (NEXT) {
#pragma unroll
            for (int m = 0; m < MPS; ++m) bq[m] = bp[2 * m * TS];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float bn[MPS];
            if (NEXT && r < 15) {
#pragma unroll
              for (int m = 0; m < MPS; ++m) bn[m] = bp[2 * ((r + 1) * MPS + m) * TS];
            }
            if ((r & 3) == 0 && MODE != 2) {
              const int q = r >> 2;
              x1 = xt[q * RR * Wp]; x2 = xt[q * RR * Wp + 1]; x3 = xb[q * RR * Wp]; x4 = xb[q * RR * Wp + 1];
            }
            if (MODE == 0 || (MODE == 1 && r < 8) || (MODE == 2 && r >= 8)) {
              // col2im into the ring (deformable_col2im_gpu_kernel, :279-335)
              const int o = r * RR * Wp;
              dm_fix36_accumulate(pt + o, cg[r] * a1);
              dm_fix36_accumulate(pt + o + 1, cg[r] * a2);
              dm_fix36_accumulate(pb + o, cg[r] * a3);
              dm_fix36_accumulate(pb + o + 1, cg[r] * a4);
            }
            if (MODE != 2) {
              // coordinate gradient (get_coordinate_weight, deform_conv_cuda_kernel.cu:145-188); outside corners read 0
              const int e = r & 3;
              const float gh = __builtin_fmaf(lw, x4[e], __builtin_fmaf(hw, x3[e], __builtin_fmaf(-lw, x2[e], -hw * x1[e])));
              const float gw = __builtin_fmaf(lh, x4[e], __builtin_fmaf(-lh, x3[e], __builtin_fmaf(hh, x2[e], -hh * x1[e])));
              acc_h = __builtin_fmaf(cg[r], gh, acc_h);
              acc_w = __builtin_fmaf(cg[r], gw, acc_w);
            }
            if (NEXT) {
#pragma unroll
              for (int m = 0; m < MPS; ++m) cgn = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r * MPS + m], bq[m], cgn, 0, 0, 0);
              if (r < 15) {
#pragma unroll
                for (int m = 0; m < MPS; ++m) bq[m] = bn[m];
              }
            }
            __builtin_amdgcn_sched_barrier(0);       // one channel at a time: registers, and the MFMAs stay spread out
          }
          if (!near) acc_h = acc_w = 0.f;
          // ---- the rare cases, lane by lane (exec may now be partial)
          if (valid) {
            bool finite = true;
#pragma unroll
            for (int r = 0; r < 16; ++r) finite = finite && __builtin_isfinite(cg[r]);
            if (!finite || !near) {
              if (!near) *far_seen = 1;
              // a far sample: corners from memory, grad_x through float atomics (the header comment's last paragraph);
              // a non-finite column gradient poisons its plane
              const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
              const int rt = h_low * W + w_low;
              dm_f32x16 cr = cg;                       // rotated, not indexed: the loop stays rolled
#pragma nounroll
              for (int r = 0; r < 16; ++r) {
                float c0 = cr[0];
                cr = __builtin_shufflevector(cr, cr, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0);
                const bool own = MODE == 0 || (MODE == 1 && r < 8) || (MODE == 2 && r >= 8);
                if (!__builtin_isfinite(c0)) { if (own) bad[r] = 1; c0 = 0.f; }
                if (near) continue;
                const float* xp = xc + (size_t)r * HW + rt;
                const float y1 = v1 ? xp[0] : 0.f, y2 = v2 ? xp[1] : 0.f, y3 = v3 ? xp[W] : 0.f, y4 = v4 ? xp[W + 1] : 0.f;
                const float gh = __builtin_fmaf(lw, y4, __builtin_fmaf(hw, y3, __builtin_fmaf(-lw, y2, -hw * y1)));
                const float gw = __builtin_fmaf(lh, y4, __builtin_fmaf(-lh, y3, __builtin_fmaf(hh, y2, -hh * y1)));
                acc_h = __builtin_fmaf(c0, gh, acc_h);
                acc_w = __builtin_fmaf(c0, gw, acc_w);
                if (!own) continue;
                float* gp = gxc + (size_t)r * HW + rt;
                if (v1) atomicAdd(gp, c0 * w1);
                if (v2) atomicAdd(gp + 1, c0 * w2);
                if (v3) atomicAdd(gp + W, c0 * w3);
                if (v4) atomicAdd(gp + W + 1, c0 * w4);
              }
            }
          }
          if (live && MODE != 2) {
            float* go = goffb + (size_t)(2 * tap) * HW + p;
            if (blk == 0) {
              go[0] = acc_h;
              go[HW] = acc_w;
            } else {
              atomicAdd(go, acc_h);
              atomicAdd(go + HW, acc_w);
            }
          }
        };
        int t = par;
        if (t < a.tiles) {
          float oh, ow;
          offsets_of(t, oh, ow);
          touch_next();                              // behind the first unit's offsets: their wait does not cover these
          dm_f32x16 cg = column_gradient(t);
          for (; t + 2 < a.tiles; t += 2) {
            float ohn, own;
            offsets_of(t + 2, ohn, own);
            dm_f32x16 cgn;
            unit(std::true_type{}, t, cg, oh, ow, cgn);
            cg = cgn; oh = ohn; ow = own;
          }
          dm_f32x16 unused;
          unit(std::false_type{}, t, cg, oh, ow, unused);
        } else {
          touch_next();
        }
        if (touched == 1.2345678e30f) *far_seen = 2;   // (never: keeps the touching loads)
      DCN_STAMP(band, 1);
      __syncthreads();                               // the band's MFMA reads and atomics are done
      DCN_STAMP(band, 2);
      flush_rows(ring_y0, more);
      DCN_STAMP(band, 3);
      if (more) {
        // ---- stage the next band (L2 hits): all of a thread's loads in flight, then its LDS stores
        const int pn = p0 + T;
        constexpr int SG = 6;                          // host: K * T / 4 <= 6 * 640
        dm_f32x4 sv[SG];
#pragma unroll
        for (int i = 0; i < SG; ++i) {
          const int j = tid + i * kFusedThreads;
          const int k = j / T4, i4 = j - k * T4;
          sv[i] = dm_f32x4{0.f, 0.f, 0.f, 0.f};
          if (j < K * T4 && pn + 4 * i4 < HW) sv[i] = *reinterpret_cast<const dm_f32x4*>(gout_n + (size_t)k * HW + pn + 4 * i4);
        }
#pragma unroll
        for (int i = 0; i < SG; ++i) {
          const int j = tid + i * kFusedThreads;
          const int k = j / T4, i4 = j - k * T4;
          if (j < K * T4) *reinterpret_cast<dm_f32x4*>(gb + k * TS + 4 * i4) = sv[i];
        }
        const int r0 = y0 + R + hr;
        for (int j = tid; j < 4 * R * Wp; j += kFusedThreads) {
          const int q = j / (R * Wp), rem = j - q * (R * Wp);
          const int rr = rem / Wp, c = rem - rr * Wp;
          const int y = r0 + rr, xc0 = c - 1;
          dm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (y >= 0 && y < H && xc0 >= 0 && xc0 < W) {
            const float* p = xc + (size_t)(4 * q) * HW + y * W + xc0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = p[(size_t)e * HW];
          }
          xq[(q * RR + (y & (RR - 1))) * Wp + c] = v;
        }
        DCN_STAMP(band, 4);
      }
      DCN_STAMP(band, 5);
      // (the barrier's workgroup-scope release waits for the stores above: a later far sample of another wave adds to
      // rows that are in the L2 this CU shares.  An agent-scope __threadfence() here writes the XCD's L2 back every
      // band: 5.3 ms per launch instead of ~1)
      __syncthreads();
      DCN_STAMP(band, 6);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
static bool fused_geometry(int C, int Cout, int H, int W, int dg, DcnBwdArgs* a, size_t* lds_bytes) {
  if (dg < 1 || C % dg || (C / dg) % 16 || (Cout != 64 && Cout != 128)) return false;
  if (W < 4 || (W & 3) || H < 4 || H * W >= (1 << 20)) return false;
  static const int r_env = getenv("DM_DCN_FUSED_R") ? atoi(getenv("DM_DCN_FUSED_R")) : 0;
  const int R = r_env > 0 ? r_env : 4;
  if (R != 2 && R != 4 && R != 8) return false;
  const int RR = 2 * R;                              // a power of two: ring slot = row & (RR - 1)
  const int hr = R / 2;
  const int T = R * W, tiles = (T + 31) / 32, TP = tiles * 32;
  const int TS = (TP % 64 == 32) ? TP : TP + 32;     // rows k and k + 1 of the band half a bank set apart
  if ((long long)Cout * (T / 4) > 6LL * kFusedThreads) return false;
  const size_t acc_b = (size_t)16 * RR * (W + 2) * 8, xq_b = (size_t)4 * RR * (W + 2) * 16, gb_b = (size_t)Cout * TS * 4;
  const size_t total = acc_b + xq_b + gb_b + 128;
  if (total > 160 * 1024) return false;
  if (a) {
    a->C = C; a->H = H; a->W = W; a->HW = H * W; a->dg = dg; a->cpg = C / dg; a->nblk = C / dg / 16;
    a->R = R; a->hr = hr; a->RR = RR; a->T = T; a->tiles = tiles; a->TS = TS; a->bands = (H + R - 1) / R;
    a->off_xq = (int)acc_b; a->off_gb = (int)(acc_b + xq_b); a->off_flags = (int)(acc_b + xq_b + gb_b);
    a->inv_w = 1.0f / (float)W;
  }
  if (lds_bytes) *lds_bytes = total;
  return true;
}

extern "C" int dm_dcn_bwd_data_fused_supported(int C, int Cout, int H, int W, int deform_groups) {
  return fused_geometry(C, Cout, H, W, deform_groups, nullptr, nullptr) ? 1 : 0;
}

extern "C" long long dm_dcn_bwd_pack_floats(int C, int Cout, int deform_groups) {
  if (deform_groups < 1 || C % deform_groups || (C / deform_groups) % 16 || (Cout & 1)) return 0;
  return (long long)deform_groups * (C / deform_groups / 16) * 5 * (Cout / 2) * 64;
}

extern "C" int dm_dcn_bwd_pack(const float* weight, int Cout, int C, int deform_groups, float* packed, dm_stream_t stream) {
  const long long total = dm_dcn_bwd_pack_floats(C, Cout, deform_groups);
  if (!weight || !packed || total <= 0) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(dcn_bwd_pack_kernel, dim3((unsigned)min((long long)4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            weight, Cout, C, deform_groups, packed);
  return dm_check_launch();
}

extern "C" int dm_dcn_bwd_data_fused(const float* x, const float* offset, const float* grad_out, const float* w_packed, int NB,
                                     int C, int Cout, int H, int W, int deform_groups, float* grad_x, float* grad_offset,
                                     dm_stream_t stream) {
  if (!x || !offset || !grad_out || !w_packed || !grad_x || !grad_offset || NB < 0) return DM_ERR_INVALID_ARG;
  DcnBwdArgs a;
  size_t lds = 0;
  if (!fused_geometry(C, Cout, H, W, deform_groups, &a, &lds)) return DM_ERR_UNSUPPORTED;
  if (((uintptr_t)grad_out | (uintptr_t)x) & 15u) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  a.x = x; a.offset = offset; a.gout = grad_out; a.wpk = w_packed; a.gx = grad_x; a.goff = grad_offset; a.NB = NB;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(grad_x, 0, (size_t)NB * C * H * W * sizeof(float), st) != hipSuccess) return DM_ERR_LAUNCH;
  static bool attr32[DM_MAX_DEVICES] = {false}, attr64[DM_MAX_DEVICES] = {false};
  const dim3 grid((unsigned)(NB * deform_groups));
  if (Cout == 64) {
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&dcn_bwd_data_fused_kernel<32>), 160 * 1024, attr32) != DM_OK) return DM_ERR_LAUNCH;
    DM_LAUNCH((dcn_bwd_data_fused_kernel<32>), grid, dim3(kFusedThreads), lds, st, a);
  } else {
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&dcn_bwd_data_fused_kernel<64>), 160 * 1024, attr64) != DM_OK) return DM_ERR_LAUNCH;
    DM_LAUNCH((dcn_bwd_data_fused_kernel<64>), grid, dim3(kFusedThreads), lds, st, a);
  }
  return dm_check_launch();
}

#ifdef DM_DCN_STAMPS
extern "C" int dm_dcn_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcn_stamps), sizeof(unsigned long long) * 64 * 10 * 8) == hipSuccess ? 0 : -1;
}
#endif
