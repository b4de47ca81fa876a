// K1/K2/K3: multi-level RoIAlign (avg, aligned=True, adaptive sampling grid), FPN level
// mapping fused into the kernels (one launch for all levels).  Three forward kernels, all
// "workgroup = RoI x channel chunk, lanes = output bins, NCHW output rows coalesced":
//
//   roi_align_tile_kernel   P*P <= 256 (14x14 mask / 7x7 bbox extraction) -- the tuned one:
//                           channel-quad interleaved LDS tile, merged (g+1)^2 stencils
//   roi_align_band_kernel   16 < P <= 64 (56x56 on P2 for MaskPre): same tile format, the RoI
//                           is walked in bands of output rows so any footprint height fits
//   roi_align_kernel        everything else (C % 4 != 0, P > 64: GT-bitmap mask targets) and
//                           the backward (scatter atomics): planar LDS tile or direct global
//                           taps, sample geometry kept in registers across the channel chunk
//
// (History: the direct-global kernel ran at 9 % of the HBM roofline on the 14x14 case, the
// planar LDS version at 16.5 %, the tile kernel at 35 %; see DESIGN.md for what bounded each.)
#include "common.h"

namespace {

struct RoiArgs {
  const float* feat[DM_MAX_LEVELS];
  float* gfeat[DM_MAX_LEVELS];
  int H[DM_MAX_LEVELS], W[DM_MAX_LEVELS];
  float scale[DM_MAX_LEVELS];
  int L, B, C;
  const float* rois;
  int N, P, sr;
  float finest;
  float* out;          // fwd: output; bwd: unused
  const float* gout;   // bwd: grad of output
  int32_t* levels;
  int CT;              // channels per workgroup
  int lds_floats;      // forward: LDS budget of the footprint tile (0 = direct global path)
  int order;           // forward tile / band kernels: 0 = workgroup b -> (RoI b / chunks, chunk b % chunks);
                       // 1 = XCD-aware (see roi_unit)
  int abl;             // ablation bits of the tile kernel (tools/micro/roi_tile_ablate.hip only, see DM_ABL)
  const float* sorted = nullptr;   // tile kernel: RoI records in processing order (roi_order_kernel), 8 floats each
};

// Ablations of the 14x14 tile kernel for the ceiling measurement (profiles/r04_roialign_ceiling.txt): the micro
// benchmark compiles THIS file once per variant with DM_ROI_ABLATE = the variant's bits (1: no global loads, 2: one tap
// instead of the stencil, 4: no output stores) -- compile-time constants: round 3 switched them at run time, and those
// uniform branches in the batch loop are exactly what makes the compiler serialise the loads (the harness
// read 80 us for the 55 us kernel).  In the library the condition is the constant false.
#ifdef DM_ROI_ABLATE
#define DM_ABL(a, bit) (((DM_ROI_ABLATE) & (bit)) != 0)
#else
#define DM_ABL(a, bit) false
#endif

// Which (RoI, channel chunk) a workgroup of the forward tile / band kernels takes.
// Workgroups are dealt round-robin to the 8 XCDs (b % 8), each with its own 4 MB L2.  In the plain order the
// chunks of one RoI sit side by side, so at any time every XCD is reading ALL the channels it owns of many RoIs:
// with 32 channels per chunk that is 8 channel quads x 1.4 MB of pyramid per XCD, and the staging reads of the
// RoIs that overlap (each map pixel is wanted by ~10 RoIs) miss the L2 and go out to the fabric again.  Order 1
// gives XCD x the chunks c with c % 8 == x and walks them chunk-major (all RoIs of a chunk before the next chunk):
// what an XCD reads at any time is one or two chunks' planes, which stay in its L2 across the RoIs, and no plane is
// fetched by more than one XCD.  Placement (b % 8) is a speed assumption only; any mapping gives the same results.
__device__ __forceinline__ void roi_unit(const RoiArgs& a, int chunks, int& k, int& chunk) {
  const int b = blockIdx.x;
  if (a.order == 1) {
    const int x = b & 7, j = b >> 3;          // chunks % 8 == 0 (checked by the launcher)
    const int lc = j / a.N;
    k = j - lc * a.N;
    chunk = lc * 8 + x;
  } else {
    k = b / chunks;
    chunk = b - k * chunks;
  }
}

// One sample coordinate along an axis -> (low index, high index, w_low, w_high).
// mmcv bilinear_interpolate rules: c < -1 or c > size -> void sample; clamp to
// >= 0; low >= size-1 -> low = high = size-1, c = low.
__device__ __forceinline__ void axis_sample(float start, float bin, int g, int p, int i, int size, int& lo, int& hi,
                                            float& wlo, float& whi) {
  float c = start + (float)p * bin + ((float)i + 0.5f) * bin / (float)g;
  const bool valid = !(c < -1.0f || c > (float)size);
  c = fmaxf(c, 0.0f);
  int l = (int)c;
  int h;
  if (l >= size - 1) {
    l = h = size - 1;
    c = (float)l;
  } else {
    h = l + 1;
  }
  float wh = c - (float)l;
  float wl = 1.0f - wh;
  if (!valid) {
    wl = 0.f;
    wh = 0.f;
    l = 0;
    h = 0;
  }
  lo = l;
  hi = h;
  wlo = wl;
  whi = wh;
}

__device__ __forceinline__ int roi_level(float x1, float y1, float x2, float y2, float finest, int L) {
  // floor(log2(sqrt(w*h)/finest + 1e-6)) clamped to [0, L-1] (single_level_roi_extractor.py:32-51)
  // with log2 the CORRECTLY ROUNDED fp32 function (what torch's CPU log2 returns at these points):
  // level >= k  <=>  round_f32(log2(t)) >= k  <=>  t >= T[k], the smallest float whose rounded
  // log2 reaches k.  Just below a power of two the true log2 is k - 1.2e-7*2^-? and rounds UP to k
  // once the floats around k are coarser than that: T[3] and T[4] sit one ulp below 8 and 16, T[5..7]
  // two ulps below 32, 64, 128 (tests/test_ops_gpu.py sweeps +-64 ulps around every threshold).
  // sqrt and the division are IEEE-rounded (hipcc default), like the reference's torch ops.
  const float s = sqrtf((x2 - x1) * (y2 - y1));
  const float t = s / finest + 1e-6f;
  const unsigned T[8] = {0u, 0x40000000u, 0x40800000u, 0x40ffffffu, 0x417fffffu, 0x41fffffeu, 0x427ffffeu, 0x42fffffeu};
  int lvl = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k)
    if (k < L && t >= __uint_as_float(T[k])) lvl = k;
  return lvl;
}

// Source window: the samples are read from a [FH][pitch] tile whose origin is
// feature pixel (fy0, fx0) -- the whole map for the global path (fy0 = fx0 = 0,
// FH = Hl, pitch = Wl), the staged footprint for the LDS path.  `csrc0` = channel
// held at f + 0 (0 for the global map, the first staged channel for LDS).
struct SrcWin {
  int fy0, fx0, FH, FW, pitch, plane, csrc0;
};

template <int G, bool BWD>
__device__ __forceinline__ void roi_bin_fast(const RoiArgs& a, const float* __restrict__ f, float* __restrict__ gf,
                                             int Hl, int Wl, float sh, float sw, float bh, float bw, int gh, int gw,
                                             float inv_count, int ph, int pw, int k, int c0, int c1, const SrcWin win) {
  int ylo[G], yhi[G], xlo[G], xhi[G];
  float wyl[G], wyh[G], wxl[G], wxh[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    ylo[i] = yhi[i] = xlo[i] = xhi[i] = 0;
    wyl[i] = wyh[i] = wxl[i] = wxh[i] = 0.f;
    if (i < gh) {
      axis_sample(sh, bh, gh, ph, i, Hl, ylo[i], yhi[i], wyl[i], wyh[i]);
      // void samples carry weight 0; the clamp only keeps their (unused) address in the tile
      ylo[i] = min(max(ylo[i] - win.fy0, 0), win.FH - 1) * win.pitch;
      yhi[i] = min(max(yhi[i] - win.fy0, 0), win.FH - 1) * win.pitch;
    }
    if (i < gw) {
      axis_sample(sw, bw, gw, pw, i, Wl, xlo[i], xhi[i], wxl[i], wxh[i]);
      xlo[i] = min(max(xlo[i] - win.fx0, 0), win.FW - 1);
      xhi[i] = min(max(xhi[i] - win.fx0, 0), win.FW - 1);
    }
  }
  const size_t plane = (size_t)win.plane;
  const int P = a.P;
  for (int c = c0; c < c1; ++c) {
    const size_t oidx = (((size_t)k * a.C + c) * P + ph) * P + pw;
    if (!BWD) {
      const float* fc = f + (size_t)(c - win.csrc0) * plane;
      float acc = 0.f;
#pragma unroll
      for (int iy = 0; iy < G; ++iy) {
        if (iy < gh) {
#pragma unroll
          for (int ix = 0; ix < G; ++ix) {
            if (ix < gw) {
              const float v1 = fc[ylo[iy] + xlo[ix]];
              const float v2 = fc[ylo[iy] + xhi[ix]];
              const float v3 = fc[yhi[iy] + xlo[ix]];
              const float v4 = fc[yhi[iy] + xhi[ix]];
              acc += wyl[iy] * wxl[ix] * v1 + wyl[iy] * wxh[ix] * v2 + wyh[iy] * wxl[ix] * v3 + wyh[iy] * wxh[ix] * v4;
            }
          }
        }
      }
      a.out[oidx] = acc * inv_count;
    } else {
      float* gc = gf + (size_t)(c - win.csrc0) * plane;
      const float g = a.gout[oidx] * inv_count;
#pragma unroll
      for (int iy = 0; iy < G; ++iy) {
        if (iy < gh) {
#pragma unroll
          for (int ix = 0; ix < G; ++ix) {
            if (ix < gw) {
              const float w1 = wyl[iy] * wxl[ix], w2 = wyl[iy] * wxh[ix], w3 = wyh[iy] * wxl[ix], w4 = wyh[iy] * wxh[ix];
              if (w1 != 0.f) atomicAdd(gc + ylo[iy] + xlo[ix], g * w1);
              if (w2 != 0.f) atomicAdd(gc + ylo[iy] + xhi[ix], g * w2);
              if (w3 != 0.f) atomicAdd(gc + yhi[iy] + xlo[ix], g * w3);
              if (w4 != 0.f) atomicAdd(gc + yhi[iy] + xhi[ix], g * w4);
            }
          }
        }
      }
    }
  }
}

template <bool BWD>
__device__ __forceinline__ void roi_bin_generic(const RoiArgs& a, const float* __restrict__ f, float* __restrict__ gf,
                                                int Hl, int Wl, float sh, float sw, float bh, float bw, int gh, int gw,
                                                float inv_count, int ph, int pw, int k, int c0, int c1) {
  const size_t plane = (size_t)Hl * Wl;
  const int P = a.P;
  for (int c = c0; c < c1; ++c) {
    const size_t oidx = (((size_t)k * a.C + c) * P + ph) * P + pw;
    const float* fc = BWD ? nullptr : f + (size_t)c * plane;
    float* gc = BWD ? gf + (size_t)c * plane : nullptr;
    const float g = BWD ? a.gout[oidx] * inv_count : 0.f;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      int yl, yh;
      float wyl, wyh;
      axis_sample(sh, bh, gh, ph, iy, Hl, yl, yh, wyl, wyh);
      for (int ix = 0; ix < gw; ++ix) {
        int xl, xh;
        float wxl, wxh;
        axis_sample(sw, bw, gw, pw, ix, Wl, xl, xh, wxl, wxh);
        if (!BWD) {
          acc += wyl * wxl * fc[yl * Wl + xl] + wyl * wxh * fc[yl * Wl + xh] + wyh * wxl * fc[yh * Wl + xl] +
                 wyh * wxh * fc[yh * Wl + xh];
        } else {
          const float w1 = wyl * wxl, w2 = wyl * wxh, w3 = wyh * wxl, w4 = wyh * wxh;
          if (w1 != 0.f) atomicAdd(gc + yl * Wl + xl, g * w1);
          if (w2 != 0.f) atomicAdd(gc + yl * Wl + xh, g * w2);
          if (w3 != 0.f) atomicAdd(gc + yh * Wl + xl, g * w3);
          if (w4 != 0.f) atomicAdd(gc + yh * Wl + xh, g * w4);
        }
      }
    }
    if (!BWD) a.out[oidx] = acc * inv_count;
  }
}

// GCLS (forward): 0 = every RoI; 1 = only RoIs with sampling grid <= 2x2 (lean
// registers -> high occupancy, which the latency-bound footprint staging needs);
// 2 = only the remaining RoIs.  The two forward launches partition the RoIs.
template <bool BWD, int GCLS>
__global__ __launch_bounds__(256) void roi_align_kernel(RoiArgs a) {
  const int chunks = (a.C + a.CT - 1) / a.CT;
  const int k = blockIdx.x / chunks;
  const int chunk = blockIdx.x - k * chunks;
  const int c0 = chunk * a.CT;
  const int c1 = min(c0 + a.CT, a.C);

  const float* r = a.rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  const int lvl = (a.L > 1) ? roi_level(x1, y1, x2, y2, a.finest, a.L) : 0;
  if (!BWD && a.levels && chunk == 0 && threadIdx.x == 0) a.levels[k] = lvl;
  const bool bad_batch = (b < 0 || b >= a.B);   // malformed batch index -> zeros, never an OOB read
  const int Hl = a.H[lvl], Wl = a.W[lvl];
  const float sc = a.scale[lvl];
  const float sw = x1 * sc - 0.5f, sh = y1 * sc - 0.5f;
  const float ew = x2 * sc - 0.5f, eh = y2 * sc - 0.5f;
  const float rw = ew - sw, rh = eh - sh;
  const int P = a.P;
  const float bh = rh / (float)P, bw = rw / (float)P;
  const int gh = a.sr > 0 ? a.sr : (int)ceilf(rh / (float)P);
  const int gw = a.sr > 0 ? a.sr : (int)ceilf(rw / (float)P);
  const float inv_count = 1.0f / (float)max(gh * gw, 1);
  const size_t img_off = bad_batch ? 0 : (size_t)b * a.C * Hl * Wl;
  const float* f = BWD ? nullptr : a.feat[lvl] + img_off;
  float* gf = BWD ? a.gfeat[lvl] + img_off : nullptr;
  const bool empty = gh <= 0 || gw <= 0 || bad_batch;
  const bool small_grid = empty || (gh <= 2 && gw <= 2);
  if (GCLS == 1 && !small_grid) return;
  if (GCLS == 2 && small_grid) return;
  const SrcWin gwin = {0, 0, Hl, Wl, Wl, Hl * Wl, 0};

  // ---- forward fast path: footprint staged in LDS ---------------------------
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bool use_lds = false;
  SrcWin lwin = gwin;
  int csub = 0;
  if (!BWD && !empty && gh <= 4 && gw <= 4 && a.lds_floats > 0) {
    // first / last sample coordinate per axis (same expression as axis_sample)
    const float yf = sh + 0.5f * bh / (float)gh;
    const float yl = sh + (float)(P - 1) * bh + ((float)(gh - 1) + 0.5f) * bh / (float)gh;
    const float xf = sw + 0.5f * bw / (float)gw;
    const float xl = sw + (float)(P - 1) * bw + ((float)(gw - 1) + 0.5f) * bw / (float)gw;
    if (!(yl < -1.0f || yf > (float)Hl || xl < -1.0f || xf > (float)Wl)) {
      const int fy0 = min((int)fmaxf(yf, 0.f), Hl - 1);
      const int fy1 = min((int)fminf(fmaxf(yl, 0.f), (float)Hl) + 2, Hl - 1);   // +1 for the high tap, +1 margin
      const int fx0 = min((int)fmaxf(xf, 0.f), Wl - 1);
      const int fx1 = min((int)fminf(fmaxf(xl, 0.f), (float)Wl) + 2, Wl - 1);
      const int FH = fy1 - fy0 + 1, FW = fx1 - fx0 + 1;
      const int pitch = FW | 1;                      // odd pitch: rows start on different banks
      if (FH * pitch <= a.lds_floats) {
        use_lds = true;
        csub = min(c1 - c0, a.lds_floats / (FH * pitch));
        lwin = {fy0, fx0, FH, FW, pitch, FH * pitch, 0};
      }
    }
  }

  if (use_lds) {
    for (int cb = c0; cb < c1; cb += csub) {
      const int nc = min(csub, c1 - cb);
      // stage the footprint of nc channels: flat index over (channel, row, 4-column group),
      // 8 x 16-byte loads in flight per thread before the LDS stores (the staging is
      // latency x concurrency bound; rows are only 4-byte aligned -> packed float4)
      {
        struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
        const int FWq = (lwin.FW + 3) >> 2;
        const int total = nc * lwin.FH * FWq;
        const float* fcb = f + (size_t)cb * Hl * Wl + (size_t)lwin.fy0 * Wl + lwin.fx0;
        for (int base = threadIdx.x; base < total; base += 8 * 256) {
          F4 v[8];
          int dsti[8], cnt[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256;
            dsti[u] = -1;
            cnt[u] = 0;
            if (idx < total) {
              const int rowi = idx / FWq;
              const int x = (idx - rowi * FWq) * 4;
              const int c = rowi / lwin.FH;
              const int r = rowi - c * lwin.FH;
              const float* src = fcb + (size_t)c * Hl * Wl + (size_t)r * Wl + x;
              dsti[u] = c * lwin.plane + r * lwin.pitch + x;
              cnt[u] = min(4, lwin.FW - x);
              if (cnt[u] == 4) {
                v[u] = *reinterpret_cast<const F4*>(src);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u].v[e] = (e < cnt[u]) ? src[e] : 0.f;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < cnt[u]) lds[dsti[u] + e] = v[u].v[e];
          }
        }
      }
      __syncthreads();
      SrcWin w = lwin;
      w.csrc0 = cb;
      for (int pos = threadIdx.x; pos < P * P; pos += blockDim.x) {
        const int ph = pos / P;
        const int pw = pos - ph * P;
        if (GCLS == 1 || (GCLS == 0 && gh <= 2 && gw <= 2))
          roi_bin_fast<2, false>(a, lds, nullptr, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, cb, cb + nc, w);
        else
          roi_bin_fast<4, false>(a, lds, nullptr, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, cb, cb + nc, w);
      }
      __syncthreads();
    }
    return;
  }

  for (int pos = threadIdx.x; pos < P * P; pos += blockDim.x) {
    const int ph = pos / P;
    const int pw = pos - ph * P;
    if (empty) {
      // empty sampling grid (degenerate RoI): mmcv's loops do not run -> 0
      if (!BWD)
        for (int c = c0; c < c1; ++c) a.out[(((size_t)k * a.C + c) * P + ph) * P + pw] = 0.f;
    } else if (GCLS != 2 && gh <= 2 && gw <= 2) {
      roi_bin_fast<2, BWD>(a, f, gf, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, c0, c1, gwin);
    } else if (GCLS != 1 && gh <= 4 && gw <= 4) {
      roi_bin_fast<4, BWD>(a, f, gf, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, c0, c1, gwin);
    } else if (GCLS != 1) {
      roi_bin_generic<BWD>(a, f, gf, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, c0, c1);
    }
  }
}

// ---------------------------------------------------------------------------
// Forward kernel for small output grids (P*P <= 256: the 14x14 mask and 7x7 bbox
// extractions).  Workgroup = one RoI x CT channels, thread = one output bin.
//
//  * The RoI's footprint is staged in LDS **channel-quad interleaved**:
//    tile[quad][row][col] is a float4 holding 4 consecutive channels of one feature
//    pixel, so ONE ds_read_b128 fetches a tap for 4 channels and adjacent lanes read
//    adjacent 16-byte words; the columns of a stencil row are instruction immediates.
//  * **Merged stencil.**  A bin's g x g bilinear samples are at most one pixel apart, so
//    they touch at most (g+1) x (g+1) feature pixels; their 4*g*g tap weights are summed
//    per pixel once (separable: Wy[r] * Wx[c], already divided by g*g) and the channel
//    loop reads (g+1)^2 taps instead of 4*g*g (9 vs 16 at g = 2, 25 vs 64 at g = 4) -- the
//    sampling is LDS-read bound, so this is the lever.  Only 2*P stencils are distinct:
//    2*P threads build them into an LDS table, every thread multiplies its row and column
//    entry.  Grids above 4 (clipped slivers: 200 feature rows x 4 columns) and fixed
//    sampling ratios with samples more than a pixel apart use a run-time loop over samples.
//  * The tile keeps G rows/columns past the last low tap: stencil cells of weight 0 must
//    still read finite values.  Those cells, and everything past the map border, are
//    clamped duplicates of the last weighted row/column -- same cache lines, no traffic.
//  * Staging: one tile pixel of one channel quad per lane -- 4 coalesced dword loads (the
//    4 channel planes) and ONE 16-byte LDS store, already interleaved (no register
//    transposition, no bank conflicts).  The LDS slot of item idx is idx; its global
//    offset does not depend on the batch and is computed once (multiply-high divisions).
//  * As many channel quads per batch as the buffer holds (bytes in flight are what the
//    staging is bound by); one buffer, 4 workgroups per CU hide each other's staging
//    round trips (see the batch loop).
// Footprints above 2048 pixels (only possible without the FPN level map) take the
// direct global path inside the same launch.
//
// Measured (512 RoIs, 1333x800 FPN, P2..P5): 68-70 us; per-workgroup timeline (s_memtime):
// setup 0.8 us, staging round trip ~1.8 us under load, 0.7 us of sampling per batch; PMC:
// VALU 35 % / LDS 25 % busy, L2 hit rate of the staging reads ~32 % (174 MB leave the L2
// per launch for 91 MB of maps: short unaligned row segments over-fetch 64-byte sectors).
constexpr int kTileFloats4 = 2048;   // float4 words of the staging buffer (32 KB): 4 workgroups share a CU (round 4 tried 1920
                                     // words + __launch_bounds__(256, 5), 96 VGPRs: FIVE per CU measured 57.0 vs 55.6 us at 512 RoIs,
                                     // 22.2 vs 20.7 at 129 -- the launch is bound by its fabric traffic, 261 MB at the copy rate)   // float4 words of LDS per workgroup (48 KB = 2 buffers of 1536 pixel-quads): 3 workgroups per CU

struct TileGeom {
  int fy0, fx0, FH, pitch;   // tile origin (feature pixel), rows, columns
  int ymax, xmax;            // last row / column that carries weight: cells past it are clamped duplicates
};

// One axis of the merged stencil: low index L of the first valid sample and the summed
// weights of pixels L .. L+G.  axis_sample zeroes both weights of a void sample.
template <int G>
__device__ __forceinline__ void axis_stencil(float start, float bin, int g, int p, int size, int& L, float (&Wt)[G + 1]) {
  int lo[G];
  float wl[G], wh[G];
  L = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    int hi;
    lo[i] = 0;
    wl[i] = wh[i] = 0.f;
    if (i < g) axis_sample(start, bin, g, p, i, size, lo[i], hi, wl[i], wh[i]);
    const bool valid = (wl[i] != 0.f) || (wh[i] != 0.f);
    if (valid) L = min(L, lo[i]);
  }
  if (L == 0x7fffffff) L = 0;
#pragma unroll
  for (int r = 0; r <= G; ++r) Wt[r] = 0.f;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int d = lo[i] - L;          // 0 .. G-1 for valid samples (samples are <= 1 pixel apart)
#pragma unroll
    for (int r = 0; r <= G; ++r) {
      if (r < G) Wt[r] += (d == r) ? wl[i] : 0.f;
      if (r > 0) Wt[r] += (d == r - 1) ? wh[i] : 0.f;
    }
  }
}

template <int G>
__device__ __forceinline__ void roi_tile_fwd(const RoiArgs& a, const float* __restrict__ fimg, int Hl, int Wl, float sh,
                                             float sw, float bh, float bw, int gh, int gw, float inv_count, int k,
                                             int c0, int c1, const TileGeom tg, float4* __restrict__ lds) {
  constexpr int kBufPx = kTileFloats4;               // pixel-quads (float4 words) of the staging buffer
  constexpr int S = G + 1;                            // merged stencil is S x S (G == 0: run-time grid)
  const int tid = threadIdx.x;
  const int P = a.P, PP = P * P;
  const bool active = tid < PP;
  const int ph = active ? (int)__umulhi((unsigned)tid, 0xFFFFFFFFu / (unsigned)P + 1u) : 0;      // tid / P (P >= 2)
  const int pw = active ? tid - ph * P : 0;
  const int plane_px = tg.FH * tg.pitch;              // <= kBufPx (checked by the caller)
  // as many channel quads per batch as the buffer holds: the staging is bound by bytes in
  // flight, so every fetch should fill the 2 x 4 x 16-byte loads each thread can issue
  const int NQ = min(kBufPx / plane_px, (c1 - c0) >> 2);      // the caller guarantees C % 4 == 0
  const int NCB = NQ * 4;

  // ---- staging.  Item = one tile pixel of one channel quad: 4 dword loads (the 4 channel
  // planes; consecutive lanes read consecutive pixels -> coalesced) and ONE 16-byte LDS
  // store, already channel-interleaved -- no register transposition.  The tile layout
  // [quad][row][col] makes the LDS slot of item `idx` simply `idx`; which pixel it is does
  // not depend on the batch, so its global byte offset is computed once.  Columns past the
  // map's last one are clamped duplicates (weight 0), like rows.
  constexpr int IPT = (kBufPx + 255) / 256;      // items per thread and batch: 8
  const size_t plane = (size_t)Hl * Wl;
  unsigned voff[IPT];
  // idx -> (quad, row, col) with multiply-high by ceil(2^32 / d): exact for idx * d < 2^32
  // (idx < 2048, d <= 2048; d >= 4 so the constants fit 32 bits); the two real divisions are uniform
  const unsigned m_plane = 0xFFFFFFFFu / (unsigned)plane_px + 1u;
  const unsigned m_pitch = 0xFFFFFFFFu / (unsigned)tg.pitch + 1u;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int idx = tid + i * 256;
    const int idc = idx < NQ * plane_px ? idx : 0;
    const int q = (int)__umulhi((unsigned)idc, m_plane);
    const int rem = idc - __mul24(q, plane_px);                     // (24-bit multiplies: full rate; 32-bit ones a quarter)
    const int r = (int)__umulhi((unsigned)rem, m_pitch);
    const int x = rem - __mul24(r, tg.pitch);
    // cells that only pad the stencil repeat the last weighted row / column: same cache lines, no extra traffic
    const int gy = min(tg.fy0 + r, tg.ymax);
    const int gx = min(tg.fx0 + x, tg.xmax);
    voff[i] = (unsigned)(__mul24(q * 4, (int)plane) + __mul24(gy, Wl) + gx) * 4u;       // bytes; H * W <= 2^23 (launcher)
  }
  float pf[IPT][4];
  // (scalar base + unsigned 32-bit lane offset: no address arithmetic per load; groups of 256 items past the batch's
  // live ones are skipped as a whole, the surplus items of the last live group re-read pixel 0 and commit it to a
  // slot of the buffer nothing samples)
  auto fetch = [&](int cb) {
    const int live = min(NQ, (c1 - cb) >> 2) * plane_px;     // items of this batch (short last batch)
    const int ngr = (live + 255) >> 8;
    const char* b0 = reinterpret_cast<const char*>(fimg + (size_t)cb * plane);   // uniform bases: the 4 channel
    const char* b1 = b0 + plane * 4;                                              // planes of a quad
    const char* b2 = b1 + plane * 4;
    const char* b3 = b2 + plane * 4;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      if (i >= ngr) continue;
      const unsigned vo = (tid + i * 256 < live) ? voff[i] : 0u;
      if (DM_ABL(a, 1)) {
        pf[i][0] = pf[i][1] = pf[i][2] = pf[i][3] = __uint_as_float(vo);
        continue;
      }
      pf[i][0] = *reinterpret_cast<const float*>(b0 + vo);
      pf[i][1] = *reinterpret_cast<const float*>(b1 + vo);
      pf[i][2] = *reinterpret_cast<const float*>(b2 + vo);
      pf[i][3] = *reinterpret_cast<const float*>(b3 + vo);
    }
  };
  auto commit = [&](int cb, int buf) {
    const int live = min(NQ, (c1 - cb) >> 2) * plane_px;
    const int ngr = (live + 255) >> 8;
    // (round 4) one unconditional vmcnt(0): every load has to be back here anyway, and it tells the compiler that none
    // is pending afterwards.  It cannot prove that the groups a fetch skipped (i >= ngr) are the groups this commit
    // skips, so it assumed a load into pf[i] might still be in flight at the NEXT fetch and put an s_waitcnt vmcnt(0)
    // in front of every group of loads there: the second and later batches of a workgroup (footprints above 512
    // pixels: a quarter of the RoIs) fetched their groups one round trip after the other.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float4* dst = lds + buf * kBufPx + tid;
#pragma unroll
    for (int i = 0; i < IPT; ++i)
      if (i < ngr) dst[i * 256] = make_float4(pf[i][0], pf[i][1], pf[i][2], pf[i][3]);
  };

  // The first batch is on its way while the stencil table is built (one wave's work, everybody else's wait).
  fetch(c0);

  // ---- stencils (channel independent).  Only 2*P of them are distinct (one per output
  // row and per output column): 2*P threads build them into a table in the -- still
  // unused -- second LDS buffer, every thread then picks its row and column entry.
  int base = 0;
  float W[G > 0 ? S * S : 1];
  if (G > 0) {
    constexpr int GG = G > 0 ? G : 1;
    float* tab = reinterpret_cast<float*>(lds + kTileFloats4);     // [2][P][8]: {L, W0 .. WG}, behind the buffer
    if (tid < 2 * P) {
      const bool xa = tid >= P;
      const int p = xa ? tid - P : tid;
      int L;
      float Wt[GG + 1];
      axis_stencil<GG>(xa ? sw : sh, xa ? bw : bh, xa ? gw : gh, p, xa ? Wl : Hl, L, Wt);
      float* e = tab + tid * 8;
      e[0] = __int_as_float(L);
#pragma unroll
      for (int r = 0; r <= GG; ++r) e[1 + r] = xa ? Wt[r] : Wt[r] * inv_count;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS only: the fetch stays in flight)
    const float* ey = tab + ph * 8;
    const float* ex = tab + (P + pw) * 8;
    const int Ly = __float_as_int(ey[0]), Lx = __float_as_int(ex[0]);
    base = min(max(Ly - tg.fy0, 0), tg.FH - S) * tg.pitch + min(max(Lx - tg.fx0, 0), tg.pitch - S);
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
      for (int c = 0; c < S; ++c) W[G > 0 ? r * S + c : 0] = ey[1 + r] * ex[1 + c];
    // (the second buffer is first written after the barrier that follows the first commit)
  }

  auto sample = [&](int cb, int buf) {
    if (active) {
      const int nq = min(NQ, (c1 - cb) >> 2);
      const float4* t = lds + buf * kBufPx;
      // (uniform base + 32-bit lane offset; C % 4 == 0 and CT % 4 == 0: a quad is never cut by c1)
      char* const ob = reinterpret_cast<char*>(a.out + ((size_t)k * a.C + cb) * PP);
      const unsigned ot = (unsigned)tid << 2;
      const size_t PB = (size_t)PP * 4;
      if (G == 0) {
        // run-time grid (slivers): samples outermost, groups of 4 channel quads accumulate per sample
        for (int q0 = 0; q0 < nq; q0 += 4) {
          float4 acc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int iy = 0; iy < gh; ++iy) {
            int lo, hi;
            float yl_w, yh_w;
            axis_sample(sh, bh, gh, ph, iy, Hl, lo, hi, yl_w, yh_w);
            const int yo = min(max(lo - tg.fy0, 0), tg.FH - 2) * tg.pitch;
            yl_w *= inv_count;
            yh_w *= inv_count;
            for (int ix = 0; ix < gw; ++ix) {
              float xl_w, xh_w;
              axis_sample(sw, bw, gw, pw, ix, Wl, lo, hi, xl_w, xh_w);
              const int xo = min(max(lo - tg.fx0, 0), tg.pitch - 2);
              const float w0 = yl_w * xl_w, w1 = yl_w * xh_w, w2 = yh_w * xl_w, w3 = yh_w * xh_w;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float4* tp = t + min(q0 + u, nq - 1) * plane_px + yo + xo;
                const float4 v1 = tp[0], v2 = tp[1], v3 = tp[tg.pitch], v4 = tp[tg.pitch + 1];
                acc[u].x += w0 * v1.x + w1 * v2.x + w2 * v3.x + w3 * v4.x;
                acc[u].y += w0 * v1.y + w1 * v2.y + w2 * v3.y + w3 * v4.y;
                acc[u].z += w0 * v1.z + w1 * v2.z + w2 * v3.z + w3 * v4.z;
                acc[u].w += w0 * v1.w + w1 * v2.w + w2 * v3.w + w3 * v4.w;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;
            if (q < nq) {
              char* const oq = ob + (size_t)(4 * q) * PB;
              *reinterpret_cast<float*>(oq + ot) = acc[u].x;
              *reinterpret_cast<float*>(oq + PB + ot) = acc[u].y;
              *reinterpret_cast<float*>(oq + 2 * PB + ot) = acc[u].z;
              *reinterpret_cast<float*>(oq + 3 * PB + ot) = acc[u].w;
            }
          }
        }
      } else {
        constexpr int kUnrollQ = (S >= 4) ? 1 : 2;
#pragma unroll kUnrollQ
        for (int q = 0; q < nq; ++q) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4* tq = t + q * plane_px + base;
          if (DM_ABL(a, 2)) {
            acc = tq[0];
          } else {
#pragma unroll
            for (int r = 0; r < S; ++r) {
              const float4* tr = tq + r * tg.pitch;
#pragma unroll
              for (int c = 0; c < S; ++c) {
                const float4 v = tr[c];
                const float wv = W[G > 0 ? r * S + c : 0];
                acc.x += wv * v.x;
                acc.y += wv * v.y;
                acc.z += wv * v.z;
                acc.w += wv * v.w;
              }
            }
          }
          if (DM_ABL(a, 4) && acc.x != 12345.678f) continue;      // (keeps the value live without the store)
          char* const oq = ob + (size_t)(4 * q) * PB;
          // (round 5) streaming stores: the output is not re-read by this kernel, and without the hint every line it writes
          // takes a place in the Infinity Cache from which a modified line has to be written back first when the caches are
          // cold and full of another kernel's data -- 86 -> 81 us from cold caches, cache-warm and the headline step unchanged
          // (round 4 had tried them cache-warm only: no difference there).  The same hint on the LOADS costs 30 us: the maps
          // are shared by neighbouring workgroups through the caches.
          __builtin_nontemporal_store(acc.x, reinterpret_cast<float*>(oq + ot));
          __builtin_nontemporal_store(acc.y, reinterpret_cast<float*>(oq + PB + ot));
          __builtin_nontemporal_store(acc.z, reinterpret_cast<float*>(oq + 2 * PB + ot));
          __builtin_nontemporal_store(acc.w, reinterpret_cast<float*>(oq + 3 * PB + ot));
        }
      }
    }
  };

  // One staging buffer: fetch -> commit -> barrier -> sample -> barrier per channel batch.
  // The staging round trip (~1.8 us under load) is not hidden inside the workgroup but by the
  // 4 workgroups that share a CU (32 KB of LDS and <= 128 VGPRs each).  Measured against a
  // double-buffered pipeline (next batch prefetched into registers, 48 KB, 3 workgroups/CU):
  // 69.6 vs 71.0 us at 512 RoIs, 23.5 vs 29.4 us at 128 -- a whole-buffer batch holds more
  // channel quads, so there are fewer dependent round trips per workgroup.
  for (int cb = c0; cb < c1; cb += NCB) {
    if (cb != c0) {
      __builtin_amdgcn_s_waitcnt(0x0F70);      // (the previous batch's stores, once, instead of in front of every group: see commit)
      fetch(cb);
    }
    commit(cb, 0);
    __syncthreads();
    sample(cb, 0);
    // LDS-only barrier: __syncthreads() would also wait (vmcnt(0)) for the acknowledgement of
    // the output stores sample() has just issued
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// Round 4: processing order of the RoIs.  With the RoIs as the caller hands them over (random positions and sizes) the
// workgroups an XCD runs at any moment stage from all over its channel planes, 4 MB of L2 keep little of it, and
// the launch fetches 160 MB from the fabric for 91 MB of maps (profiles/r04_roi_traffic.txt); walked by (level, 32-pixel
// rows, x) neighbouring workgroups share their footprints: 93 MB fetched -- the whole launch moves 1.0 x its algorithmic
// bytes -- and 56 -> 47 us with 32 channels per workgroup.  The keys are ranked by counting (N <= 1024; a first version
// with ONE workgroup of 1024 threads took 17 us: one CU's vector ALU for N x N / 4 compares) and the RoIs written in that
// order as 8-float records {batch, x1, y1, x2, y2, index,
// level, 0}, one scalar load for the extraction's workgroups; it also writes levels_out.  The order changes no result.
// The default for 192 RoIs or more when the caller passes a workspace (DM_ROI_SORT=0: off): with the first ordering kernel
// (64 RoIs x 4 threads, 7.2 us) the launch and the dependency behind it cost what the ordered extraction gains; with 16
// threads per RoI and unique keys it takes 5 us and the pair runs in 50.7 us against 57 (profiles/r04_roi_exp.txt (l)).
constexpr int kOrderMaxRois = 1024;

// workgroup = 16 RoIs x 16 threads each; every workgroup holds all N keys in LDS (computing them costs less than a
// second launch), a RoI's sixteen threads each count a sixteenth of the keys below it.  The keys carry the RoI's index in
// their low ten bits: unique, one compare per key.  (64 RoIs x 4 threads with index tie-breaks: 7.2 us for 512 RoIs.)
constexpr int kOrderTpr = 16;                      // threads per RoI
__global__ __launch_bounds__(256) void roi_order_kernel(RoiArgs a, float* __restrict__ sorted) {
  __shared__ __attribute__((aligned(16))) unsigned keys[kOrderMaxRois];
  const int tid = threadIdx.x;
  auto key_of = [&](int t, float (&rec)[5], int& lvl) {
    const float* r = a.rois + (size_t)t * 5;
#pragma unroll
    for (int i = 0; i < 5; ++i) rec[i] = r[i];
    lvl = (a.L > 1) ? roi_level(rec[1], rec[2], rec[3], rec[4], a.finest, a.L) : 0;
    const float cy = 0.5f * (rec[2] + rec[4]), cx = 0.5f * (rec[1] + rec[3]);
    const unsigned row = (unsigned)fminf(fmaxf(cy * (1.0f / 32.0f), 0.f), 127.f);            // 32-pixel rows, 7 bits
    const unsigned col = (unsigned)fminf(fmaxf(cx * (1.0f / 8.0f), 0.f), 511.f);             // 8-pixel columns, 9 bits
    const unsigned img = (unsigned)min(max((int)rec[0], 0), 3);                 // (beyond 4 images the order is only coarser)
    return (img << 30) | ((unsigned)lvl << 26) | (row << 19) | (col << 10) | (unsigned)t;     // t < 1024: unique keys
  };
  const int npad = (a.N + 3) & ~3;
  for (int t = tid; t < npad; t += 256) {
    float rec[5];
    int lvl;
    keys[t] = t < a.N ? key_of(t, rec, lvl) : 0xFFFFFFFFu;      // (padding keys: never below a real key)
  }
  __syncthreads();
  const int t = blockIdx.x * (256 / kOrderTpr) + tid / kOrderTpr, part = tid % kOrderTpr;
  float rec[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int lvl = 0;
  unsigned key = 0u;
  if (t < a.N) key = key_of(t, rec, lvl);
  const int n4 = npad >> 2;
  const uint4* k4 = reinterpret_cast<const uint4*>(keys);
  int rank = 0;
  for (int j = part; j < n4; j += kOrderTpr) {
    const uint4 q = k4[j];
    rank += (q.x < key) + (q.y < key) + (q.z < key) + (q.w < key);
  }
#pragma unroll
  for (int m = 1; m < kOrderTpr; m <<= 1) rank += __shfl_xor(rank, m, 64);
  if (t < a.N && part == 0) {
    if (a.levels) a.levels[t] = lvl;
    float* o = sorted + (size_t)rank * 8;
    reinterpret_cast<float4*>(o)[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
    reinterpret_cast<float4*>(o)[1] = make_float4(rec[4], (float)t, (float)lvl, 0.f);
  }
}

__global__ __launch_bounds__(256, 4) void roi_align_tile_kernel(RoiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int chunks = (a.C + a.CT - 1) / a.CT;
  int k, chunk;
  roi_unit(a, chunks, k, chunk);
  const int c0 = chunk * a.CT;
  const int c1 = min(c0 + a.CT, a.C);
  // (with a workspace the launcher has ordered the RoIs by level and position, roi_order_kernel: unit k of the grid is
  // the k-th RoI of that order; its record carries the RoI's own index and its level)
  const float* r = a.sorted ? a.sorted + (size_t)k * 8 : a.rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  int lvl;
  if (a.sorted) {
    k = (int)r[5];
    lvl = (int)r[6];
  } else {
    lvl = (a.L > 1) ? roi_level(x1, y1, x2, y2, a.finest, a.L) : 0;
    if (a.levels && chunk == 0 && threadIdx.x == 0) a.levels[k] = lvl;
  }
  const bool bad_batch = (b < 0 || b >= a.B);
  int Hl = a.H[0], Wl = a.W[0];
  float sc = a.scale[0];
  const float* flvl = a.feat[0];
#pragma unroll
  for (int l = 1; l < DM_MAX_LEVELS; ++l)
    if (lvl == l) {
      Hl = a.H[l];
      Wl = a.W[l];
      sc = a.scale[l];
      flvl = a.feat[l];
    }
  const float sw = x1 * sc - 0.5f, sh = y1 * sc - 0.5f;
  const float ew = x2 * sc - 0.5f, eh = y2 * sc - 0.5f;
  const float rw = ew - sw, rh = eh - sh;
  const int P = a.P, PP = P * P;
  const float bh = rh / (float)P, bw = rw / (float)P;
  const int gh = a.sr > 0 ? a.sr : (int)ceilf(rh / (float)P);
  const int gw = a.sr > 0 ? a.sr : (int)ceilf(rw / (float)P);
  const float inv_count = 1.0f / (float)max(gh * gw, 1);
  const float* fimg = flvl + (bad_batch ? 0 : (size_t)b * a.C * Hl * Wl);
  if (gh <= 0 || gw <= 0 || bad_batch) {
    // empty sampling grid (degenerate RoI) / malformed batch index -> zeros
    for (int i = threadIdx.x; i < (c1 - c0) * PP; i += blockDim.x) a.out[((size_t)k * a.C + c0) * PP + i] = 0.f;
    return;
  }
  {
    // first / last sample coordinate per axis (the expression of axis_sample at its extremes)
    const float yf = sh + 0.5f * bh / (float)gh;
    const float yl = sh + (float)(P - 1) * bh + ((float)(gh - 1) + 0.5f) * bh / (float)gh;
    const float xf = sw + 0.5f * bw / (float)gw;
    const float xl = sw + (float)(P - 1) * bw + ((float)(gw - 1) + 0.5f) * bw / (float)gw;
    const int G = max(gh, gw);
    // the merged stencil needs samples <= 1 pixel apart (always true for the adaptive grid)
    const bool merged = G <= 4 && bh <= (float)gh && bw <= (float)gw;
    const int pad = merged ? G : 1;          // cells past the last low tap (merged stencil: G, per-sample: 1)
    TileGeom tg;
    tg.fy0 = min((int)fmaxf(fminf(yf, yl), 0.f), Hl - 1);
    tg.fx0 = min((int)fmaxf(fminf(xf, xl), 0.f), Wl - 1);
    const int ylast = min((int)fmaxf(fmaxf(yf, yl), 0.f), Hl - 1);     // last low tap
    const int xlast = min((int)fmaxf(fmaxf(xf, xl), 0.f), Wl - 1);
    tg.FH = ylast + pad - tg.fy0 + 1;
    tg.pitch = xlast + pad - tg.fx0 + 1;
    tg.ymax = min(ylast + 1, Hl - 1);
    tg.xmax = min(xlast + 1, Wl - 1);
    const int px = tg.FH * tg.pitch;
#define DM_ROI_TILE(GG) roi_tile_fwd<GG>(a, fimg, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, k, c0, c1, tg, lds4)
    // with the FPN level map the footprint stays below ~1700 pixels (a 200 x 4 sliver)
    if (px <= kTileFloats4) {
      if (!merged) DM_ROI_TILE(0);
      else if (G == 1) DM_ROI_TILE(1);
      else if (G == 2) DM_ROI_TILE(2);
      else if (G == 3) DM_ROI_TILE(3);
      else DM_ROI_TILE(4);
      return;
    }
#undef DM_ROI_TILE
  }
  for (int pos = threadIdx.x; pos < PP; pos += blockDim.x) {
    const int ph = pos / P;
    const int pw = pos - ph * P;
    roi_bin_generic<false>(a, fimg, nullptr, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, c0, c1);
  }
}

// ---------------------------------------------------------------------------
// Forward kernel for larger output grids (16 < P <= 64: the 56x56 extraction on P2 that
// feeds MaskPre).  Same LDS tile format, merged stencils and staging as the
// 14x14 kernel, generalised in two ways:
//   * **bands**: the footprint of a large RoI (up to 200 x 336 feature pixels) does not fit
//     LDS, but a band of R output rows only needs (R * bin_h + g + 2) feature rows; the
//     workgroup walks the bands and re-stages per band (rows shared by two bands are read twice);
//   * several output bins per thread: the stencils of all P rows and P columns sit in an LDS
//     table, a thread rebuilds the (g+1)^2 products of each bin it owns from two table
//     entries and reuses them for the channel quads of the batch.
// RoIs wider than the buffer allows even for a one-row band take the direct global path.
constexpr int kBandTabFloats4 = 2 * 64 * 8 / 4;      // stencil table for P <= 64

template <int G, int TB>
__device__ __forceinline__ void roi_band_fwd(const RoiArgs& a, const float* __restrict__ fimg, int Hl, int Wl, float sh,
                                             float sw, float bh, float bw, int gh, int gw, float inv_count, int k,
                                             int c0, int fx0, int pitch, int xmax, int rows_per_band,
                                             int pw0, int Cw, float4* __restrict__ lds) {
  // pw0, Cw: the block of output columns this call produces (the whole width unless the RoI is too wide for a tile)
  // One channel quad (c0 .. c0+3) per workgroup.  A tile slot's (row, column) does not depend on the band, so the
  // division is done once per RoI (packed: row << 20 | map column) and a band's gather offset is five full-rate
  // VALU instructions per slot; loads and stores are a scalar base + a 32-bit lane offset.  (SQ counters,
  // profiles/r03_sq_pmc_roi.txt: the kernel issued 2700 VALU instructions per wave where its stencils need 500 --
  // per band and slot two magic divisions and five 32-bit multiplies at a quarter of the rate, and a 64-bit
  // address add in front of every load and store -- and was VALU-bound: 0.35 M of its 0.52 M cycles.)
  static_assert(TB % 256 == 0, "tile slots are dealt 256 at a time");
  constexpr int S = G + 1;
  constexpr int GG = G > 0 ? G : 1;
  constexpr int IPT = TB / 256;
  const int tid = threadIdx.x;
  const int P = a.P, PP = P * P;
  const size_t plane = (size_t)Hl * Wl;
  const int pad = G > 0 ? G : 1;
  const unsigned m_pitch = 0xFFFFFFFFu / (unsigned)pitch + 1u;
  const unsigned m_cw = Cw > 1 ? 0xFFFFFFFFu / (unsigned)Cw + 1u : 0u;
  unsigned slot[IPT];                                   // row in the tile << 20 | column in the map (H * W <= 2^23)
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int idx = tid + i * 256;
    const int r = (int)__umulhi((unsigned)idx, m_pitch);
    const int x = idx - __mul24(r, pitch);
    slot[i] = ((unsigned)r << 20) | (unsigned)min(fx0 + x, xmax);
  }
  const char* const b0 = reinterpret_cast<const char*>(fimg + (size_t)c0 * plane);
  const char* const b1 = b0 + plane * 4;
  const char* const b2 = b1 + plane * 4;
  const char* const b3 = b2 + plane * 4;
  char* const o0 = reinterpret_cast<char*>(a.out + ((size_t)k * a.C + c0) * PP);
  char* const o1 = o0 + PP * 4;
  char* const o2 = o1 + PP * 4;
  char* const o3 = o2 + PP * 4;
  // A band's tile: rows fy0 .. fy0 + FH - 1 of the map (first / last sample of the band: the expressions of
  // axis_sample), nsl groups of 256 slots.
  struct Band { int R, fy0, FH, ymax, nsl; };
  auto band_of = [&](int ph0) {
    Band g;
    g.R = min(rows_per_band, P - ph0);
    const float yf = sh + (float)ph0 * bh + 0.5f * bh / (float)gh;
    const float yl = sh + (float)(ph0 + g.R - 1) * bh + ((float)(gh - 1) + 0.5f) * bh / (float)gh;
    g.fy0 = min((int)fmaxf(fminf(yf, yl), 0.f), Hl - 1);
    const int ylast = min((int)fmaxf(fmaxf(yf, yl), 0.f), Hl - 1);
    g.FH = ylast + pad - g.fy0 + 1;
    g.ymax = min(ylast + 1, Hl - 1);
    g.nsl = (min(g.FH * pitch, TB) + 255) >> 8;            // FH * pitch <= TB by the choice of rows_per_band
    return g;
  };
  // every slot's offset lies inside the map (rows past the tile or the map repeat row ymax, columns are clamped)
  // and all TB slots exist in LDS: whole groups are skipped, nothing is predicated
  float pf[IPT][4];
  auto fetch = [&](const Band& g) {
#pragma unroll
    for (int i = 0; i < IPT; ++i)
      if (i < g.nsl) {
        const int gy = min(g.fy0 + (int)(slot[i] >> 20), g.ymax);
        const unsigned vo = ((unsigned)__mul24(gy, Wl) + (slot[i] & 0xFFFFFu)) << 2;
        pf[i][0] = *reinterpret_cast<const float*>(b0 + vo);
        pf[i][1] = *reinterpret_cast<const float*>(b1 + vo);
        pf[i][2] = *reinterpret_cast<const float*>(b2 + vo);
        pf[i][3] = *reinterpret_cast<const float*>(b3 + vo);
      }
  };
  // The bands of a RoI used to be a chain of dependent round trips (fetch, stage, compute, fetch ...): the next
  // band's fetch is now in flight while this band's bins are computed and stored.
  Band g = band_of(0);
  fetch(g);
  // (the first band is on its way while one wave builds the stencil table)
  float* tab = reinterpret_cast<float*>(lds + TB);     // [2][P][8]: {L, W0 .. WG}
  if (G > 0 && pw0 == 0) {                                        // (same table for every column block of the RoI)
    if (tid < 2 * P) {
      const bool xa = tid >= P;
      const int p = xa ? tid - P : tid;
      int L;
      float Wt[GG + 1];
      axis_stencil<GG>(xa ? sw : sh, xa ? bw : bh, xa ? gw : gh, p, xa ? Wl : Hl, L, Wt);
      float* e = tab + tid * 8;
      e[0] = __int_as_float(L);
#pragma unroll
      for (int r = 0; r <= GG; ++r) e[1 + r] = xa ? Wt[r] : Wt[r] * inv_count;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS only: the fetch stays in flight)
  }
  for (int ph0 = 0; ph0 < P; ph0 += rows_per_band) {
    const int R = g.R, fy0 = g.fy0, FH = g.FH;
#pragma unroll
    for (int i = 0; i < IPT; ++i)
      if (i < g.nsl) lds[tid + i * 256] = make_float4(pf[i][0], pf[i][1], pf[i][2], pf[i][3]);
    __syncthreads();
    if (ph0 + rows_per_band < P) {
      g = band_of(ph0 + rows_per_band);
      fetch(g);
    }
    // Full-width bands are one contiguous run of every channel plane: lanes are dealt by absolute position, so that a
    // wave's 64 dwords are a 256-byte-aligned piece (4 full 64-byte write requests instead of 5 with two partial
    // ones: the L2 counted 26 % more write requests than the output has 64-byte pieces)
    const int skew = Cw == P ? (int)(((reinterpret_cast<uintptr_t>(o0) >> 2) + (unsigned)__mul24(ph0, P)) & 63u) : 0;
    for (int j = tid; j < R * Cw + skew; j += 256) {
      const int i = j - skew;
      if (i < 0) continue;
      const int pr = Cw > 1 ? (int)__umulhi((unsigned)i, m_cw) : i;
      const int ph = ph0 + pr, pw = pw0 + i - __mul24(pr, Cw);
      const unsigned oo = (unsigned)(__mul24(ph, P) + pw) << 2;        // byte offset inside a channel plane
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (G > 0) {
        const float* ey = tab + ph * 8;
        const float* ex = tab + (P + pw) * 8;
        const int Ly = __float_as_int(ey[0]), Lx = __float_as_int(ex[0]);
        const float4* tq = lds + __mul24(min(max(Ly - fy0, 0), FH - S), pitch) + min(max(Lx - fx0, 0), pitch - S);
        float Wx[S];
#pragma unroll
        for (int c = 0; c < S; ++c) Wx[c] = ex[1 + c];
#pragma unroll
        for (int r = 0; r < S; ++r) {
          const float4* tr = tq + r * pitch;
          const float wy = ey[1 + r];
#pragma unroll
          for (int c = 0; c < S; ++c) {
            const float4 v = tr[c];
            const float wv = wy * Wx[c];
            acc.x += wv * v.x;
            acc.y += wv * v.y;
            acc.z += wv * v.z;
            acc.w += wv * v.w;
          }
        }
      } else {
        for (int iy = 0; iy < gh; ++iy) {
          int lo, hi;
          float yl_w, yh_w;
          axis_sample(sh, bh, gh, ph, iy, Hl, lo, hi, yl_w, yh_w);
          const int yo = min(max(lo - fy0, 0), FH - 2) * pitch;
          yl_w *= inv_count;
          yh_w *= inv_count;
          for (int ix = 0; ix < gw; ++ix) {
            float xl_w, xh_w;
            axis_sample(sw, bw, gw, pw, ix, Wl, lo, hi, xl_w, xh_w);
            const int xo = min(max(lo - fx0, 0), pitch - 2);
            const float4* tp = lds + yo + xo;
            const float4 v1 = tp[0], v2 = tp[1], v3 = tp[pitch], v4 = tp[pitch + 1];
            const float w0 = yl_w * xl_w, w1 = yl_w * xh_w, w2 = yh_w * xl_w, w3 = yh_w * xh_w;
            acc.x += w0 * v1.x + w1 * v2.x + w2 * v3.x + w3 * v4.x;
            acc.y += w0 * v1.y + w1 * v2.y + w2 * v3.y + w3 * v4.y;
            acc.z += w0 * v1.z + w1 * v2.z + w2 * v3.z + w3 * v4.z;
            acc.w += w0 * v1.w + w1 * v2.w + w2 * v3.w + w3 * v4.w;
          }
        }
      }
      *reinterpret_cast<float*>(o0 + oo) = acc.x;
      *reinterpret_cast<float*>(o1 + oo) = acc.y;
      *reinterpret_cast<float*>(o2 + oo) = acc.z;
      *reinterpret_cast<float*>(o3 + oo) = acc.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

template <int TB>
__device__ __forceinline__ void band_plan(const RoiArgs& a, int P, float bh, float bw, int gh, int gw, int& ncb, int& Cwb,
                                          int& R) {
  const int G = max(gh, gw);
  const bool merged = G <= 4 && bh <= (float)gh && bw <= (float)gw;
  const int pad = merged ? G : 1;
  const float abh = fmaxf(fabsf(bh), 1e-6f), abw = fmaxf(fabsf(bw), 1e-6f);
  // fewest column blocks (of equal width, a multiple of 4 columns where P allows) whose tiles hold >= 2 output rows
  ncb = 1; Cwb = P; R = 0;
  for (int n = 1; n <= P; n *= 2) {
    const int cw = ((P + n - 1) / n + 3) & ~3;
    const int pitch = (int)floorf((float)cw * abw) + pad + 3;          // >= the tile pitch of any block of cw columns
    const int fh_max = TB / pitch;
    const int r = (int)fminf((float)P, floorf((float)(fh_max - pad - 2) / abh));
    if (r > R) { R = r; ncb = (P + cw - 1) / cw; Cwb = cw; }
    if (r >= min(P, 2) || cw <= 4) break;
  }
}

template <int TB, int WPC>
__global__ __launch_bounds__(256, WPC) void roi_align_band_kernel(RoiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int chunks = (a.C + a.CT - 1) / a.CT;
  int k, chunk;
  roi_unit(a, chunks, k, chunk);
  const int P = a.P, PP = P * P;
  const int c0 = chunk * a.CT;
  const int c1 = min(c0 + a.CT, a.C);
  const float* r = a.rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  const int lvl = (a.L > 1) ? roi_level(x1, y1, x2, y2, a.finest, a.L) : 0;
  if (a.levels && chunk == 0 && threadIdx.x == 0) a.levels[k] = lvl;
  const bool bad_batch = (b < 0 || b >= a.B);
  int Hl = a.H[0], Wl = a.W[0];
  float sc = a.scale[0];
  const float* flvl = a.feat[0];
#pragma unroll
  for (int l = 1; l < DM_MAX_LEVELS; ++l)
    if (lvl == l) {
      Hl = a.H[l];
      Wl = a.W[l];
      sc = a.scale[l];
      flvl = a.feat[l];
    }
  const float sw = x1 * sc - 0.5f, sh = y1 * sc - 0.5f;
  const float ew = x2 * sc - 0.5f, eh = y2 * sc - 0.5f;
  const float rw = ew - sw, rh = eh - sh;
  const float bh = rh / (float)P, bw = rw / (float)P;
  const int gh = a.sr > 0 ? a.sr : (int)ceilf(rh / (float)P);
  const int gw = a.sr > 0 ? a.sr : (int)ceilf(rw / (float)P);
  const float inv_count = 1.0f / (float)max(gh * gw, 1);
  const float* fimg = flvl + (bad_batch ? 0 : (size_t)b * a.C * Hl * Wl);
  if (gh <= 0 || gw <= 0 || bad_batch) {
    for (int i = threadIdx.x; i < (c1 - c0) * PP; i += blockDim.x) a.out[((size_t)k * a.C + c0) * PP + i] = 0.f;
    return;
  }
  {
    const int G = max(gh, gw);
    const bool merged = G <= 4 && bh <= (float)gh && bw <= (float)gw;
    const int pad = merged ? G : 1;
    int ncb, Cwb, R;
    band_plan<TB>(a, P, bh, bw, gh, gw, ncb, Cwb, R);
    if (ncb == 1) {
      // the whole width in one tile: as many rows per band as the tile holds, or a fraction of that (a.order >= 2,
      // experiments: bands of 1/2 .. 1/4 of the height measure the same as full ones, within the noise)
      const float xf = sw + 0.5f * bw / (float)gw;
      const float xl = sw + (float)(P - 1) * bw + ((float)(gw - 1) + 0.5f) * bw / (float)gw;
      const int fx0 = min((int)fmaxf(fminf(xf, xl), 0.f), Wl - 1);
      const int xlast = min((int)fmaxf(fmaxf(xf, xl), 0.f), Wl - 1);
      const int pitch = xlast + pad - fx0 + 1;
      const int fh_max = TB / pitch;
      const float abh = fmaxf(fabsf(bh), 1e-6f);
      R = (int)fminf((float)P, floorf((float)(fh_max - pad - 2) / abh));
      const int dv = a.order >= 2 ? a.order - 1 : 1;
      const int R2 = (int)fminf((float)P, floorf((float)(fh_max / dv - pad - 2) / abh));
      if (R2 >= 4 && dv > 1) R = R2;
    }
    if (R >= 1) {
      for (int cb = 0; cb < ncb; ++cb) {
        const int pw0 = cb * Cwb, Cw = min(Cwb, P - pw0);
        const float xf = sw + (float)pw0 * bw + 0.5f * bw / (float)gw;
        const float xl = sw + (float)(pw0 + Cw - 1) * bw + ((float)(gw - 1) + 0.5f) * bw / (float)gw;
        const int fx0 = min((int)fmaxf(fminf(xf, xl), 0.f), Wl - 1);
        const int xlast = min((int)fmaxf(fmaxf(xf, xl), 0.f), Wl - 1);
        const int pitch = xlast + pad - fx0 + 1;
        const int xmax = min(xlast + 1, Wl - 1);
        if ((R * fmaxf(fabsf(bh), 1e-6f) + (float)(pad + 2)) * (float)pitch > (float)TB) { R = 0; break; }   // (bounds of band_plan: never)
#define DM_ROI_BAND(GG) roi_band_fwd<GG, TB>(a, fimg, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, k, c0, fx0, pitch, xmax, R, pw0, Cw, lds4)
        if (!merged) DM_ROI_BAND(0);
        else if (G == 1) DM_ROI_BAND(1);
        else if (G == 2) DM_ROI_BAND(2);
        else if (G == 3) DM_ROI_BAND(3);
        else DM_ROI_BAND(4);
#undef DM_ROI_BAND
      }
      if (R >= 1) return;
    }
  }
  for (int pos = threadIdx.x; pos < PP; pos += blockDim.x) {
    const int ph = pos / P;
    const int pw = pos - ph * P;
    roi_bin_generic<false>(a, fimg, nullptr, Hl, Wl, sh, sw, bh, bw, gh, gw, inv_count, ph, pw, k, c0, c1);
  }
}

// ---------------------------------------------------------------------------
// Adjoint of the RoIAlign in GATHER form (round 5), for output grids up to 64 x 64.
// Why: the training step extracted [256, 256, 56, 56] = 822 MB of P2 for MaskPre (K2) and read it twice more (conv1 and
// its weight gradient).  conv1 is 1x1 and RoIAlign linear, so conv1 runs on the P2 map instead (12x fewer pixels) and 128
// channels are extracted; the price is that conv1's weight gradient then needs the ADJOINT of the 56 x 56 extraction,
// G = A^T g_y1 on the map.  The scatter kernel below (roi_align_kernel<true>) issues one float atomic per sample tap:
// 103 M outputs x ~12 taps = 1.2 G atomics = 3.8 ms at the memory-side atomic rate -- which is why round 2 kept the
// commutation for inference only.
// Here a workgroup owns (RoI, channel quad): the RoI's gradient planes sit in LDS as [pixel] float4, and a thread owns a
// CELL of the RoI's footprint on the map.  The RoIAlign is separable, out = Ay . M . Ax^T with banded Ay, Ax (a sample
// touches the two pixels around it), so cell (Y, X) sums wy * wx * g[py][px] over the samples whose low or high tap is Y
// (resp. X): along an axis the sample coordinate grows with the sample index, so those are two runs of consecutive
// samples, found by a lower-bound search once per footprint row / column.  The sum is formed in registers in a fixed
// order and leaves with ONE float atomic per cell and channel (RoIs overlap on the map): 167 M for the training step's
// 256 RoIs x 128 channels instead of 1.2 G.
// Samples per axis beyond kAdjMaxSamples (P * g: g > 6 at P = 56) and sampling grids of non-positive size take the
// per-tap atomics of the generic path inside this kernel.
constexpr int kAdjMaxSamples = 384;

__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(RoiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int P = a.P, PP = P * P;
  const int chunks = a.C >> 2;
  const int k = blockIdx.x / chunks;
  const int c0 = (blockIdx.x - k * chunks) << 2;
  const int tid = threadIdx.x;
  const float* r = a.rois + (size_t)k * 5;
  const int b = (int)r[0];
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  const int lvl = (a.L > 1) ? roi_level(x1, y1, x2, y2, a.finest, a.L) : 0;
  if (b < 0 || b >= a.B) return;
  int Hl = a.H[0], Wl = a.W[0];
  float sc = a.scale[0];
  float* glvl = a.gfeat[0];
#pragma unroll
  for (int l = 1; l < DM_MAX_LEVELS; ++l)
    if (lvl == l) {
      Hl = a.H[l];
      Wl = a.W[l];
      sc = a.scale[l];
      glvl = a.gfeat[l];
    }
  const float sw = x1 * sc - 0.5f, sh = y1 * sc - 0.5f;
  const float ew = x2 * sc - 0.5f, eh = y2 * sc - 0.5f;
  const float rw = ew - sw, rh = eh - sh;
  const float bh = rh / (float)P, bw = rw / (float)P;
  const int gh = a.sr > 0 ? a.sr : (int)ceilf(rh / (float)P);
  const int gw = a.sr > 0 ? a.sr : (int)ceilf(rw / (float)P);
  if (gh <= 0 || gw <= 0) return;                       // (the forward wrote zeros: no gradient)
  // a box that is not finite, or so large that its sampling grid exceeds 4096 samples per bin and axis (229 000 feature
  // pixels at P = 56: no clipped proposal comes near): (int)ceilf(inf) is not defined, P * g overflows an int and slips
  // past the table test below, and the per-tap path would loop over g * g samples per bin without end -- no gradient
  // for it.  Written so that a NaN fails the test.
  if (!(fabsf(rh) <= 4096.f * (float)P) || !(fabsf(rw) <= 4096.f * (float)P)) return;
  const float inv_count = 1.0f / (float)(gh * gw);
  const size_t plane = (size_t)Hl * Wl;
  float* const gimg = glvl + ((size_t)b * a.C + c0) * plane;
  const float* const go = a.gout + ((size_t)k * a.C + c0) * PP;
  const long long ny64 = (long long)P * gh, nx64 = (long long)P * gw;
  const int ny = (int)min(ny64, (long long)kAdjMaxSamples + 1), nx = (int)min(nx64, (long long)kAdjMaxSamples + 1);

  // the RoI's four gradient planes -> LDS [pixel] float4
  float4* const gq = lds4;
  for (int i = tid; i < PP; i += 256) gq[i] = make_float4(go[i], go[PP + i], go[2 * PP + i], go[3 * PP + i]);

  // generic path: one atomic per tap (huge sampling grids; a fixed sampling_ratio on a box of non-positive size or with
  // samples more than a pixel apart)
  auto generic = [&]() {
    for (int pos = tid; pos < PP; pos += 256) {
      const int ph = pos / P, pw = pos - ph * P;
      const float4 g4 = gq[pos];
      for (int iy = 0; iy < gh; ++iy) {
        int yl, yh;
        float wyl, wyh;
        axis_sample(sh, bh, gh, ph, iy, Hl, yl, yh, wyl, wyh);
        for (int ix = 0; ix < gw; ++ix) {
          int xl, xh;
          float wxl, wxh;
          axis_sample(sw, bw, gw, pw, ix, Wl, xl, xh, wxl, wxh);
          const float w[4] = {wyl * wxl * inv_count, wyl * wxh * inv_count, wyh * wxl * inv_count, wyh * wxh * inv_count};
          const int off[4] = {yl * Wl + xl, yl * Wl + xh, yh * Wl + xl, yh * Wl + xh};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (w[t] == 0.f) continue;
            atomicAdd(gimg + off[t], w[t] * g4.x);
            atomicAdd(gimg + plane + off[t], w[t] * g4.y);
            atomicAdd(gimg + 2 * plane + off[t], w[t] * g4.z);
            atomicAdd(gimg + 3 * plane + off[t], w[t] * g4.w);
          }
        }
      }
    }
  };
  // The tables below rely on the sample coordinate growing with the sample index.  With the adaptive grid (sr == 0) a
  // bin holds g = ceil(bin) samples, so consecutive samples are bin / g in (0.5, 1] pixels apart: monotone in fp32.  A
  // FIXED sampling_ratio on a tiny box puts them bin / g apart, and below an ulp of the coordinate (~3e-5 px on P2) the
  // last sample of bin p can round above the first of bin p + 1 across a pixel boundary -- taps would be dropped or
  // counted twice.  Such boxes take the per-tap path.
  const bool dense_fixed = a.sr > 0 && (!(bh > 1e-3f * (float)gh) || !(bw > 1e-3f * (float)gw));
  if (ny > kAdjMaxSamples || nx > kAdjMaxSamples || !(bh > 0.f) || !(bw > 0.f) || dense_fixed) {
    __syncthreads();
    generic();
    return;
  }

  // per-axis sample tables: low tap (monotone in the sample index; a void sample keeps the order and has zero weights),
  // high tap, the two weights (the y weights carry 1 / (gh * gw))
  int* const lo_y = reinterpret_cast<int*>(gq + PP);
  int* const hi_y = lo_y + kAdjMaxSamples;
  float* const wl_y = reinterpret_cast<float*>(hi_y + kAdjMaxSamples);
  float* const wh_y = wl_y + kAdjMaxSamples;
  int* const lo_x = reinterpret_cast<int*>(wh_y + kAdjMaxSamples);
  int* const hi_x = lo_x + kAdjMaxSamples;
  float* const wl_x = reinterpret_cast<float*>(hi_x + kAdjMaxSamples);
  float* const wh_x = wl_x + kAdjMaxSamples;
  // entry lists by cell: (output row / column, weight), the entries of a cell contiguous; first entry of a cell in off_*
  int* const ep_y = reinterpret_cast<int*>(wh_x + kAdjMaxSamples);
  float* const ew_y = reinterpret_cast<float*>(ep_y + 2 * kAdjMaxSamples);
  int* const ep_x = reinterpret_cast<int*>(ew_y + 2 * kAdjMaxSamples);
  float* const ew_x = reinterpret_cast<float*>(ep_x + 2 * kAdjMaxSamples);
  int* const off_y = reinterpret_cast<int*>(ew_x + 2 * kAdjMaxSamples);      // [Hf + 1]
  int* const off_x = off_y + (kAdjMaxSamples + 2);                            // [Wf + 1]  (a footprint has at most n + 1 rows)
  for (int s_ = tid; s_ < ny + nx; s_ += 256) {
    const bool xa = s_ >= ny;
    const int s = xa ? s_ - ny : s_;
    const int g = xa ? gw : gh, size = xa ? Wl : Hl;
    const int p = s / g, i = s - p * g;
    int lo, hi;
    float wl, wh;
    axis_sample(xa ? sw : sh, xa ? bw : bh, g, p, i, size, lo, hi, wl, wh);
    if (wl == 0.f && wh == 0.f) {
      // void sample (axis_sample reports taps 0, 0): keep its place in the order -- before the map or after it
      const float c = (xa ? sw : sh) + (float)p * (xa ? bw : bh) + ((float)i + 0.5f) * (xa ? bw : bh) / (float)g;
      lo = hi = c < 0.f ? 0 : size - 1;
    }
    if (xa) {
      lo_x[s] = lo; hi_x[s] = hi; wl_x[s] = wl; wh_x[s] = wh;
    } else {
      lo_y[s] = lo; hi_y[s] = hi; wl_y[s] = wl * inv_count; wh_y[s] = wh * inv_count;
    }
  }
  __syncthreads();
  const int Y0 = lo_y[0], Hf = hi_y[ny - 1] - Y0 + 1;
  const int X0 = lo_x[0], Wf = hi_x[nx - 1] - X0 + 1;
  if (Hf > kAdjMaxSamples + 1 || Wf > kAdjMaxSamples + 1) {      // (samples more than a pixel apart: a fixed sampling_ratio)
    generic();
    return;
  }
  // first entry of each footprint row / column: the samples whose low tap lies below it + those whose high tap does
  auto lower = [](const int* t, int n, int v) {      // first index with t[i] >= v
    int l = 0, h = n;
    while (l < h) {
      const int m = (l + h) >> 1;
      if (t[m] < v) l = m + 1; else h = m;
    }
    return l;
  };
  for (int c_ = tid; c_ < Hf + 1 + Wf + 1; c_ += 256) {
    const bool xa = c_ >= Hf + 1;
    const int c = xa ? c_ - (Hf + 1) : c_;
    const int v = (xa ? X0 : Y0) + c;
    const int* lo = xa ? lo_x : lo_y;
    const int* hi = xa ? hi_x : hi_y;
    const int n = xa ? nx : ny;
    const int a0 = lower(lo, n, v), a1 = lower(lo, n, v + 1), h0 = lower(hi, n, v), h1 = lower(hi, n, v + 1);
    (xa ? off_x : off_y)[c] = a0 + h0;
    const int last = xa ? Wf : Hf;
    if (c < last) {
      int* ep = xa ? ep_x : ep_y;
      float* ew = xa ? ew_x : ew_y;
      const float* wl = xa ? wl_x : wl_y;
      const float* wh = xa ? wh_x : wh_y;
      const int g = xa ? gw : gh;
      int e = a0 + h0;
      for (int s = h0; s < h1; ++s, ++e) { ep[e] = s / g; ew[e] = wh[s]; }      // high taps first: the lower sample indices
      for (int s = a0; s < a1; ++s, ++e) { ep[e] = s / g; ew[e] = wl[s]; }
    }
  }
  __syncthreads();
  const int cells = Hf * Wf;
  const unsigned m_wf = Wf > 1 ? 0xFFFFFFFFu / (unsigned)Wf + 1u : 0u;
  for (int cell = tid; cell < cells; cell += 256) {
    const int yy = Wf > 1 ? (int)__umulhi((unsigned)cell, m_wf) : cell;
    const int xx = cell - yy * Wf;
    const int ey0 = off_y[yy], ey1 = off_y[yy + 1], ex0 = off_x[xx], ex1 = off_x[xx + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ey = ey0; ey < ey1; ++ey) {
      const float wy = ew_y[ey];
      const float4* row = gq + ep_y[ey] * P;
      float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ex = ex0; ex < ex1; ++ex) {
        const float wx = ew_x[ex];
        const float4 v = row[ep_x[ex]];
        racc.x += wx * v.x;
        racc.y += wx * v.y;
        racc.z += wx * v.z;
        racc.w += wx * v.w;
      }
      acc.x += wy * racc.x;
      acc.y += wy * racc.y;
      acc.z += wy * racc.z;
      acc.w += wy * racc.w;
    }
    float* const o = gimg + (size_t)(Y0 + yy) * Wl + (X0 + xx);
    atomicAdd(o, acc.x);
    atomicAdd(o + plane, acc.y);
    atomicAdd(o + 2 * plane, acc.z);
    atomicAdd(o + 3 * plane, acc.w);
  }
}

// LDS bytes of roi_align_bwd_gather_kernel: the gradient quads + the tables above
static inline size_t roi_adj_lds_bytes(int P) {
  return (size_t)P * P * 16 + (size_t)kAdjMaxSamples * 4 * (8 + 8) + (size_t)(kAdjMaxSamples + 2) * 4 * 2;
}

// The two knobs of the forward launchers, read from the environment ONCE (first launch) and clamped;
// dm_reload_env_knobs() re-reads them (tools that sweep settings in one process).  Neither changes a result.
// (Rounds 3-4 had eleven: chunk size, workgroup orders, nontemporal stores, the units / plan / persistent kernels.  What
// they measured is in docs/HISTORY.md and profiles/r04_roi_*; the settings that won are the constants below.)
struct RoiKnobs {
  int sort;         // DM_ROI_SORT: 1 (default) = order the RoIs by level and position when the caller passes a workspace: the
                    // extraction drops from 57 to 47 us, the ordering launch in front of it costs 4-5 back (50.7 us in a graph of 20)
  int sort_min;     // DM_ROI_SORT_MIN: fewest RoIs worth the extra launch (default 192)
};
RoiKnobs g_roi_knobs;
bool g_roi_knobs_loaded = false;

int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const int v = atoi(e);
  return v < lo ? lo : (v > hi ? hi : v);
}

void roi_load_knobs() {
  RoiKnobs k;
  k.sort = env_int("DM_ROI_SORT", 1, 0, 1);
  k.sort_min = env_int("DM_ROI_SORT_MIN", 192, 1, 1024);
  g_roi_knobs = k;
  g_roi_knobs_loaded = true;
}

const RoiKnobs& roi_knobs() {
  if (!g_roi_knobs_loaded) roi_load_knobs();      // (a race between two host threads repeats an idempotent load)
  return g_roi_knobs;
}

int fill_args(RoiArgs& a, const int* H, const int* W, const float* spatial_scales, int num_levels, int B, int C,
              const float* rois, int N, int P, int sampling_ratio, float finest_scale) {
  if (!H || !W || !spatial_scales || (!rois && N > 0)) return DM_ERR_INVALID_ARG;
  if (num_levels < 1 || num_levels > DM_MAX_LEVELS || B <= 0 || C <= 0 || N < 0 || P <= 0 || sampling_ratio < 0)
    return DM_ERR_INVALID_ARG;
  a.L = num_levels; a.B = B; a.C = C; a.rois = rois; a.N = N; a.P = P; a.sr = sampling_ratio;
  a.finest = finest_scale;
  for (int l = 0; l < DM_MAX_LEVELS; ++l) {
    a.H[l] = l < num_levels ? H[l] : 0;
    a.W[l] = l < num_levels ? W[l] : 0;
    a.scale[l] = l < num_levels ? spatial_scales[l] : 0.f;
    a.feat[l] = nullptr;
    a.gfeat[l] = nullptr;
    if (l < num_levels && (H[l] <= 0 || W[l] <= 0)) return DM_ERR_INVALID_ARG;
  }
  // channels per workgroup: enough workgroups to fill 256 CUs several times over
  a.CT = (P * P >= 1024) ? 4 : 16;
  a.lds_floats = 0;
  a.order = 0;
  a.abl = 0;
  a.out = nullptr; a.gout = nullptr; a.levels = nullptr;
  return DM_OK;
}

}  // namespace

extern "C" int dm_reload_env_knobs(void) {
  roi_load_knobs();
  return DM_OK;
}

namespace {
int roi_align_fwd_impl(const float* const* feats, const int* H, const int* W, const float* spatial_scales,
                       int num_levels, int B, int C, const float* rois, int N, int P, int sampling_ratio,
                       float finest_scale, float* out, int32_t* levels_out, void* workspace, long long workspace_bytes,
                       dm_stream_t stream) {
  RoiArgs a;
  int rc = fill_args(a, H, W, spatial_scales, num_levels, B, C, rois, N, P, sampling_ratio, finest_scale);
  if (rc != DM_OK) return rc;
  if (N == 0) return DM_OK;
  if (!feats || !out) return DM_ERR_INVALID_ARG;
  for (int l = 0; l < num_levels; ++l) {
    if (!feats[l]) return DM_ERR_INVALID_ARG;
    a.feat[l] = feats[l];
  }
  if (N == 0) return DM_OK;
  a.out = out;
  a.levels = levels_out;
  const RoiKnobs& kn = roi_knobs();
  bool tile_ok = P * P <= 256 && P >= 2 && C % 4 == 0;
  for (int l = 0; l < num_levels; ++l) tile_ok = tile_ok && (long long)H[l] * W[l] <= (1 << 23);   // 32-bit staging offsets
  if (tile_ok && kn.sort && P * P >= 128 && workspace && N >= kn.sort_min && N <= kOrderMaxRois && workspace_bytes >= (long long)N * 32 &&
      (((uintptr_t)workspace) & 15) == 0) {
    // RoIs walked by level and position (roi_order_kernel), 32 channels per workgroup, XCD-aware chunk-major order
    float* sorted = reinterpret_cast<float*>(workspace);
    DM_LAUNCH(roi_order_kernel, dim3(dm_ceil_div(N, 256 / kOrderTpr)), dim3(256), 0, (hipStream_t)stream, a, sorted);
    int rco = dm_check_launch();
    if (rco != DM_OK) return rco;
    a.levels = nullptr;
    a.sorted = sorted;
    a.CT = 32;
    a.order = 1;
    int chunks = dm_ceil_div(C, a.CT);
    if (chunks % 8 != 0) a.order = 0;
    DM_LAUNCH(roi_align_tile_kernel, dim3(N * chunks), dim3(256), (kTileFloats4 + 64) * sizeof(float4), (hipStream_t)stream, a);
    return dm_check_launch();
  }
  if (tile_ok) {
    // 16 channels per workgroup in the XCD-aware order (roi_unit): XCD x walks the chunks c = x (mod 8) chunk-major,
    // so the planes it is staging from stay in its L2 across the RoIs that share them.  Same time as round 2's 32
    // channels in launch order (16 .. 256 channels swept at 128 .. 2048 RoIs), 29 % fewer bytes fetched from
    // the fabric (FETCH_SIZE 113.6 -> 81.2 MB per launch, round 3's ceiling measurement).
    a.CT = 16;
    a.order = 1;
    int chunks = dm_ceil_div(C, a.CT);
    if (chunks % 8 != 0) a.order = 0;
    DM_LAUNCH(roi_align_tile_kernel, dim3(N * chunks), dim3(256), (kTileFloats4 + 64) * sizeof(float4), (hipStream_t)stream, a);
    return dm_check_launch();
  }
  bool band_ok = P > 16 && P <= 64 && C % 4 == 0;
  for (int l = 0; l < num_levels; ++l) band_ok = band_ok && (long long)H[l] * W[l] <= (1 << 23) && W[l] < (1 << 20);      // (the band kernel packs a map column into 20 bits)
  if (band_ok) {
    // one channel quad per workgroup (swept 4 .. 32 at 128 RoIs on P2: 0.32 / 0.45 / 0.79 / 1.5 ms): the
    // bands of a large RoI are a long chain of dependent staging round trips, so the parallelism has to
    // come from the number of workgroups
    a.CT = 4;            // (roi_band_fwd is written for exactly one quad)
    const int chunks = dm_ceil_div(C, a.CT);
    a.order = 0;                  // full-height bands in launch order (XCD-aware chunk-major and shorter bands measured slower, round 4)
    // (a 28 KB tile with five workgroups per CU -- 96 VGPRs, 92 bytes of scratch -- measured slower: 202 vs 188 us)
    DM_LAUNCH((roi_align_band_kernel<kTileFloats4, 4>), dim3(N * chunks), dim3(256),
              (kTileFloats4 + kBandTabFloats4) * sizeof(float4), (hipStream_t)stream, a);
    return dm_check_launch();
  }
  // Large output grids (56x56 extraction): planar LDS staging where the footprint fits,
  // direct global taps otherwise; two launches partition the RoIs by sampling-grid class.
  a.CT = (P * P >= 1024) ? 8 : 16;
  a.lds_floats = 6 * 1024;
  const int chunks = dm_ceil_div(C, a.CT);
  DM_LAUNCH((roi_align_kernel<false, 1>), dim3(N * chunks), dim3(256), a.lds_floats * sizeof(float), (hipStream_t)stream, a);
  int rc1 = dm_check_launch();
  if (rc1 != DM_OK) return rc1;
  DM_LAUNCH((roi_align_kernel<false, 2>), dim3(N * chunks), dim3(256), a.lds_floats * sizeof(float), (hipStream_t)stream, a);
  return dm_check_launch();
}
}  // namespace

extern "C" int dm_roi_align_fwd(const float* const* feats, const int* H, const int* W, const float* spatial_scales,
                                int num_levels, int B, int C, const float* rois, int N, int P, int sampling_ratio,
                                float finest_scale, float* out, int32_t* levels_out, dm_stream_t stream) {
  return roi_align_fwd_impl(feats, H, W, spatial_scales, num_levels, B, C, rois, N, P, sampling_ratio, finest_scale, out,
                            levels_out, nullptr, 0, stream);
}

extern "C" long long dm_roi_align_workspace_bytes(int N, int P) {
  if (N <= 0 || P < 2 || P * P > 256) return 0;
  return (long long)N * 32;      // ordered RoI records (roi_order_kernel)
}

extern "C" int dm_roi_align_fwd_ws(const float* const* feats, const int* H, const int* W, const float* spatial_scales,
                                   int num_levels, int B, int C, const float* rois, int N, int P, int sampling_ratio,
                                   float finest_scale, float* out, int32_t* levels_out, void* workspace,
                                   long long workspace_bytes, dm_stream_t stream) {
  if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace)) return DM_ERR_INVALID_ARG;
  return roi_align_fwd_impl(feats, H, W, spatial_scales, num_levels, B, C, rois, N, P, sampling_ratio, finest_scale, out,
                            levels_out, workspace, workspace_bytes, stream);
}

extern "C" int dm_roi_align_bwd(const float* grad_out, float* const* grad_feats, const int* H, const int* W,
                                const float* spatial_scales, int num_levels, int B, int C, const float* rois, int N,
                                int P, int sampling_ratio, float finest_scale, dm_stream_t stream) {
  RoiArgs a;
  int rc = fill_args(a, H, W, spatial_scales, num_levels, B, C, rois, N, P, sampling_ratio, finest_scale);
  if (rc != DM_OK) return rc;
  if (N == 0) return DM_OK;
  if (!grad_feats || !grad_out) return DM_ERR_INVALID_ARG;
  for (int l = 0; l < num_levels; ++l) {
    if (!grad_feats[l]) return DM_ERR_INVALID_ARG;
    a.gfeat[l] = grad_feats[l];
  }
  if (N == 0) return DM_OK;
  a.gout = grad_out;
  if (P > 16 && P <= 64 && C % 4 == 0 && (long long)N * (C / 4) <= 0x7fffffffLL) {
    // larger output grids (the 56 x 56 extraction in front of MaskPre): gather form, one atomic per footprint cell
    static bool raised[DM_MAX_DEVICES] = {false};
    const size_t lds = roi_adj_lds_bytes(P);
    if (dm_ensure_lds_limit(reinterpret_cast<const void*>(&roi_align_bwd_gather_kernel), (int)lds, raised) != DM_OK) return DM_ERR_LAUNCH;
    DM_LAUNCH(roi_align_bwd_gather_kernel, dim3((unsigned)(N * (C / 4))), dim3(256), lds, (hipStream_t)stream, a);
    return dm_check_launch();
  }
  const int chunks = dm_ceil_div(C, a.CT);
  DM_LAUNCH((roi_align_kernel<true, 0>), dim3(N * chunks), dim3(256), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}
