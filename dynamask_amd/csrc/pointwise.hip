// HBM-bound kernels of the mask-head path: point sample (K4), class-gathered
// logits (K7), x2 bilinear upsample (K11), inference boundary merge (K15),
// Gumbel selector (K10), DetailTarget (K13), mask losses (K12).
// All are coalesced along the innermost (x / pixel) dimension; reductions use
// wave shuffles + one atomic per workgroup.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide sum (result valid in thread 0). blockDim.x multiple of 64, <= 1024.
__device__ __forceinline__ float block_sum(float v, float* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (wave == 0) {
    r = (lane < (int)(blockDim.x >> 6)) ? smem[lane] : 0.f;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;
}

// A product that stays a product: hipcc contracts a * b - c into one fma wherever it sees both (-ffp-contract=fast, and
// __fmul_rn is a plain multiplication to it), per kernel as its scheduler likes -- a source coordinate rs * o whose
// fraction is then taken by a subtraction came out an ulp apart in two kernels that must agree.  The pragma clears the
// 'contract' flag of this one multiplication; it survives inlining.
__device__ __forceinline__ float dm_mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// One bilinear x2 value, spelled as an explicit fma chain: the three kernels that interpolate (generic, half-pixel fast
// path, logits of the upsampled stage) must round alike -- left to the compiler, the same source expression contracted
// differently next to different code and the fused exit differed from upsample + logits in the last bit.
__device__ __forceinline__ float dm_up2x_interp(float hy, float ly, float hx, float lx, float a, float b, float c, float d) {
  const float top = __builtin_fmaf(lx, b, hx * a);
  const float bot = __builtin_fmaf(lx, d, hx * c);
  return __builtin_fmaf(ly, bot, hy * top);
}

// ------------------------------------------------------------------ K4
// grid_sample(bilinear, zeros, align_corners=False) at RoI-relative pixel
// centres.  Thread = one sample point; its 4 taps/weights are computed once and
// reused over the channel chunk of the workgroup.
__device__ __forceinline__ void point_sample_body(int bid, const float* __restrict__ feat, int B, int C, int H, int W,
                                                  const float* __restrict__ rois, int N, int S, float scale,
                                                  float* __restrict__ out, int CT) {
  // Thread = one sample point of the flat (RoI, position) list (S*S = 196 would leave a quarter of a 256-thread
  // workgroup idle per RoI); the two taps of a row come as ONE 8-byte load from the pair base column
  // cb = clamp(x0, 0, W-2), with the weights moved to the pair's slots (a tap outside the map keeps weight 0, so the
  // expression below is the reference's four-term sum): half the gather instructions of a load per tap.
  const int chunks = (C + CT - 1) / CT;
  const int chunk = bid % chunks;
  const int pb = bid / chunks;
  const int SS = S * S;
  const long long flat = (long long)pb * blockDim.x + threadIdx.x;
  if (flat >= (long long)N * SS) return;
  const int n = (int)(flat / SS), pos = (int)(flat - (long long)n * SS);
  const int iy = pos / S, ix = pos - iy * S;
  const float* r = rois + (size_t)n * 5;
  const int b = (int)r[0];
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  const int c0 = chunk * CT, c1 = min(c0 + CT, C);
  float* o = out + ((size_t)n * C) * SS + pos;
  if (b < 0 || b >= B) {
    for (int c = c0; c < c1; ++c) o[(size_t)c * SS] = 0.f;
    return;
  }
  // affine_grid(align_corners=False) base coordinate, then (g+1)/2 -> [0,1]
  const float gx0 = ((float)(2 * ix + 1)) / (float)S - 1.0f;
  const float gy0 = ((float)(2 * iy + 1)) / (float)S - 1.0f;
  float px = (gx0 + 1.0f) / 2.0f;
  float py = (gy0 + 1.0f) / 2.0f;
  px = px * (x2 - x1) + x1;            // absolute image point
  py = py * (y2 - y1) + y1;
  px = px / (float)W * scale;          // relative to the feature map
  py = py / (float)H * scale;
  const float gx = px * 2.0f - 1.0f;   // grid_sample coordinate
  const float gy = py * 2.0f - 1.0f;
  const float sx = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
  const float sy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
  const float fx = floorf(sx), fy = floorf(sy);
  // keep the int conversion safe for far-away points (all taps void anyway)
  const float cfx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
  const float cfy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
  const bool far = (cfx != fx) || (cfy != fy);
  const int x0 = (int)cfx, y0 = (int)cfy;
  const int x1i = x0 + 1, y1i = y0 + 1;
  const float lx = sx - fx, ly = sy - fy;
  float w_nw = (1.f - lx) * (1.f - ly), w_ne = lx * (1.f - ly), w_sw = (1.f - lx) * ly, w_se = lx * ly;
  const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1i >= 0 && x1i < W;
  const bool oky0 = y0 >= 0 && y0 < H, oky1 = y1i >= 0 && y1i < H;
  if (far || !okx0 || !oky0) w_nw = 0.f;
  if (far || !okx1 || !oky0) w_ne = 0.f;
  if (far || !okx0 || !oky1) w_sw = 0.f;
  if (far || !okx1 || !oky1) w_se = 0.f;
  const float* f = feat + ((size_t)b * C) * H * W;
  const size_t plane = (size_t)H * W;
  if (W >= 2) {
    struct __attribute__((packed, aligned(4))) F2 { float a, b; };
    const int cb = min(max(x0, 0), W - 2);
    const int rt = min(max(y0, 0), H - 1), rb = min(max(y1i, 0), H - 1);
    // slot a = column cb, slot b = column cb + 1: x0 == cb (inside), x1i == cb (x0 = -1) or x0 == cb + 1 (x0 = W - 1)
    const bool in = (cb == x0);
    const float ta = in ? w_nw : (x1i == cb ? w_ne : 0.f), tb = in ? w_ne : (x0 == cb + 1 ? w_nw : 0.f);
    const float ba = in ? w_sw : (x1i == cb ? w_se : 0.f), bb = in ? w_se : (x0 == cb + 1 ? w_sw : 0.f);
    const int ot = rt * W + cb, ob = rb * W + cb;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
      const float* fc = f + (size_t)c * plane;
      const F2 top = *reinterpret_cast<const F2*>(fc + ot);
      const F2 bot = *reinterpret_cast<const F2*>(fc + ob);
      o[(size_t)c * SS] = top.a * ta + top.b * tb + bot.a * ba + bot.b * bb;
    }
    return;
  }
  const int o_nw = (w_nw != 0.f) ? y0 * W + x0 : 0, o_ne = (w_ne != 0.f) ? y0 * W + x1i : 0;
  const int o_sw = (w_sw != 0.f) ? y1i * W + x0 : 0, o_se = (w_se != 0.f) ? y1i * W + x1i : 0;
  for (int c = c0; c < c1; ++c) {
    const float* fc = f + (size_t)c * plane;
    o[(size_t)c * SS] = fc[o_nw] * w_nw + fc[o_ne] * w_ne + fc[o_sw] * w_sw + fc[o_se] * w_se;
  }
}

__global__ __launch_bounds__(256) void point_sample_kernel(const float* __restrict__ feat, int B, int C, int H, int W,
                                                           const float* __restrict__ rois, int N, int S, float scale,
                                                           float* __restrict__ out, int CT, int pos_blocks) {
  point_sample_body(blockIdx.x, feat, B, C, H, W, rois, N, S, scale, out, CT);
}

// ------------------------------------------------------------------ K7
// A workgroup = 64 pixels of one RoI x four interleaved channel subsets (one per wave); the two weight rows W[label] are
// workgroup-uniform (scalar loads).  The waves' partial sums meet in LDS and are added in wave order.  (Round 2 gave a
// thread a pixel and ALL channels: one 256-thread workgroup per 14 x 14 RoI, 8 waves per CU with four loads in flight
// each -- 1.2 TB/s on a kernel that only streams x.)
__device__ __forceinline__ void class_logits_body(int bid, float (&red)[3][64][2], const float* __restrict__ x, int N, int C, int HW,
                                                  const float* __restrict__ wi, const float* __restrict__ bi,
                                                  const float* __restrict__ wd, const float* __restrict__ bd,
                                                  int num_classes, const int64_t* __restrict__ labels,
                                                  float* __restrict__ inst, float* __restrict__ det,
                                                  float* __restrict__ sig, int sig_ct, int sig_off,
                                                  int pix_blocks) {
  const int n = bid / pix_blocks;
  const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the weights of the channel loop are scalar loads)
  const int p = (bid - n * pix_blocks) * 64 + lane;
  const bool ok = p < HW;
  int lab = (int)labels[n];
  lab = min(max(lab, 0), num_classes - 1);
  const float* wri = wi + (size_t)lab * C;
  const float* wrd = wd + (size_t)lab * C;
  const float* xp = x + (size_t)n * C * HW + min(p, HW - 1);
  float ai = 0.f, ad = 0.f;
#pragma unroll 8
  for (int c = part; c < C; c += 4) {
    const float v = xp[(size_t)c * HW];
    ai += wri[c] * v;
    ad += wrd[c] * v;
  }
  if (part > 0) {
    red[part - 1][lane][0] = ai;
    red[part - 1][lane][1] = ad;
  }
  __syncthreads();
  if (part > 0 || !ok) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ai += red[k][lane][0];
    ad += red[k][lane][1];
  }
  ai += bi[lab];
  ad += bd[lab];
  inst[(size_t)n * HW + p] = ai;
  det[(size_t)n * HW + p] = ad;
  if (sig) {
    sig[((size_t)n * sig_ct + sig_off) * HW + p] = sigmoidf_(ai);
    sig[((size_t)n * sig_ct + sig_off + 1) * HW + p] = sigmoidf_(ad);
  }
}

__global__ __launch_bounds__(256) void class_logits_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                           const float* __restrict__ wi, const float* __restrict__ bi,
                                                           const float* __restrict__ wd, const float* __restrict__ bd,
                                                           int num_classes, const int64_t* __restrict__ labels,
                                                           float* __restrict__ inst, float* __restrict__ det,
                                                           float* __restrict__ sig, int sig_ct, int sig_off,
                                                           int pix_blocks) {
  __shared__ float red[3][64][2];
  class_logits_body(blockIdx.x, red, x, N, C, HW, wi, bi, wd, bd, num_classes, labels, inst, det, sig, sig_ct, sig_off, pix_blocks);
}

// The head of an SFM stage in ONE launch (round 6): the point sample of the stage's semantic map (K4) and the two
// class-gathered logits of its instance features (K7) share no data -- both only feed the fusion convolution behind them
// (dynamask_head.py:104-116).  Workgroups [0, ps_blocks) run the body of point_sample_kernel, the rest that of
// class_logits_kernel: the same code, so the same bits, one launch less per stage in a chain of 5-10 us launches.
struct StageHeadArgs {
  const float* feat; int B, Cs, H, W; const float* rois; int N, S; float scale; float* ps_out; int CT;
  const float* x; int C, HW; const float *wi, *bi, *wd, *bd; int num_classes; const int64_t* labels;
  float *inst, *det, *sig; int sig_ct, sig_off, pix_blocks;
  int ps_blocks;
};
__global__ __launch_bounds__(256) void stage_head_kernel(StageHeadArgs a) {
  __shared__ float red[3][64][2];
  if ((int)blockIdx.x < a.ps_blocks) {
    point_sample_body(blockIdx.x, a.feat, a.B, a.Cs, a.H, a.W, a.rois, a.N, a.S, a.scale, a.ps_out, a.CT);
    return;
  }
  class_logits_body(blockIdx.x - a.ps_blocks, red, a.x, a.N, a.C, a.HW, a.wi, a.bi, a.wd, a.bd, a.num_classes, a.labels, a.inst,
                    a.det, a.sig, a.sig_ct, a.sig_off, a.pix_blocks);
}

// K7 at twice the resolution of its input: logits of relu(upsample2x(x)) without the upsampled tensor (the last stage
// before an exit only feeds the exit's two logit maps: at the fixed 28 x 28 exit the upsampled features were 205 MB written
// by one kernel and read once by the next).  A lane owns two horizontally adjacent INPUT pixels -- a 2 x 4 block of
// outputs, as upsample2x_half_kernel, same expression, so the interpolated values have the same bits; a workgroup = 64
// such items of one RoI x four interleaved channel subsets (one per wave), partial sums added in wave order through LDS.
__global__ __launch_bounds__(256) void class_logits_up2x_kernel(const float* __restrict__ x, int item_blocks, int C, int H, int W,
                                                                const float* __restrict__ wi, const float* __restrict__ bi,
                                                                const float* __restrict__ wd, const float* __restrict__ bd,
                                                                int num_classes, const int64_t* __restrict__ labels,
                                                                float* __restrict__ inst, float* __restrict__ det) {
  __shared__ float red[3][16][64];
  const int n = blockIdx.x / item_blocks;
  const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the channel loop's addresses and weights are uniform)
  const int Wh = W >> 1, items = H * Wh;
  const int it = (blockIdx.x - n * item_blocks) * 64 + lane;
  const bool ok = it < items;
  const int item = min(it, items - 1);
  const int y = item / Wh, xp = item - y * Wh;
  int lab = (int)labels[n];
  lab = min(max(lab, 0), num_classes - 1);
  const float* wri = wi + (size_t)lab * C;
  const float* wrd = wd + (size_t)lab * C;
  const int ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const int x0 = 2 * xp;
  struct __attribute__((packed, aligned(4))) F2 { float a, b; };
  const bool first = x0 == 0, last = x0 + 2 > W - 1;       // column pair at the left / right border of the row
  const int la = first ? 0 : x0 - 1, lb = last ? x0 : x0 + 1;
  // byte offsets of the two 8-byte loads of each row inside a channel plane (scalar plane base + 32-bit lane offset)
  const unsigned oa[3] = {(unsigned)(ym * W + la) * 4u, (unsigned)(y * W + la) * 4u, (unsigned)(yp * W + la) * 4u};
  const unsigned ob[3] = {(unsigned)(ym * W + lb) * 4u, (unsigned)(y * W + lb) * 4u, (unsigned)(yp * W + lb) * 4u};
  float acc[16];      // [branch][dy][e]
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const float* px = x + (size_t)n * C * H * W;
  // a channel's six 8-byte loads (three rows x the column pairs (x0-1, x0) and (x0+1, x0+2), clamped to the row) are
  // issued one channel ahead of the arithmetic that uses them
  auto fetch = [&](int c, F2 (&A)[3], F2 (&B)[3]) {
    const char* p = reinterpret_cast<const char*>(px + (size_t)c * H * W);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      A[r] = *reinterpret_cast<const F2*>(p + oa[r]);
      B[r] = *reinterpret_cast<const F2*>(p + ob[r]);
    }
  };
  auto accumulate = [&](int c, const F2 (&A)[3], const F2 (&B)[3], float on) {
    float v[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      v[r][0] = A[r].a;
      v[r][1] = first ? A[r].a : A[r].b;
      v[r][2] = last ? B[r].b : B[r].a;
      v[r][3] = B[r].b;
    }
    const float wci = wri[c] * on, wcd = wrd[c] * on;      // on = 1, or 0 for the padding half of an odd channel count
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      // (compile-time indices into v[][]: at the top row rows 0 and 1 are the same row of the map, at the left border
      // columns 0 and 1 the same column -- the weight goes to zero instead of the index moving, same operands either way;
      // run-time indices put v[][] in scratch memory, a store and a reload per channel in this loop)
      const int ia = dy == 0 ? 0 : 1, ib = dy == 0 ? 1 : 2;
      float ly = dy == 0 ? 0.75f : 0.25f;
      if (dy == 0 && y == 0) ly = 0.f;
      const float hy = 1.f - ly;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ja = (e == 0) ? 0 : (e == 3 ? 2 : 1);
        const int jb = ja + 1;
        float lx = (e & 1) ? 0.25f : 0.75f;
        if (e == 0 && x0 == 0) lx = 0.f;
        const float hx = 1.f - lx;
        const float r = fmaxf(dm_up2x_interp(hy, ly, hx, lx, v[ia][ja], v[ia][jb], v[ib][ja], v[ib][jb]), 0.f);
        acc[dy * 4 + e] += wci * r;
        acc[8 + dy * 4 + e] += wcd * r;
      }
    }
  };
  // (the prefetch is unconditional -- past the last channel it re-reads the current one: a branch around it, or around
  // the second half, makes the compiler wait for the loads it has just issued, at the join)
  F2 A0[3], B0[3], A1[3], B1[3];
  fetch(min(part, C - 1), A0, B0);
  for (int c = part; c < C; c += 8) {
    const bool two = c + 4 < C;
    const int c1 = two ? c + 4 : c;
    fetch(c1, A1, B1);
    __builtin_amdgcn_sched_barrier(0);             // (or the scheduler sinks the loads to their first use again)
    accumulate(c, A0, B0, 1.f);
    __builtin_amdgcn_sched_barrier(0);
    fetch(c + 8 < C ? c + 8 : c, A0, B0);
    __builtin_amdgcn_sched_barrier(0);
    accumulate(c1, A1, B1, two ? 1.f : 0.f);       // (no branch: the loop body is one basic block, its waits count loads)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (part > 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) red[part - 1][k][lane] = acc[k];
  }
  __syncthreads();
  if (part > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] += red[w][k][lane];
  const float b_i = bi[lab], b_d = bd[lab];
  const size_t o = ((size_t)n * 2 * H + 2 * y) * (size_t)(2 * W) + 4 * xp;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    *reinterpret_cast<dm_f32x4*>(inst + o + (size_t)dy * 2 * W) =
        dm_f32x4{acc[dy * 4] + b_i, acc[dy * 4 + 1] + b_i, acc[dy * 4 + 2] + b_i, acc[dy * 4 + 3] + b_i};
    *reinterpret_cast<dm_f32x4*>(det + o + (size_t)dy * 2 * W) =
        dm_f32x4{acc[8 + dy * 4] + b_d, acc[8 + dy * 4 + 1] + b_d, acc[8 + dy * 4 + 2] + b_d, acc[8 + dy * 4 + 3] + b_d};
  }
}

// ------------------------------------------------------------------ K11
// One thread = 4 consecutive output pixels of a row (one 16-B store); 32-bit index
// math; the two source rows are re-used across the 4 outputs.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, int NC, int H, int W, int ac,
                                                         int relu, float* __restrict__ out) {
  const int OH = 2 * H, OW = 2 * W;          // OW is a multiple of 2; vector path needs OW % 4 == 0
  const int OWq = (OW + 3) >> 2;
  const long long total = (long long)NC * OH * OWq;
  const float rh = ac ? (OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f) : 0.5f;
  const float rw = ac ? (OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f) : 0.5f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int xq = (int)(idx % OWq);
    const long long t = idx / OWq;
    const int oy = (int)(t % OH);
    const long long nc = t / OH;
    const float sy = ac ? dm_mul_rn(rh, (float)oy) : fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);      // (ac: a rounded product, see boundary_merge_kernel)
    const int y0 = (int)sy;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const float* r0 = in + nc * H * W + (long long)y0 * W;
    const float* r1 = in + nc * H * W + (long long)y1 * W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ox = xq * 4 + e;
      const float sx = ac ? dm_mul_rn(rw, (float)ox) : fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
      int x0 = (int)sx;
      x0 = min(x0, W - 1);
      const int x1 = x0 + ((x0 < W - 1) ? 1 : 0);
      const float lx = sx - (float)x0, hx = 1.f - lx;
      float r = dm_up2x_interp(hy, ly, hx, lx, r0[x0], r0[x1], r1[x0], r1[x1]);
      if (relu) r = fmaxf(r, 0.f);
      v[e] = r;
    }
    float* o = out + (nc * OH + oy) * (long long)OW + xq * 4;
    if ((OW & 3) == 0) {
      *reinterpret_cast<dm_f32x4*>(o) = dm_f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (xq * 4 + e < OW) o[e] = v[e];
    }
  }
}

// Half-pixel (align_corners=False) x2 fast path, W even: a thread owns two horizontally adjacent
// input pixels, i.e. a 2 x 4 block of outputs, and reads the 3 x 4 input neighbourhood once
// (1.5 loads per output instead of 4); two 16-byte stores.  The interpolation weights are the
// exact constants 0.25 / 0.75 the generic expression produces, combined in the same order, so
// the results are bit-identical to upsample2x_kernel.
__global__ __launch_bounds__(256) void upsample2x_half_kernel(const float* __restrict__ in, long long NC, int H, int W,
                                                              int relu, float* __restrict__ out) {
  const int Wh = W >> 1;
  const long long total = NC * H * Wh;
  const int OW = 2 * W;
  const bool small = total < (1LL << 31);       // 32-bit index arithmetic (two 64-bit divisions per item cost more than its eight outputs)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int xp, y;
    long long nc;
    if (small) {
      const unsigned u = (unsigned)idx, t = u / (unsigned)Wh;
      xp = (int)(u - t * (unsigned)Wh);
      const unsigned n32 = t / (unsigned)H;
      y = (int)(t - n32 * (unsigned)H);
      nc = n32;
    } else {
      xp = (int)(idx % Wh);
      const long long t = idx / Wh;
      y = (int)(t % H);
      nc = t / H;
    }
    const float* p = in + nc * H * W;
    const int ym = max(y - 1, 0), yp = min(y + 1, H - 1);
    const int x0 = 2 * xp;
    // the four columns (x0-1, x0, x0+1, x0+2, clamped to the row) as two 8-byte loads
    struct __attribute__((packed, aligned(4))) F2 { float a, b; };
    const bool first = x0 == 0, last = x0 + 2 > W - 1;
    const int la = first ? 0 : x0 - 1, lb = last ? x0 : x0 + 1;
    float v[3][4];
    const int rows[3] = {ym, y, yp};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* pr = p + (long long)rows[r] * W;
      const F2 pa = *reinterpret_cast<const F2*>(pr + la);
      const F2 pb = *reinterpret_cast<const F2*>(pr + lb);
      v[r][0] = pa.a;
      v[r][1] = first ? pa.a : pa.b;
      v[r][2] = last ? pb.b : pb.a;
      v[r][3] = pb.b;
    }
    // output column ox = 2*x0 + e: (left tap, right tap, lx); at the borders the generic code clamps
    // the source coordinate to 0 (lx = 0) or repeats the last column (x1 == x0)
    float o[2][4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      // oy = 2y + dy: rows (y-1, y) with ly = 0.75, or (y, y+1) with ly = 0.25; oy = 0 -> ly = 0
      // (compile-time indices: at y = 0 rows 0 and 1 of v are the same row of the map, at x0 = 0 columns 0 and 1 the same
      // column, so the clamped cases only zero the weight -- same operands, no run-time indexing of v[][])
      const int ia = dy == 0 ? 0 : 1, ib = dy == 0 ? 1 : 2;
      float ly = dy == 0 ? 0.75f : 0.25f;
      if (dy == 0 && y == 0) ly = 0.f;                           // sy clamped to 0: y0 = 0
      const float hy = 1.f - ly;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // e: 0 -> (x0-1, x0, .75)  1 -> (x0, x0+1, .25)  2 -> (x0, x0+1, .75)  3 -> (x0+1, x0+2, .25)
        const int ja = (e == 0) ? 0 : (e == 3 ? 2 : 1);
        const int jb = ja + 1;
        float lx = (e & 1) ? 0.25f : 0.75f;
        if (e == 0 && x0 == 0) lx = 0.f;                         // sx clamped to 0
        const float hx = 1.f - lx;
        float r = dm_up2x_interp(hy, ly, hx, lx, v[ia][ja], v[ia][jb], v[ib][ja], v[ib][jb]);
        if (relu) r = fmaxf(r, 0.f);
        o[dy][e] = r;
      }
    }
    float* op = out + (nc * 2 * H + 2 * y) * (long long)OW + 4 * xp;
    *reinterpret_cast<dm_f32x4*>(op) = dm_f32x4{o[0][0], o[0][1], o[0][2], o[0][3]};
    *reinterpret_cast<dm_f32x4*>(op + OW) = dm_f32x4{o[1][0], o[1][1], o[1][2], o[1][3]};
  }
}

// ------------------------------------------------------------------ K15
// One workgroup per RoI.  LDS holds the coarse logits and the non-boundary map.
__global__ __launch_bounds__(256) void boundary_merge_kernel(const float* __restrict__ coarse, float* __restrict__ fine,
                                                             int n, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cl = lds;            // [S*S] coarse logits
  float* nb = lds + S * S;    // [S*S] 1 = not a boundary pixel
  const int r = blockIdx.x;
  const float* c = coarse + (size_t)r * S * S;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) cl[i] = c[i];
  __syncthreads();
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    const int y = i / S, x = i - y * S;
    const bool m = sigmoidf_(cl[i]) >= 0.5f;
    // generate_block_target(boundary_width=1): 3x3 Laplacian on the zero-padded
    // mask (pos) and on 1 - padded mask (neg: padding counts as background=1)
    int sum_in = 0, cnt_in = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < S && xx >= 0 && xx < S) {
          cnt_in++;
          sum_in += (sigmoidf_(cl[yy * S + xx]) >= 0.5f) ? 1 : 0;
        }
      }
    // pos: m && (9*m - sum3x3(mask_padded)) >= 1  <=> m && sum_in < 9
    // neg: !m && (9 - sum3x3(1 - mask_padded)) ... = !m && some in-bounds neighbour is foreground
    const bool pos = m && (sum_in < 9);
    const bool neg = (!m) && (sum_in > 0);
    nb[i] = (pos || neg) ? 0.f : 1.f;
    (void)cnt_in;
  }
  __syncthreads();
  const int OS = 2 * S;
  const float rs = OS > 1 ? (float)(S - 1) / (float)(OS - 1) : 0.f;
  float* f = fine + (size_t)r * OS * OS;
  for (int i = threadIdx.x; i < OS * OS; i += blockDim.x) {
    const int oy = i / OS, ox = i - oy * OS;
    const float sy = dm_mul_rn(rs, (float)oy), sx = dm_mul_rn(rs, (float)ox);      // (rounded products: no fma contraction with the subtraction below, as F.interpolate computes them, and the same bits in every kernel that merges)
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < S - 1) ? 1 : 0), x1 = x0 + ((x0 < S - 1) ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    // (the explicit fma chain of dm_up2x_interp: boundary_merge_chain_kernel below must produce the same bits)
    const float nbv = dm_up2x_interp(hy, ly, hx, lx, nb[y0 * S + x0], nb[y0 * S + x1], nb[y1 * S + x0], nb[y1 * S + x1]);
    if (nbv >= 0.5f) {
      f[i] = dm_up2x_interp(hy, ly, hx, lx, cl[y0 * S + x0], cl[y0 * S + x1], cl[y1 * S + x0], cl[y1 * S + x1]);
    }
  }
}

// K15 for the whole inference tail in ONE launch (round 6): the two dependent merges S -> 2S -> 4S of
// dynamask_roi_head.py:138-149 and, optionally, the final align_corners x2 upsample of the last stage's logits
// (dynamask_head.py:240-243) that produces the 4S x 4S "fine" logits in the first place.  The launch sequence it replaces
// -- upsample2x (instance), upsample2x (detail, unused by inference), boundary_merge(S), boundary_merge(2S) -- ran one
// workgroup per RoI: 34 us for the 112 x 112 merge whether 16 or 100 RoIs, behind the join of the RoI streams.
// Here a workgroup owns a band of MC_ROWS output rows of one RoI and recomputes what the band depends on: the
// coarsest logits whole (S x S: 784 floats), the merged 2S x 2S rows its outputs interpolate between plus one halo row
// either side (the 3 x 3 boundary stencil), nothing else.  The merged 2S x 2S logits are NOT written back: the
// reference overwrites them in place, but they are temporaries there (only the 4S x 4S result is used).
// Every value is computed by the expressions of boundary_merge_kernel / upsample2x_kernel, operand for operand: the
// result has the bits of the four launches it replaces.
constexpr int MC_ROWS = 16;
__device__ __forceinline__ bool mc_boundary(const float* __restrict__ cl, int rows_lo, int rows_hi, int row_off, int S, int y, int x) {
  // generate_block_target(boundary_width=1) on the mask sigmoid(cl) >= 0.5 (see boundary_merge_kernel); rows of the image
  // are [rows_lo, rows_hi), the LDS copy starts at image row row_off
  const bool m = sigmoidf_(cl[(y - row_off) * S + x]) >= 0.5f;
  int sum_in = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy >= rows_lo && yy < rows_hi && xx >= 0 && xx < S) sum_in += (sigmoidf_(cl[(yy - row_off) * S + xx]) >= 0.5f) ? 1 : 0;
    }
  return (m && sum_in < 9) || (!m && sum_in > 0);
}

__global__ __launch_bounds__(256) void boundary_merge_chain_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                                   const float* __restrict__ fin2, float* __restrict__ out4,
                                                                   int n, int S, int bands) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int S2 = 2 * S, S4 = 4 * S;
  const int r = blockIdx.x / bands, band = blockIdx.x - r * bands;
  const int oy0 = band * MC_ROWS, oy1 = min(oy0 + MC_ROWS, S4);
  const float rs1 = S2 > 1 ? (float)(S - 1) / (float)(S2 - 1) : 0.f;
  const float rs2 = S4 > 1 ? (float)(S2 - 1) / (float)(S4 - 1) : 0.f;
  // rows of the 2S grid the band's outputs interpolate between, and one halo row either side for their 3 x 3 stencils
  const int ya = (int)dm_mul_rn(rs2, (float)oy0);
  const int yl = (int)dm_mul_rn(rs2, (float)(oy1 - 1));
  const int yb = yl + ((yl < S2 - 1) ? 1 : 0);
  const int ra = max(ya - 1, 0), rb = min(yb + 1, S2 - 1);
  const int mrows = rb - ra + 1, nrows = yb - ya + 1;
  float* cl1 = lds;                       // [S * S]      coarsest logits
  float* nb1 = cl1 + S * S;               // [S * S]      1 = not a boundary pixel
  float* m2 = nb1 + S * S;                // [mrows * 2S] merged 2S logits, image rows ra .. rb
  float* nb2 = m2 + (MC_ROWS / 2 + 5) * S2;  // [nrows * 2S] rows ya .. yb
  const float* c1 = p1 + (size_t)r * S * S;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) cl1[i] = c1[i];
  __syncthreads();
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    const int y = i / S, x = i - y * S;
    nb1[i] = mc_boundary(cl1, 0, S, 0, S, y, x) ? 0.f : 1.f;
  }
  __syncthreads();
  const float* f2 = p2 + (size_t)r * S2 * S2;
  for (int i = threadIdx.x; i < mrows * S2; i += blockDim.x) {
    const int yy = i / S2, ox = i - yy * S2, oy = ra + yy;
    const float sy = dm_mul_rn(rs1, (float)oy), sx = dm_mul_rn(rs1, (float)ox);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < S - 1) ? 1 : 0), x1 = x0 + ((x0 < S - 1) ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float nbv = dm_up2x_interp(hy, ly, hx, lx, nb1[y0 * S + x0], nb1[y0 * S + x1], nb1[y1 * S + x0], nb1[y1 * S + x1]);
    float v = f2[oy * S2 + ox];
    if (nbv >= 0.5f) v = dm_up2x_interp(hy, ly, hx, lx, cl1[y0 * S + x0], cl1[y0 * S + x1], cl1[y1 * S + x0], cl1[y1 * S + x1]);
    m2[i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrows * S2; i += blockDim.x) {
    const int yy = i / S2, x = i - yy * S2;
    nb2[i] = mc_boundary(m2, 0, S2, ra, S2, ya + yy, x) ? 0.f : 1.f;
  }
  __syncthreads();
  float* o = out4 + (size_t)r * S4 * S4;
  const float* f4 = fin2 ? fin2 + (size_t)r * S2 * S2 : nullptr;
  for (int i = threadIdx.x; i < (oy1 - oy0) * S4; i += blockDim.x) {
    const int yy = i / S4, ox = i - yy * S4, oy = oy0 + yy;
    const float sy = dm_mul_rn(rs2, (float)oy), sx = dm_mul_rn(rs2, (float)ox);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + ((y0 < S2 - 1) ? 1 : 0), x1 = x0 + ((x0 < S2 - 1) ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* nr0 = nb2 + (y0 - ya) * S2;
    const float* nr1 = nb2 + (y1 - ya) * S2;
    const float nbv = dm_up2x_interp(hy, ly, hx, lx, nr0[x0], nr0[x1], nr1[x0], nr1[x1]);
    if (nbv >= 0.5f) {
      const float* mr0 = m2 + (y0 - ra) * S2;
      const float* mr1 = m2 + (y1 - ra) * S2;
      o[oy * S4 + ox] = dm_up2x_interp(hy, ly, hx, lx, mr0[x0], mr0[x1], mr1[x0], mr1[x1]);
    } else if (f4) {
      // the fine logits do not exist yet: upsample2x_kernel's align_corners value of the last stage's 2S x 2S logits
      // (same source coordinates: its ratio (H - 1) / (OH - 1) is rs2, and min(x0, W - 1) never binds for ox < 4S)
      o[oy * S4 + ox] = dm_up2x_interp(hy, ly, hx, lx, f4[y0 * S2 + x0], f4[y0 * S2 + x1], f4[y1 * S2 + x0], f4[y1 * S2 + x1]);
    }
  }
}

// ------------------------------------------------------------------ K10
__global__ void gumbel_kernel(const float* __restrict__ logits, const float* __restrict__ U, int N, int K, float T,
                              float* __restrict__ y_soft, float* __restrict__ one_hot, int32_t* __restrict__ index) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float eps = 1e-20f;
  float z[8];
  float zmax = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float g = -logf(-logf(U[n * K + k] + eps) + eps);
    z[k] = (logits[n * K + k] + g) / T;
    zmax = fmaxf(zmax, z[k]);
  }
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    z[k] = expf(z[k] - zmax);
    s += z[k];
  }
  int best = 0;
  float bv = -1.f;
  for (int k = 0; k < K; ++k) {
    const float y = z[k] / s;
    y_soft[n * K + k] = y;
    if (y > bv) {  // strict: first maximum wins, as torch.max
      bv = y;
      best = k;
    }
  }
  index[n] = best;
  for (int k = 0; k < K; ++k) one_hot[n * K + k] = (k == best) ? 1.f : 0.f;
}

// ------------------------------------------------------------------ K13
// grid = (N, row bands): a workgroup stages its band of the mask plus 3 halo rows (the stride-2 Laplacian of pixel
// (y, x) is taken at row 2 * floor(y * S2 / S), within [y - 2, y]) -- one workgroup per RoI left three quarters of
// the chip idle at 256 RoIs.
__global__ __launch_bounds__(256) void detail_target_kernel(const float* __restrict__ masks, int N, int S, float f0,
                                                            float f1, const float* __restrict__ fuse_dev, int band_rows,
                                                            float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (fuse_dev) {          // the fuse kernel is a weight-decayed parameter: read its current value on the device
    f0 = fuse_dev[0];
    f1 = fuse_dev[1];
  }
  float* m = lds;  // rows [r_lo, r_hi) of the mask
  const int r = blockIdx.x;
  const int y_begin = blockIdx.y * band_rows, y_end = min(y_begin + band_rows, S);
  const int r_lo = max(y_begin - 3, 0), r_hi = min(y_end + 3, S);
  const float* src = masks + (size_t)r * S * S;
  for (int i = r_lo * S + threadIdx.x; i < r_hi * S; i += blockDim.x) m[i - r_lo * S] = src[i];
  __syncthreads();
  auto lap = [&](int y, int x) -> float {
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < S && xx >= 0 && xx < S) acc += ((dy == 0 && dx == 0) ? 8.f : -1.f) * m[(yy - r_lo) * S + xx];
      }
    return fmaxf(acc, 0.f);
  };
  const int S2 = (S - 1) / 2 + 1;  // stride-2 conv output size (pad 1, k 3)
  for (int i = y_begin * S + threadIdx.x; i < y_end * S; i += blockDim.x) {
    const int y = i / S, x = i - y * S;
    const float b1 = lap(y, x) > 0.1f ? 1.f : 0.f;
    // F.interpolate(nearest): src = floor(dst * in/out)
    const int y2 = min((int)floorf((float)y * ((float)S2 / (float)S)), S2 - 1);
    const int x2 = min((int)floorf((float)x * ((float)S2 / (float)S)), S2 - 1);
    const float b2 = lap(2 * y2, 2 * x2) > 0.1f ? 1.f : 0.f;
    const float fz = f0 * b1 + f1 * b2;
    out[(size_t)r * S * S + i] = fz > 0.1f ? 1.f : 0.f;
  }
}

// ------------------------------------------------------------------ K12
// grid = (pixel blocks, N).  Each workgroup leaves its two partial sums in part[(n * gridDim.x + block) * 2 ..]:
// BCE-with-logits, un-weighted eps-BCE.  mask_loss_finish_kernel adds them in a fixed order (round 3; round 2 added
// with float atomics, whose order -- and with it the last bits of the loss and of d loss / d mask_labels -- changed
// from run to run).
__global__ __launch_bounds__(256) void mask_loss_kernel(const float* __restrict__ ip, const float* __restrict__ dp,
                                                        const float* __restrict__ it, const float* __restrict__ dt,
                                                        const float* __restrict__ weight, int N, int HW,
                                                        float* __restrict__ part,
                                                        float* __restrict__ gi, float* __restrict__ gd) {
  __shared__ float red[16];
  const int n = blockIdx.y;
  const float w = weight[n];
  const float eps = 1e-10f;
  float s_bce = 0.f, s_det = 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const size_t i = (size_t)n * HW + p;
    const float x = ip[i], t = it[i];
    // F.binary_cross_entropy_with_logits: (1-t)*x + max(-x,0) + log(exp(-mv) + exp(-x-mv))
    const float mv = fmaxf(-x, 0.f);
    s_bce += (1.f - t) * x + mv + logf(expf(-mv) + expf(-x - mv));
    const float xd = dp[i], td = dt[i];
    const float s = sigmoidf_(xd);
    s_det += -(td * logf(s + eps) + (1.f - td) * logf(1.f - s + eps));
    if (gi) gi[i] = sigmoidf_(x) - t;
    if (gd) {
      const float ds = s * (1.f - s);
      gd[i] = -w * (td * ds / (s + eps) - (1.f - td) * ds / (1.f - s + eps));
    }
  }
  const float b = block_sum(s_bce, red);
  const float d = block_sum(s_det, red);
  if (threadIdx.x == 0) {
    part[((size_t)n * gridDim.x + blockIdx.x) * 2] = b;
    part[((size_t)n * gridDim.x + blockIdx.x) * 2 + 1] = d;
  }
}

// one workgroup: per_roi[n] = its blocks' eps-BCE partials in block order; sums[0] += sum_n BCE, sums[1] += sum_n w*d,
// RoIs dealt to the threads by stride and the threads' sums added by a fixed tree
__global__ __launch_bounds__(256) void mask_loss_finish_kernel(const float* __restrict__ part, const float* __restrict__ weight,
                                                               int N, int pb, float* __restrict__ sums,
                                                               float* __restrict__ per_roi) {
  __shared__ float sa[256], sb[256];
  const int t = threadIdx.x;
  float xa = 0.f, xb = 0.f;
  for (int n = t; n < N; n += 256) {
    float b = 0.f, d = 0.f;
    for (int j = 0; j < pb; ++j) {
      b += part[((size_t)n * pb + j) * 2];
      d += part[((size_t)n * pb + j) * 2 + 1];
    }
    if (per_roi) per_roi[n] += d;
    xa += b;
    xb += weight[n] * d;
  }
  sa[t] = xa; sb[t] = xb;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (t < s2) { sa[t] += sa[t + s2]; sb[t] += sb[t + s2]; }
    __syncthreads();
  }
  if (t == 0) { sums[0] += sa[0]; sums[1] += sb[0]; }
}

// K12, one stage of DynaCrossEntropyLoss with its normalisers folded in (round 3).  Round 2's kernel left sums and
// unscaled gradients and the host finished a stage with ~30 small tensor operations -- about 130 launches of a few
// microseconds between the end of the forward and the start of the backward, the one stretch of the step in which
// the GPU sits idle (1.3 ms).  Here the stage weight w_n = mask_labels[n, col], its normaliser
// den = sum_n w_n + 1e-5 (cross_entropy_loss.py:462; every workgroup adds the N weights itself, in one fixed order)
// and the constants are applied where the values are produced:
//   grad_inst = (sigmoid(x) - t) / n_el                                  (d mean-BCE / d logits)
//   grad_det  = -w_n * detail_w / (HW * den) * d eps-BCE / d logit
// and the finishing workgroup writes the stage's loss terms and d loss / d mask_labels[:, col].
__global__ __launch_bounds__(256) void mask_loss_stage_kernel(const float* __restrict__ ip, const float* __restrict__ dp,
                                                              const float* __restrict__ it, const float* __restrict__ dt,
                                                              const float* __restrict__ ml, int K, int col, int N, int HW,
                                                              float detail_w, float inv_nel, float* __restrict__ part,
                                                              float* __restrict__ gi, float* __restrict__ gd) {
  __shared__ float red[16];
  __shared__ float den_s;
  const int n = blockIdx.y;
  float ws = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) ws += ml[(size_t)i * K + col];
  const float wsum = block_sum(ws, red);
  if (threadIdx.x == 0) den_s = wsum + 1e-5f;
  __syncthreads();
  const float w = ml[(size_t)n * K + col];
  const float gscale = detail_w / ((float)HW * den_s);
  const float eps = 1e-10f;
  float s_bce = 0.f, s_det = 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const size_t i = (size_t)n * HW + p;
    const float x = ip[i], t = it[i];
    const float mv = fmaxf(-x, 0.f);
    s_bce += (1.f - t) * x + mv + logf(expf(-mv) + expf(-x - mv));
    const float xd = dp[i], td = dt[i];
    const float s = sigmoidf_(xd);
    s_det += -(td * logf(s + eps) + (1.f - td) * logf(1.f - s + eps));
    if (gi) gi[i] = (sigmoidf_(x) - t) * inv_nel;
    if (gd) {
      const float ds = s * (1.f - s);
      gd[i] = -(w * gscale) * (td * ds / (s + eps) - (1.f - td) * ds / (1.f - s + eps));
    }
  }
  const float b = block_sum(s_bce, red);
  const float d = block_sum(s_det, red);
  if (threadIdx.x == 0) {
    part[((size_t)n * gridDim.x + blockIdx.x) * 2] = b;
    part[((size_t)n * gridDim.x + blockIdx.x) * 2 + 1] = d;
  }
}

// one workgroup: terms[0] = mean BCE of this stage, terms[1] += detail_w * (N / n_el) / den * sum_n w_n d_n,
// grad_ml[n, col] = d_n * detail_w / (HW * den); partials added in a fixed order
__global__ __launch_bounds__(256) void mask_loss_stage_finish_kernel(const float* __restrict__ part, const float* __restrict__ ml,
                                                                     int K, int col, int N, int HW, int pb, float detail_w,
                                                                     float inv_nel, float* __restrict__ terms,
                                                                     float* __restrict__ grad_ml) {
  __shared__ float sa[256], sb[256], sw[256];
  const int t = threadIdx.x;
  float ws = 0.f;
  for (int i = t; i < N; i += 256) ws += ml[(size_t)i * K + col];
  sw[t] = ws;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (t < s2) sw[t] += sw[t + s2];
    __syncthreads();
  }
  // (the stage kernel's workgroups took the same sum through block_sum: wave sums, then the waves -- another order.
  //  Both are fixed; the two dens may differ in the last bit, which only moves the loss value against its gradient
  //  by one ulp.)
  const float gscale = detail_w / ((float)HW * (sw[0] + 1e-5f));
  float xa = 0.f, xb = 0.f;
  for (int n = t; n < N; n += 256) {
    float b = 0.f, d = 0.f;
    for (int j = 0; j < pb; ++j) {
      b += part[((size_t)n * pb + j) * 2];
      d += part[((size_t)n * pb + j) * 2 + 1];
    }
    grad_ml[(size_t)n * K + col] = d * gscale;
    xa += b;
    xb += ml[(size_t)n * K + col] * d;
  }
  sa[t] = xa; sb[t] = xb;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (t < s2) { sa[t] += sa[t + s2]; sb[t] += sb[t + s2]; }
    __syncthreads();
  }
  if (t == 0) { terms[0] = sa[0] * inv_nel; terms[1] += sb[0] * gscale; }
}

// nearest x2 (nn.Upsample(scale_factor=2, mode='nearest'), FCNMaskHead upsample_cfg type 'nearest',
// fcn_mask_head.py:88-96), its adjoint (sum of the 2x2 block), and the space-to-depth of a x2 map
// (out[n, (dy*2+dx)*C + c, y, x] = in[n, c, 2y+dy, 2x+dx]) that turns the 2x2 stride-2 deconvolution's
// backward into plain 1x1 GEMMs.
__global__ __launch_bounds__(256) void nearest2x_kernel(const float* __restrict__ in, long long total_out, int H, int W,
                                                        float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_out) return;
  const int OW = 2 * W, OH = 2 * H;
  const int ox = (int)(idx % OW);
  const int oy = (int)((idx / OW) % OH);
  const long long nc = idx / ((long long)OW * OH);
  out[idx] = in[(nc * H + (oy >> 1)) * W + (ox >> 1)];
}

__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const float* __restrict__ gout, long long total_in, int H, int W,
                                                            float* __restrict__ gin) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_in) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  const long long nc = idx / ((long long)W * H);
  const float* g = gout + (nc * 2 * H + 2 * y) * (2 * W) + 2 * x;
  gin[idx] = (g[0] + g[1]) + (g[2 * W] + g[2 * W + 1]);
}

__global__ __launch_bounds__(256) void unshuffle2x_kernel(const float* __restrict__ in, int NB, int C, int H, int W,
                                                          float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // over out [NB, 4C, H, W]
  const long long total = (long long)NB * 4 * C * H * W;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  const int oc = (int)((idx / ((long long)W * H)) % (4 * C));
  const int n = (int)(idx / ((long long)W * H * 4 * C));
  const int d = oc / C, c = oc - d * C;
  out[idx] = in[(((size_t)n * C + c) * 2 * H + 2 * y + (d >> 1)) * (2 * W) + 2 * x + (d & 1)];
}

}  // namespace

extern "C" int dm_point_sample_fwd(const float* feat, int B, int C, int H, int W, const float* rois, int N, int S,
                                   float spatial_scale, float* out, dm_stream_t stream) {
  if (!feat || !rois || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || N < 0 || S <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  const int CT = 16;
  const int chunks = dm_ceil_div(C, CT);
  const long long pos_blocks = ((long long)N * S * S + 255) / 256;
  if (pos_blocks * chunks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH(point_sample_kernel, dim3((unsigned)(pos_blocks * chunks)), dim3(256), 0, (hipStream_t)stream,
                     feat, B, C, H, W, rois, N, S, spatial_scale, out, CT, (int)pos_blocks);
  return dm_check_launch();
}

extern "C" int dm_class_logits_fwd(const float* x, int N, int C, int HW, const float* w_inst, const float* b_inst,
                                   const float* w_det, const float* b_det, int num_classes, const int64_t* labels,
                                   float* inst, float* det, float* sig_out, int sig_ch_total, int sig_ch_offset,
                                   dm_stream_t stream) {
  if (!x || !w_inst || !b_inst || !w_det || !b_det || !labels || !inst || !det) return DM_ERR_INVALID_ARG;
  if (N < 0 || C <= 0 || HW <= 0 || num_classes <= 0) return DM_ERR_INVALID_ARG;
  if (sig_out && (sig_ch_offset < 0 || sig_ch_offset + 2 > sig_ch_total)) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  const int pix_blocks = dm_ceil_div(HW, 64);
  if ((long long)N * pix_blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH(class_logits_kernel, dim3((unsigned)(N * pix_blocks)), dim3(256), 0, (hipStream_t)stream, x, N, C,
                     HW, w_inst, b_inst, w_det, b_det, num_classes, labels, inst, det, sig_out, sig_ch_total,
                     sig_ch_offset, pix_blocks);
  return dm_check_launch();
}

// (ABI 25) dm_point_sample_fwd + dm_class_logits_fwd of one SFM stage as one launch (see stage_head_kernel).
extern "C" int dm_stage_head_fwd(const float* sem, int B, int Cs, int H, int W, const float* rois, int N, int S, float spatial_scale,
                                 float* sampled, const float* x, int C, const float* w_inst, const float* b_inst, const float* w_det,
                                 const float* b_det, int num_classes, const int64_t* labels, float* inst, float* det, float* sig_out,
                                 int sig_ch_total, int sig_ch_offset, dm_stream_t stream) {
  if (!sem || !rois || !sampled || B <= 0 || Cs <= 0 || H <= 0 || W <= 0 || N < 0 || S <= 0) return DM_ERR_INVALID_ARG;
  if (!x || !w_inst || !b_inst || !w_det || !b_det || !labels || !inst || !det || C <= 0 || num_classes <= 0) return DM_ERR_INVALID_ARG;
  if (sig_out && (sig_ch_offset < 0 || sig_ch_offset + 2 > sig_ch_total)) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  StageHeadArgs a;
  a.feat = sem; a.B = B; a.Cs = Cs; a.H = H; a.W = W; a.rois = rois; a.N = N; a.S = S; a.scale = spatial_scale; a.ps_out = sampled;
  a.CT = 16;
  a.x = x; a.C = C; a.HW = S * S; a.wi = w_inst; a.bi = b_inst; a.wd = w_det; a.bd = b_det; a.num_classes = num_classes;
  a.labels = labels; a.inst = inst; a.det = det; a.sig = sig_out; a.sig_ct = sig_ch_total; a.sig_off = sig_ch_offset;
  a.pix_blocks = dm_ceil_div(S * S, 64);
  const long long ps = (((long long)N * S * S + 255) / 256) * dm_ceil_div(Cs, a.CT), cl = (long long)N * a.pix_blocks;
  if (ps + cl > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
  a.ps_blocks = (int)ps;
  DM_LAUNCH(stage_head_kernel, dim3((unsigned)(ps + cl)), dim3(256), 0, (hipStream_t)stream, a);
  return dm_check_launch();
}

extern "C" int dm_class_logits_up2x_fwd(const float* x, int N, int C, int H, int W, const float* w_inst, const float* b_inst,
                                        const float* w_det, const float* b_det, int num_classes, const int64_t* labels,
                                        float* inst, float* det, dm_stream_t stream) {
  if (!x || !w_inst || !b_inst || !w_det || !b_det || !labels || !inst || !det) return DM_ERR_INVALID_ARG;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || num_classes <= 0) return DM_ERR_INVALID_ARG;
  if ((W & 1) || H < 2 || W < 2) return DM_ERR_UNSUPPORTED;      // the half-pixel fast path of dm_upsample2x_bilinear_fwd
  if (N == 0) return DM_OK;
  const int item_blocks = dm_ceil_div(H * (W / 2), 64);
  if ((long long)N * item_blocks > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH(class_logits_up2x_kernel, dim3((unsigned)(N * item_blocks)), dim3(256), 0, (hipStream_t)stream, x, item_blocks, C, H, W,
            w_inst, b_inst, w_det, b_det, num_classes, labels, inst, det);
  return dm_check_launch();
}

extern "C" int dm_upsample2x_bilinear_fwd(const float* in, int NC, int H, int W, int align_corners, int relu, float* out,
                                          dm_stream_t stream) {
  if (!in || !out || NC < 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NC == 0) return DM_OK;
  if (!align_corners && (W & 1) == 0 && H >= 2 && W >= 2) {
    const size_t items = (size_t)NC * H * (W / 2);
    const int blk = (int)min((size_t)dm_ceil_div((long long)items, 256), (size_t)65536);
    DM_LAUNCH(upsample2x_half_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, in, (long long)NC, H, W, relu, out);
    return dm_check_launch();
  }
  const size_t total = (size_t)NC * 2 * H * ((2 * W + 3) / 4);
  const int blocks = (int)min((size_t)dm_ceil_div((long long)total, 256), (size_t)32768);
  DM_LAUNCH(upsample2x_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, NC, H, W, align_corners,
                     relu, out);
  return dm_check_launch();
}

extern "C" int dm_boundary_merge(const float* coarse, float* fine, int n, int S, dm_stream_t stream) {
  if (!coarse || !fine || n < 0 || S <= 0) return DM_ERR_INVALID_ARG;
  if ((size_t)2 * S * S * sizeof(float) > 64 * 1024) return DM_ERR_UNSUPPORTED;
  if (n == 0) return DM_OK;
  DM_LAUNCH(boundary_merge_kernel, dim3(n), dim3(256), 2 * S * S * sizeof(float), (hipStream_t)stream, coarse,
                     fine, n, S);
  return dm_check_launch();
}

// (ABI 25) The inference tail in one launch: out_4s = merge(merge(p_s -> p_2s) -> fine), fine = the align_corners x2
// upsample of final_2s when that is given (then out_4s is write-only), else out_4s itself (merged in place).
extern "C" int dm_boundary_merge_chain(const float* p_s, const float* p_2s, const float* final_2s, float* out_4s, int n, int S,
                                       dm_stream_t stream) {
  if (!p_s || !p_2s || !out_4s || n < 0 || S <= 1) return DM_ERR_INVALID_ARG;
  const size_t lds_bytes = sizeof(float) * ((size_t)2 * S * S + (size_t)(MC_ROWS / 2 + 5) * 2 * S * 2);
  if (lds_bytes > 64 * 1024) return DM_ERR_UNSUPPORTED;
  if (n == 0) return DM_OK;
  const int bands = dm_ceil_div(4 * S, MC_ROWS);
  if ((long long)n * bands > 0x7fffffffLL) return DM_ERR_UNSUPPORTED;
  DM_LAUNCH(boundary_merge_chain_kernel, dim3((unsigned)(n * bands)), dim3(256), lds_bytes, (hipStream_t)stream, p_s, p_2s, final_2s,
            out_4s, n, S, bands);
  return dm_check_launch();
}

extern "C" int dm_gumbel_select_fwd(const float* logits, const float* U, int N, int K, float temperature, float* y_soft,
                                    float* one_hot, int32_t* index, dm_stream_t stream) {
  if (!logits || !U || !y_soft || !one_hot || !index || N < 0 || K <= 0 || K > 8 || temperature <= 0.f)
    return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  DM_LAUNCH(gumbel_kernel, dim3(dm_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, logits, U, N, K,
                     temperature, y_soft, one_hot, index);
  return dm_check_launch();
}

extern "C" int dm_detail_target(const float* masks, int N, int S, float fuse0, float fuse1, const float* fuse_dev,
                                float* out, dm_stream_t stream) {
  if (!masks || !out || N < 0 || S <= 0) return DM_ERR_INVALID_ARG;
  if ((size_t)S * S * sizeof(float) > 64 * 1024) return DM_ERR_UNSUPPORTED;
  if (N == 0) return DM_OK;
  // bands of rows so that a few hundred RoIs still give every CU several workgroups
  int bands = 1;
  while (bands < 8 && (long long)N * bands < 4LL * dm_num_cus() && dm_ceil_div(S, bands * 2) >= 8) bands *= 2;
  const int band_rows = dm_ceil_div(S, bands);
  bands = dm_ceil_div(S, band_rows);
  DM_LAUNCH(detail_target_kernel, dim3(N, bands), dim3(256), (size_t)(band_rows + 6) * S * sizeof(float), (hipStream_t)stream,
            masks, N, S, fuse0, fuse1, fuse_dev, band_rows, out);
  return dm_check_launch();
}

extern "C" long long dm_mask_loss_scratch_floats(int N) { return N >= 0 ? (long long)N * 16 : -1; }

extern "C" int dm_mask_loss_fwd_bwd(const float* inst_pred, const float* det_pred, const float* inst_tgt,
                                    const float* det_tgt, const float* weight, int N, int HW, float* sums,
                                    float* per_roi_det, float* grad_inst, float* grad_det, float* scratch,
                                    dm_stream_t stream) {
  if (!inst_pred || !det_pred || !inst_tgt || !det_tgt || !weight || !sums || !scratch || N < 0 || HW <= 0)
    return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  // as few workgroups per RoI as still fill the chip (at most 8: the scratch holds 8 pairs per RoI)
  int pb = min(dm_ceil_div(HW, 256), 8);
  while (pb > 1 && (long long)N * (pb / 2) >= dm_num_cus()) pb /= 2;
  DM_LAUNCH(mask_loss_kernel, dim3(pb, N), dim3(256), 0, (hipStream_t)stream, inst_pred, det_pred, inst_tgt,
                     det_tgt, weight, N, HW, scratch, grad_inst, grad_det);
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(mask_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, weight, N, pb, sums,
            per_roi_det);
  return dm_check_launch();
}

extern "C" int dm_mask_loss_stage(const float* inst_pred, const float* det_pred, const float* inst_tgt,
                                  const float* det_tgt, const float* mask_labels, int num_stages, int stage, int N, int HW,
                                  float detail_weight, float* loss_terms, float* grad_mask_labels, float* grad_inst,
                                  float* grad_det, float* scratch, dm_stream_t stream) {
  if (!inst_pred || !det_pred || !inst_tgt || !det_tgt || !mask_labels || !loss_terms || !grad_mask_labels || !scratch ||
      N < 0 || HW <= 0 || num_stages <= 0 || stage < 0 || stage >= num_stages)
    return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  int pb = min(dm_ceil_div(HW, 256), 8);
  while (pb > 1 && (long long)N * (pb / 2) >= dm_num_cus()) pb /= 2;
  const float inv_nel = 1.0f / ((float)N * (float)HW);
  DM_LAUNCH(mask_loss_stage_kernel, dim3(pb, N), dim3(256), 0, (hipStream_t)stream, inst_pred, det_pred, inst_tgt, det_tgt,
            mask_labels, num_stages, stage, N, HW, detail_weight, inv_nel, scratch, grad_inst, grad_det);
  int rc = dm_check_launch();
  if (rc != DM_OK) return rc;
  DM_LAUNCH(mask_loss_stage_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, mask_labels,
            num_stages, stage, N, HW, pb, detail_weight, inv_nel, loss_terms, grad_mask_labels);
  return dm_check_launch();
}

// ------------------------------------------------------------------ K14
// class-balance entropy  cb = sum_k p_k log(p_k + 1e-10),  p = colsum / total
// (losses/cross_entropy_loss.py:478-481) and its gradient wrt mask_labels
// (identical for every RoI).  One workgroup; N <= a few hundred.
namespace {
__global__ __launch_bounds__(256) void class_balance_kernel(const float* __restrict__ ml, int N, int K,
                                                            float* __restrict__ loss, float* __restrict__ grad) {
  __shared__ float red[16];
  __shared__ float col[8];
  __shared__ float gk[8];
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) s += ml[n * K + k];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) col[k] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float T = 0.f;
    for (int k = 0; k < K; ++k) T += col[k];
    float cb = 0.f, dot = 0.f;
    float fp[8];
    for (int k = 0; k < K; ++k) {
      const float p = col[k] / T;
      cb += p * logf(p + 1e-10f);
      fp[k] = logf(p + 1e-10f) + p / (p + 1e-10f);
      dot += fp[k] * p;
    }
    loss[0] = cb;
    for (int k = 0; k < K; ++k) gk[k] = (fp[k] - dot) / T;
  }
  __syncthreads();
  if (grad)
    for (int i = threadIdx.x; i < N * K; i += blockDim.x) grad[i] = gk[i % K];
}

// K10 backward: y_hard = (one_hot - y).detach() + y  ->  d/dlogits of the
// softmax((logits+g)/T) branch only.
__global__ void gumbel_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, int N, int K, float T,
                                  float* __restrict__ glogits) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float dot = 0.f;
  for (int k = 0; k < K; ++k) dot += y[n * K + k] * gy[n * K + k];
  for (int k = 0; k < K; ++k) glogits[n * K + k] = y[n * K + k] * (gy[n * K + k] - dot) / T;
}
}  // namespace

extern "C" int dm_class_balance_fwd_bwd(const float* mask_labels, int N, int K, float* loss, float* grad,
                                        dm_stream_t stream) {
  if (!mask_labels || !loss || N <= 0 || K <= 0 || K > 8) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(class_balance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mask_labels, N, K, loss, grad);
  return dm_check_launch();
}

extern "C" int dm_gumbel_select_bwd(const float* y_soft, const float* grad_y, int N, int K, float temperature,
                                    float* grad_logits, dm_stream_t stream) {
  if (!y_soft || !grad_y || !grad_logits || N < 0 || K <= 0 || temperature <= 0.f) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  DM_LAUNCH(gumbel_bwd_kernel, dim3(dm_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, y_soft, grad_y, N, K,
                     temperature, grad_logits);
  return dm_check_launch();
}

// ---------------------------------------------------------------------------
// Callers either side of the path.
namespace {

// rois[n] = (gt_index, clip(x1,0,maxw), clip(y1,0,maxh), clip(x2,0,maxw), clip(y2,0,maxh))
// (DynaMaskHead.get_targets, dynamask_head.py:248-261, + BitmapMasks.crop_and_resize
// rois assembly, core/mask/structures.py:270-276)
__global__ void mask_target_rois_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ inds, int N, float maxw,
                                        float maxh, float* __restrict__ rois) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  rois[n * 5 + 0] = (float)inds[n];
  rois[n * 5 + 1] = fminf(fmaxf(boxes[n * 4 + 0], 0.f), maxw);
  rois[n * 5 + 2] = fminf(fmaxf(boxes[n * 4 + 1], 0.f), maxh);
  rois[n * 5 + 3] = fminf(fmaxf(boxes[n * 4 + 2], 0.f), maxw);
  rois[n * 5 + 4] = fminf(fmaxf(boxes[n * 4 + 3], 0.f), maxh);
}

__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ x, size_t n, float thr, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = x[i] >= thr ? 1.f : 0.f;
}

// K18 paste: grid_sample(bilinear, zeros, align_corners=False) of each mask at the
// image pixel centres mapped into its box, thresholded.  grid = (row blocks, N).
// (_do_paste_mask, fcn_mask_head.py:240-308 with skip_empty=False, + the >= thr of
// get_seg_masks, dynamask_head.py:333-334)
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                          int N, int mh, int mw, int img_h, int img_w, float thr,
                                                          int apply_sigmoid, uint8_t* __restrict__ out) {
  const int n = blockIdx.y;
  const float x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
  const float* m = masks + (size_t)n * mh * mw;
  uint8_t* o = out + (size_t)n * img_h * img_w;
  const size_t total = (size_t)img_h * img_w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int py = (int)(i / img_w), px = (int)(i - (size_t)py * img_w);
    float gx = ((float)px + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
    float gy = ((float)py + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
    if (isinf(gx)) gx = 0.f;      // degenerate boxes (reference zeroes inf coordinates, :283-288)
    if (isinf(gy)) gy = 0.f;
    const float sx = ((gx + 1.f) * (float)mw - 1.f) / 2.f;
    const float sy = ((gy + 1.f) * (float)mh - 1.f) / 2.f;
    float v = 0.f;
    if (sx > -1.f && sx < (float)mw && sy > -1.f && sy < (float)mh) {
      const float fx = floorf(sx), fy = floorf(sy);
      const int ix = (int)fx, iy = (int)fy;
      const float lx = sx - fx, ly = sy - fy;
      auto at = [&](int yy, int xx) -> float {
        if (yy < 0 || yy >= mh || xx < 0 || xx >= mw) return 0.f;
        const float t = m[yy * mw + xx];
        return apply_sigmoid ? 1.f / (1.f + expf(-t)) : t;
      };
      v = at(iy, ix) * (1.f - lx) * (1.f - ly) + at(iy, ix + 1) * lx * (1.f - ly) + at(iy + 1, ix) * (1.f - lx) * ly +
          at(iy + 1, ix + 1) * lx * ly;
    }
    o[i] = v >= thr ? 1 : 0;
  }
}

}  // namespace

extern "C" int dm_mask_target_rois(const float* boxes, const int64_t* gt_inds, int N, float max_w, float max_h, float* rois,
                                   dm_stream_t stream) {
  if (N < 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!boxes || !gt_inds || !rois) return DM_ERR_INVALID_ARG;
  DM_LAUNCH(mask_target_rois_kernel, dim3(dm_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, boxes, gt_inds, N, max_w,
            max_h, rois);
  return dm_check_launch();
}

extern "C" int dm_threshold_ge(const float* x, long long count, float thr, float* out, dm_stream_t stream) {
  if (count < 0) return DM_ERR_INVALID_ARG;
  if (count == 0) return DM_OK;
  if (!x || !out) return DM_ERR_INVALID_ARG;
  const int blocks = (int)min((long long)dm_ceil_div(count, 256), 16384LL);
  DM_LAUNCH(threshold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)count, thr, out);
  return dm_check_launch();
}

extern "C" int dm_paste_masks(const float* masks, const float* boxes, int N, int mask_h, int mask_w, int img_h, int img_w,
                              float threshold, int apply_sigmoid, uint8_t* out, dm_stream_t stream) {
  if (N < 0 || mask_h <= 0 || mask_w <= 0 || img_h <= 0 || img_w <= 0) return DM_ERR_INVALID_ARG;
  if (N == 0) return DM_OK;
  if (!masks || !boxes || !out) return DM_ERR_INVALID_ARG;
  const int bx = min(dm_ceil_div((long long)img_h * img_w, 256 * 4), 1024);
  DM_LAUNCH(paste_masks_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, masks, boxes, N, mask_h, mask_w, img_h, img_w,
            threshold, apply_sigmoid, out);
  return dm_check_launch();
}

extern "C" int dm_upsample2x_nearest_fwd(const float* in, int NC, int H, int W, float* out, dm_stream_t stream) {
  if (!in || !out || NC < 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NC == 0) return DM_OK;
  const long long total = (long long)NC * 4 * H * W;
  DM_LAUNCH(nearest2x_kernel, dim3((unsigned)dm_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, total, H, W, out);
  return dm_check_launch();
}

extern "C" int dm_upsample2x_nearest_bwd(const float* grad_out, int NC, int H, int W, float* grad_in, dm_stream_t stream) {
  if (!grad_out || !grad_in || NC < 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NC == 0) return DM_OK;
  const long long total = (long long)NC * H * W;
  DM_LAUNCH(nearest2x_bwd_kernel, dim3((unsigned)dm_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, grad_out, total,
            H, W, grad_in);
  return dm_check_launch();
}

extern "C" int dm_pixel_unshuffle2x(const float* in, int NB, int C, int H, int W, float* out, dm_stream_t stream) {
  if (!in || !out || NB < 0 || C <= 0 || H <= 0 || W <= 0) return DM_ERR_INVALID_ARG;
  if (NB == 0) return DM_OK;
  const long long total = (long long)NB * 4 * C * H * W;
  DM_LAUNCH(unshuffle2x_kernel, dim3((unsigned)dm_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, NB, C, H, W, out);
  return dm_check_launch();
}
