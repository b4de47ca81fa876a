// Microbenchmark: what does work issued next to fp32 MFMAs cost on gfx950, and does it matter HOW it is issued?
// 2 workgroups of 4 waves per CU; per iteration a wave issues 16 v_mfma_f32_32x32x2_f32 (4 independent accumulators)
// plus one of:
//   mode 0  4 ds_read_b128 (the conv kernels' operand fetch), placed by the compiler
//   mode 1  the same, s_setprio 1 around the MFMA cluster
//   mode 2  the same, one read after every 4th MFMA (sched_group_barrier)
//   mode 3  8 ds_read_b64 (same bytes, twice the instructions)
//   mode 4  2 ds_read_b128 + 2 global_load_dwordx4 (operand A straight from L1/L2)
//   mode 5  4 global_load_dwordx4, no LDS
//   mode 6  mode 0 + 16 independent v_fma_f32
//   mode 7  mode 0 + 16 v_fma_f32, interleaved one per MFMA
//   mode 8  mode 0 + 8 ds_read2_b32 (a gather's reads)
//   mode 9  nothing (fixed operands)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mix.hip -o gpurun_out/mfma_mix && gpurun_out/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int iters, const f32x4* __restrict__ g, float* out, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  f32x2* l2 = reinterpret_cast<f32x2*>(lds);
  for (int i = threadIdx.x; i < 2048; i += 256) l4[i] = f32x4{1.f + i, 2.f, 3.f, 4.f};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 fr[2][4];
  for (int j = 0; j < 4; ++j) fr[0][j] = l4[wave * 256 + j * 64 + lane];
  for (int j = 0; j < 4; ++j) fr[1][j] = fr[0][j];
  float va[16];
  for (int j = 0; j < 16; ++j) va[j] = (float)(lane + j);
  f32x2 gsum = {0.f, 0.f};
  const long long c0 = clock64(), w0 = wall_clock64();
  auto step = [&](auto cur_c, int it) {
    constexpr int cur = MODE == 9 ? 0 : decltype(cur_c)::value;      // compile-time: a run-time index turns the fragment arrays into selects
    const int base = ((it + 1) & 7) * 512 + wave * 64;
    if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 6 || MODE == 7 || MODE == 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[cur ^ 1][j] = l4[(base + j * 64 + lane) & 2047];
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 a = l2[((base + j * 64) * 2 + lane) & 4095], b = l2[((base + j * 64) * 2 + 64 + lane) & 4095];
        fr[cur ^ 1][j] = f32x4{a[0], a[1], b[0], b[1]};
      }
    }
    if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fr[cur ^ 1][j] = g[(base + j * 64 + lane) & 2047];
#pragma unroll
      for (int j = 2; j < 4; ++j) fr[cur ^ 1][j] = l4[(base + j * 64 + lane) & 2047];
    }
    if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[cur ^ 1][j] = g[(base + j * 64 + lane) & 2047];
    }
    if (MODE == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* p = lds + (((base + j * 37) * 4 + lane * 3) & 8191);
        gsum[0] += p[0];
        gsum[1] += p[1];
      }
    }
    if (MODE == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][0][e], fr[cur][2][e], acc[0], 0, 0, 0);
      if (MODE == 7) { va[4 * e] = __builtin_fmaf(va[4 * e], 1.0001f, 0.5f); }
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][0][e], fr[cur][3][e], acc[1], 0, 0, 0);
      if (MODE == 7) { va[4 * e + 1] = __builtin_fmaf(va[4 * e + 1], 1.0001f, 0.5f); }
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][1][e], fr[cur][2][e], acc[2], 0, 0, 0);
      if (MODE == 7) { va[4 * e + 2] = __builtin_fmaf(va[4 * e + 2], 1.0001f, 0.5f); }
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][1][e], fr[cur][3][e], acc[3], 0, 0, 0);
      if (MODE == 7) { va[4 * e + 3] = __builtin_fmaf(va[4 * e + 3], 1.0001f, 0.5f); }
      if (MODE == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      }
      if (MODE == 7) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
    }
    if (MODE == 1) __builtin_amdgcn_s_setprio(0);
    if (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 16; ++j) va[j] = __builtin_fmaf(va[j], 1.0001f, 0.5f);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    step(std::integral_constant<int, 0>{}, it);
    step(std::integral_constant<int, 1>{}, it + 1);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - w0;
  }
  float s = gsum[0] + gsum[1];
  for (int j = 0; j < 16; ++j) s += va[j];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what) {
  float* out;
  long long* clk;
  f32x4* g;
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&clk, 16);
  hipMalloc(&g, 2048 * 16);
  hipMemset(g, 0, 2048 * 16);
  const int iters = 4000, blocks = 512;
  const size_t lds_bytes = 65536;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256, lds_bytes>>>(iters, g, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256, lds_bytes>>>(iters, g, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double frac = flops / (ms * 1e-3) / (256.0 * 4 * 64 * ghz * 1e9);
  printf("mode %d  %-62s %.3f ms  %6.1f TFLOP/s  %.2f GHz  %5.1f %% of the MFMA rate  (+%4.0f clk per 16 MFMA)\n", MODE, what, ms,
         flops / ms / 1e9, ghz, 100.0 * frac, 1024.0 / frac - 1024.0);
  hipFree(out); hipFree(clk); hipFree(g);
}
int main() {
  run<9>("nothing else (fixed operands)");
  run<0>("4 ds_read_b128");
  run<1>("4 ds_read_b128, s_setprio 1 around the MFMAs");
  run<2>("4 ds_read_b128, one after every 4th MFMA");
  run<3>("8 ds_read_b64");
  run<4>("2 ds_read_b128 + 2 global_load_dwordx4");
  run<5>("4 global_load_dwordx4");
  run<6>("4 ds_read_b128 + 16 v_fma_f32 after the MFMAs");
  run<7>("4 ds_read_b128 + 16 v_fma_f32, one per MFMA");
  run<8>("4 ds_read_b128 + 8 ds_read2_b32 (gather)");
  return 0;
}
