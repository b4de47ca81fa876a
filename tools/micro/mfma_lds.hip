// Microbenchmark: does LDS read traffic cost fp32 MFMA throughput on gfx950?
// Two workgroups of 4 waves per CU (2 waves per SIMD, as in the conv kernels).  Per iteration a wave
// issues 16 v_mfma_f32_32x32x2_f32 (4 independent accumulators) and L conflict-free ds_read_b128
// (lane-linear addresses).  L = 4 is the conv kernel's ratio (2 A + 2 B fragments per 16 MFMAs); the
// extra reads of L > 4 stand for a gather sharing the LDS pipe (DCN).  The fragments of iteration
// i + 1 are read before the MFMAs of iteration i, like the kernels do.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_lds.hip -o gpurun_out/mfma_lds && gpurun_out/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int L, int OCC>
__global__ __launch_bounds__(256, OCC) void k(int iters, float* out, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  for (int i = threadIdx.x; i < 2048; i += 256) l4[i] = f32x4{1.f + i, 2.f, 3.f, 4.f};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f32x4 fr[2][4];
  for (int j = 0; j < 4; ++j) fr[0][j] = l4[wave * 256 + j * 64 + lane];
  f32x4 dummy = {0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const int base = ((it + 1) & 7) * 512 + wave * 64;
    if (L < 0) {
      // no operand traffic at all: the same fragment registers every iteration
    } else if (L >= 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[cur ^ 1][j] = l4[(base + j * 64 + lane) & 2047];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[cur ^ 1][j] = fr[cur][j];
#pragma unroll
      for (int j = 0; j < L; ++j) fr[cur ^ 1][j] = l4[(base + j * 64 + lane) & 2047];
    }
#pragma unroll
    for (int j = 4; j < L; ++j) {
      const f32x4 v = l4[(base + 1024 + j * 64 + lane) & 2047];
      dummy += v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cur = L < 0 ? 0 : (it & 1);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][0][e], fr[cur][2][e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][0][e], fr[cur][3][e], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][1][e], fr[cur][2][e], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[cur][1][e], fr[cur][3][e], acc[3], 0, 0, 0);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = clock64() - c0;          // shader clock
    clk[1] = wall_clock64() - w0;     // constant 100 MHz
  }
  float s = dummy[0] + dummy[1] + dummy[2] + dummy[3];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int L, int OCC>
void run() {
  float* out;
  long long* clk;
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&clk, 16);
  const int iters = 4000, blocks = 256 * OCC;
  const size_t lds_bytes = OCC == 1 ? 98304 : OCC == 2 ? 65536 : OCC == 3 ? 49152 : 36864;          // OCC workgroups per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<L, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<L, OCC><<<blocks, 256, lds_bytes>>>(iters, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<L, OCC><<<blocks, 256, lds_bytes>>>(iters, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
  const double lds_bytes_read = (double)blocks * 4 * iters * (L < 0 ? 0 : L) * 1024;
  long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  printf("%d wg/CU  L=%2d ds_read_b128 per 16 MFMA: %.3f ms  %.1f TFLOP/s  clock %.2f GHz -> %.1f %% of the MFMA rate at that clock  LDS %.0f B/clk/CU\n",
         OCC, L, ms, flops / ms / 1e9, ghz, 100.0 * flops / (ms * 1e-3) / (256.0 * 4 * 64 * ghz * 1e9), lds_bytes_read / (ms * 1e-3) / 256 / (ghz * 1e9));
  hipFree(out);
  hipFree(clk);
}
int main() {
  run<0, 2>(); run<2, 2>(); run<4, 2>(); run<8, 2>(); run<16, 2>(); run<32, 2>();
  run<0, 3>(); run<4, 3>(); run<8, 3>(); run<16, 3>();
  run<0, 1>(); run<4, 1>(); run<0, 4>(); run<4, 4>();
  run<-1, 1>(); run<-1, 2>(); run<-1, 4>();      // L = -1: MFMAs only, fixed operand registers
  return 0;
}
