// Microbenchmark: LDS atomic add throughput for f32 / u32 / u64 at random addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned* out) {
  __shared__ unsigned long long lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned a = (s >> 10) & 4095;
    if (MODE == 0) atomicAdd(reinterpret_cast<float*>(lds) + a, 1.0f);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(lds) + a, 1u);
    if (MODE == 2) atomicAdd(lds + a, 1ull);
    if (MODE == 3) reinterpret_cast<float*>(lds)[a] += 1.0f;   // non-atomic RMW for reference
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)lds[1];
}
template <int MODE> void run(const char* name) {
  unsigned* out; hipMalloc(&out, 4096 * 4);
  const int iters = 2000, blocks = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(iters, out); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)blocks * 256 * iters;
  printf("%-10s %.3f ms  %.2f G lane-atomics/s  (%.3f per clk per CU @2.1GHz)\n", name, ms, n / ms / 1e6, n / (ms * 1e-3) / 256 / 2.1e9);
}
int main() { run<0>("ds f32"); run<1>("ds u32"); run<2>("ds u64"); run<3>("rmw f32"); return 0; }
