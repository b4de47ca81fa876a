// Microbenchmark: what HBM gives a kernel that only reads, only writes, or copies (gfx950).  The write-dominated kernels of
// the path (the 56x56 column-gradient GEMM, deformable im2col, RoIAlign 56x56) sit at 2.2-2.9 TB/s: is that the part's
// write rate or theirs?  2 GiB buffers (8x the Infinity Cache), 16-byte accesses, grid-stride, 2048 workgroups x 256.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/hbm_rw.hip -o /tmp/hbm_rw && /tmp/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, size_t n, float* out) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ a, size_t n, float v) {
  const f32x4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = x;
}
__global__ __launch_bounds__(256) void k_write_nt(f32x4* __restrict__ a, size_t n, float v) {
  const f32x4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(x, a + i);
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// the access shape of the NCHW tile kernels: a wave stores 2 rows x 128 contiguous bytes, rows 12.5 KB apart
__global__ __launch_bounds__(256) void k_write_rows(float* __restrict__ a, size_t rows, int hw, float v) {
  const int lane = threadIdx.x & 31, r2 = threadIdx.x >> 5;          // 8 row slots x 32 pixels per workgroup pass
  const size_t tiles = (size_t)hw / 32;
  for (size_t t = blockIdx.x; t < tiles * (rows / 8); t += gridDim.x) {
    const size_t rb = (t / tiles) * 8, px = (t % tiles) * 32;
    a[(rb + r2) * hw + px + lane] = v;
  }
}
// the store shape of a TRANSPOSED accumulator tile (pixels along the registers): a lane stores 4 consecutive pixels of ITS
// row as one 16-byte piece, lanes 0-31 = 32 rows, lanes 32-63 the next 4 pixels; four such instructions complete a 128-byte line
__global__ __launch_bounds__(256) void k_write_cols(float* __restrict__ a, size_t rows, int hw, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
  const size_t tiles = (size_t)hw / 32;
  const f32x4 x = {v, v, v, v};
  for (size_t t = blockIdx.x; t < tiles * (rows / 128); t += gridDim.x) {
    const size_t rb = (t / tiles) * 128 + wave * 32, px = (t % tiles) * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a + (rb + l31) * hw + px + 8 * g + 4 * hi) = x;
  }
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const size_t bytes = 2ull << 30, n = bytes / 16;
  f32x4 *a, *b; float* out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  const int grid = 2048;
  float ms;
  ms = timeit([&] { k_read<<<grid, 256>>>(a, n, out); });            printf("read only        %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { k_write<<<grid, 256>>>(a, n, 1.f); });           printf("write only       %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { k_write_nt<<<grid, 256>>>(a, n, 1.f); });        printf("write only, nt   %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { k_copy<<<grid, 256>>>(a, b, n); });              printf("copy             %.3f ms  %.2f TB/s (read + write)\n", ms, 2.0 * bytes / ms / 1e9);
  const int hw = 3136; const size_t rows = bytes / 4 / hw / 8 * 8;
  ms = timeit([&] { k_write_rows<<<8192, 256>>>((float*)a, rows, hw, 1.f); });
  printf("write, 128-byte row pieces 12.5 KB apart (the tile kernels' store shape)  %.3f ms  %.2f TB/s\n", ms, (double)rows * (hw / 32 * 32) * 4 / ms / 1e9);
  const size_t rows2 = rows / 128 * 128;
  ms = timeit([&] { k_write_cols<<<8192, 256>>>((float*)a, rows2, hw, 1.f); });
  printf("write, 16-byte pieces per lane across 32 rows (a transposed tile's store shape)  %.3f ms  %.2f TB/s\n", ms, (double)rows2 * (hw / 32 * 32) * 4 / ms / 1e9);
  return 0;
}
