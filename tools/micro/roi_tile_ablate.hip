// Ceiling measurement of the 14x14 multi-level RoIAlign kernel (reproducible evidence).
// Compiles the LIBRARY's roi_align.hip with -DDM_ROI_ABLATE=<variant>, ONE BINARY PER VARIANT (eight; the round-4 driver
// script went with the prune of round 6, profiles/r04_roialign_ceiling.txt is what it wrote): the ablation bits are compile-time constants (DM_ABL in roi_align.hip), variant 0 is the product kernel
// instruction for instruction.  Bit 1 = no global loads (the staging commits register garbage), bit 2 = one LDS tap per
// output instead of the merged stencil, bit 4 = no output stores.  Workload = bench.py's: FPN maps of a
// 1333x800 image (P2..P5, 256 channels, random), the 512 RoIs of synth.make_rois(seed=1) (rois_512_1333x800.txt).
//
//   (pass the product's flags -- dynamask_amd/build.py FLAGS: no packed fp32 -- or variant 0 is not the product's codegen)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Idynamask_amd/csrc -DDM_ROI_ABLATE=3 tools/micro/roi_tile_ablate.hip -o gpurun_out/roi_tile_ablate_3
//   gpurun_out/roi_tile_ablate_3 tools/micro/rois_512_1333x800.txt          # that variant, 14x14
//   ROI_P=7 ...                                                              # the 7x7 bbox extraction
#ifndef DM_ROI_ABLATE
#define DM_ROI_ABLATE 0
#endif
#include "../../dynamask_amd/csrc/roi_align.hip"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char* dm_error_string(int) { return "error"; }

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s rois.txt\n", argv[0]); return 2; }
  std::vector<float> rois;
  {
    FILE* f = fopen(argv[1], "r");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    char line[512];
    while (fgets(line, sizeof line, f)) {
      if (line[0] == '#') continue;
      float v[5];
      if (sscanf(line, "%f %f %f %f %f", v, v + 1, v + 2, v + 3, v + 4) == 5) rois.insert(rois.end(), v, v + 5);
    }
    fclose(f);
  }
  const int N = (int)rois.size() / 5, C = 256, B = 1;
  const int P = getenv("ROI_P") ? atoi(getenv("ROI_P")) : 14;
  const int H[4] = {200, 100, 50, 25}, W[4] = {336, 168, 84, 42};       // ceil(800 / s), ceil(1333 / s) padded as the FPN does
  const float scales[4] = {1.f / 4, 1.f / 8, 1.f / 16, 1.f / 32};
  float* feats[4];
  size_t map_bytes = 0;
  for (int l = 0; l < 4; ++l) {
    const size_t n = (size_t)B * C * H[l] * W[l];
    std::vector<float> h(n);
    unsigned s = 12345u + l;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMalloc(&feats[l], n * 4));
    CK(hipMemcpy(feats[l], h.data(), n * 4, hipMemcpyHostToDevice));
    map_bytes += n * 4;
  }
  float *d_rois, *d_out;
  int* d_lv;
  CK(hipMalloc(&d_rois, rois.size() * 4));
  CK(hipMemcpy(d_rois, rois.data(), rois.size() * 4, hipMemcpyHostToDevice));
  const size_t out_bytes = (size_t)N * C * P * P * 4;
  CK(hipMalloc(&d_out, out_bytes));
  CK(hipMalloc(&d_lv, N * 4));
  // algorithmic bytes as bench.py counts them: output + rois + per-RoI footprints, the read capped by the maps
  std::vector<int> lv(N);
  int rc = dm_roi_align_fwd(feats, H, W, scales, 4, B, C, d_rois, N, P, 0, 56.f, d_out, d_lv, nullptr);
  if (rc) { printf("dm_roi_align_fwd rc %d\n", rc); return 1; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(lv.data(), d_lv, N * 4, hipMemcpyDeviceToHost));
  double foot = 0, touched = 0;
  bool seen[4] = {false, false, false, false};
  const int strides[4] = {4, 8, 16, 32};
  for (int k = 0; k < N; ++k) {
    const int l = lv[k];
    const float* r = &rois[5 * k];
    const double w = ceil((r[3] - r[1]) / strides[l]) + 2, h = ceil((r[4] - r[2]) / strides[l]) + 2;
    foot += 4.0 * C * fmin(w * h, (double)H[l] * W[l]);
    if (!seen[l]) { seen[l] = true; touched += 4.0 * C * H[l] * W[l]; }
  }
  const double alg = (double)out_bytes + N * 20.0 + fmin(foot, touched);
  printf("N=%d RoIs, P=%d: output %.1f MB, maps %.1f MB, footprints %.1f MB -> algorithmic %.1f MB\n", N, P, out_bytes / 1e6,
         map_bytes / 1e6, foot / 1e6, alg / 1e6);
  const char* names[8] = {"full kernel", "no global loads", "one tap (no stencil)", "no loads, one tap", "no stores",
                          "no loads, no stores", "one tap, no stores", "no loads, one tap, no stores"};
  const int reps = getenv("ROI_REPS") ? atoi(getenv("ROI_REPS")) : 50;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {
    const int v = DM_ROI_ABLATE;
    for (int i = 0; i < 5; ++i) dm_roi_align_fwd(feats, H, W, scales, 4, B, C, d_rois, N, P, 0, 56.f, d_out, nullptr, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int t = 0; t < 5; ++t) {
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < reps; ++i) dm_roi_align_fwd(feats, H, W, scales, 4, B, C, d_rois, N, P, 0, 56.f, d_out, nullptr, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = fminf(best, ms / reps);
      sum += ms / reps;
    }
    const double us = best * 1e3;
    printf("abl=%d  %-30s  %7.1f us (best of 5 x %d back-to-back launches; mean %.1f)  %6.2f TB/s algorithmic%s\n", v, names[v], us,
           reps, sum / 5 * 1e3, alg / us / 1e6, v ? "" : "  <- the product kernel");
  }
  return 0;
}
