// Library identity / error strings.
#include "common.h"

extern "C" const char* dm_error_string(int code) {
  switch (code) {
    case DM_OK: return "ok";
    case DM_ERR_INVALID_ARG: return "invalid argument (shape, null pointer or unsupported parameter)";
    case DM_ERR_LAUNCH: return "HIP kernel launch failed";
    case DM_ERR_UNSUPPORTED: return "configuration not supported by libdynamask_hip";
    default: return "unknown error";
  }
}

extern "C" int dm_abi_version(void) { return 27; }

// How this library was compiled: the compiler and the product-wide flag set dynamask_amd/build.py passed (it hands them
// over as -DDM_BUILD_FLAGS="..."; a recipe that does not say what it used yields "flags=unknown").  The host binding
// refuses a library whose string lacks "-packed-fp32-ops" (build.py FLAGS: a compiler-generated v_pk_fma_f32 dropped a
// product beside other queues, profiles/r05_race_hunt.txt), so that a build recipe which silently drops the flag is
// caught at load time and not by a gradient that is occasionally off by one product.
#ifndef DM_BUILD_FLAGS
#define DM_BUILD_FLAGS "unknown"
#endif
extern "C" const char* dm_build_info(void) {
  return "libdynamask_hip abi=27 arch=gfx950 compiler=" __clang_version__ " flags=" DM_BUILD_FLAGS;
}

// ---------------------------------------------------------------------------
// Optimiser step on the flat mask-head parameter buffer: SGD with momentum and
// weight decay (mmcv OptimizerHook + torch.optim.SGD of the reference config,
// configs/dynamask/coco/r50-dynamask-1x.py: lr 0.02, momentum 0.9, wd 1e-4),
// with the 1/world gradient averaging folded in (grad_scale).
namespace {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  long long n, float lr, float momentum, float wd, float gscale,
                                                  int first_step) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float d = g[i] * gscale + wd * p[i];
    float buf = first_step ? d : momentum * m[i] + d;   // torch.optim.SGD: buf = d on the first step
    m[i] = buf;
    p[i] -= lr * buf;
  }
}
}  // namespace

extern "C" int dm_sgd_momentum_step(float* params, const float* grads, float* momentum_buf, long long count, float lr,
                                    float momentum, float weight_decay, float grad_scale, int first_step,
                                    dm_stream_t stream) {
  if (!params || !grads || !momentum_buf || count < 0) return DM_ERR_INVALID_ARG;
  if (count == 0) return DM_OK;
  const int blocks = (int)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
  DM_LAUNCH(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, count, lr, momentum,
            weight_decay, grad_scale, first_step);
  return dm_check_launch();
}
