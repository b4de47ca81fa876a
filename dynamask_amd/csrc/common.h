// Shared helpers for the gfx950 kernels of libdynamask_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dynamask_hip.h"

#define DM_WAVE 64

static inline int dm_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipGetLastError() is sticky per thread: a benign error left behind by another
// library's HIP call (torch probing peer access, an event query returning
// "not ready") would otherwise be blamed on our launch.  Clear it first.
#define DM_LAUNCH(...)        \
  do {                        \
    (void)hipGetLastError();  \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

// Post-launch check: the launch itself is asynchronous; configuration errors
// (bad grid, too much LDS) surface here.
static inline int dm_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DM_OK : DM_ERR_LAUNCH;
}

typedef float dm_f32x16 __attribute__((ext_vector_type(16)));
typedef float dm_f32x4 __attribute__((ext_vector_type(4)));
