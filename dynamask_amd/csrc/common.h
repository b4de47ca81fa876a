// Shared helpers for the gfx950 kernels of libdynamask_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

// Compute units of the current device (MI355X: 256), read once.
static inline int dm_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

typedef float dm_f32x16 __attribute__((ext_vector_type(16)));
typedef float dm_f32x4 __attribute__((ext_vector_type(4)));
