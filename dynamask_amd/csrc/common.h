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

// Per-device facts.  One process may drive several devices (a rehearsal of ranks on one box, a
// single-process multi-GPU caller): compute-unit counts and "this kernel's LDS limit has been
// raised" flags are properties of a device, not of the process, so they are kept per device.
// (A benign race between two host threads only repeats an idempotent call.)
#define DM_MAX_DEVICES 64

static inline int dm_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DM_MAX_DEVICES) dev = 0;
  return dev;
}

// Compute units of the current device (MI355X: 256).
static inline int dm_num_cus() {
  static int cus[DM_MAX_DEVICES] = {0};
  const int dev = dm_current_device();
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}

// Raise a kernel's dynamic-LDS limit once per device.  `flags` is a static array of
// DM_MAX_DEVICES bools owned by the call site (one per kernel).
static inline int dm_ensure_lds_limit(const void* kernel, int bytes, bool* flags) {
  const int dev = dm_current_device();
  if (!flags[dev]) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return DM_ERR_LAUNCH;
    flags[dev] = true;
  }
  return DM_OK;
}

// ---- order-independent accumulation (deterministic mode; dynamask_hip.h "Deterministic accumulation") ----
// The *_fx entry points add into 64-bit fixed-point cells (value * 2^36, two's complement through the u64 atomic)
// instead of float cells: integer addition is associative, so the sum does not depend on the order in which
// workgroups arrive.  A non-finite addend becomes 2^62 (dm_fx_to_float turns any |cell| >= 2^61 into NaN).
#define DM_FX_ONE 68719476736.0 /* 2^36 */
// g * w -> the same 2^-36 fixed point in eight fp32 / integer instructions (the fp64 route -- a double multiply and
// __double2ll_rn, for which there is no hardware convert -- is ~11 and the DCN col2im is bound by exactly these: 5.3e8
// vector instructions per launch, its VALU pipe 100 % busy).  The caller passes w16 = w * 2^4 and w36 = w * 2^36 (exact
// scalings, shared by the channels of a sample): t = g * w16 and p = g * w36 are the same rounded product at two scales;
// hi = rne(t) is bits 63..32 (+ a borrow), p - hi * 2^32 is exact in fp32 (|.| <= 2^31, a multiple of p's ulp) and its
// integer part is the signed low word.  Finite g * w with |g * w| < 2^27; the fraction below 2^-36 is truncated.
__device__ __forceinline__ unsigned long long dm_fix36_mul(float g, float w16, float w36) {
  const float hi_f = __builtin_rintf(g * w16);
  const float rem = __builtin_fmaf(hi_f, -4294967296.0f, g * w36);
  const int lo = (int)rem;
  const int hi = (int)hi_f + (lo >> 31);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// |value| * 2^36 from t16 = value * 2^4 in four instructions: the high word is trunc(|t16|), the low word its fraction
// * 2^32 (exact for a non-negative number: a subset of its mantissa bits; a negative one has no such split -- 1 - 2^-40
// is not a float -- which is why the sign goes into the choice of ds_add_u64 / ds_sub_u64 instead: dm_fix36_accumulate).
// |value| < 2^27; the fraction below 2^-36 is truncated towards zero, as in dm_fix36_mul.
__device__ __forceinline__ unsigned long long dm_fix36_abs16(float t16) {
  const float m = __builtin_fabsf(t16);
  const unsigned hi = (unsigned)m;
  const unsigned lo = (unsigned)(__builtin_amdgcn_fractf(m) * 4294967296.0f);
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
__device__ __forceinline__ void dm_fix36_accumulate(unsigned long long* cell, float t16) {
  const unsigned long long mag = dm_fix36_abs16(t16);
  if (t16 < 0.f) atomicSub(cell, mag);
  else atomicAdd(cell, mag);
}

__device__ __forceinline__ unsigned long long dm_to_fx(float v) {
  const long long q = __builtin_isfinite(v) ? __double2ll_rn((double)v * DM_FX_ONE) : (1LL << 62);
  return (unsigned long long)q;
}
// accumulate v into cell idx of an accumulator that is float (fx = false) or 64-bit fixed point (fx = true)
__device__ __forceinline__ void dm_acc_add(float* base, size_t idx, float v, bool fx) {
  if (fx) atomicAdd(reinterpret_cast<unsigned long long*>(base) + idx, dm_to_fx(v));
  else atomicAdd(base + idx, v);
}

typedef float dm_f32x16 __attribute__((ext_vector_type(16)));
typedef float dm_f32x4 __attribute__((ext_vector_type(4)));
