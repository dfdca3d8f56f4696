// Shared helpers for libdtt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dtt_hip.h"

#define DTT_WAVE 64

void dtt_set_error(const char* fmt, ...);

// Reference launchers print and return 0 on a launch error (correlation_cuda_kernel.cu:362-368);
// same contract, the message is kept for dtt_last_error().
#define DTT_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t _e = hipGetLastError();                                      \
    if (_e != hipSuccess) {                                                 \
      dtt_set_error("%s: launch failed: %s", name, hipGetErrorString(_e)); \
      return 0;                                                             \
    }                                                                       \
  } while (0)

#define DTT_REQUIRE(cond, ...)    \
  do {                            \
    if (!(cond)) {                \
      dtt_set_error(__VA_ARGS__); \
      return 0;                   \
    }                             \
  } while (0)

// Optional per-kernel timing hook (bench.py / rocprof cross-check): when a tag is attached, the launcher of
// the kernel with that tag brackets each launch with hipEventRecord on the launch stream.
void dtt_prof_begin(const char* tag, hipStream_t stream);
void dtt_prof_end(const char* tag, hipStream_t stream);

// "hipFuncSetAttribute already applied" flags: function attributes (the dynamic-LDS limit) are per DEVICE, so a
// process that drives several GPUs needs one flag per device.  Racing first calls merely set the attribute twice.
struct DttDeviceOnce {
  bool done[64] = {};
  bool& here() {
    int d = 0;
    (void)hipGetDevice(&d);
    return done[d & 63];
  }
};

// compute units of the current device, cached per device (hipGetDeviceProperties costs ~10 us of host time per call)
int dtt_device_cus();

static inline int dtt_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware remap of a linear workgroup id (8 XCDs; block b runs on XCD b % 8): gives each XCD a
// contiguous chunk of the logical grid so neighbouring tiles share that XCD's L2.  Bijective for any n.
__device__ __forceinline__ int dtt_xcd_remap(int bid, int n) {
  const int nx = 8;
  int q = n / nx, r = n % nx;
  int xcd = bid % nx, k = bid / nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
