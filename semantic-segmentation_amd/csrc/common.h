// Shared helpers for libssmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ssmi.h"

namespace ssmi {

void set_error(const char* fmt, ...);
// Records which kernel variant the calling thread's last convolution entry point dispatched (profiling aid).
void set_last_kernel(const char* fmt, ...);

#define SSMI_REQUIRE(cond, code, ...)      \
  do {                                     \
    if (!(cond)) {                         \
      ::ssmi::set_error(__VA_ARGS__);      \
      return (code);                       \
    }                                      \
  } while (0)

#define SSMI_LAUNCH_CHECK()                                                        \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::ssmi::set_error("%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SSMI_E_HIP;                                                           \
    }                                                                              \
  } while (0)

#define SSMI_HIP(call)                                                             \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      ::ssmi::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return SSMI_E_HIP;                                                           \
    }                                                                              \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave-wide (64 lanes) sum.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace ssmi
