// fpsg_common.h -- shared helpers of libfpsg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fpsg_hip.h"

namespace fpsg {

// thread-local message for fpsg_last_error(); defined in capi.hip
void set_error(const char* fmt, ...);

inline bool misaligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }

// Checks the launch that was just enqueued; never synchronises.
inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}

#define FPSG_REQUIRE_PTR(p)                                   \
  do {                                                        \
    if ((p) == nullptr) {                                     \
      fpsg::set_error("%s: null pointer '%s'", __func__, #p); \
      return FPSG_E_NULL;                                     \
    }                                                         \
    if (fpsg::misaligned4(p)) {                               \
      fpsg::set_error("%s: '%s' not 4-byte aligned", __func__, #p); \
      return FPSG_E_ALIGN;                                    \
    }                                                         \
  } while (0)

#define FPSG_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      fpsg::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ v2f fma_rn(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

}  // namespace fpsg
