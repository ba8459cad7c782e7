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

// Streams through tensors larger than the 256 MB Infinity Cache (touched once per pass) use non-temporal loads and
// stores: the data does not sweep the cache, and the passes run 13-25 % faster (profiles/r03/nontemporal_streams.txt);
// a tensor that fits the cache stays there for its consumer and must NOT get the hint (15-75 % slower).  NT is a
// template parameter of the kernels (behind a run-time branch the compiler merges the two accesses and drops the
// hint); the launchers pick the instantiation with `beyond_cache(bytes of the largest tensor of the pass)`.
#ifndef FPSG_NT_BYTES
#define FPSG_NT_BYTES ((size_t)300 << 20)
#endif
inline bool beyond_cache(size_t bytes) { return bytes > (size_t)FPSG_NT_BYTES; }
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ v2f fma_rn(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Lane exchange v[lane ^ M] without touching memory: DPP inside a row of 16 lanes (M = 4, 8 as two
// mirrors: (l^7)^3 = l^4, (l^15)^7 = l^8), the swizzle crossbar for 16, a half-wave swap for 32.
template <int M>
__device__ __forceinline__ unsigned lane_xor(unsigned v) {
  const int x = (int)v;
  if (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  if (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  if (M == 4) {
    const int h = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false);               // row_half_mirror: l^7
    return (unsigned)__builtin_amdgcn_update_dpp(h, h, 0x1B, 0xF, 0xF, false);             // quad_perm [3,2,1,0]: ^3
  }
  if (M == 8) {
    const int h = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false);               // row_mirror: l^15
    return (unsigned)__builtin_amdgcn_update_dpp(h, h, 0x141, 0xF, 0xF, false);            // row_half_mirror: ^7
  }
  if (M == 16) return (unsigned)__builtin_amdgcn_ds_swizzle(x, (16 << 10) | 0x1F);         // bit mode: xor 16
  return (unsigned)__shfl_xor(x, 32, 64);
}


// Sum over the 64 lanes of a wave, every lane receives the total; a fixed balanced tree (deterministic).
__device__ __forceinline__ float wave_sum(float v) {
  v += __uint_as_float(lane_xor<1>(__float_as_uint(v)));
  v += __uint_as_float(lane_xor<2>(__float_as_uint(v)));
  v += __uint_as_float(lane_xor<4>(__float_as_uint(v)));
  v += __uint_as_float(lane_xor<8>(__float_as_uint(v)));
  v += __uint_as_float(lane_xor<16>(__float_as_uint(v)));
  v += __uint_as_float(lane_xor<32>(__float_as_uint(v)));
  return v;
}

// Sums of four values over the 64 lanes, delivered in lane 63 (other lanes: partial sums): the row_shr / row_bcast DPP
// tree -- six dependent adds per value, no LDS crossbar operation, the four chains interleaved so that a DPP operand
// is never read within two instructions of its write (the first s_nop covers the producers).  Fixed order.
__device__ __forceinline__ void wave_sum4_to_last(float& a, float& b, float& c, float& d) {
#define FPSG_DPP_ADD4(ctrl)                          \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n"              \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n"              \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n"              \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n"
  asm volatile("s_nop 1\n"
               FPSG_DPP_ADD4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               FPSG_DPP_ADD4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               FPSG_DPP_ADD4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               FPSG_DPP_ADD4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               FPSG_DPP_ADD4("row_bcast:15 row_mask:0xa bank_mask:0xf")
               FPSG_DPP_ADD4("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef FPSG_DPP_ADD4
}

}  // namespace fpsg
