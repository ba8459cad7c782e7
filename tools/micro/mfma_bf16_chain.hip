// Micro-benchmark behind K10 (round 5): how fast do v_mfma_f32_32x32x16_bf16 issue in the access pattern of the split GEMM?
//   mode 0: 8 accumulators, each MFMA on the next accumulator (independent)       -- the pipe's rate
//   mode 1: 8 accumulators, 6 consecutive MFMAs per accumulator (the split GEMM's chain), operands in registers
//   mode 2: mode 1 with the 18 fragment reads per 48 MFMAs from LDS (ds_read_b128), no barrier
//   mode 3: mode 2 with a workgroup barrier per 48 MFMAs
//   mode 4: the persistent kernel's order: per block of 6 MFMAs the next block's 3 fragment reads, barrier per 48
//   mode 5: mode 4 with fragments carried as dwords and cast at the MFMA (no v_perm)
// 512 threads (2 waves per SIMD), one workgroup per CU.   hipcc -O3 --offload-arch=gfx950 mfma_bf16_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* in, float* out, int iters) {
  __shared__ u32x4 lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 fa[12], fb[6];
  for (int i = 0; i < 12; ++i) fa[i] = __builtin_bit_cast(bf16x8, lds[(i * 64 + lane) & 4095]);
  for (int i = 0; i < 6; ++i) fb[i] = __builtin_bit_cast(bf16x8, lds[(i * 64 + lane + 1024) & 4095]);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {
      const int base = (it & 1) * 2048 + (tid >> 6) * 64;
#pragma unroll
      for (int i = 0; i < 12; ++i) fa[i] = __builtin_bit_cast(bf16x8, lds[(base + i * 96 + lane) & 4095]);
#pragma unroll
      for (int i = 0; i < 6; ++i) fb[i] = __builtin_bit_cast(bf16x8, lds[(base + 1200 + i * 96 + lane) & 4095]);
    }
    if (MODE >= 4) {
      const int base = (it & 1) * 2048 + (tid >> 6) * 32;
      u32x4 a3[3], b3[2][3];
#pragma unroll
      for (int t = 0; t < 3; ++t) a3[t] = lds[(base + t * 600 + lane) & 4095];
#pragma unroll
      for (int t = 0; t < 3; ++t) b3[0][t] = lds[(base + 1800 + t * 300 + lane) & 4095];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) {
#pragma unroll
          for (int t = 0; t < 3; ++t) b3[(j + 1) & 1][t] = lds[(base + 1800 + t * 300 + (j + 1) * 32 + lane) & 4095];
        }
        f32x16 c = acc[j];
#define MF(x, y) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)
        MF(b3[j & 1][0], a3[2]); MF(b3[j & 1][1], a3[1]); MF(b3[j & 1][2], a3[0]);
        MF(b3[j & 1][0], a3[1]); MF(b3[j & 1][1], a3[0]); MF(b3[j & 1][0], a3[0]);
#undef MF
        acc[j] = c;
        if (MODE == 4) __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      continue;
    }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < 8; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(a >> 1) * 3 + q % 3], fb[(a & 1) * 3 + q / 2], acc[a], 0, 0, 0);
    } else {
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int i = a >> 1, j = a & 1;
        f32x16 c = acc[a];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 2], fb[j * 3 + 0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 1], fb[j * 3 + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 0], fb[j * 3 + 2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 1], fb[j * 3 + 0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 0], fb[j * 3 + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i * 3 + 0], fb[j * 3 + 0], c, 0, 0, 0);
        acc[a] = c;
      }
    }
    if (MODE == 3) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 512 + tid] = s;
}

int main() {
  u32x4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 512 * 4);
  std::vector<unsigned> h(4096 * 4);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x3fff3fffu) | 0x3c003c00u; }   // random bf16 pairs near 1
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, in, out, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, in, out, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, in, out, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, in, out, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, in, out, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 256.0 * 8 * iters * 48 * 2.0 * 32 * 32 * 16;
      if (rep == 2) printf("mode %d: %.3f ms  %.1f TFLOP/s bf16  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", mode, ms,
                           flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 48.0 * 2));
    }
  }
  return 0;
}
