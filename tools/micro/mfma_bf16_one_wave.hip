// Micro-benchmark behind K10's specialised-wave form (round 5): ONE multiplying wave per SIMD (256 threads per CU) --
// how fast do v_mfma_f32_32x32x16_bf16 issue when the chain per accumulator is dependent?
//   mode 0: 8 accumulators round robin (independent neighbours)
//   mode 1: 6 consecutive MFMAs per accumulator (K10's chain as written)
//   mode 2: two accumulators alternate (dependent distance 2), 6 each
//   mode 3: four accumulators alternate (distance 4)
//   mode 4: mode 2 + the consumer's fragment reads (6 + 12 ds_read_b128 per 48 MFMAs) + a barrier per 48
//   mode 5: mode 1 + the same reads and barrier
// hipcc -O3 --offload-arch=gfx950 mfma_bf16_one_wave.hip -o mfma_one_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const u32x4* in, float* out, int iters) {
  __shared__ u32x4 lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += THREADS) lds[i] = in[i];
  __syncthreads();
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  u32x4 fa[2][3], fb[4][3];
  for (int i = 0; i < 2; ++i)
    for (int t = 0; t < 3; ++t) fa[i][t] = lds[(i * 192 + t * 64 + lane) & 4095];
  for (int j = 0; j < 4; ++j)
    for (int t = 0; t < 3; ++t) fb[j][t] = lds[(1024 + j * 192 + t * 64 + lane) & 4095];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
      const int base = (it & 1) * 2048 + (tid >> 6) * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) fa[i][t] = lds[(base + i * 32 + t * 600 + lane) & 4095];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t) fb[j][t] = lds[(base + 1800 + t * 300 + j * 32 + lane) & 4095];
    }
    if (MODE == 0) {
      // products (t, u) of the six, accumulators round robin
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = MFMA(fb[a >> 1][TB[q]], fa[a & 1][TA[q]], acc[a]);
    } else if (MODE == 1 || MODE == 5) {
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int j = a >> 1, i = a & 1;
        f32x16 c = acc[a];
        c = MFMA(fb[j][0], fa[i][2], c); c = MFMA(fb[j][1], fa[i][1], c); c = MFMA(fb[j][2], fa[i][0], c);
        c = MFMA(fb[j][0], fa[i][1], c); c = MFMA(fb[j][1], fa[i][0], c); c = MFMA(fb[j][0], fa[i][0], c);
        acc[a] = c;
      }
    } else if (MODE == 2 || MODE == 4) {
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x16 c0 = acc[2 * j], c1 = acc[2 * j + 1];
#pragma unroll
        for (int q = 0; q < 6; ++q) { c0 = MFMA(fb[j][TB[q]], fa[0][TA[q]], c0); c1 = MFMA(fb[j][TB[q]], fa[1][TA[q]], c1); }
        acc[2 * j] = c0; acc[2 * j + 1] = c1;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[4 * jj + a] = MFMA(fb[2 * jj + (a >> 1)][TB[q]], fa[a & 1][TA[q]], acc[4 * jj + a]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE >= 4) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * THREADS + tid] = s;
}

template <int MODE, int THREADS>
void run(const u32x4* in, float* out, int iters, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double waves = THREADS / 64, flop = 256.0 * waves * iters * 48 * 2.0 * 32 * 32 * 16;
  printf("mode %d, %d waves per SIMD: %.3f ms  %.1f TFLOP/s bf16  (%.1f cycles per MFMA per SIMD at 2.4 GHz)  %s\n", MODE, THREADS / 256, ms,
         flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 48.0 * (THREADS / 256)), what);
}

int main() {
  u32x4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 512 * 4);
  std::vector<unsigned> h(4096 * 4);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x3fff3fffu) | 0x3c003c00u; }
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  run<0, 256>(in, out, iters, "8 accumulators round robin");
  run<1, 256>(in, out, iters, "6 consecutive MFMAs per accumulator");
  run<2, 256>(in, out, iters, "two accumulators alternate");
  run<3, 256>(in, out, iters, "four accumulators alternate");
  run<4, 256>(in, out, iters, "two alternate + 18 fragment reads + barrier per 48");
  run<5, 256>(in, out, iters, "6 consecutive + 18 fragment reads + barrier per 48");
  run<0, 512>(in, out, iters, "8 accumulators round robin");
  run<1, 512>(in, out, iters, "6 consecutive MFMAs per accumulator");
  run<2, 512>(in, out, iters, "two accumulators alternate");
  run<4, 512>(in, out, iters, "two alternate + 18 fragment reads + barrier per 48");
  return 0;
}
