// Do scalar fp32 VALU instructions hide in the issue gaps of v_mfma_f32_16x16x4_f32 with ONE wave per SIMD?
// Each wave runs ITER x (36 MFMAs on 36 accumulators, FILL v_fma_f32 after each MFMA); prints cycles per MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_fill.hip -o /tmp/mfma_fill && /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FILL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, long long* cyc) {
  v4f acc[36];
  for (int i = 0; i < 36; ++i) acc[i] = (v4f){0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float f[12];
  for (int i = 0; i < 12; ++i) f[i] = i + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < FILL; ++j)
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(i * FILL + j) % 12]) : "v"(b));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0;
  for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 12; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int FILL, int WAVES>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<FILL, WAVES><<<256, 64 * WAVES>>>(out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<FILL, WAVES><<<256, 64 * WAVES>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 36.0 * iters;
  printf("waves/SIMD %d  fill %2d: %7.1f ns per MFMA slot (wall), s_memtime %6.1f ticks per MFMA (100 MHz ticks x%.1f)\n",
         WAVES / 4, FILL, ms * 1e6 / n, (double)c / n, 1.0);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0, 4>(out, cyc); run<2, 4>(out, cyc); run<4, 4>(out, cyc); run<6, 4>(out, cyc); run<8, 4>(out, cyc); run<12, 4>(out, cyc);
  run<0, 8>(out, cyc); run<6, 8>(out, cyc); run<12, 8>(out, cyc);
  return 0;
}
