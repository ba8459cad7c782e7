// Can scalar fp32 VALU work of ONE wave run under the v_mfma_f32_16x16x4_f32 stream of ANOTHER wave on the
// same SIMD?  Workgroup of 8 waves (2 per SIMD): waves 0-3 issue only MFMAs, waves 4-7 only v_fma_f32.
// Prints cycles for each kind alone and for both together.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

// mode bit 0: MFMA waves work, bit 1: VALU waves work; vper = VALU instructions per MFMA-equivalent slot
template <int VPER>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  v4f acc[36];
  for (int i = 0; i < 36; ++i) acc[i] = (v4f){0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float f[12];
  for (int i = 0; i < 12; ++i) f[i] = i + threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 36; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 36 * VPER; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i % 12]) : "v"(b));
      }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0;
  for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 12; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

template <int VPER>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    k<VPER><<<256, 512>>>(out, 10, mode, cyc);
    hipDeviceSynchronize();
    k<VPER><<<256, 512>>>(out, iters, mode, cyc);
    hipDeviceSynchronize();
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double n = 36.0 * iters;
    printf("VALU per MFMA slot %2d  mode %s: MFMA wave %6.1f cycles per MFMA, VALU wave %6.1f cycles per slot (%4.2f per v_fma)\n", VPER,
           mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both     ", c[0] / n, c[1] / n, c[1] / n / VPER);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc);
  return 0;
}
