// valu_rate.hip -- microbenchmark: issue rate of the FP32 VALU instructions K1 is built
// from (plain vs packed), at 1/2/4/8 waves per SIMD, on every CU.  Standalone:
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
// Prints wave-instructions per ns per SIMD and the implied cycles per instruction at the
// measured shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;
constexpr int OPS_PER_ITER = 32;

enum Op { FMA, PK_FMA, PK_ADD, PK_MUL, MIN3, ADD, CNDMASK, K1MIX };

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* clk) {
  v2f a[8];
  float s[8];
  float seed = out[threadIdx.x & 7];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (v2f){seed + i, seed - i}; s[i] = seed * i; }
  v2f b = {seed, seed + 1.0f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "v"(b.x));
        if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(b.y));
        if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b.x));
      }
    }
    if (OP == K1MIX) {
      // the K1 inner body: per 2 pairs {3 pk_add, pk_mul, 2 pk_fma, min3}; 4 copies + 4 extra
      // = 32 instructions per iteration
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v2f dx, dy, dz, d;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(a[i]), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(a[i + 4]), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(a[(i + 1) & 7]), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d) : "v"(dx));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dy));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dz));
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(d.x), "v"(d.y));
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(s[i + 4]) : "v"(d.y), "v"(d.x));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
void run(const char* name, float* out, unsigned long long* clk) {
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps;  // 256-thread blocks: 1 wave per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    unsigned long long h[2];
    CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
    const double winstr = (double)ITERS * OPS_PER_ITER * wps;  // per SIMD
    const double ns = ms * 1e6;
    printf("%-8s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.3f wave-instr/ns/SIMD  = %.2f cycles/instr\n",
           name, wps, ms, ghz, winstr / ns, ns * ghz / winstr);
  }
}

int main() {
  float* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMemset(out, 0, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&clk, 16));
  run<FMA>("fma", out, clk);
  run<PK_FMA>("pk_fma", out, clk);
  run<PK_ADD>("pk_add", out, clk);
  run<PK_MUL>("pk_mul", out, clk);
  run<MIN3>("min3", out, clk);
  run<ADD>("add", out, clk);
  run<CNDMASK>("cndmask", out, clk);
  run<K1MIX>("k1mix", out, clk);
  return 0;
}
