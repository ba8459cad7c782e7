// winograd_dw.hip -- K6w: WEIGHT gradient of a Winograd F(4x4,3x3) convolution with 64 input channels in one pass
// (gfx950): dU[xi][k][c] = sum over tiles of dM[xi][k][tile] * V[xi][c][tile], with both transform-domain operands
// formed in registers -- dM = A dY A^T from the 4x4 tile of dy, V = B^T d B from the 6x6 patch of the layer's input --
// and fed straight to v_mfma_f32_16x16x4_f32.  Neither dM nor V (2.25 x the activation tensor each: 1.07 GB for
// conv1_2 of VGG16, src/models/image_net.py:14) reaches HBM; the three-kernel form spends an input transform, a
// grad-output transform and a GEMM with a 116,032-long reduction on them (1.1 ms; this kernel: see DESIGN.md).
//
//   * a wave owns one (16 output channels) x (16 input channels) block of dU for all 36 transform points: 144
//     accumulator registers (AGPRs), carried over the wave's range of tiles; an MFMA step reduces over 4 tiles:
//     lane (kk, col) forms dM of (output channel kb*16 + col, tile kk of the step) -- the A operand -- and V of (input
//     channel wave*16 + col, the same tile) -- the B operand; the four waves of a workgroup share the dy tiles;
//   * a step's four tiles are horizontally adjacent (the tile row length must be a multiple of 4);
//   * a step's operands are staged in LDS by LDS-DMA loads (coalesced along the image rows); two workgroups per CU
//     cover each other's DMA round trips; out-of-image rows and columns enter the transform with weight 0 (as in
//     winograd_fused.hip); the input may be the PRE-BatchNorm tensor of a folded layer (ACT);
//   * workgroup (kb, range) writes its block as one partial [36][16][64]; a second kernel sums the partials of a
//     block in a fixed order in fp64.  Deterministic.
// Workgroups of the same tile range sit on one XCD (id % 8), so the input tensor is fetched once per range.
// fp32 MFMAs and fp32 VALU instructions do not overlap on gfx950: scalar transforms, few instructions, products as
// in-place asm blocks on AGPR tuples (compiled with -fno-slp-vectorize; see winograd_fused.hip).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kDwThreads = 256;

// B^T d for one column (winograd.hip's Wino<4>::in), d[0] entering through `c0` (4, or 0 for a masked row/column)
__device__ __forceinline__ void in4(const float (&d)[6], float c0, float (&t)[6]) {
  const float p = fma_rn(-4.0f, d[2], d[4]), q = fma_rn(-4.0f, d[1], d[3]);
  const float r = d[4] - d[2], s = d[3] - d[1];
  t[0] = fma_rn(c0, d[0], fma_rn(-5.0f, d[2], d[4]));
  t[1] = p + q;
  t[2] = p - q;
  t[3] = fma_rn(2.0f, s, r);
  t[4] = fma_rn(-2.0f, s, r);
  t[5] = fma_rn(4.0f, d[1], fma_rn(-5.0f, d[3], d[5]));
}
// A y for one column of the output gradient (winograd.hip's Wino<4>::gout)
__device__ __forceinline__ void gout4(const float (&y)[4], float (&r)[6]) {
  const float e = y[0] + y[2], o = y[1] + y[3];
  r[0] = y[0];
  r[1] = e + o;
  r[2] = e - o;
  const float e4 = fma_rn(4.0f, y[2], y[0]), o4 = fma_rn(8.0f, y[3], 2.0f * y[1]);
  r[3] = e4 + o4;
  r[4] = e4 - o4;
  r[5] = y[3];
}

// LDS staging of one MFMA step (4 horizontally adjacent tiles), filled by LDS-DMA loads (buffer_load_dwordx4 ... lds:
// a wave instruction moves 64 x 16 bytes from per-lane global addresses to 1 KB of consecutive LDS):
//   x : [6 patch rows][64 channels][6 chunks of 4 floats] -- chunk 0 / 5 hold the halo columns, chunks 1..4 the four
//       tiles; 36 wave instructions.  A lane's 16-byte reads (channel = its column of 16 lanes) are 24 floats apart
//       between neighbouring lanes: 2-way bank conflicts, no padding -- 40 KB per stage so that TWO stages of TWO
//       workgroups fill the CU's 160 KB exactly;
//   dy: [16 output channels][16 chunks], chunk (row a, tile kk) of channel k stored at slot (4a + kk + (k & 7)) & 15 (the
//       channels are 64 floats = a whole bank cycle apart; the rotation spreads 8 lanes over the 8 bank groups); 4
//       wave instructions.
// Reading the operands straight from global memory costs one cache access per LANE (a lane's neighbours hold other
// channels, 200 KB apart): 64 accesses per instruction bound the first version at 1.2 ms for conv1_2's shape.
constexpr int kXInstr = 36, kGInstr = 4;
constexpr int kStageFloats = (kXInstr + kGInstr) * 64 * 4;          // 40 KB per buffer
constexpr int kDma = 10;                             // DMA instructions per wave and step: 9 of x, 1 of dy

template <bool ACT>
__global__ __launch_bounds__(kDwThreads) void wino4_dw_c64_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  int K, int H, int W, int Th, int Tw, int n_img,
                                                                  long n_steps, int KB, int spr,
                                                                  float* __restrict__ part /*[R][36][K][64]*/,
                                                                  const float* __restrict__ chan,
                                                                  const float* __restrict__ pre_bias) {
  constexpr int C = 64;
  extern __shared__ __attribute__((aligned(16))) float stage[];      // two buffers of kStageFloats
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 4, col = lane & 15;
  const int i = blockIdx.x;
  const int kb = (i >> 3) % KB;
  const int rg = (i & 7) + 8 * (i / (8 * KB));              // tile range of this workgroup
  const long s0 = (long)rg * spr;
  const long s1 = (s0 + spr) < n_steps ? (s0 + spr) : n_steps;
  const int cx = wave * 16 + col;                           // input channel of this lane's B operand
  float* out = part + (((size_t)rg * 36) * K + kb * 16 + 4 * kk) * C + cx;
  if (s0 >= s1) {                                // a range without steps (more ranges than steps): a zero partial
#pragma unroll
    for (int xi = 0; xi < 36; ++xi)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((size_t)xi * K + r) * C] = 0.0f;
    return;
  }
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)((size_t)n_img * C * H * W * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (int)((size_t)n_img * K * H * W * sizeof(float)), 0x00020000);
  float asc = 1.0f, ash = 0.0f, apb = 0.0f;
  if (ACT) { asc = chan[cx]; ash = chan[C + cx]; apb = pre_bias ? pre_bias[cx] : 0.0f; }

  // ---- this wave's share of a step's DMA instructions: x instructions wave, wave + 4, ..., wave + 32 and dy instruction
  // `wave`.  Per instruction a lane fetches one 16-byte chunk, the same chunk of the step's window in every step:
  // byte offset = rowoff[d] (everything but the tile column; recomputed when the tile row changes) + 16 * tw0.
  int fix[kDma];             // float offset of the lane's chunk for (image 0, tile row 0, tile column 0), without its row
  int rsel[9];               // x: the patch row 0..5 of the chunk (clamped rows change with the tile row)
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    const int p = (wave + 4 * d) * 64 + lane;                          // LDS chunk slot = [row][channel][chunk]
    const int r = p / 384, rem = p - r * 384, ch = rem / 6, q = rem - 6 * ch;
    rsel[d] = r;
    fix[d] = ch * H * W + 4 * q - 4;
  }
  {
    const int p = wave * 64 + lane, k = p >> 4, j = ((p & 15) - (k & 7)) & 15;    // slot -> (row a, tile kk) of channel k
    fix[9] = ((kb * 16 + k) * H + (j >> 2)) * W + 4 * (j & 3);
  }
  // the step's position: tile index 4*s = (image n, tile row th, tile column tw0); all wave-uniform
  int tw0, th;
  unsigned n;
  {
    const unsigned p0 = (unsigned)(4 * s0);
    tw0 = (int)(p0 % (unsigned)Tw);
    const unsigned q = p0 / (unsigned)Tw;
    th = (int)(q % (unsigned)Th);
    n = q / (unsigned)Th;
  }
  uint32_t rowoff[kDma];
  auto place_row = [&]() {
    const int r0 = 4 * th - 1;
    const int xbase = (int)(n * (unsigned)(C * H * W)), gbase = (int)(n * (unsigned)(K * H * W)) + 4 * th * W;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const int row = r0 + rsel[d];
      rowoff[d] = (uint32_t)(xbase + fix[d] + (row < 0 ? 0 : (row >= H ? H - 1 : row)) * W) * 4u;
    }
    rowoff[9] = (uint32_t)(gbase + fix[9]) * 4u;
  };
  place_row();
  auto issue = [&](int buf) {                    // the DMA of the step at (n, th, tw0) into stage buffer `buf`
    const uint32_t cadv = (uint32_t)tw0 * 16u;   // wave-uniform; added per lane: a buffer load's range check looks at the
                                                 // lane offset alone (the window's first chunk starts at -16 B in row 0)
    float* sb = stage + buf * kStageFloats;
#pragma unroll
    for (int d = 0; d < 9; ++d)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(sb + (wave + 4 * d) * 256), 16,
                                               rowoff[d] + cadv, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(grsrc, (__attribute__((address_space(3))) void*)(sb + (kXInstr + wave) * 256), 16,
                                             rowoff[9] + cadv, 0, 0, 0);
  };
  auto advance = [&]() {
    tw0 += 4;
    if (tw0 >= Tw) {                             // wave-uniform: next tile row (or image)
      tw0 = 0;
      if (++th >= Th) { th = 0; ++n; }
      place_row();
    }
  };
  // zeroed once per kernel and accumulated in place by every step.  (A first step with C = 0 writing them as fresh
  // outputs lets the register allocator put copies right behind the asm block, i.e. reads of MFMA results without the
  // wait states it cannot know about.)
  v4f acc[36];
#pragma unroll
  for (int xi = 0; xi < 36; ++xi) acc[xi] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
  auto act = [&](float v) { return ACT ? __builtin_fmaxf(fma_rn(v + apb, asc, ash), 0.0f) : v; };
  int gslot[4];                                  // the lane's four dy chunks in its channel's rotated slots
#pragma unroll
  for (int a = 0; a < 4; ++a) gslot[a] = (4 * a + kk + (col & 7)) & 15;
  auto compute = [&](const float* sb, int tw, int thc, v4f (&accr)[36]) {
    // Out-of-image rows / columns enter the transform with weight 0 (as in winograd_fused.hip): rows r0+1 .. r0+4 are
    // always inside; the top row through the 4 of B^T's first row, the left column likewise in the second pass, the
    // bottom row and the right column by one multiply per element.
    const int r0 = 4 * thc - 1;
    const float c4t = r0 >= 0 ? 4.0f : 0.0f, mb = r0 + 5 < H ? 1.0f : 0.0f;
    const float c4l = tw > 0 ? 4.0f : 0.0f, mr = tw < Tw - 1 ? 1.0f : 0.0f, mr5 = mr * mb;
    // B operand: V = B^T d B of the 6x6 input patch of (channel cx, tile kk): three 16-byte LDS reads per row
    const v4f* xs = reinterpret_cast<const v4f*>(sb) + cx * 6 + kk;
    float d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const v4f lf = xs[r * 384], md = xs[r * 384 + 1], rt = xs[r * 384 + 2];
      const float l = act(lf[3]), m0 = act(md[0]), m1 = act(md[1]), m2 = act(md[2]), m3 = act(md[3]), rr = act(rt[0]);
      d[r][0] = r == 5 ? l * mb : l;                                 // (the left edge enters through c4l below)
      d[r][1] = r == 5 ? m0 * mb : m0;
      d[r][2] = r == 5 ? m1 * mb : m1;
      d[r][3] = r == 5 ? m2 * mb : m2;
      d[r][4] = r == 5 ? m3 * mb : m3;
      d[r][5] = rr * (r == 5 ? mr5 : mr);
    }
    float t[6][6];        // t[j][i]: column j after the transform along rows
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float colv[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
      in4(colv, c4t, t[j]);
    }
    // A operand: dM = A dY A^T of the 4x4 tile of dy of (output channel kb*16 + col, tile kk)
    const v4f* gs = reinterpret_cast<const v4f*>(sb + kXInstr * 256) + col * 16;
    v4f g[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) g[a] = gs[gslot[a]];
    float u[4][6];        // u[j][i]: column j after the transform along rows
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float colv[4] = {g[0][j], g[1][j], g[2][j], g[3][j]};
      gout4(colv, u[j]);
    }
#pragma unroll
    for (int i2 = 0; i2 < 6; ++i2) {
      const float rowb[6] = {t[0][i2], t[1][i2], t[2][i2], t[3][i2], t[4][i2], t[5][i2]};
      const float rowa[4] = {u[0][i2], u[1][i2], u[2][i2], u[3][i2]};
      float ob[6], oa[6];
      in4(rowb, c4l, ob);
      gout4(rowa, oa);
      // six products of the row as one block, in place on AGPR tuples; s_nop 1: the two wait states between a VALU
      // write and the MFMA reading it (the compiler does not see MFMAs inside asm)
      const int x0 = 6 * i2;
      asm volatile(
          "s_nop 1\n\t"
          "v_mfma_f32_16x16x4_f32 %0, %6, %12, %0\n\t"
          "v_mfma_f32_16x16x4_f32 %1, %7, %13, %1\n\t"
          "v_mfma_f32_16x16x4_f32 %2, %8, %14, %2\n\t"
          "v_mfma_f32_16x16x4_f32 %3, %9, %15, %3\n\t"
          "v_mfma_f32_16x16x4_f32 %4, %10, %16, %4\n\t"
          "v_mfma_f32_16x16x4_f32 %5, %11, %17, %5"
          : "+a"(accr[x0]), "+a"(accr[x0 + 1]), "+a"(accr[x0 + 2]), "+a"(accr[x0 + 3]), "+a"(accr[x0 + 4]),
            "+a"(accr[x0 + 5])
          : "v"(oa[0]), "v"(oa[1]), "v"(oa[2]), "v"(oa[3]), "v"(oa[4]), "v"(oa[5]),
            "v"(ob[0]), "v"(ob[1]), "v"(ob[2]), "v"(ob[3]), "v"(ob[4]), "v"(ob[5]));
    }
  };
  // ---- steps: two stage buffers, the DMA of step s + 1 under the transforms and products of step s, one barrier per
  // step (the next buffer is complete for every wave, and every wave is done reading the buffer that the DMA after next
  // overwrites); 80 KB of LDS and 105 + 144 registers let TWO workgroups share a CU, which covers the DMA instructions'
  // issue cost (~60-180 cycles each, 10 per wave and step: 46 % issue stalls with one workgroup per CU).
  issue(0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  int buf = 0;
  for (long s = s0; s < s1; ++s) {
    const int tw_now = tw0 + kk, th_now = th;
    if (s + 1 < s1) {
      advance();
      issue(buf ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(stage + buf * kStageFloats, tw_now, th_now, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    buf ^= 1;
  }
  // the compiler does not see MFMAs in the asm statements: cover the MFMA-write -> VALU-read distance by hand
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // accumulator element r of acc[xi] = dU[xi][kb*16 + 4*kk + r][wave*16 + col]
#pragma unroll
  for (int xi = 0; xi < 36; ++xi)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[((size_t)xi * K + r) * C] = acc[xi][r];
}

// dU[e] = sum over the R partials, in range order, in fp64
__global__ __launch_bounds__(256) void wino_dw_reduce_kernel(const float* __restrict__ part, int R, long n,
                                                             float* __restrict__ dU) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  double a = 0.0;
  for (int r = 0; r < R; ++r) a += (double)part[(size_t)r * n + e];
  dU[e] = (float)a;
}

// Tile ranges per channel block: two workgroups per CU at most, at least ~32 steps per range (each range costs a
// [36][16][64] partial written and read back: 75 MB for 128 ranges at K = 64), a multiple of 8 (XCD placement).
long dw_ranges(int KB, long n_steps) {
  long cap = ((512 + KB - 1) / KB + 7) / 8 * 8;
  long want = ((n_steps + 31) / 32 + 7) / 8 * 8;
  return want < cap ? want : cap;
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_wino_dw_fused_workspace_floats(int N, int K, int H, int W) {
  if (N <= 0 || K <= 0 || K % 16 || H <= 0 || W <= 0) return 0;
  if (H % 4 || W % 16) return 0;
  return (size_t)fpsg::dw_ranges(K / 16, (long)N * (H / 4) * (W / 4) / 4) * 36 * K * 64;
}

extern "C" int fpsg_wino_dw_fused(const float* x, const float* chan, const float* pre_bias, const float* dy, int N, int C,
                                  int K, int H, int W, float* dU, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  const char* fn = "fpsg_wino_dw_fused";
  FPSG_REQUIRE(C == 64, FPSG_E_SHAPE, "%s: C must be 64 (got %d)", fn, C);
  FPSG_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && K <= 1024 && H > 0 && W > 0 && H % 4 == 0 && W % 16 == 0, FPSG_E_SHAPE,
               "%s: K a positive multiple of 16 (<= 1024), H a multiple of 4 and W of 16 (four tiles per step lie "
               "in one tile row; got K=%d H=%d W=%d)", fn, K, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(dU); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, FPSG_E_ALIGN,
               "%s: x and dy must be 16-byte aligned", fn);
  FPSG_REQUIRE(!misaligned4(chan) && !misaligned4(pre_bias) && !misaligned4(dU) && !misaligned4(ws), FPSG_E_ALIGN,
               "%s: chan / pre_bias / dU / ws not 4-byte aligned", fn);
  FPSG_REQUIRE((size_t)N * (K > C ? K : C) * H * W * sizeof(float) < ((size_t)1 << 31), FPSG_E_LIMIT,
               "%s: x and dy must be below 2 GiB each (32-bit lane offsets; got N=%d K=%d H=%d W=%d)", fn, N, K, H, W);
  const int Th = H / 4, Tw = W / 4, KB = K / 16;
  const long n_steps = (long)N * Th * Tw / 4;
  const long R = dw_ranges(KB, n_steps);
  const int spr = (int)((n_steps + R - 1) / R);
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)(R * KB));
  const size_t lds_bytes = 2 * (size_t)kStageFloats * sizeof(float);
  typedef void (*kern_t)(const float*, const float*, int, int, int, int, int, int, long, int, int, float*, const float*,
                         const float*);
  const kern_t kern = chan ? wino4_dw_c64_kernel<true> : wino4_dw_c64_kernel<false>;
  const hipError_t lds_optin = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_optin != hipSuccess) {
    set_error("%s: cannot reserve %zu B of LDS: %s", fn, lds_bytes, hipGetErrorString(lds_optin));
    return static_cast<int>(lds_optin);
  }
  hipLaunchKernelGGL(kern, grid, dim3(kDwThreads), lds_bytes, s, x, dy, K, H, W, Th, Tw, N, n_steps, KB, spr, ws, chan,
                     chan ? pre_bias : nullptr);
  int rc = launch_status(fn);
  if (rc) return rc;
  const long n = 36L * K * 64;
  hipLaunchKernelGGL(wino_dw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, (int)R, n, dU);
  return launch_status("fpsg_wino_dw_fused(reduce)");
}
