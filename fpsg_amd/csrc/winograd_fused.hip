// winograd_fused.hip -- K6f: Winograd F(4x4,3x3) convolution for 64 input channels in ONE kernel:
// input transform, the 36 transform-domain products on the fp32 MFMA pipes and the output
// transform, without the transform-domain tensors ever reaching HBM (gfx950).
//
// With 64 channels the library GEMMs of the three-kernel form (winograd.hip) are memory-bound:
// for conv1_2 of VGG16 (64 -> 64 @224x224, 37 images; src/models/image_net.py:14) V and M are
// 1.07 GB each, i.e. 4.3 GB of traffic around 34 GFLOP of products, next to 0.95 GB for the
// image tensors themselves.  Here:
//   * a workgroup owns 16 output channels: its slice U[36][16][64] of the transformed filter
//     (144 KiB) sits in LDS for the workgroup's lifetime, laid out as MFMA A fragments;
//   * a wave owns 16 consecutive tiles at a time.  For each group of 4 input channels a lane
//     (channel c = 4*step + lane/16, tile = lane%16) loads its tile's 6x6 patch -- interior columns
//     as one aligned vector, halo columns from the neighbouring lanes -- transforms it (the 36
//     values ARE the B fragments of v_mfma_f32_16x16x4_f32 for that step) and issues 36 MFMAs,
//     one per transform point, accumulating M[xi][16 k][16 tiles] in 144 accumulator VGPRs;
//   * after the 16 channel steps a lane holds M[0..35] for its tile and 4 output channels:
//     the output transform runs in registers and the 4x4 pixels are stored as aligned vectors.
// Workgroup -> (slice, tile range) mapping keeps the workgroups of one tile range on one XCD
// (round-robin dispatch: id % 8), so that the re-reads of x by the other slices hit that L2.
// Same arithmetic per element as winograd.hip's transforms; the channel sum runs in the MFMA's
// k order.  Deterministic.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kFusedThreads = 256;
constexpr int kSteps = 16;          // 64 input channels / 4 per MFMA step
constexpr int kUldsFloats = 36 * kSteps * 64;

// the same transform on two independent columns per instruction (v_pk_fma_f32 / v_pk_add_f32)
__device__ __forceinline__ void in4x2(const v2f (&d)[6], v2f (&t)[6]) {
  const v2f c4 = {4.0f, 4.0f}, cm5 = {-5.0f, -5.0f}, cm4 = {-4.0f, -4.0f}, c2 = {2.0f, 2.0f};
  t[0] = fma_rn(c4, d[0], fma_rn(cm5, d[2], d[4]));
  t[1] = fma_rn(cm4, d[1] + d[2], d[3] + d[4]);
  t[2] = fma_rn(c4, d[1] - d[2], d[4] - d[3]);
  t[3] = fma_rn(c2, d[3] - d[1], d[4] - d[2]);
  t[4] = fma_rn(c2, d[1] - d[3], d[4] - d[2]);
  t[5] = fma_rn(c4, d[1], fma_rn(cm5, d[3], d[5]));
}
// lane i <- lane i-1 / i+1 inside its row of 16 lanes (DPP row_shr:1 / row_shl:1): one VALU op
__device__ __forceinline__ float from_left(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, false));
}
__device__ __forceinline__ void out4(const float (&m)[6], float (&s)[4]) {
  const float p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
  s[0] = (m[0] + p12) + p34;
  s[1] = fma_rn(2.0f, m34, m12);
  s[2] = fma_rn(4.0f, p34, p12);
  s[3] = fma_rn(8.0f, m34, m12) + m[5];
}

// ACT: x is the PRE-BatchNorm output of the previous convolution; the patch values are
// relu(fma(x + pre_bias[c], scale[c], shift[c])) (K5's apply arithmetic), padding stays zero.
template <bool ACT>
__global__ __launch_bounds__(kFusedThreads) void wino4_fused_c64_kernel(const float* __restrict__ x,
                                                                        const float* __restrict__ U /*[36][K][64]*/,
                                                                        int K, int H, int W, int Th, int Tw, long P,
                                                                        int S, float* __restrict__ y,
                                                                        const float* __restrict__ chan,
                                                                        const float* __restrict__ pre_bias) {
  extern __shared__ __attribute__((aligned(16))) float ulds[];      // [36][kSteps][64] A fragments | [3][64] scale, shift, pre-bias
  constexpr int C = 64;
  float* actp = ulds + kUldsFloats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, col = lane & 15;
  const int i = blockIdx.x;
  const int slice = (i >> 3) % S;
  const int j = (i & 7) + 8 * (i / (8 * S));                        // index among the slice's workgroups
  const int per_slice = gridDim.x / S;
  const int k0 = slice * 16;
  // ulds[(c4*64 + lane)*36 + xi] = U[xi][k0 + lane%16][4*c4 + lane/16]: a lane's 36 A fragments of a
  // step are contiguous (nine 16-byte LDS reads; 8 lanes x 16 B cover the 32 banks once)
  for (int e = tid; e < kUldsFloats; e += kFusedThreads) {               // reads of U coalesced (c fastest)
    const int xi = e / (16 * C), rem = e - xi * (16 * C);
    const int kl = rem >> 6, c = rem & 63;
    const int c4 = c >> 2, kq = c & 3;
    ulds[((size_t)c4 * 64 + kq * 16 + kl) * 36 + xi] = U[((size_t)xi * K + k0 + kl) * C + c];
  }
  if (ACT && tid < C) {
    actp[tid] = chan[tid];
    actp[C + tid] = chan[C + tid];
    actp[2 * C + tid] = pre_bias ? pre_bias[tid] : 0.0f;
  }
  __syncthreads();
  const long G = (P + 15) >> 4;
  for (long g = (long)j * 4 + wave; g < G; g += 4L * per_slice) {
    const long p_raw = g * 16 + col;
    const bool live = p_raw < P;
    const long p = live ? p_raw : P - 1;
    const int tw = (int)(p % Tw);
    const long q = p / Tw;
    const int th = (int)(q % Th);
    const long n = q / Th;
    const bool has_left = tw > 0, has_right = tw < Tw - 1;
    const bool left_lane = has_left && col > 0, right_lane = has_right && col < 15;
    const int r0 = 4 * th - 1;
    v4f acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    const float* xn = x + (((size_t)n * C + kk) * H) * W + 4 * tw;      // channel kk of step 0
    const size_t step_stride = (size_t)4 * H * W;                        // 4 channels per step
    const bool edge_l = has_left && !left_lane, edge_r = has_right && !right_lane;
    // rows r0+1 .. r0+4 are the tile's own output rows: always inside the image; only the halo rows
    // r0 (top) and r0+5 (bottom) can fall outside and are then read from a valid row and zeroed
    const bool top_in = r0 >= 0, bot_in = r0 + 5 < H;
    int roff[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int row = r0 + r;
      roff[r] = (row < 0 ? 0 : (row >= H ? H - 1 : row)) * W;
    }

    // the patch of one channel step as loaded: interior vectors + the two edge-lane halo columns.
    // One wave per SIMD (the U slice fills the LDS), so the ~2 us of load latency is covered by
    // issuing the loads two steps ahead of their use.
    struct Raw { v4f mid[6]; float e[6]; };     // e: the halo column of a lane at the edge of its row of 16
    // no branches around the loads: the compiler then counts outstanding loads exactly
    // (s_waitcnt vmcnt(n) per buffer instead of vmcnt(0)); lanes without an edge halo re-read
    // their own first / last interior element
    // (a lane is at most at one edge: column 0 needs the element left of its vector, column 15 the
    // one right of it)
    const int eoff = edge_l ? -1 : (edge_r ? 4 : 0);
    auto load_raw = [&](int c4, Raw& w) {
      const float* xp = xn + (size_t)c4 * step_stride;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        w.mid[r] = *reinterpret_cast<const v4f*>(xp + roff[r]);
        w.e[r] = xp[roff[r] + eoff];
      }
    };
    auto compute = [&](int c4, const Raw& w) {
      // this step's 36 A fragments: issued first, in flight under the input transform
      v4f a[9];
      {
        const v4f* up = reinterpret_cast<const v4f*>(ulds + ((size_t)c4 * 64 + lane) * 36);
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) a[q4] = up[q4];
      }
      float asc = 1.0f, ash = 0.0f, apb = 0.0f;
      if (ACT) { asc = actp[4 * c4 + kk]; ash = actp[C + 4 * c4 + kk]; apb = actp[2 * C + 4 * c4 + kk]; }
      // d[r][c] as column pairs for the packed transforms: dp[r][j] = (d[r][2j], d[r][2j+1])
      v2f dp[6][3];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const bool rin = r == 0 ? top_in : (r == 5 ? bot_in : true);
        v4f mid = w.mid[r];
        float e = w.e[r];
        if (ACT) {
#pragma unroll
          for (int u = 0; u < 4; ++u) mid[u] = __builtin_fmaxf(fma_rn(mid[u] + apb, asc, ash), 0.0f);
          e = __builtin_fmaxf(fma_rn(e + apb, asc, ash), 0.0f);
        }
        const float sl = from_left(mid[3]);
        const float sr = from_right(mid[0]);
        const float lft = left_lane ? sl : (edge_l ? e : 0.0f);
        const float rgt = right_lane ? sr : (edge_r ? e : 0.0f);
        dp[r][0] = (v2f){rin ? lft : 0.0f, rin ? mid[0] : 0.0f};
        dp[r][1] = (v2f){rin ? mid[1] : 0.0f, rin ? mid[2] : 0.0f};
        dp[r][2] = (v2f){rin ? mid[3] : 0.0f, rin ? rgt : 0.0f};
      }
      // transform along rows (down the columns), two columns per instruction: tp[i][j] = (t[i][2j], t[i][2j+1])
      v2f tp[6][3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        v2f colv[6], o[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) colv[r] = dp[r][j];
        in4x2(colv, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tp[i][j] = o[i];
      }
      // transform along columns, two rows per instruction, and the 12 products of those rows
#pragma unroll
      for (int ip = 0; ip < 3; ++ip) {
        v2f rowv[6], o[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          rowv[2 * j] = (v2f){tp[2 * ip][j][0], tp[2 * ip + 1][j][0]};
          rowv[2 * j + 1] = (v2f){tp[2 * ip][j][1], tp[2 * ip + 1][j][1]};
        }
        in4x2(rowv, o);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const int xi = 6 * (2 * ip + h) + c;
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[xi >> 2][xi & 3], o[c][h], acc[xi], 0, 0, 0);
          }
        }
      }
    };
    // sched_barrier: keep each step's loads where they are written (the scheduler otherwise hoists
    // all of them to the top and spills)
#define FPSG_LOAD(st, buf) load_raw((st) < kSteps ? (st) : kSteps - 1, buf); __builtin_amdgcn_sched_barrier(0)
#define FPSG_COMPUTE(st, buf) compute(st, buf); __builtin_amdgcn_sched_barrier(0)
    Raw ra, rb, rc;
    FPSG_LOAD(0, ra);
    FPSG_LOAD(1, rb);
#pragma unroll 1
    for (int c4 = 0; c4 < kSteps - 1; c4 += 3) {          // steps 0 .. 14, loads two steps ahead
      FPSG_LOAD(c4 + 2, rc);
      FPSG_COMPUTE(c4, ra);
      FPSG_LOAD(c4 + 3, ra);                              // <= 15
      FPSG_COMPUTE(c4 + 1, rb);
      FPSG_LOAD(c4 + 4, rb);                              // clamped to the last step
      FPSG_COMPUTE(c4 + 2, rc);
    }
    FPSG_COMPUTE(kSteps - 1, ra);
#undef FPSG_COMPUTE
#undef FPSG_LOAD
    // output transform: accumulator element r of acc[xi] = M[xi][k0 + 4*kk + r][tile col]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[6][4];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float m[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) m[a] = acc[6 * a + c][r];
        out4(m, s[c]);
      }
      if (live) {
        float* yp = y + ((((size_t)n * K + k0 + 4 * kk + r) * H) + 4 * th) * W + 4 * tw;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float rowv[6], o[4];
#pragma unroll
          for (int c = 0; c < 6; ++c) rowv[c] = s[c][a];
          out4(rowv, o);
          *reinterpret_cast<v4f*>(yp + (size_t)a * W) = (v4f){o[0], o[1], o[2], o[3]};
        }
      }
    }
  }
}

}  // namespace
}  // namespace fpsg

static int wino_conv_fused_launch(const char* fn, const float* x, const float* chan, const float* pre_bias,
                                  const float* U, int N, int C, int K, int H, int W, float* y, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(C == 64, FPSG_E_SHAPE, "%s: C must be 64 (got %d)", fn, C);
  FPSG_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && K <= 1024 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, FPSG_E_SHAPE,
               "%s: K a positive multiple of 16 (<= 1024), H and W positive multiples of 4 "
               "(got K=%d H=%d W=%d)", fn, K, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(U); FPSG_REQUIRE_PTR(y);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, FPSG_E_ALIGN,
               "%s: x and y must be 16-byte aligned", fn);
  const long P = (long)N * (H / 4) * (W / 4);
  const int S = K / 16;
  const size_t lds_bytes = (size_t)(kUldsFloats + 3 * 64) * sizeof(float);
  const void* kern = chan ? reinterpret_cast<const void*>(wino4_fused_c64_kernel<true>)
                          : reinterpret_cast<const void*>(wino4_fused_c64_kernel<false>);
  const hipError_t lds_optin = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_optin != hipSuccess) {
    set_error("%s: cannot reserve %zu B of LDS: %s", fn, lds_bytes, hipGetErrorString(lds_optin));
    return static_cast<int>(lds_optin);
  }
  // one workgroup per CU (LDS), a multiple of 8*S workgroups; never more than the tile groups need
  const long G = (P + 15) / 16;
  long per_slice = (256 + S - 1) / S;
  const long need = (G + 3) / 4;
  if (per_slice > need) per_slice = need;
  per_slice = ((per_slice + 7) / 8) * 8;
  dim3 grid((unsigned)(per_slice * S));
  if (chan)
    hipLaunchKernelGGL(wino4_fused_c64_kernel<true>, grid, dim3(kFusedThreads), lds_bytes,
                       static_cast<hipStream_t>(stream), x, U, K, H, W, H / 4, W / 4, P, S, y, chan, pre_bias);
  else
    hipLaunchKernelGGL(wino4_fused_c64_kernel<false>, grid, dim3(kFusedThreads), lds_bytes,
                       static_cast<hipStream_t>(stream), x, U, K, H, W, H / 4, W / 4, P, S, y, nullptr, nullptr);
  return launch_status(fn);
}

extern "C" int fpsg_wino_conv_fused(const float* x, const float* U, int N, int C, int K, int H, int W, float* y,
                                    fpsg_stream_t stream) {
  return wino_conv_fused_launch("fpsg_wino_conv_fused", x, nullptr, nullptr, U, N, C, K, H, W, y, stream);
}

extern "C" int fpsg_wino_conv_fused_act(const float* x, const float* chan, const float* pre_bias, const float* U,
                                        int N, int C, int K, int H, int W, float* y, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(!misaligned4(pre_bias), FPSG_E_ALIGN, "fpsg_wino_conv_fused_act: pre_bias not 4-byte aligned");
  return wino_conv_fused_launch("fpsg_wino_conv_fused_act", x, chan, pre_bias, U, N, C, K, H, W, y, stream);
}
