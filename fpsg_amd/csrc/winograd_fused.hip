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
//     one per transform point, accumulating M[xi][16 k][16 tiles] in 144 accumulator registers (AGPRs);
//   * after the 16 channel steps a lane holds M[0..35] for its tile and 4 output channels:
//     the output transform runs in registers and the 4x4 pixels are stored as aligned vectors.
// Workgroup -> (slice, tile range) mapping keeps the workgroups of one tile range on one XCD
// (round-robin dispatch: id % 8), so that the re-reads of x by the other slices hit that L2.
// fp32 MFMAs and fp32 VALU instructions do not overlap on gfx950 (tools/micro/mfma_f32_*.hip): a step costs its 36
// MFMAs PLUS every vector instruction, so the code below is written for few instructions -- scalar transform on the
// columns as loaded, 12-operation B^T d, out-of-image rows / columns through transform weights instead of selects,
// the products as in-place asm blocks on AGPR tuples, buffer loads without address arithmetic.
// Same arithmetic per element as winograd.hip's transforms (masked rows / columns enter as x 0 instead of a selected
// 0: the sign of a zero may differ); the channel sum runs in the MFMA's k order.  Deterministic.
#include <type_traits>

#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kFusedThreads = 256;
constexpr int kSteps = 16;          // 64 input channels / 4 per MFMA step
constexpr int kUldsFloats = 36 * kSteps * 64;

// lane i <- lane i-1 / i+1 inside its row of 16 lanes (DPP row_shr:1 / row_shl:1, one VALU op); the lanes
// without a neighbour in their row (column 0 / 15) keep `keep`
__device__ __forceinline__ float from_left_or(float keep, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right_or(float keep, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x101, 0xF, 0xF, false));
}
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
__device__ __forceinline__ void out4(const float (&m)[6], float (&s)[4]) {
  const float p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
  s[0] = (m[0] + p12) + p34;
  s[1] = fma_rn(2.0f, m34, m12);
  s[2] = fma_rn(4.0f, p34, p12);
  s[3] = fma_rn(8.0f, m34, m12) + m[5];
}

// ACT: x is the PRE-BatchNorm output of the previous convolution; the patch values are
// relu(fma(x + pre_bias[c], scale[c], shift[c])) (K5's apply arithmetic), padding stays zero.
// STATS: per output channel the partial sums sum(y + out_bias[k]), sum((y + out_bias[k])^2) over the workgroup's
// tiles -> parts[k][index of the workgroup in its slice][2] for the BatchNorm that follows (fpsg_bn_stats with
// parts): a lane keeps the sums of its 4 channels over its tile groups, then its row of 16 lanes (fixed DPP tree),
// then the four waves in order.  Deterministic.
template <bool ACT, bool STATS, bool NT = false>
__global__ __launch_bounds__(kFusedThreads) void wino4_fused_c64_kernel(const float* __restrict__ x,
                                                                        const float* __restrict__ U /*[36][K][64]*/,
                                                                        int K, int H, int W, int Th, int Tw, long P,
                                                                        int S, float* __restrict__ y,
                                                                        const float* __restrict__ chan,
                                                                        const float* __restrict__ pre_bias,
                                                                        const float* __restrict__ out_bias,
                                                                        float* __restrict__ parts) {
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
  // A thread owns (output channel kl, channel group c4) for all 36 transform points: 36 vector loads issued
  // together (a workgroup reads 4 KB contiguous per point; one round trip instead of one per element), then
  // four points at a time go to LDS as 16-byte writes (the points are the fastest LDS index).
  {
    const int kl = tid >> 4, c4 = tid & 15;
    const float* up = U + ((size_t)k0 + kl) * C + 4 * c4;
    v4f ub[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) ub[xi] = *reinterpret_cast<const v4f*>(up + (size_t)xi * K * C);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      v4f* dst = reinterpret_cast<v4f*>(ulds + ((size_t)c4 * 64 + kq * 16 + kl) * 36);
#pragma unroll
      for (int x4 = 0; x4 < 9; ++x4)
        dst[x4] = (v4f){ub[4 * x4][kq], ub[4 * x4 + 1][kq], ub[4 * x4 + 2][kq], ub[4 * x4 + 3][kq]};
    }
  }
  if (ACT && tid < C) {
    actp[tid] = chan[tid];
    actp[C + tid] = chan[C + tid];
    actp[2 * C + tid] = pre_bias ? pre_bias[tid] : 0.0f;
  }
  __syncthreads();
  // x as a buffer resource: loads are `descriptor + 32-bit lane offset + scalar step offset` (no address VALU)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)(P * 16 * C * sizeof(float)), 0x00020000);
  float st0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, st1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ob[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (STATS && out_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ob[r] = out_bias[k0 + 4 * kk + r];
  }
  const long G = (P + 15) >> 4;
  // What a lane needs to know about its tile in a group of 16: where it is, and how it enters the transform.
  // Rows r0+1 .. r0+4 are the tile's own output rows: always inside the image; only the halo rows r0 (top) and
  // r0+5 (bottom) can fall outside.  They are then read from a valid row and enter the transform with weight 0:
  // B^T uses row 0 only as 4*d[0] (the 4 becomes 0) and row 5 only as +d[5] (multiplied by 0); the halo columns
  // likewise (x 0 when the tile touches the image's left / right edge).
  struct Tile {
    uint32_t boff[6], eoffb[6];     // byte offsets of the six patch rows inside channel kk of step 0 / of the edge element
    float c4t, mb, c4l, mr, mr5;
    int tw, th;
    unsigned n;
    bool live;
  };
  auto locate = [&](long g, Tile& t) {
    const long p_raw = g * 16 + col;
    t.live = p_raw < P;
    const long p = t.live ? p_raw : P - 1;
    const unsigned pu = (unsigned)p;                                    // P < 2^20 (the 4 GiB check of the host)
    t.tw = (int)(pu % (unsigned)Tw);
    const unsigned q = pu / (unsigned)Tw;
    t.th = (int)(q % (unsigned)Th);
    t.n = q / (unsigned)Th;
    const bool has_left = t.tw > 0, has_right = t.tw < Tw - 1;
    const bool edge_l = has_left && col == 0, edge_r = has_right && col == 15;
    const int r0 = 4 * t.th - 1;
    const bool top_in = r0 >= 0, bot_in = r0 + 5 < H;
    t.c4t = top_in ? 4.0f : 0.0f;
    t.mb = bot_in ? 1.0f : 0.0f;
    t.c4l = has_left ? 4.0f : 0.0f;
    t.mr = has_right ? 1.0f : 0.0f;
    t.mr5 = t.mr * t.mb;
    // the tensor is below 4 GiB (checked by the host): a step's loads are `uniform base + 32-bit lane offset`
    const uint32_t lane_elem = (uint32_t)((((size_t)t.n * C + kk) * H) * W + 4 * t.tw);
    // a lane is at most at one edge of its row of 16: column 0 needs the element left of its vector, column
    // 15 the one right of it; the others re-read their own first element (never used)
    const uint32_t eadd = edge_l ? (uint32_t)-4 : (edge_r ? 16u : 0u);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int row = r0 + r;
      t.boff[r] = (lane_elem + (uint32_t)((row < 0 ? 0 : (row >= H ? H - 1 : row)) * W)) * 4u;
      t.eoffb[r] = t.boff[r] + eadd;
    }
  };
  const uint32_t step_bytes = (uint32_t)(4 * H * W) * 4u;              // 4 channels per step
  struct Raw { v4f mid[6]; float e[6]; };
  auto load_raw = [&](const Tile& t, int c4, Raw& w) {
    const uint32_t sbase = (uint32_t)c4 * step_bytes;                  // wave-uniform: the buffer load's soffset
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      w.mid[r] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, t.boff[r], sbase, 0));
      w.e[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, t.eoffb[r], sbase, 0));
    }
  };
  const long g_stride = 4L * per_slice;
  Tile cur, nxt;
  Raw ra, rb;
  long g = (long)j * 4 + wave;
  if (g < G) {
    locate(g, cur);
    load_raw(cur, 0, ra);
    load_raw(cur, 1, rb);
  }
  for (; g < G; g += g_stride) {
    // the next group's first two patches are loaded under this group's last steps and its output transform
    // (a group otherwise starts with both buffers cold: ~7 % of the kernel parked at s_waitcnt)
    locate(g + g_stride < G ? g + g_stride : g, nxt);
    const float c4t = cur.c4t, mb = cur.mb, c4l = cur.c4l, mr = cur.mr, mr5 = cur.mr5;
    const bool live = cur.live;
    const int tw = cur.tw, th = cur.th;
    const unsigned n = cur.n;
    v4f acc[36];                                   // written by step 0 (C = 0), accumulated in place afterwards

    // The patch of one channel step as loaded: interior vectors + the halo element of an edge lane.  One wave per
    // SIMD (the U slice fills the LDS), so the load latency is covered by issuing the loads one to two steps ahead of
    // their use; no branches around the loads (the compiler then counts outstanding loads exactly: s_waitcnt
    // vmcnt(n) per buffer instead of vmcnt(0)).
    // (accr: the accumulators as a parameter -- asm operands in a generic lambda cannot name captured arrays)
    auto compute = [&](int c4, const Raw& w, v4f (&accr)[36], auto first) {
      // this step's 36 A fragments: issued first (pinned by the sched_barrier), in flight under the input transform
      v4f a[9];
      {
        const v4f* up = reinterpret_cast<const v4f*>(ulds + ((size_t)c4 * 64 + lane) * 36);
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) a[q4] = up[q4];
      }
      __builtin_amdgcn_sched_barrier(0);
      float asc = 1.0f, ash = 0.0f, apb = 0.0f;
      if (ACT) { asc = actp[4 * c4 + kk]; ash = actp[C + 4 * c4 + kk]; apb = actp[2 * C + 4 * c4 + kk]; }
      // Scalar fp32 only (the file is compiled with -fno-slp-vectorize): fp32 MFMAs and fp32 VALU operations do
      // not overlap on this hardware (tools/micro/mfma_f32_*.hip), so the step costs MFMA time + VALU issue
      // time, and a packed v_pk_*_f32 beside MFMAs costs more than the two scalar operations it replaces.
      float d[6][6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float m0 = w.mid[r][0], m1 = w.mid[r][1], m2 = w.mid[r][2], m3 = w.mid[r][3], e = w.e[r];
        if (ACT) {
          m0 = __builtin_fmaxf(fma_rn(m0 + apb, asc, ash), 0.0f);
          m1 = __builtin_fmaxf(fma_rn(m1 + apb, asc, ash), 0.0f);
          m2 = __builtin_fmaxf(fma_rn(m2 + apb, asc, ash), 0.0f);
          m3 = __builtin_fmaxf(fma_rn(m3 + apb, asc, ash), 0.0f);
          e = __builtin_fmaxf(fma_rn(e + apb, asc, ash), 0.0f);
        }
        const float lft = from_left_or(e, m3);                     // column 0 of its row of 16 keeps e
        d[r][0] = r == 5 ? lft * mb : lft;                         // (the left edge enters through c4l below)
        d[r][5] = from_right_or(e, m0) * (r == 5 ? mr5 : mr);      // column 15 keeps e
        d[r][1] = r == 5 ? m0 * mb : m0;
        d[r][2] = r == 5 ? m1 * mb : m1;
        d[r][3] = r == 5 ? m2 * mb : m2;
        d[r][4] = r == 5 ? m3 * mb : m3;
      }
      // transform along rows (down the columns)
      float t[6][6];        // t[j][i]: column j after the transform along rows
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float colv[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        in4(colv, c4t, t[j]);
      }
      // transform along columns, one row of 6 transform points at a time, and that row's 6 products
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float rowv[6] = {t[0][i], t[1][i], t[2][i], t[3][i], t[4][i], t[5][i]};
        float o[6];
        in4(rowv, c4l, o);
        // The row's six products as one block, in place on AGPR tuples.  (Through the builtin the register
        // allocator moves the 36 accumulators between AGPR tuples every step, ~60 v_accvgpr_* beside 36 MFMAs.)
        // The compiler does not see MFMAs inside asm: s_nop 1 covers the two wait states gfx950 needs between
        // a VALU write of o[] and the MFMA reading it; the accumulator reads after the last step are covered below.
        {
          const int x0 = 6 * i, l0 = x0;
#define FPSG_A(c) a[(l0 + (c)) >> 2][(l0 + (c)) & 3]
          if constexpr (decltype(first)::value) {
            asm volatile(
                "s_nop 1\n\t"
                "v_mfma_f32_16x16x4_f32 %0, %6, %12, 0\n\t"
                "v_mfma_f32_16x16x4_f32 %1, %7, %13, 0\n\t"
                "v_mfma_f32_16x16x4_f32 %2, %8, %14, 0\n\t"
                "v_mfma_f32_16x16x4_f32 %3, %9, %15, 0\n\t"
                "v_mfma_f32_16x16x4_f32 %4, %10, %16, 0\n\t"
                "v_mfma_f32_16x16x4_f32 %5, %11, %17, 0"
                : "=&a"(accr[x0]), "=&a"(accr[x0 + 1]), "=&a"(accr[x0 + 2]), "=&a"(accr[x0 + 3]), "=&a"(accr[x0 + 4]),
                  "=&a"(accr[x0 + 5])
                : "v"(FPSG_A(0)), "v"(FPSG_A(1)), "v"(FPSG_A(2)), "v"(FPSG_A(3)), "v"(FPSG_A(4)), "v"(FPSG_A(5)),
                  "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]));
          } else {
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
                : "v"(FPSG_A(0)), "v"(FPSG_A(1)), "v"(FPSG_A(2)), "v"(FPSG_A(3)), "v"(FPSG_A(4)), "v"(FPSG_A(5)),
                  "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]));
          }
#undef FPSG_A
        }
      }
    };
    // sched_barrier: keep each step's loads where they are written (the scheduler otherwise hoists
    // all of them to the top and spills)
#define FPSG_LOAD(st, buf) load_raw(cur, st, buf); __builtin_amdgcn_sched_barrier(0)
#define FPSG_LOAD_NEXT(st, buf) load_raw(nxt, st, buf); __builtin_amdgcn_sched_barrier(0)
#define FPSG_COMPUTE(st, buf) compute(st, buf, acc, std::false_type{}); __builtin_amdgcn_sched_barrier(0)
#define FPSG_COMPUTE_FIRST(st, buf) compute(st, buf, acc, std::true_type{}); __builtin_amdgcn_sched_barrier(0)
    __builtin_amdgcn_sched_barrier(0);
    FPSG_COMPUTE_FIRST(0, ra);                            // step 0 writes the accumulators (C = 0)
    FPSG_LOAD(2, ra);
    FPSG_COMPUTE(1, rb);
    FPSG_LOAD(3, rb);
#pragma unroll 1
    for (int c4 = 2; c4 < kSteps - 2; c4 += 2) {          // steps 2 .. 13; a step's loads are issued one to two steps ahead
      FPSG_COMPUTE(c4, ra);
      FPSG_LOAD(c4 + 2, ra);                              // <= 15
      FPSG_COMPUTE(c4 + 1, rb);
      FPSG_LOAD(c4 + 3, rb);
    }
    FPSG_COMPUTE(kSteps - 2, ra);
    FPSG_LOAD_NEXT(0, ra);
    FPSG_COMPUTE(kSteps - 1, rb);
    FPSG_LOAD_NEXT(1, rb);
#undef FPSG_COMPUTE
#undef FPSG_COMPUTE_FIRST
#undef FPSG_LOAD_NEXT
#undef FPSG_LOAD
    // the compiler does not see MFMAs in the asm statements: cover the MFMA-write -> VALU-read distance by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
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
          st_stream<NT>(reinterpret_cast<v4f*>(yp + (size_t)a * W), (v4f){o[0], o[1], o[2], o[3]});
          if (STATS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float v = o[u] + ob[r];
              st0[r] += v;
              st1[r] = fma_rn(v, v, st1[r]);
            }
          }
        }
      }
    }
    cur = nxt;
  }
  if (STATS) {
    __syncthreads();                                      // every wave is done with the U slice: reuse its LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a0 = st0[r], a1 = st1[r];
      a0 += __uint_as_float(lane_xor<1>(__float_as_uint(a0))); a1 += __uint_as_float(lane_xor<1>(__float_as_uint(a1)));
      a0 += __uint_as_float(lane_xor<2>(__float_as_uint(a0))); a1 += __uint_as_float(lane_xor<2>(__float_as_uint(a1)));
      a0 += __uint_as_float(lane_xor<4>(__float_as_uint(a0))); a1 += __uint_as_float(lane_xor<4>(__float_as_uint(a1)));
      a0 += __uint_as_float(lane_xor<8>(__float_as_uint(a0))); a1 += __uint_as_float(lane_xor<8>(__float_as_uint(a1)));
      if (col == 0) {
        ulds[((wave * 4 + kk) * 4 + r) * 2] = a0;
        ulds[((wave * 4 + kk) * 4 + r) * 2 + 1] = a1;
      }
    }
    __syncthreads();
    if (tid < 16) {                                       // tid = 4 * kk + r: channel k0 + tid
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t0 += ulds[(w * 16 + tid) * 2]; t1 += ulds[(w * 16 + tid) * 2 + 1]; }
      float* out = parts + ((size_t)(k0 + tid) * per_slice + j) * 2;
      out[0] = t0;
      out[1] = t1;
    }
  }
}

}  // namespace
}  // namespace fpsg

static long wino_fused_per_slice(long P, int S) {
  // one workgroup per CU (LDS), a multiple of 8*S workgroups; never more than the tile groups need
  const long G = (P + 15) / 16;
  long per_slice = (256 + S - 1) / S;
  const long need = (G + 3) / 4;
  if (per_slice > need) per_slice = need;
  return ((per_slice + 7) / 8) * 8;
}

static int wino_conv_fused_launch(const char* fn, const float* x, const float* chan, const float* pre_bias,
                                  const float* U, int N, int C, int K, int H, int W, float* y, const float* out_bias,
                                  float* parts, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(C == 64, FPSG_E_SHAPE, "%s: C must be 64 (got %d)", fn, C);
  FPSG_REQUIRE(N > 0 && K > 0 && K % 16 == 0 && K <= 1024 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, FPSG_E_SHAPE,
               "%s: K a positive multiple of 16 (<= 1024), H and W positive multiples of 4 "
               "(got K=%d H=%d W=%d)", fn, K, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(U); FPSG_REQUIRE_PTR(y);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(U)) & 15) == 0,
               FPSG_E_ALIGN, "%s: x, y and U must be 16-byte aligned", fn);
  FPSG_REQUIRE(!misaligned4(out_bias) && !misaligned4(parts), FPSG_E_ALIGN, "%s: out_bias / parts not 4-byte aligned", fn);
  FPSG_REQUIRE((size_t)N * C * H * W * sizeof(float) < ((size_t)1 << 32), FPSG_E_LIMIT,
               "%s: x must be below 4 GiB (32-bit lane offsets; got N=%d H=%d W=%d)", fn, N, H, W);
  const long P = (long)N * (H / 4) * (W / 4);
  const int S = K / 16;
  const size_t lds_bytes = (size_t)(kUldsFloats + 3 * 64) * sizeof(float);
  typedef void (*kern_t)(const float*, const float*, int, int, int, int, int, long, int, float*, const float*,
                         const float*, const float*, float*);
  // y beyond the Infinity Cache (64 -> 64 @224 at 37 images: 475 MB) is written with non-temporal stores
  const bool nt = beyond_cache((size_t)N * K * H * W * sizeof(float));
  const kern_t kern =
      nt ? (chan ? (parts ? wino4_fused_c64_kernel<true, true, true> : wino4_fused_c64_kernel<true, false, true>)
                 : (parts ? wino4_fused_c64_kernel<false, true, true> : wino4_fused_c64_kernel<false, false, true>))
         : (chan ? (parts ? wino4_fused_c64_kernel<true, true> : wino4_fused_c64_kernel<true, false>)
                 : (parts ? wino4_fused_c64_kernel<false, true> : wino4_fused_c64_kernel<false, false>));
  const hipError_t lds_optin = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_optin != hipSuccess) {
    set_error("%s: cannot reserve %zu B of LDS: %s", fn, lds_bytes, hipGetErrorString(lds_optin));
    return static_cast<int>(lds_optin);
  }
  dim3 grid((unsigned)(wino_fused_per_slice(P, S) * S));
  hipLaunchKernelGGL(kern, grid, dim3(kFusedThreads), lds_bytes, static_cast<hipStream_t>(stream), x, U, K, H, W, H / 4,
                     W / 4, P, S, y, chan, pre_bias, out_bias, parts);
  return launch_status(fn);
}

extern "C" int fpsg_wino_conv_fused(const float* x, const float* U, int N, int C, int K, int H, int W, float* y,
                                    fpsg_stream_t stream) {
  return wino_conv_fused_launch("fpsg_wino_conv_fused", x, nullptr, nullptr, U, N, C, K, H, W, y, nullptr, nullptr, stream);
}

extern "C" int fpsg_wino_conv_fused_act(const float* x, const float* chan, const float* pre_bias, const float* U,
                                        int N, int C, int K, int H, int W, float* y, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(!misaligned4(pre_bias), FPSG_E_ALIGN, "fpsg_wino_conv_fused_act: pre_bias not 4-byte aligned");
  return wino_conv_fused_launch("fpsg_wino_conv_fused_act", x, chan, pre_bias, U, N, C, K, H, W, y, nullptr, nullptr, stream);
}

extern "C" int fpsg_wino_conv_fused_parts(int N, int K, int H, int W) {
  if (N <= 0 || K <= 0 || K % 16 || H <= 0 || W <= 0 || H % 4 || W % 4) return 0;
  return (int)wino_fused_per_slice((long)N * (H / 4) * (W / 4), K / 16);
}

extern "C" int fpsg_wino_conv_fused_stats(const float* x, const float* chan, const float* pre_bias, const float* U,
                                          int N, int C, int K, int H, int W, float* y, const float* out_bias,
                                          float* parts, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE_PTR(parts);
  FPSG_REQUIRE(!misaligned4(pre_bias), FPSG_E_ALIGN, "fpsg_wino_conv_fused_stats: pre_bias not 4-byte aligned");
  return wino_conv_fused_launch("fpsg_wino_conv_fused_stats", x, chan, pre_bias, U, N, C, K, H, W, y, out_bias, parts, stream);
}
