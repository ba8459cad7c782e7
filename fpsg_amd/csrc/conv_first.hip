// conv_first.hip -- K8: weight gradient of the first VGG convolution (3 -> 64 channels, 3x3,
// padding 1; torchvision vgg16_bn.features[0], src/models/image_net.py:14), for gfx950.
//
// dw[k][c][a][b] = sum over (n, h, w) of dy[n,k,h,w] * x[n,c,h+a-1,w+b-1] is a [64 x P] . [P x 27]
// product with a 1.86 M-long reduction (37 images of 224x224) and a 1,728-element result: 6.4
// GFLOP over 475 MB of dy -- HBM-bound by two orders of magnitude.  The library runs it as an
// NHWC implicit GEMM behind two layout transposes of dy (0.73 ms); here dy is read exactly once:
//   * a workgroup walks over row segments of PX pixels (112 for VGG's 224-pixel rows, else 64); dy[64 ch][PX px]
//     is staged coalesced into LDS (448-B / 256-B runs per channel) together with the 3 x 3 x (PX+2) input patch of
//     the segment; the next segment's vectors are already on their way into registers while this one is multiplied;
//   * wave w owns output channels 16w .. 16w+15: PX/4 steps of v_mfma_f32_16x16x4_f32 per segment
//     with A = dy (channel x 4 pixels) and B = the im2col patch (4 pixels x 16 taps; taps 27..31
//     are zero), two tap tiles -> 8 accumulator registers carried over all segments;
//   * each workgroup writes one [64][32] partial; a second kernel sums the partials in a fixed
//     order in fp64.  Deterministic.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kFirstThreads = 256;

// PX = pixels per segment: 112 when the rows are whole multiples of it (VGG's 224: two 448-byte runs per channel row,
// no idle pixels), 64 otherwise.  The NEXT segment's dy vectors and patch elements are fetched into registers before
// the current segment's MFMAs start, so the HBM latency runs beside the matrix work instead of in front of it.
// FOLD (round 4): dy is not given -- it is the BatchNorm + ReLU backward of the layer behind this convolution,
//   dy = k1 dz + k2 (y + b) + k3,  dz = ga * [ (y + b) scale + shift > 0 ],
// formed while the tile is written to LDS from the convolution's own output y, the gradient ga of the (never stored)
// activation and seven per-channel constants (K5's dx pass, bn_apply_kernel<1>, same arithmetic bit for bit): the 475 MB
// of dy are neither written by K5 nor read here (fpsg_conv_first_dw_fold).
struct FoldArgs {
  const float* y;        // [N,64,H,W] the convolution's output (pre-BatchNorm)
  const float* chan;     // [4][64] scale, shift, mean, rstd
  const float* coef;     // [3][64] k1, k2, k3
  const float* pre_bias; // [64] or null
};

template <int PX, bool FOLD>
__global__ __launch_bounds__(kFirstThreads) void conv_first_dw_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ dy, FoldArgs fa, int H, int W,
                                                                      int segs_per_row, int n_segs,
                                                                      float* __restrict__ part /*[grid][64][32]*/) {
  constexpr int LD = PX + 4;                      // LDS row stride: 2 lanes per bank for the A reads, the minimum
  constexpr int VPR = PX / 4;                     // 16-byte vectors per channel row of the tile
  constexpr int NV = 64 * VPR / kFirstThreads;    // vectors per thread (4 or 7)
  constexpr int PE = 9 * (PX + 2);                // patch elements: 3 channels x 3 rows x (PX + 2) columns
  constexpr int NP = (PE + kFirstThreads - 1) / kFirstThreads;
  static_assert(64 * VPR % kFirstThreads == 0, "tile vectors divide over the threads");
  __shared__ float dyt[64 * LD];
  __shared__ float xt[9 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, col = lane & 15;
  // B operand: tap index 16*nt + col -> (c, a, b); its patch element for pixel p is xt[(3c+a)*LD + p + b]
  int boff[2];
  bool bval[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tap = 16 * nt + col;
    bval[nt] = tap < 27;
    const int t = bval[nt] ? tap : 0;
    boff[nt] = (t / 3) * LD + (t % 3);            // t/3 = 3c + a, t%3 = b
  }
  // this thread's slots of a tile: vector v = it*256 + tid -> (channel v / VPR, pixels 4*(v % VPR) ..), patch
  // element e = it*256 + tid -> (row cr = e / (PX+2), column e % (PX+2))
  int vch[NV], vq4[NV], pcr[NP], pj[NP];
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int v = it * kFirstThreads + tid;
    vch[it] = v / VPR;
    vq4[it] = (v - vch[it] * VPR) * 4;
  }
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int e = it * kFirstThreads + tid;
    pcr[it] = e < PE ? e / (PX + 2) : -1;
    pj[it] = e - (e / (PX + 2)) * (PX + 2);
  }
  v4f rv[NV];
  v4f ry[FOLD ? NV : 1];
  float rp[NP];
  // FOLD: the constants of this thread's channels (a slot's channel does not change from segment to segment)
  float fsc[FOLD ? NV : 1], fsh[FOLD ? NV : 1], fk1[FOLD ? NV : 1], fk2[FOLD ? NV : 1], fk3[FOLD ? NV : 1], fb[FOLD ? NV : 1];
  if constexpr (FOLD) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = vch[it];
      fsc[it] = fa.chan[c]; fsh[it] = fa.chan[64 + c];
      fk1[it] = fa.coef[c]; fk2[it] = fa.coef[64 + c]; fk3[it] = fa.coef[128 + c];
      fb[it] = fa.pre_bias ? fa.pre_bias[c] : 0.0f;
    }
  }
  auto fetch = [&](int seg) {
    const int row = seg / segs_per_row;             // (n, h)
    const int w0 = (seg - row * segs_per_row) * PX;
    const int n = row / H;
    const int h = row - n * H;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      rv[it] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};       // W % 4 == 0: a vector is inside or outside as a whole
      if constexpr (FOLD) ry[it] = rv[it];
      if (w0 + vq4[it] < W) {
        const size_t o = (((size_t)n * 64 + vch[it]) * H + h) * W + w0 + vq4[it];
        rv[it] = *reinterpret_cast<const v4f*>(dy + o);
        if constexpr (FOLD) ry[it] = *reinterpret_cast<const v4f*>(fa.y + o);
      } else if constexpr (FOLD) {
        ry[it] = (v4f){-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};     // outside the row: flagged below, enters as 0
      }
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int c = pcr[it] / 3, r = pcr[it] - 3 * c;
      const int hh = h + r - 1, ww = w0 + pj[it] - 1;
      rp[it] = 0.0f;
      if (pcr[it] >= 0 && hh >= 0 && hh < H && ww >= 0 && ww < W) rp[it] = x[(((size_t)n * 3 + c) * H + hh) * W + ww];
    }
  };
  v4f acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
  int seg = blockIdx.x;
  if (seg < n_segs) fetch(seg);
  for (; seg < n_segs; seg += gridDim.x) {
    __syncthreads();                                // the previous segment's tiles are consumed
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      v4f v = rv[it];
      if constexpr (FOLD) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv = ry[it][u] + fb[it];
          const float dz = rv[it][u] * (fma_rn(xv, fsc[it], fsh[it]) > 0.0f ? 1.0f : 0.0f);
          const float dx = fma_rn(fk1[it], dz, fma_rn(fk2[it], xv, fk3[it]));
          v[u] = ry[it][u] == -3.0e38f ? 0.0f : dx;                 // pixels beyond the row contribute nothing
        }
      }
      *reinterpret_cast<v4f*>(dyt + vch[it] * LD + vq4[it]) = v;
    }
#pragma unroll
    for (int it = 0; it < NP; ++it)
      if (pcr[it] >= 0) xt[pcr[it] * LD + pj[it]] = rp[it];
    __syncthreads();
    if (seg + (int)gridDim.x < n_segs) fetch(seg + gridDim.x);
    const float* ap = dyt + (16 * wave + col) * LD + kk;
    const float* bp0 = xt + boff[0] + kk;
    const float* bp1 = xt + boff[1] + kk;
#pragma unroll
    for (int s = 0; s < PX / 4; ++s) {
      const float a = ap[4 * s];
      const float b0 = bval[0] ? bp0[4 * s] : 0.0f;
      const float b1 = bval[1] ? bp1[4 * s] : 0.0f;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[1], 0, 0, 0);
    }
  }
  // D layout: element r of acc[nt] = out[channel 16*wave + 4*kk + r][tap 16*nt + col]
  float* pp = part + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) pp[(16 * wave + 4 * kk + r) * 32 + 16 * nt + col] = acc[nt][r];
}

// dw[k][tap] (tap = 9c + 3a + b < 27) = sum of the workgroups' partials in a fixed order, fp64.
// One workgroup per output channel k: thread (g, tap) sums the partials p = g, g+32, ... (each
// (p, k) row of 32 taps is one 128-byte read), the 32 groups are combined through LDS in order.
__global__ __launch_bounds__(1024) void conv_first_dw_reduce_kernel(const float* __restrict__ part, int n_part,
                                                                    float* __restrict__ dw /*[64][27]*/) {
  __shared__ double acc[32][32];
  const int k = blockIdx.x, tap = threadIdx.x & 31, g = threadIdx.x >> 5;
  double a = 0.0;
#pragma unroll 8
  for (int p = g; p < n_part; p += 32) a += (double)part[((size_t)p * 64 + k) * 32 + tap];
  acc[g][tap] = a;
  __syncthreads();
  if (g == 0 && tap < 27) {
    double t = acc[0][tap];
#pragma unroll
    for (int q = 1; q < 32; ++q) t += acc[q][tap];
    dw[k * 27 + tap] = (float)t;
  }
}


// ---- K8f: the FORWARD of the same layer ----------------------------------------------------------------------
// y[n,k,h,w] = sum over (c, a, b) of w[k][c][a][b] * x[n,c,h+a-1,w+b-1], zero padding: 27 multiply-adds per output and
// 64 outputs per input pixel -- bound by the write of y (475 MB at 37 images).  A thread owns 4 adjacent pixels: its
// 3 x 3 x 6 input patch sits in registers (rows as aligned 16-byte loads, the two halo columns from the neighbouring
// lanes; every load is issued before the first use), the 64 x 27 weights are wave-uniform (scalar loads), and each
// output channel is one 16-byte store per thread, a wave writing 1 KB contiguous.  Per output the taps are added in
// (c, a, b) order by an fma chain starting from 0 (the oracle of tests/test_winograd_gpu.py restates exactly this).
// STATS: per channel the partial sums of y + bias[k] and its square over the workgroup's 1024 pixels ->
// parts[k][blockIdx.x][2] for the BatchNorm that follows (fpsg_bn_stats with parts).  Deterministic.
// the 27 weights (and the bias) of output channel k: k is wave-uniform, so these are scalar loads
__device__ __forceinline__ void first_load_weights(const float* __restrict__ wt, const float* __restrict__ bias, int k,
                                                   float (&w)[27], float& b) {
#pragma unroll
  for (int i = 0; i < 27; ++i) w[i] = wt[k * 27 + i];
  b = bias ? bias[k] : 0.0f;
}

// pins the wait for a channel's scalar loads at this point of the program
__device__ __forceinline__ void first_wait_weights(const float (&w)[27], float b) {
  asm volatile("" ::"s"(w[0]), "s"(w[15]), "s"(w[23]), "s"(w[25]), "s"(w[26]), "s"(b));
  __builtin_amdgcn_sched_barrier(0);
}

// one output channel of the thread's 4 pixels: taps added in (c, a, b) order by an fma chain starting from 0
__device__ __forceinline__ void first_channel(const float (&d)[3][3][6], const float (&w)[27], float (&o)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) o[p] = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float wv = w[(c * 3 + a) * 3 + b];
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = fma_rn(wv, d[c][a][b + p], o[p]);
      }
}

// a lane's contribution to the BatchNorm sums of y + bias over its 4 pixels (0 for the lanes beyond the tensor)
__device__ __forceinline__ void first_lane_sums(const float (&o)[4], float bk, bool live, float& s0, float& s1) {
  s0 = 0.0f;
  s1 = 0.0f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float v = o[p] + bk;
    s0 += v;
    s1 = fma_rn(v, v, s1);
  }
  if (!live) s0 = s1 = 0.0f;
}

template <bool STATS, bool NT = false>
__global__ __launch_bounds__(kFirstThreads) void conv_first_fwd_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ wt /*[64][27]*/, int H,
                                                                       int W, long Q, float* __restrict__ y,
                                                                       const float* __restrict__ bias,
                                                                       float* __restrict__ parts) {
  __shared__ float red[kFirstThreads / 64][64][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long q_raw = (long)blockIdx.x * kFirstThreads + tid;
  const bool live = q_raw < Q;
  const long q = live ? q_raw : Q - 1;          // dead lanes shadow the last quad: every lane takes part in the shuffles
  const int W4 = W >> 2;
  const int w4 = (int)(q % W4);
  const long t = q / W4;
  const int h = (int)(t % H);
  const long n = t / H;
  const bool has_left = w4 > 0, has_right = w4 < W4 - 1;
  const bool left_lane = has_left && lane > 0, right_lane = has_right && lane < 63;
  v4f mid[3][3];
  float el[3][3], er[3][3];
  const float* rp[3][3];
  bool rin[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int r = h + a - 1;
    rin[a] = r >= 0 && r < H;
    const int rc = rin[a] ? r : h;               // rows outside the image: a valid row, zeroed below
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rp[c][a] = x + (((size_t)n * 3 + c) * H + rc) * W + 4 * w4;
      mid[c][a] = *reinterpret_cast<const v4f*>(rp[c][a]);
      el[c][a] = er[c][a] = 0.0f;
    }
  }
  if (has_left && !left_lane) {                  // the wave's edge lanes fetch their halo column (one branch each:
#pragma unroll                                   // loads behind per-element branches would serialise their waits)
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) el[c][a] = rp[c][a][-1];
  }
  if (has_right && !right_lane) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) er[c][a] = rp[c][a][4];
  }
  float d[3][3][6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v4f m = mid[c][a];
      if (!rin[a]) m = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      float lft = __shfl_up(m[3], 1, 64), rgt = __shfl_down(m[0], 1, 64);
      if (!left_lane) lft = (has_left && rin[a]) ? el[c][a] : 0.0f;
      if (!right_lane) rgt = (has_right && rin[a]) ? er[c][a] : 0.0f;
      d[c][a][0] = lft; d[c][a][1] = m[0]; d[c][a][2] = m[1]; d[c][a][3] = m[2]; d[c][a][4] = m[3]; d[c][a][5] = rgt;
    }
  }
  const size_t plane = (size_t)H * W;
  float* yp = y + ((size_t)n * 64 * H + h) * W + 4 * w4;
  // Two output channels per round; a channel's 27 weights (+ bias) are fetched one channel ahead of their use, so
  // the scalar loads' latency is covered by the previous channel's 54 packed fmas.
  float wa[27], wb[27], ba = 0.0f, bb = 0.0f;
  first_load_weights(wt, STATS ? bias : nullptr, 0, wa, ba);
  for (int k = 0; k < 64; k += 2) {
    // (scalar loads return out of order, so a wait for one means a wait for all: the wait for the channel about to
    // be used comes first -- its loads were issued a channel ago -- and only then the next channel's loads go out)
    first_wait_weights(wa, ba);
    first_load_weights(wt, STATS ? bias : nullptr, k + 1, wb, bb);
    __builtin_amdgcn_sched_barrier(0);
    float oa[4], ob[4];
    first_channel(d, wa, oa);
    if (live) st_stream<NT>(reinterpret_cast<v4f*>(yp + (size_t)k * plane), (v4f){oa[0], oa[1], oa[2], oa[3]});
    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
    if (STATS) first_lane_sums(oa, ba, live, a0, a1);
    first_wait_weights(wb, bb);
    first_load_weights(wt, STATS ? bias : nullptr, k + 2 < 64 ? k + 2 : 63, wa, ba);
    __builtin_amdgcn_sched_barrier(0);
    first_channel(d, wb, ob);
    if (live) st_stream<NT>(reinterpret_cast<v4f*>(yp + (size_t)(k + 1) * plane), (v4f){ob[0], ob[1], ob[2], ob[3]});
    if (STATS) {
      first_lane_sums(ob, bb, live, b0, b1);
      wave_sum4_to_last(a0, a1, b0, b1);
      if (lane == 63) {
        *reinterpret_cast<v2f*>(&red[wave][k][0]) = (v2f){a0, a1};
        *reinterpret_cast<v2f*>(&red[wave][k + 1][0]) = (v2f){b0, b1};
      }
    }
  }
  if (STATS) {
    __syncthreads();
    if (tid < 64) {
      float t0 = red[0][tid][0], t1 = red[0][tid][1];
#pragma unroll
      for (int w2 = 1; w2 < kFirstThreads / 64; ++w2) { t0 += red[w2][tid][0]; t1 += red[w2][tid][1]; }
      float* out = parts + ((size_t)tid * gridDim.x + blockIdx.x) * 2;
      out[0] = t0;
      out[1] = t1;
    }
  }
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_conv_first_dw_workspace_floats(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)2048 * 64 * 32;
}

static int conv_first_dw_launch(const char* fn, const float* x, const float* dy, fpsg::FoldArgs fa, int N, int C, int K,
                                int H, int W, float* dw, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  const bool fold = fa.y != nullptr;
  FPSG_REQUIRE(C == 3 && K == 64, FPSG_E_SHAPE, "%s: the 3 -> 64 channel layer only (got %d -> %d)", fn, C, K);
  FPSG_REQUIRE(N > 0 && H > 0 && W > 0 && W % 4 == 0, FPSG_E_SHAPE,
               "%s: N, H positive and W a positive multiple of 4 (got %d,%d,%d)", fn, N, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(dw); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(fa.y) & 15) == 0, FPSG_E_ALIGN,
               "%s: dy / y / ga must be 16-byte aligned", fn);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int px = W % 112 == 0 ? 112 : 64;
  const int segs_per_row = (W + px - 1) / px;
  const long n_segs_l = (long)N * H * segs_per_row;
  FPSG_REQUIRE(n_segs_l < (1L << 31), FPSG_E_LIMIT, "fpsg_conv_first_dw: %ld row segments beyond the 32-bit walk", n_segs_l);
  const int n_segs = (int)n_segs_l;
  // as many workgroups as are resident at once (LDS: 4 per CU with 112-pixel tiles, 8 with 64-pixel tiles); each walks
  // over its share of the segments.  (Non-temporal loads of dy were measured here: 144 -> 185 us; plain loads.)
  const int resident = px == 112 ? 1024 : 2048;
  const int blocks = n_segs < resident ? n_segs : resident;
  if (px == 112 && fold)
    hipLaunchKernelGGL((conv_first_dw_kernel<112, true>), dim3(blocks), dim3(kFirstThreads), 0, s, x, dy, fa, H, W, segs_per_row, n_segs, ws);
  else if (px == 112)
    hipLaunchKernelGGL((conv_first_dw_kernel<112, false>), dim3(blocks), dim3(kFirstThreads), 0, s, x, dy, fa, H, W, segs_per_row, n_segs, ws);
  else if (fold)
    hipLaunchKernelGGL((conv_first_dw_kernel<64, true>), dim3(blocks), dim3(kFirstThreads), 0, s, x, dy, fa, H, W, segs_per_row, n_segs, ws);
  else
    hipLaunchKernelGGL((conv_first_dw_kernel<64, false>), dim3(blocks), dim3(kFirstThreads), 0, s, x, dy, fa, H, W, segs_per_row, n_segs, ws);
  int rc = launch_status(fn);
  if (rc) return rc;
  hipLaunchKernelGGL(conv_first_dw_reduce_kernel, dim3(64), dim3(1024), 0, s, ws, blocks, dw);
  return launch_status(fn);
}

extern "C" int fpsg_conv_first_dw(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                                  float* ws, fpsg_stream_t stream) {
  return conv_first_dw_launch("fpsg_conv_first_dw", x, dy, fpsg::FoldArgs{nullptr, nullptr, nullptr, nullptr}, N, C, K, H, W,
                              dw, ws, stream);
}

extern "C" int fpsg_conv_first_dw_fold(const float* x, const float* y, const float* ga, const float* chan,
                                       const float* coef, const float* pre_bias, int N, int C, int K, int H, int W,
                                       float* dw, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(coef);
  FPSG_REQUIRE(!misaligned4(pre_bias), FPSG_E_ALIGN, "fpsg_conv_first_dw_fold: pre_bias not 4-byte aligned");
  return conv_first_dw_launch("fpsg_conv_first_dw_fold", x, ga, FoldArgs{y, chan, coef, pre_bias}, N, C, K, H, W, dw, ws,
                              stream);
}

extern "C" int fpsg_conv_first_parts(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0 || W % 4) return 0;
  const long Q = (long)N * H * (W / 4);
  return (int)((Q + fpsg::kFirstThreads - 1) / fpsg::kFirstThreads);
}

extern "C" int fpsg_conv_first_fwd(const float* x, const float* w, int N, int C, int K, int H, int W, float* y,
                                   const float* bias, float* parts, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(C == 3 && K == 64, FPSG_E_SHAPE, "fpsg_conv_first_fwd: the 3 -> 64 channel layer only (got %d -> %d)", C, K);
  FPSG_REQUIRE(N > 0 && H > 0 && W > 0 && W % 4 == 0, FPSG_E_SHAPE,
               "fpsg_conv_first_fwd: N, H positive and W a positive multiple of 4 (got %d,%d,%d)", N, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(w); FPSG_REQUIRE_PTR(y);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_conv_first_fwd: x and y must be 16-byte aligned");
  FPSG_REQUIRE(!misaligned4(bias) && !misaligned4(parts) && !misaligned4(w), FPSG_E_ALIGN,
               "fpsg_conv_first_fwd: w / bias / parts not 4-byte aligned");
  const long Q = (long)N * H * (W / 4);
  const long blocks = (Q + kFirstThreads - 1) / kFirstThreads;
  FPSG_REQUIRE(blocks < (1L << 31), FPSG_E_LIMIT, "fpsg_conv_first_fwd: %ld workgroups beyond the grid limit", blocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool nt = beyond_cache((size_t)N * 64 * H * W * sizeof(float));      // y: 475 MB at 37 images, written once
  if (parts && nt)
    hipLaunchKernelGGL((conv_first_fwd_kernel<true, true>), dim3((unsigned)blocks), dim3(kFirstThreads), 0, s, x, w, H, W, Q, y, bias, parts);
  else if (parts)
    hipLaunchKernelGGL(conv_first_fwd_kernel<true>, dim3((unsigned)blocks), dim3(kFirstThreads), 0, s, x, w, H, W, Q, y, bias, parts);
  else if (nt)
    hipLaunchKernelGGL((conv_first_fwd_kernel<false, true>), dim3((unsigned)blocks), dim3(kFirstThreads), 0, s, x, w, H, W, Q, y, nullptr, nullptr);
  else
    hipLaunchKernelGGL(conv_first_fwd_kernel<false>, dim3((unsigned)blocks), dim3(kFirstThreads), 0, s, x, w, H, W, Q, y, nullptr, nullptr);
  return launch_status("fpsg_conv_first_fwd");
}
