// conv_first.hip -- K8: weight gradient of the first VGG convolution (3 -> 64 channels, 3x3,
// padding 1; torchvision vgg16_bn.features[0], src/models/image_net.py:14), for gfx950.
//
// dw[k][c][a][b] = sum over (n, h, w) of dy[n,k,h,w] * x[n,c,h+a-1,w+b-1] is a [64 x P] . [P x 27]
// product with a 1.86 M-long reduction (37 images of 224x224) and a 1,728-element result: 6.4
// GFLOP over 475 MB of dy -- HBM-bound by two orders of magnitude.  The library runs it as an
// NHWC implicit GEMM behind two layout transposes of dy (0.73 ms); here dy is read exactly once:
//   * a workgroup walks over row segments of 64 pixels; dy[64 ch][64 px] is staged coalesced into
//     LDS (256-B runs per channel) together with the 3 x 3 x 66 input patch of the segment;
//   * wave w owns output channels 16w .. 16w+15: 16 steps of v_mfma_f32_16x16x4_f32 per segment
//     with A = dy (channel x 4 pixels) and B = the im2col patch (4 pixels x 16 taps; taps 27..31
//     are zero), two tap tiles -> 8 accumulator registers carried over all segments;
//   * each workgroup writes one [64][32] partial; a second kernel sums the partials in a fixed
//     order in fp64.  Deterministic.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kPx = 64;                 // pixels per segment
constexpr int kDyLd = kPx + 4;          // LDS row stride of the dy tile: 2 lanes per bank, the minimum
constexpr int kXLd = kPx + 4;           // row stride of the input patch (66 used)
constexpr int kFirstThreads = 256;

__global__ __launch_bounds__(kFirstThreads) void conv_first_dw_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ dy, int H, int W,
                                                                      int segs_per_row, long n_segs,
                                                                      float* __restrict__ part /*[grid][64][32]*/) {
  __shared__ float dyt[64 * kDyLd];
  __shared__ float xt[9 * kXLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, col = lane & 15;
  // B operand: tap index 16*nt + col -> (c, a, b); its patch element for pixel p is xt[(3c+a)*kXLd + p + b]
  int boff[2];
  bool bval[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tap = 16 * nt + col;
    bval[nt] = tap < 27;
    const int t = bval[nt] ? tap : 0;
    boff[nt] = (t / 3) * kXLd + (t % 3);          // t/3 = 3c + a, t%3 = b
  }
  v4f acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
  for (long seg = blockIdx.x; seg < n_segs; seg += gridDim.x) {
    const long row = seg / segs_per_row;            // (n, h)
    const int w0 = (int)(seg - row * segs_per_row) * kPx;
    const long n = row / H;
    const int h = (int)(row - n * H);
    __syncthreads();                                // the previous segment's tiles are consumed
    // dy tile: thread -> (channel = it*16 + tid/16, pixels 4*(tid%16) .. +3)
    {
      const int q4 = (tid & 15) * 4;
      const bool in = w0 + q4 < W;                  // W % 4 == 0: a vector is inside or outside as a whole
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ch = it * 16 + (tid >> 4);
        v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (in) v = *reinterpret_cast<const v4f*>(dy + (((size_t)n * 64 + ch) * H + h) * W + w0 + q4);
        *reinterpret_cast<v4f*>(dyt + ch * kDyLd + q4) = v;
      }
    }
    // input patch: 3 channels x 3 rows x 66 columns (zero outside the image)
    for (int e = tid; e < 9 * (kPx + 2); e += kFirstThreads) {
      const int cr = e / (kPx + 2), j = e - cr * (kPx + 2);
      const int c = cr / 3, r = cr - 3 * c;
      const int hh = h + r - 1, ww = w0 + j - 1;
      xt[cr * kXLd + j] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? x[(((size_t)n * 3 + c) * H + hh) * W + ww] : 0.0f;
    }
    __syncthreads();
    const float* ap = dyt + (16 * wave + col) * kDyLd + kk;
    const float* bp0 = xt + boff[0] + kk;
    const float* bp1 = xt + boff[1] + kk;
#pragma unroll
    for (int s = 0; s < kPx / 4; ++s) {
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

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_conv_first_dw_workspace_floats(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)2048 * 64 * 32;
}

extern "C" int fpsg_conv_first_dw(const float* x, const float* dy, int N, int C, int K, int H, int W, float* dw,
                                  float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(C == 3 && K == 64, FPSG_E_SHAPE, "fpsg_conv_first_dw: the 3 -> 64 channel layer only (got %d -> %d)", C, K);
  FPSG_REQUIRE(N > 0 && H > 0 && W > 0 && W % 4 == 0, FPSG_E_SHAPE,
               "fpsg_conv_first_dw: N, H positive and W a positive multiple of 4 (got %d,%d,%d)", N, H, W);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(dw); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, FPSG_E_ALIGN, "fpsg_conv_first_dw: dy must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int segs_per_row = (W + kPx - 1) / kPx;
  const long n_segs = (long)N * H * segs_per_row;
  const int blocks = n_segs < 2048 ? (int)n_segs : 2048;   // 8 workgroups per CU: staging of one overlaps the MFMAs of others
  hipLaunchKernelGGL(conv_first_dw_kernel, dim3(blocks), dim3(kFirstThreads), 0, s, x, dy, H, W, segs_per_row, n_segs, ws);
  int rc = launch_status("fpsg_conv_first_dw(partials)");
  if (rc) return rc;
  hipLaunchKernelGGL(conv_first_dw_reduce_kernel, dim3(64), dim3(1024), 0, s, ws, blocks, dw);
  return launch_status("fpsg_conv_first_dw(reduce)");
}
