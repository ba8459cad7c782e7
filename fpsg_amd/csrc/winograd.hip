// winograd.hip -- K6: Winograd F(2x2, 3x3) transforms for the deep 3x3 convolutions of the
// VGG16-BN trunk (reference src/models/image_net.py:14, torchvision vgg16_bn.features: thirteen
// Conv2d(3x3, padding 1, stride 1); here the ones with >= 256 channels), gfx950.
//
// conv(x, w)[n,k] = sum_c x[n,c] * w[k,c] (3x3, pad 1) is evaluated per 2x2 output tile as
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A ,   d = the 4x4 input tile at stride 2
// i.e. 16 independent GEMMs  M[xi] = U[xi] (K x C)  *  V[xi] (C x P)   over the P = N*(H/2)*(W/2)
// tiles, with 2.25x fewer multiplications than the direct form.  The GEMMs are plain, large fp32
// matrix products and run on the MFMA pipes through the caller's BLAS (hipBLASLt via torch.bmm:
// 110-120 TFLOP/s at these shapes); this file holds the data-movement halves, which are pure HBM
// streams:
//   input transform   x  [N,C,H,W]   -> V  [16,C,P]     (reads 1x, writes 4x the tensor)
//   output transform  M  [16,K,P]    -> y  [N,K,H,W]    (reads 4x, writes 1x)
//   filter transform  w  [K,C,3,3]   -> U  [16,K,C]     (or, for the data gradient, the
//                                                        180-degree-rotated, transposed filter)
// and, for the weight gradient  dU[xi] = dM[xi] (K x P) * V[xi]^T (P x C):
//   grad-output transform  dy [N,K,H,W] -> dM [16,K,P]  (dM = A dY A^T per tile)
//   filter-grad transform  dU [16,K,C]  -> dw [K,C,3,3] (dw = G^T dU G)
// Tile p = (n*Th + th)*Tw + tw covers output rows 2th..2th+1, cols 2tw..2tw+1 and input rows
// 2th-1..2th+2, cols 2tw-1..2tw+2 (zero outside the image).  Threads run along p, so every
// transform-domain access is a coalesced 4-byte stream and the image-side accesses of a wave
// cover contiguous row segments.  Deterministic (no atomics).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kWinoThreads = 256;

__global__ __launch_bounds__(kWinoThreads) void wino_input_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                   int Th, int Tw, long P, float* __restrict__ V) {
  const long p = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  if (p >= P) return;
  const int c = blockIdx.y;
  const int tw = (int)(p % Tw);
  const long q = p / Tw;
  const int th = (int)(q % Th);
  const long n = q / Th;
  const float* xp = x + ((size_t)n * C + c) * H * W;
  const int r0 = 2 * th - 1, c0 = 2 * tw - 1;
  float d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + i;
    const bool rin = r >= 0 && r < H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c0 + j;
      d[i][j] = (rin && cc >= 0 && cc < W) ? xp[(size_t)r * W + cc] : 0.0f;
    }
  }
  float t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j];
    t[1][j] = d[1][j] + d[2][j];
    t[2][j] = d[2][j] - d[1][j];
    t[3][j] = d[1][j] - d[3][j];
  }
  const size_t plane = (size_t)C * P;
  float* vp = V + (size_t)c * P + p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vp[(size_t)(4 * i + 0) * plane] = t[i][0] - t[i][2];
    vp[(size_t)(4 * i + 1) * plane] = t[i][1] + t[i][2];
    vp[(size_t)(4 * i + 2) * plane] = t[i][2] - t[i][1];
    vp[(size_t)(4 * i + 3) * plane] = t[i][1] - t[i][3];
  }
}

__global__ __launch_bounds__(kWinoThreads) void wino_output_kernel(const float* __restrict__ M, int K, int H, int W,
                                                                    int Th, int Tw, long P, float* __restrict__ y) {
  const long p = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  if (p >= P) return;
  const int k = blockIdx.y;
  const int tw = (int)(p % Tw);
  const long q = p / Tw;
  const int th = (int)(q % Th);
  const long n = q / Th;
  const size_t plane = (size_t)K * P;
  const float* mp = M + (size_t)k * P + p;
  float m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = mp[(size_t)(4 * i + j) * plane];
  float s[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
    s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
  }
  float* yp = y + (((size_t)n * K + k) * H + 2 * th) * W + 2 * tw;
  v2f o0, o1;
  o0[0] = (s[0][0] + s[0][1]) + s[0][2];
  o0[1] = (s[0][1] - s[0][2]) - s[0][3];
  o1[0] = (s[1][0] + s[1][1]) + s[1][2];
  o1[1] = (s[1][1] - s[1][2]) - s[1][3];
  *reinterpret_cast<v2f*>(yp) = o0;
  *reinterpret_cast<v2f*>(yp + W) = o1;
}

// dM = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
__global__ __launch_bounds__(kWinoThreads) void wino_grad_output_kernel(const float* __restrict__ dy, int K, int H,
                                                                         int W, int Th, int Tw, long P,
                                                                         float* __restrict__ dM) {
  const long p = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  if (p >= P) return;
  const int k = blockIdx.y;
  const int tw = (int)(p % Tw);
  const long q = p / Tw;
  const int th = (int)(q % Th);
  const long n = q / Th;
  const float* yp = dy + (((size_t)n * K + k) * H + 2 * th) * W + 2 * tw;
  const v2f y0 = *reinterpret_cast<const v2f*>(yp);
  const v2f y1 = *reinterpret_cast<const v2f*>(yp + W);
  float r[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    r[0][j] = y0[j];
    r[1][j] = y0[j] + y1[j];
    r[2][j] = y0[j] - y1[j];
    r[3][j] = -y1[j];
  }
  const size_t plane = (size_t)K * P;
  float* mp = dM + (size_t)k * P + p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mp[(size_t)(4 * i + 0) * plane] = r[i][0];
    mp[(size_t)(4 * i + 1) * plane] = r[i][0] + r[i][1];
    mp[(size_t)(4 * i + 2) * plane] = r[i][0] - r[i][1];
    mp[(size_t)(4 * i + 3) * plane] = -r[i][1];
  }
}

// U = G g G^T per (k, c); flip != 0: the filter of the data gradient, g'[c][k] = rot180(g[k][c]),
// written as U [16, C, K].
__global__ void wino_filter_kernel(const float* __restrict__ w, int K, int C, int flip, float* __restrict__ U) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * C) return;
  const int k = (int)(e / C), c = (int)(e % C);
  const float* g = w + (size_t)e * 9;
  float a[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = flip ? g[(2 - i) * 3 + (2 - j)] : g[i * 3 + j];
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0][j] = a[0][j];
    t[1][j] = 0.5f * ((a[0][j] + a[1][j]) + a[2][j]);
    t[2][j] = 0.5f * ((a[0][j] - a[1][j]) + a[2][j]);
    t[3][j] = a[2][j];
  }
  const size_t plane = (size_t)K * C;
  float* up = U + (flip ? (size_t)c * K + k : (size_t)k * C + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    up[(size_t)(4 * i + 0) * plane] = t[i][0];
    up[(size_t)(4 * i + 1) * plane] = 0.5f * ((t[i][0] + t[i][1]) + t[i][2]);
    up[(size_t)(4 * i + 2) * plane] = 0.5f * ((t[i][0] - t[i][1]) + t[i][2]);
    up[(size_t)(4 * i + 3) * plane] = t[i][2];
  }
}

// dw = G^T dU G per (k, c)
__global__ void wino_filter_grad_kernel(const float* __restrict__ dU, int K, int C, float* __restrict__ dw) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * C) return;
  const size_t plane = (size_t)K * C;
  float u[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[i][j] = dU[(size_t)(4 * i + j) * plane + e];
  float a[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
    a[1][j] = 0.5f * (u[1][j] - u[2][j]);
    a[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
  }
  float* g = dw + (size_t)e * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g[i * 3 + 0] = a[i][0] + 0.5f * (a[i][1] + a[i][2]);
    g[i * 3 + 1] = 0.5f * (a[i][1] - a[i][2]);
    g[i * 3 + 2] = 0.5f * (a[i][1] + a[i][2]) + a[i][3];
  }
}

int check_image(const char* fn, int N, int C, int H, int W) {
  FPSG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0, FPSG_E_SHAPE,
               "%s: N,C positive and H,W positive and even (got %d,%d,%d,%d)", fn, N, C, H, W);
  const long P = (long)N * (H / 2) * (W / 2);
  FPSG_REQUIRE(C <= 65535 && (P + kWinoThreads - 1) / kWinoThreads < (1L << 31), FPSG_E_LIMIT,
               "%s: C=%d or tile count %ld beyond the grid limits", fn, C, P);
  return 0;
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_wino_input_transform(const float* x, int N, int C, int H, int W, float* V, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_input_transform", N, C, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(V);
  const long P = (long)N * (H / 2) * (W / 2);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), C);
  hipLaunchKernelGGL(wino_input_kernel, grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W,
                     H / 2, W / 2, P, V);
  return launch_status("fpsg_wino_input_transform");
}

extern "C" int fpsg_wino_output_transform(const float* M, int N, int K, int H, int W, float* y, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_output_transform", N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(M); FPSG_REQUIRE_PTR(y);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(y) & 7) == 0, FPSG_E_ALIGN, "fpsg_wino_output_transform: y must be 8-byte aligned");
  const long P = (long)N * (H / 2) * (W / 2);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), K);
  hipLaunchKernelGGL(wino_output_kernel, grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), M, K, H, W,
                     H / 2, W / 2, P, y);
  return launch_status("fpsg_wino_output_transform");
}

extern "C" int fpsg_wino_grad_output_transform(const float* dy, int N, int K, int H, int W, float* dM,
                                               fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_grad_output_transform", N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(dM);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 7) == 0, FPSG_E_ALIGN, "fpsg_wino_grad_output_transform: dy must be 8-byte aligned");
  const long P = (long)N * (H / 2) * (W / 2);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), K);
  hipLaunchKernelGGL(wino_grad_output_kernel, grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), dy, K, H,
                     W, H / 2, W / 2, P, dM);
  return launch_status("fpsg_wino_grad_output_transform");
}

extern "C" int fpsg_wino_filter_transform(const float* w, int K, int C, int flip_transpose, float* U,
                                          fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(K > 0 && C > 0, FPSG_E_SHAPE, "fpsg_wino_filter_transform: K,C must be positive (got %d,%d)", K, C);
  FPSG_REQUIRE_PTR(w); FPSG_REQUIRE_PTR(U);
  const long n = (long)K * C;
  hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, K, C, flip_transpose, U);
  return launch_status("fpsg_wino_filter_transform");
}

extern "C" int fpsg_wino_filter_grad_transform(const float* dU, int K, int C, float* dw, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(K > 0 && C > 0, FPSG_E_SHAPE, "fpsg_wino_filter_grad_transform: K,C must be positive (got %d,%d)", K, C);
  FPSG_REQUIRE_PTR(dU); FPSG_REQUIRE_PTR(dw);
  const long n = (long)K * C;
  hipLaunchKernelGGL(wino_filter_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dU, K, C, dw);
  return launch_status("fpsg_wino_filter_grad_transform");
}
