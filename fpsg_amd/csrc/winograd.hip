// winograd.hip -- K6: Winograd F(m x m, 3x3) transforms, m = 2 or 4, for the deep 3x3 convolutions of the
// VGG16-BN trunk (reference src/models/image_net.py:14, torchvision vgg16_bn.features: thirteen
// Conv2d(3x3, padding 1, stride 1); here the ones with >= 256 channels), gfx950.
//
// conv(x, w)[n,k] = sum_c x[n,c] * w[k,c] (3x3, pad 1) is evaluated per m x m output tile as
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A ,   d = the (m+2)x(m+2) input tile at stride m
// i.e. (m+2)^2 independent GEMMs  M[xi] = U[xi] (K x C)  *  V[xi] (C x P)  over the
// P = N*(H/m)*(W/m) tiles, with 2.25x (m=2) or 4x (m=4) fewer multiplications than the direct
// form; the transform-domain tensors are 4x (m=2) or 2.25x (m=4) the image tensors.  The GEMMs are plain, large fp32
// matrix products and run on the MFMA pipes through the caller's BLAS (hipBLASLt via torch.bmm:
// 110-120 TFLOP/s at these shapes); this file holds the data-movement halves, which are pure HBM
// streams:
//   input transform   x  [N,C,H,W]   -> V  [A*A,C,P]    (A = m+2)
//   output transform  M  [A*A,K,P]   -> y  [N,K,H,W]
//   filter transform  w  [K,C,3,3]   -> U  [A*A,K,C]    (or, for the data gradient, the
//                                                        180-degree-rotated, transposed filter)
// and, for the weight gradient  dU[xi] = dM[xi] (K x P) * V[xi]^T (P x C):
//   grad-output transform  dy [N,K,H,W] -> dM [A*A,K,P] (dM = A dY A^T per tile)
//   filter-grad transform  dU [A*A,K,C] -> dw [K,C,3,3] (dw = G^T dU G)
// Tile p = (n*Th + th)*Tw + tw covers output rows m*th..m*th+m-1, cols m*tw..m*tw+m-1 and input
// rows m*th-1..m*th+m, cols m*tw-1..m*tw+m (zero outside the image).  Threads run along p, so every
// transform-domain access is a coalesced 4-byte stream and the image-side accesses of a wave
// cover contiguous row segments.  Deterministic (no atomics).
#include "fpsg_common.h"

namespace fpsg {
namespace {

// V, dM (written once, read once by the batched GEMM) and M (read once): when such a tensor is larger than the 256 MB
// Infinity Cache -- the 128-channel layers at 112 x 112 and 37 images: 534 MB -- non-temporal stores / loads keep it
// from sweeping the cache (input transform 168 -> 125 us, 58 -> 77 % of 8 TB/s; grad-output 157 -> 124; output 156 ->
// 140); a tensor that fits stays cached for its consumer, and the same hint costs 15-75 % there
// (profiles/r03/k6_transforms_nontemporal.txt).  A template parameter: behind a run-time branch the compiler merges the
// two stores and drops the hint.
constexpr size_t kStreamBytes = (size_t)300 << 20;
template <bool NT>
__device__ __forceinline__ void stream_store(float* p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT>
__device__ __forceinline__ float stream_load(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

constexpr int kWinoThreads = 256;

// One-dimensional transforms of F(m, 3), m = 2 and 4 (interpolation points 0, +-1 [, +-2], inf;
// the matrices of Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks").  A = m + 2.
//   in  : B^T d      (A -> A)      out : A^T m     (A -> m)
//   flt : G g        (3 -> A)      gout: A y       (m -> A)      fgrad: G^T u    (A -> 3)
template <int M> struct Wino;

template <> struct Wino<2> {
  static constexpr int A = 4;
  static __device__ __forceinline__ void in(const float (&d)[4], float (&t)[4]) {
    t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
  }
  static __device__ __forceinline__ void out(const float (&m)[4], float (&s)[2]) {
    s[0] = (m[0] + m[1]) + m[2]; s[1] = (m[1] - m[2]) - m[3];
  }
  static __device__ __forceinline__ void flt(const float (&g)[3], float (&u)[4]) {
    u[0] = g[0]; u[1] = 0.5f * ((g[0] + g[1]) + g[2]); u[2] = 0.5f * ((g[0] - g[1]) + g[2]); u[3] = g[2];
  }
  static __device__ __forceinline__ void gout(const float (&y)[2], float (&r)[4]) {
    r[0] = y[0]; r[1] = y[0] + y[1]; r[2] = y[0] - y[1]; r[3] = -y[1];
  }
  static __device__ __forceinline__ void fgrad(const float (&u)[4], float (&a)[3]) {
    a[0] = u[0] + 0.5f * (u[1] + u[2]); a[1] = 0.5f * (u[1] - u[2]); a[2] = 0.5f * (u[1] + u[2]) + u[3];
  }
};

template <> struct Wino<4> {
  static constexpr int A = 6;
  // 12 operations: rows 1,2 = (d4 - 4 d2) +- (d3 - 4 d1), rows 3,4 = (d4 - d2) +- 2 (d3 - d1)
  static __device__ __forceinline__ void in(const float (&d)[6], float (&t)[6]) {
    const float p = fma_rn(-4.0f, d[2], d[4]), q = fma_rn(-4.0f, d[1], d[3]);
    const float r = d[4] - d[2], s = d[3] - d[1];
    t[0] = fma_rn(4.0f, d[0], fma_rn(-5.0f, d[2], d[4]));
    t[1] = p + q;
    t[2] = p - q;
    t[3] = fma_rn(2.0f, s, r);
    t[4] = fma_rn(-2.0f, s, r);
    t[5] = fma_rn(4.0f, d[1], fma_rn(-5.0f, d[3], d[5]));
  }
  static __device__ __forceinline__ void out(const float (&m)[6], float (&s)[4]) {
    const float p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
    s[0] = (m[0] + p12) + p34;
    s[1] = fma_rn(2.0f, m34, m12);
    s[2] = fma_rn(4.0f, p34, p12);
    s[3] = fma_rn(8.0f, m34, m12) + m[5];
  }
  static __device__ __forceinline__ void flt(const float (&g)[3], float (&u)[6]) {
    const float e = g[0] + g[2];
    u[0] = 0.25f * g[0];
    u[1] = (-1.0f / 6.0f) * (e + g[1]);
    u[2] = (-1.0f / 6.0f) * (e - g[1]);
    const float f = fma_rn(1.0f / 24.0f, g[0], (1.0f / 6.0f) * g[2]);
    u[3] = fma_rn(1.0f / 12.0f, g[1], f);
    u[4] = fma_rn(-1.0f / 12.0f, g[1], f);
    u[5] = g[2];
  }
  static __device__ __forceinline__ void gout(const float (&y)[4], float (&r)[6]) {
    const float e = y[0] + y[2], o = y[1] + y[3];
    r[0] = y[0];
    r[1] = e + o;
    r[2] = e - o;
    const float e4 = fma_rn(4.0f, y[2], y[0]), o4 = fma_rn(8.0f, y[3], 2.0f * y[1]);
    r[3] = e4 + o4;
    r[4] = e4 - o4;
    r[5] = y[3];
  }
  static __device__ __forceinline__ void fgrad(const float (&u)[6], float (&a)[3]) {
    const float p12 = u[1] + u[2], m12 = u[1] - u[2], p34 = u[3] + u[4], m34 = u[3] - u[4];
    a[0] = fma_rn(0.25f, u[0], fma_rn(-1.0f / 6.0f, p12, (1.0f / 24.0f) * p34));
    a[1] = fma_rn(-1.0f / 6.0f, m12, (1.0f / 12.0f) * m34);
    a[2] = fma_rn(-1.0f / 6.0f, p12, fma_rn(1.0f / 6.0f, p34, u[5]));
  }
};

struct TileIndex { long n; int th, tw; };
__device__ __forceinline__ TileIndex tile_of(long p, int Th, int Tw) {
  TileIndex t;
  t.tw = (int)(p % Tw);
  const long q = p / Tw;
  t.th = (int)(q % Th);
  t.n = q / Th;
  return t;
}

// Each thread loads the M interior columns of its tile's rows as ONE aligned vector (a wave reads a
// contiguous row segment) and takes the two halo columns from the neighbouring lanes' vectors;
// only lanes at a wave edge inside an image row fetch their halo from memory.
// chan != null: the tensor is the PRE-BatchNorm output of the previous convolution and the transform
// reads a = relu(fma(x + pre_bias[c], scale[c], shift[c])) instead (exactly the value K5's apply pass
// would have stored; padding stays zero) -- the BatchNorm + ReLU apply pass between two convolutions of a
// VGG stage (one read + one write of the activation tensor) is folded into this load.
// RAG (m = 4 only): H, W even but not multiples of 4 -- the tile grid is ceil(H/4) x ceil(W/4), pixels beyond the
// image enter as zeros (the last tile row / column is half empty); rows are then only 8-byte aligned, so a tile row is
// read as two 8-byte halves and the second half of a last-column tile is not read at all.
// GO (round 4): x is an output GRADIENT whose two Winograd transforms the backward of a convolution needs -- B^T d B of the
// 6 x 6 patches (the data gradient's "input" transform) and A g A^T of the 4 x 4 tiles (the weight gradient's): the tile is
// the patch's interior, so one read serves both and dM [A*A, C, P] is written beside V (wino_grad_output_kernel's
// arithmetic on the same values: bit-identical to the two separate launches).
template <int M, bool ACT, bool NT = false, bool RAG = false, bool GO = false>
__global__ __launch_bounds__(kWinoThreads) void wino_input_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                   int Th, int Tw, long P, long Ps, float* __restrict__ V,
                                                                   const float* __restrict__ chan,
                                                                   const float* __restrict__ pre_bias,
                                                                   float* __restrict__ dM = nullptr) {
  constexpr int A = Wino<M>::A;
  typedef float vin __attribute__((ext_vector_type(M)));
  const long p_raw = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  const bool live = p_raw < P;
  const long p = live ? p_raw : P - 1;          // dead lanes shadow the last tile: every lane shuffles
  const int c = blockIdx.y;
  const TileIndex ti = tile_of(p, Th, Tw);
  const float* xp = x + ((size_t)ti.n * C + c) * H * W;
  const int r0 = M * ti.th - 1, c0 = M * ti.tw;
  const int lane = threadIdx.x & (kWave - 1);
  const bool has_left = ti.tw > 0, has_right = ti.tw < Tw - 1;
  const bool left_lane = has_left && lane > 0;              // lane-1 holds tile (th, tw-1)
  const bool right_lane = has_right && lane < kWave - 1;    // lane+1 holds tile (th, tw+1)
  float sc = 1.0f, sh = 0.0f, pb = 0.0f;
  if (ACT) { sc = chan[c]; sh = chan[C + c]; pb = pre_bias ? pre_bias[c] : 0.0f; }
  auto act = [&](float v) { return ACT ? __builtin_fmaxf(fma_rn(v + pb, sc, sh), 0.0f) : v; };
  // All loads of the tile are issued before the first use: rows outside the image read a clamped (valid)
  // row and are zeroed afterwards, so no load sits behind a branch that would serialise the waits.
  vin mid[A];
  bool rin[A];
  const float* rp[A];
  const bool hi_in = !RAG || c0 + 2 < W;                    // (RAG) columns c0 + 2, c0 + 3 of this tile are inside the image
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const int r = r0 + i;
    rin[i] = r >= 0 && r < H;
    rp[i] = xp + (size_t)(rin[i] ? r : (r < 0 ? 0 : H - 1)) * W + c0;
    if constexpr (RAG) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      const v2 lo = *reinterpret_cast<const v2*>(rp[i]);
      v2 hi = {0.0f, 0.0f};
      if (hi_in) hi = *reinterpret_cast<const v2*>(rp[i] + 2);
      mid[i][0] = lo[0]; mid[i][1] = lo[1]; mid[i][M - 2] = hi[0]; mid[i][M - 1] = hi[1];
    } else {
      mid[i] = *reinterpret_cast<const vin*>(rp[i]);
    }
  }
  float el[A], er[A];                                       // halo columns of the lanes at a wave edge
#pragma unroll
  for (int i = 0; i < A; ++i) el[i] = er[i] = 0.0f;
  if (has_left && !left_lane) {
#pragma unroll
    for (int i = 0; i < A; ++i) el[i] = rp[i][-1];
  }
  if (has_right && !right_lane) {
#pragma unroll
    for (int i = 0; i < A; ++i) er[i] = rp[i][M];
  }
  float d[A][A];
#pragma unroll
  for (int i = 0; i < A; ++i) {
#pragma unroll
    for (int j = 0; j < M; ++j) mid[i][j] = (rin[i] && (!RAG || j < 2 || hi_in)) ? act(mid[i][j]) : 0.0f;
    float lft = __shfl_up(mid[i][M - 1], 1, kWave);
    float rgt = __shfl_down(mid[i][0], 1, kWave);
    if (!left_lane) lft = (has_left && rin[i]) ? act(el[i]) : 0.0f;
    if (!right_lane) rgt = (has_right && rin[i]) ? act(er[i]) : 0.0f;
    d[i][0] = lft;
#pragma unroll
    for (int j = 0; j < M; ++j) d[i][1 + j] = mid[i][j];
    d[i][A - 1] = rgt;
  }
  float t[A][A];        // t[j][i]: column j of the tile after the transform along rows
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float col[A];
#pragma unroll
    for (int i = 0; i < A; ++i) col[i] = d[i][j];
    Wino<M>::in(col, t[j]);
  }
  const size_t plane = (size_t)C * Ps;
  if (!live) {                                   // the pad columns P .. Ps-1 of every row are written as zeros
    if (p_raw < Ps) {
#pragma unroll
      for (int q = 0; q < A * A; ++q) {
        stream_store<NT>(V + (size_t)c * Ps + p_raw + (size_t)q * plane, 0.0f);
        if constexpr (GO) stream_store<NT>(dM + (size_t)c * Ps + p_raw + (size_t)q * plane, 0.0f);
      }
    }
    return;
  }
  float* vp = V + (size_t)c * Ps + p;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float row[A], v[A];
#pragma unroll
    for (int j = 0; j < A; ++j) row[j] = t[j][i];
    Wino<M>::in(row, v);
#pragma unroll
    for (int j = 0; j < A; ++j) stream_store<NT>(vp + (size_t)(A * i + j) * plane, v[j]);
  }
  if constexpr (GO) {
    static_assert(!ACT, "an output gradient is not activated");
    // the tile = rows 1 .. M of the patch, its interior columns (zero beyond the image: mid[] was zeroed above)
    float r[M][A];        // r[j][i]: column j after the transform along rows
#pragma unroll
    for (int j = 0; j < M; ++j) {
      float col[M];
#pragma unroll
      for (int i = 0; i < M; ++i) col[i] = mid[1 + i][j];
      Wino<M>::gout(col, r[j]);
    }
    float* mp = dM + (size_t)c * Ps + p;
#pragma unroll
    for (int i = 0; i < A; ++i) {
      float row[M], o[A];
#pragma unroll
      for (int j = 0; j < M; ++j) row[j] = r[j][i];
      Wino<M>::gout(row, o);
#pragma unroll
      for (int j = 0; j < A; ++j) stream_store<NT>(mp + (size_t)(A * i + j) * plane, o[j]);
    }
  }
}

// STATS: the kernel also accumulates, per output channel, the partial sums a following BatchNorm needs --
// sum(y + bias[k]) and sum((y + bias[k])^2) over the workgroup's 256 tiles -> parts[k][blockIdx.x][2] (per thread 16
// pixels in row order, then the wave's fixed tree, then the four waves in order: deterministic).  K5's
// statistics pass over y (one read of the tensor) is then not needed (fpsg_bn_stats with parts).
// BWD (with STATS): y is the gradient of relu(bn(xpre + bias)) -- the data gradient of the convolution that consumed
// that activation -- and the sums are the ones BatchNorm's backward needs, in the arithmetic of K5's own pass
// (bn_reduce_kernel, MODE 1): dz = y * [fma(x, scale, shift) > 0], sum(dz) and sum(dz * (x - mean) * rstd) with
// x = xpre + bias[k]; chan = [4][K] scale, shift, mean, rstd.  K5's backward then starts at its finalize
// (fpsg_bn_act_bwd_parts) and reads neither tensor for the sums.
// RAG (m = 4 only, see wino_input_kernel): only the pixels inside the image are written (and counted in the sums).
template <int M, bool STATS, bool BWD = false, bool NT = false, bool RAG = false>
__global__ __launch_bounds__(kWinoThreads) void wino_output_kernel(const float* __restrict__ Mt, int K, int H, int W,
                                                                    int Th, int Tw, long P, long Ps, float* __restrict__ y,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ parts,
                                                                    const float* __restrict__ xpre = nullptr,
                                                                    const float* __restrict__ chan = nullptr) {
  constexpr int A = Wino<M>::A;
  typedef float vout __attribute__((ext_vector_type(M)));
  __shared__ float red[2 * (kWinoThreads / kWave)];
  const long p_raw = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  const bool live = p_raw < P;
  if (!STATS && !live) return;
  const long p = live ? p_raw : P - 1;          // (STATS) dead lanes stay for the reduction and add nothing
  const int k = blockIdx.y;
  const TileIndex ti = tile_of(p, Th, Tw);
  const size_t plane = (size_t)K * Ps;
  const float* mp = Mt + (size_t)k * Ps + p;
  float s[A][M];        // s[j][i]: column j after the transform along rows
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float m[A];
#pragma unroll
    for (int i = 0; i < A; ++i) m[i] = stream_load<NT>(mp + (size_t)(A * i + j) * plane);
    Wino<M>::out(m, s[j]);
  }
  const size_t yoff = (((size_t)ti.n * K + k) * H + M * ti.th) * W + M * ti.tw;
  float* yp = y + yoff;
  const float b = (STATS && bias) ? bias[k] : 0.0f;
  float sc = 0.0f, sh = 0.0f, mu = 0.0f, rs = 0.0f;
  vout xrow[M];
  const bool hi_in = !RAG || M * ti.tw + 2 < W;             // (RAG) the tile's last two columns are inside the image
  bool row_in[M];
#pragma unroll
  for (int i = 0; i < M; ++i) row_in[i] = !RAG || M * ti.th + i < H;
  if constexpr (BWD) {
    sc = chan[k]; sh = chan[K + k]; mu = chan[2 * K + k]; rs = chan[3 * K + k];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if constexpr (RAG) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
        if (row_in[i]) {
          lo = *reinterpret_cast<const v2*>(xpre + yoff + (size_t)i * W);
          if (hi_in) hi = *reinterpret_cast<const v2*>(xpre + yoff + (size_t)i * W + 2);
        }
        xrow[i][0] = lo[0]; xrow[i][1] = lo[1]; xrow[i][M - 2] = hi[0]; xrow[i][M - 1] = hi[1];
      } else {
        xrow[i] = *reinterpret_cast<const vout*>(xpre + yoff + (size_t)i * W);   // dead lanes: the last tile
      }
    }
  }
  float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    float row[A], o[M];
#pragma unroll
    for (int j = 0; j < A; ++j) row[j] = s[j][i];
    Wino<M>::out(row, o);
    vout ov;
#pragma unroll
    for (int j = 0; j < M; ++j) ov[j] = o[j];
    if constexpr (RAG) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      if (live && row_in[i]) {
        *reinterpret_cast<v2*>(yp + (size_t)i * W) = (v2){o[0], o[1]};
        if (hi_in) *reinterpret_cast<v2*>(yp + (size_t)i * W + 2) = (v2){o[M - 2], o[M - 1]};
      }
    } else {
      if (live) *reinterpret_cast<vout*>(yp + (size_t)i * W) = ov;
    }
    if constexpr (BWD) {
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const bool in = live && row_in[i] && (j < 2 || hi_in);
        const float xv = xrow[i][j] + b;
        const float dz = (in && fma_rn(xv, sc, sh) > 0.0f) ? o[j] : 0.0f;
        a0 += dz;
        a1 = fma_rn(dz, (xv - mu) * rs, a1);
      }
    } else if (STATS) {
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const bool in = live && row_in[i] && (j < 2 || hi_in);
        const float v = in ? o[j] + b : 0.0f;
        a0 += v;
        a1 = fma_rn(v, v, a1);
      }
    }
  }
  if (STATS) {
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { red[2 * wave] = a0; red[2 * wave + 1] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float t0 = red[0], t1 = red[1];
#pragma unroll
      for (int w = 1; w < kWinoThreads / kWave; ++w) { t0 += red[2 * w]; t1 += red[2 * w + 1]; }
      float* out = parts + ((size_t)k * gridDim.x + blockIdx.x) * 2;
      out[0] = t0;
      out[1] = t1;
    }
  }
}

// dM = A dY A^T per tile
template <int M, bool NT = false, bool RAG = false>
__global__ __launch_bounds__(kWinoThreads) void wino_grad_output_kernel(const float* __restrict__ dy, int K, int H,
                                                                         int W, int Th, int Tw, long P, long Ps,
                                                                         float* __restrict__ dM) {
  constexpr int A = Wino<M>::A;
  typedef float vin __attribute__((ext_vector_type(M)));
  const long p = (long)blockIdx.x * kWinoThreads + threadIdx.x;
  const int k = blockIdx.y;
  if (p >= P) {                                  // the pad columns P .. Ps-1 of every row are written as zeros
    if (p < Ps) {
#pragma unroll
      for (int q = 0; q < A * A; ++q) stream_store<NT>(dM + (size_t)k * Ps + p + (size_t)q * (size_t)K * Ps, 0.0f);
    }
    return;
  }
  const TileIndex ti = tile_of(p, Th, Tw);
  const float* yp = dy + (((size_t)ti.n * K + k) * H + M * ti.th) * W + M * ti.tw;
  float yv[M][M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if constexpr (RAG) {                                     // pixels beyond the image: zero gradient
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
      if (M * ti.th + i < H) {
        lo = *reinterpret_cast<const v2*>(yp + (size_t)i * W);
        if (M * ti.tw + 2 < W) hi = *reinterpret_cast<const v2*>(yp + (size_t)i * W + 2);
      }
      yv[i][0] = lo[0]; yv[i][1] = lo[1]; yv[i][M - 2] = hi[0]; yv[i][M - 1] = hi[1];
    } else {
      const vin v = *reinterpret_cast<const vin*>(yp + (size_t)i * W);
#pragma unroll
      for (int j = 0; j < M; ++j) yv[i][j] = v[j];
    }
  }
  float r[M][A];        // r[j][i]: column j after the transform along rows
#pragma unroll
  for (int j = 0; j < M; ++j) {
    float col[M];
#pragma unroll
    for (int i = 0; i < M; ++i) col[i] = yv[i][j];
    Wino<M>::gout(col, r[j]);
  }
  const size_t plane = (size_t)K * Ps;
  float* mp = dM + (size_t)k * Ps + p;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float row[M], o[A];
#pragma unroll
    for (int j = 0; j < M; ++j) row[j] = r[j][i];
    Wino<M>::gout(row, o);
#pragma unroll
    for (int j = 0; j < A; ++j) stream_store<NT>(mp + (size_t)(A * i + j) * plane, o[j]);
  }
}

// U = G g G^T per (k, c); flip != 0: the filter of the data gradient, g'[c][k] = rot180(g[k][c]),
// written as U [A*A, C, K].
template <int M>
__device__ __forceinline__ void wino_filter_one(const float* __restrict__ w, int K, int C, int flip, float* __restrict__ U,
                                                long e) {
  constexpr int A = Wino<M>::A;
  // threads run along the fastest axis of the OUTPUT (c, or k when transposing) so that the A*A
  // plane writes are coalesced; the transposing form then reads its 36-byte filters K*36 bytes apart
  const int k = flip ? (int)(e % K) : (int)(e / C), c = flip ? (int)(e / K) : (int)(e % C);
  const float* g = w + ((size_t)k * C + c) * 9;
  float t[3][A];        // t[j][i]: column j after the transform along rows
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) col[i] = flip ? g[(2 - i) * 3 + (2 - j)] : g[i * 3 + j];
    Wino<M>::flt(col, t[j]);
  }
  const size_t plane = (size_t)K * C;
  float* up = U + (flip ? (size_t)c * K + k : (size_t)k * C + c);
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float row[3], o[A];
#pragma unroll
    for (int j = 0; j < 3; ++j) row[j] = t[j][i];
    Wino<M>::flt(row, o);
#pragma unroll
    for (int j = 0; j < A; ++j) up[(size_t)(A * i + j) * plane] = o[j];
  }
}

template <int M>
__global__ void wino_filter_kernel(const float* __restrict__ w, int K, int C, int flip, float* __restrict__ U) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * C) return;
  wino_filter_one<M>(w, K, C, flip, U, e);
}

// Every filter transform of a step in one launch.  jobs [n][6] int64: w pointer, U pointer, K, C, 2*m + flip, first
// workgroup of the job (ascending; job j owns the workgroups [first_j, first_{j+1})).
constexpr int kFilterJobFields = 6;
__global__ __launch_bounds__(256) void wino_filter_batch_kernel(const long long* __restrict__ jobs, int n_jobs) {
  int j = 0;
  while (j + 1 < n_jobs && (long long)blockIdx.x >= jobs[(j + 1) * kFilterJobFields + 5]) ++j;    // wave-uniform
  const long long* job = jobs + (size_t)j * kFilterJobFields;
  const float* w = reinterpret_cast<const float*>(job[0]);
  float* U = reinterpret_cast<float*>(job[1]);
  const int K = (int)job[2], C = (int)job[3], mf = (int)job[4];
  const long e = ((long)blockIdx.x - (long)job[5]) * 256 + threadIdx.x;
  if (e >= (long)K * C) return;
  if ((mf >> 1) == 2) wino_filter_one<2>(w, K, C, mf & 1, U, e);
  else wino_filter_one<4>(w, K, C, mf & 1, U, e);
}

// dw = G^T dU G per (k, c)
template <int M>
__global__ void wino_filter_grad_kernel(const float* __restrict__ dU, int K, int C, float* __restrict__ dw) {
  constexpr int A = Wino<M>::A;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * C) return;
  const size_t plane = (size_t)K * C;
  float a[A][3];        // a[j][i]: column j after the transform along rows
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float col[A];
#pragma unroll
    for (int i = 0; i < A; ++i) col[i] = dU[(size_t)(A * i + j) * plane + e];
    Wino<M>::fgrad(col, a[j]);
  }
  float* g = dw + (size_t)e * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float row[A], o[3];
#pragma unroll
    for (int j = 0; j < A; ++j) row[j] = a[j][i];
    Wino<M>::fgrad(row, o);
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i * 3 + j] = o[j];
  }
}

// Tiles per image side: ceil -- with m = 4 an even H or W that is not a multiple of 4 gets a half-empty last tile (RAG).
inline int tiles_of(int H, int m) { return (H + m - 1) / m; }
inline bool ragged(int m, int H, int W) { return m == 4 && ((H | W) & 3) != 0; }

int check_image(const char* fn, int m, int N, int C, int H, int W) {
  FPSG_REQUIRE(m == 2 || m == 4, FPSG_E_SHAPE, "%s: m must be 2 or 4 (got %d)", fn, m);
  FPSG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, FPSG_E_SHAPE,
               "%s: N,C positive and H,W positive and even (got %d,%d,%d,%d)", fn, N, C, H, W);
  const long P = (long)N * ((H + m - 1) / m) * ((W + m - 1) / m);
  FPSG_REQUIRE(C <= 65535 && (P + kWinoThreads - 1) / kWinoThreads < (1L << 31), FPSG_E_LIMIT,
               "%s: C=%d or tile count %ld beyond the grid limits", fn, C, P);
  return 0;
}

}  // namespace
}  // namespace fpsg

#define FPSG_WINO_IMAGE_LAUNCH(KERNEL, CH, ...)                                                              \
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);                                                                \
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), CH);                                          \
  if (m == 2) hipLaunchKernelGGL((KERNEL<2>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), __VA_ARGS__); \
  else hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), __VA_ARGS__)

extern "C" int fpsg_wino_input_transform(int m, const float* x, int N, int C, int H, int W, float* V, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_input_transform", m, N, C, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(V);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_input_transform: x must be 16-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_input_transform: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((Ps + kWinoThreads - 1) / kWinoThreads), C);
  if (m == 2) hipLaunchKernelGGL((wino_input_kernel<2, false>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_input_kernel<4, false, false, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr);
  else if ((size_t)36 * C * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_input_kernel<4, false, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr);
  else hipLaunchKernelGGL((wino_input_kernel<4, false>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr);
  return launch_status("fpsg_wino_input_transform");
}

extern "C" int fpsg_wino_input_transform_act(int m, const float* x, const float* chan, const float* pre_bias, int N,
                                             int C, int H, int W, float* V, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_input_transform_act", m, N, C, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(V); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(!misaligned4(pre_bias), FPSG_E_ALIGN, "fpsg_wino_input_transform_act: pre_bias not 4-byte aligned");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_input_transform_act: x must be 16-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_input_transform_act: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((Ps + kWinoThreads - 1) / kWinoThreads), C);
  if (m == 2) hipLaunchKernelGGL((wino_input_kernel<2, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, chan, pre_bias);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_input_kernel<4, true, false, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, chan, pre_bias);
  else if ((size_t)36 * C * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_input_kernel<4, true, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, chan, pre_bias);
  else hipLaunchKernelGGL((wino_input_kernel<4, true>), grid, dim3(kWinoThreads), 0, static_cast<hipStream_t>(stream), x, C, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, chan, pre_bias);
  return launch_status("fpsg_wino_input_transform_act");
}

extern "C" int fpsg_wino_output_transform(int m, const float* M, int N, int K, int H, int W, float* y, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_output_transform", m, N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(M); FPSG_REQUIRE_PTR(y);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_output_transform: y must be 16-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_output_transform: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (m == 2) hipLaunchKernelGGL((wino_output_kernel<2, false>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, nullptr, nullptr);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_output_kernel<4, false, false, false, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, nullptr, nullptr);
  else if ((size_t)36 * K * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_output_kernel<4, false, false, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, nullptr, nullptr);
  else hipLaunchKernelGGL((wino_output_kernel<4, false>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, nullptr, nullptr);
  return launch_status("fpsg_wino_output_transform");
}

extern "C" int fpsg_wino_stats_parts(int m, int N, int H, int W) {
  if ((m != 2 && m != 4) || N <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2) return 0;
  const long P = (long)N * fpsg::tiles_of(H, m) * fpsg::tiles_of(W, m);
  return (int)((P + fpsg::kWinoThreads - 1) / fpsg::kWinoThreads);
}

extern "C" int fpsg_wino_output_transform_stats(int m, const float* M, int N, int K, int H, int W, float* y,
                                                const float* bias, float* parts, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_output_transform_stats", m, N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(M); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(parts);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_output_transform_stats: y must be 16-byte aligned");
  FPSG_REQUIRE(!misaligned4(bias) && !misaligned4(parts), FPSG_E_ALIGN, "fpsg_wino_output_transform_stats: bias / parts not 4-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_output_transform_stats: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (m == 2) hipLaunchKernelGGL((wino_output_kernel<2, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, bias, parts);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_output_kernel<4, true, false, false, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, bias, parts);
  else if ((size_t)36 * K * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_output_kernel<4, true, false, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, bias, parts);
  else hipLaunchKernelGGL((wino_output_kernel<4, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, bias, parts);
  return launch_status("fpsg_wino_output_transform_stats");
}

extern "C" int fpsg_wino_output_transform_bwd_stats(int m, const float* M, int N, int K, int H, int W, float* y,
                                                    const float* xpre, const float* pre_bias, const float* chan,
                                                    float* parts, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_output_transform_bwd_stats", m, N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(M); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(parts); FPSG_REQUIRE_PTR(xpre); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(xpre)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_wino_output_transform_bwd_stats: y and xpre must be 16-byte aligned");
  FPSG_REQUIRE(!misaligned4(pre_bias) && !misaligned4(parts) && !misaligned4(chan), FPSG_E_ALIGN,
               "fpsg_wino_output_transform_bwd_stats: pre_bias / chan / parts not 4-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_output_transform_bwd_stats: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((P + kWinoThreads - 1) / kWinoThreads), K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (m == 2) hipLaunchKernelGGL((wino_output_kernel<2, true, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, pre_bias, parts, xpre, chan);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_output_kernel<4, true, true, false, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, pre_bias, parts, xpre, chan);
  else if ((size_t)36 * K * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_output_kernel<4, true, true, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, pre_bias, parts, xpre, chan);
  else hipLaunchKernelGGL((wino_output_kernel<4, true, true>), grid, dim3(kWinoThreads), 0, hs, M, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, y, pre_bias, parts, xpre, chan);
  return launch_status("fpsg_wino_output_transform_bwd_stats");
}

extern "C" int fpsg_wino_grad_output_transform(int m, const float* dy, int N, int K, int H, int W, float* dM, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_grad_output_transform", m, N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(dM);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_grad_output_transform: dy must be 16-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_grad_output_transform: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((Ps + kWinoThreads - 1) / kWinoThreads), K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (m == 2) hipLaunchKernelGGL((wino_grad_output_kernel<2>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, dM);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_grad_output_kernel<4, false, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, dM);
  else if ((size_t)36 * K * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_grad_output_kernel<4, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, dM);
  else hipLaunchKernelGGL((wino_grad_output_kernel<4>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, dM);
  return launch_status("fpsg_wino_grad_output_transform");
}

extern "C" int fpsg_wino_grad_transforms(int m, const float* dy, int N, int K, int H, int W, float* V, float* dM, long ldp,
    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_image("fpsg_wino_grad_transforms", m, N, K, H, W);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(V); FPSG_REQUIRE_PTR(dM);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, FPSG_E_ALIGN, "fpsg_wino_grad_transforms: dy must be 16-byte aligned");
  const long P = (long)N * tiles_of(H, m) * tiles_of(W, m);
  const long Ps = ldp ? ldp : P;
  FPSG_REQUIRE(Ps >= P, FPSG_E_SHAPE, "fpsg_wino_grad_transforms: row stride %ld must be 0 or at least the %ld tiles", ldp, P);
  dim3 grid((unsigned)((Ps + kWinoThreads - 1) / kWinoThreads), K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (m == 2) hipLaunchKernelGGL((wino_input_kernel<2, false, false, false, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr, dM);
  else if (ragged(m, H, W)) hipLaunchKernelGGL((wino_input_kernel<4, false, false, true, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr, dM);
  else if ((size_t)36 * K * Ps * sizeof(float) > kStreamBytes) hipLaunchKernelGGL((wino_input_kernel<4, false, true, false, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr, dM);
  else hipLaunchKernelGGL((wino_input_kernel<4, false, false, false, true>), grid, dim3(kWinoThreads), 0, hs, dy, K, H, W, tiles_of(H, m), tiles_of(W, m), P, Ps, V, nullptr, nullptr, dM);
  return launch_status("fpsg_wino_grad_transforms");
}

extern "C" int fpsg_wino_filter_transform(int m, const float* w, int K, int C, int flip_transpose, float* U,
                                          fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(m == 2 || m == 4, FPSG_E_SHAPE, "fpsg_wino_filter_transform: m must be 2 or 4 (got %d)", m);
  FPSG_REQUIRE(K > 0 && C > 0, FPSG_E_SHAPE, "fpsg_wino_filter_transform: K,C must be positive (got %d,%d)", K, C);
  FPSG_REQUIRE_PTR(w); FPSG_REQUIRE_PTR(U);
  const long n = (long)K * C;
  dim3 grid((unsigned)((n + 255) / 256));
  if (m == 2) hipLaunchKernelGGL(wino_filter_kernel<2>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, K, C, flip_transpose, U);
  else hipLaunchKernelGGL(wino_filter_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, K, C, flip_transpose, U);
  return launch_status("fpsg_wino_filter_transform");
}

extern "C" int fpsg_wino_filter_transform_batch(const int64_t* jobs, int n_jobs, long total_blocks, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(n_jobs > 0 && total_blocks > 0 && total_blocks < (1L << 31), FPSG_E_SHAPE,
               "fpsg_wino_filter_transform_batch: n_jobs and total_blocks must be positive (got %d, %ld)", n_jobs, total_blocks);
  FPSG_REQUIRE_PTR(jobs);
  hipLaunchKernelGGL(wino_filter_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(jobs), n_jobs);
  return launch_status("fpsg_wino_filter_transform_batch");
}

extern "C" int fpsg_wino_filter_grad_transform(int m, const float* dU, int K, int C, float* dw, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(m == 2 || m == 4, FPSG_E_SHAPE, "fpsg_wino_filter_grad_transform: m must be 2 or 4 (got %d)", m);
  FPSG_REQUIRE(K > 0 && C > 0, FPSG_E_SHAPE, "fpsg_wino_filter_grad_transform: K,C must be positive (got %d,%d)", K, C);
  FPSG_REQUIRE_PTR(dU); FPSG_REQUIRE_PTR(dw);
  const long n = (long)K * C;
  dim3 grid((unsigned)((n + 255) / 256));
  if (m == 2) hipLaunchKernelGGL(wino_filter_grad_kernel<2>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), dU, K, C, dw);
  else hipLaunchKernelGGL(wino_filter_grad_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), dU, K, C, dw);
  return launch_status("fpsg_wino_filter_grad_transform");
}
