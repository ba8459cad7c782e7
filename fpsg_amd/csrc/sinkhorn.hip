// sinkhorn.hip -- K2b: the soft-min operator of the Sinkhorn loop, for gfx950.
// The reference's evaluation "EMD" is neuralnet_pytorch.metrics.emd_loss(sinkhorn=True)
// (src/models/utils.py:12-13), i.e. geomloss.SamplesLoss() -- a debiased Sinkhorn divergence
// with cost |x-y|^2/2 whose whole cost is the repeated evaluation of
//     out[b,i] = -eps * log sum_j exp( h[b,j] - |x_i - y_j|^2 / (2 eps) )
// geomloss' "tensorized" backend materialises the [B,N,M] cost matrices (16 MB per 2048-point
// pair, several of them); here nothing of size N x M exists: one launch per soft-min, owners
// one per lane, the summed cloud (+ its log-weights) staged in LDS as SoA and split over the
// 16 waves of a workgroup, a running (max, sum) pair per lane in base-2 exponent space
// (v_exp_f32 / v_log_f32 are base 2), merged across waves in a fixed order => deterministic.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kSmWaves = 16;
constexpr int kSmThreads = 64 * kSmWaves;
constexpr int kSmTile = 2048;
constexpr int kSmChunk = 8;      // candidates per rescale of the running sum
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__global__ __launch_bounds__(kSmThreads) void softmin_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ h, int N, int M,
                                                             float k2, float eps,
                                                             float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sx[kSmTile], sy[kSmTile], sz[kSmTile], sh[kSmTile];
  __shared__ float pm[kSmWaves][64], ps[kSmWaves][64];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = blockIdx.x * 64 + lane;
  const int oc = o < N ? o : N - 1;
  const float* __restrict__ xb = x + (size_t)b * N * 3;
  const float* __restrict__ yb = y + (size_t)b * M * 3;
  const float* __restrict__ hb = h + (size_t)b * M;
  const float px = xb[3 * oc], py = xb[3 * oc + 1], pz = xb[3 * oc + 2];
  float m = -__builtin_inff(), s = 0.0f;    // running max (base-2 exponent) and scaled sum
  for (int t0 = 0; t0 < M; t0 += kSmTile) {
    if (t0) __syncthreads();
    const int cnt = (M - t0) < kSmTile ? (M - t0) : kSmTile;
    const int padded = (cnt + kSmChunk - 1) / kSmChunk * kSmChunk;
    for (int e = tid; e < padded; e += kSmThreads) {
      const bool in = e < cnt;
      sx[e] = in ? yb[3 * (t0 + e)] : 0.0f;
      sy[e] = in ? yb[3 * (t0 + e) + 1] : 0.0f;
      sz[e] = in ? yb[3 * (t0 + e) + 2] : 0.0f;
      sh[e] = in ? hb[t0 + e] * kLog2e : -__builtin_inff();   // padding never contributes
    }
    __syncthreads();
    const int chunks = padded / kSmChunk;
    const int per = (chunks + kSmWaves - 1) / kSmWaves;
    const int lo = wave * per;
    const int hi = (lo + per) < chunks ? (lo + per) : chunks;
    for (int c = lo; c < hi; ++c) {
      float v[kSmChunk];
      float cmax = -__builtin_inff();
#pragma unroll
      for (int u = 0; u < kSmChunk; ++u) {
        const int l = c * kSmChunk + u;
        const float dx = sx[l] - px, dy = sy[l] - py, dz = sz[l] - pz;   // LDS broadcast reads
        const float d2 = fma_rn(dz, dz, fma_rn(dy, dy, dx * dx));
        v[u] = fma_rn(-k2, d2, sh[l]);
        cmax = __builtin_fmaxf(cmax, v[u]);
      }
      const float mn = __builtin_fmaxf(m, cmax);
      if (mn > -__builtin_inff()) {          // else: nothing finite yet, keep (m, s) = (-inf, 0)
        float acc = s * __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
        for (int u = 0; u < kSmChunk; ++u) acc += __builtin_amdgcn_exp2f(v[u] - mn);
        s = acc;
        m = mn;
      }
    }
  }
  pm[wave][lane] = m;
  ps[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || o >= N) return;
  float mm = pm[0][lane];
#pragma unroll
  for (int w = 1; w < kSmWaves; ++w) mm = __builtin_fmaxf(mm, pm[w][lane]);
  float ss = 0.0f;
#pragma unroll
  for (int w = 0; w < kSmWaves; ++w)      // fixed order
    ss += (pm[w][lane] > -__builtin_inff()) ? ps[w][lane] * __builtin_amdgcn_exp2f(pm[w][lane] - mm) : 0.0f;
  out[(size_t)b * N + o] = -eps * kLn2 * (mm + __builtin_amdgcn_logf(ss));
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_softmin(const float* x, const float* y, const float* h, int B, int N, int M,
                            float eps, float* out, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE,
               "fpsg_softmin: B,N,M must be positive (got %d,%d,%d)", B, N, M);
  FPSG_REQUIRE(eps > 0.0f, FPSG_E_SHAPE, "fpsg_softmin: eps must be positive (got %g)", (double)eps);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_softmin: B=%d exceeds 65535", B);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(h); FPSG_REQUIRE_PTR(out);
  const float k2 = 0.5f / eps * 1.4426950408889634f;   // |x-y|^2/(2 eps) in base-2 exponent units
  hipLaunchKernelGGL(softmin_kernel, dim3((N + 63) / 64, B), dim3(kSmThreads), 0,
                     static_cast<hipStream_t>(stream), x, y, h, N, M, k2, eps, out);
  return launch_status("fpsg_softmin");
}
