// emd.hip -- K2: approximate-assignment EMD solver (auction-style soft matching over 10
// temperature levels, Fan/Su/Guibas "approxmatch" + "matchcost") for gfx950.
// Stands where the reference calls neuralnet_pytorch.metrics.emd_loss
// (src/models/utils.py:12-13 <- src/models/few_shot.py:168).  Specification: see
// oracle_emd_approx in oracle/fpsg_oracle.c (parity with neuralnet_pytorch is UNPINNED).
//
// MI355X-first structure (not the one-block-per-cloud loop of the CUDA original):
//   * the N x M soft-assignment matrix is never stored: cost and gradients are accumulated
//     level by level, so memory is O(N+M) instead of 16 MB per 2048-point pair;
//   * every level is three dependent sweeps over the N x M pairs (row normalisers,
//     column consumption, assignment); each sweep is ONE launch spread over
//     B x ceil(owners/64) workgroups of 16 waves -- the 64 owner points of a workgroup sit
//     one per lane, the other cloud (+ its per-point weight) is staged in LDS as SoA and
//     split 16 ways across the waves; partial sums are merged in LDS in fixed wave order,
//     so results are deterministic (no float atomics);
//   * the per-pair work is FP32 VALU + one v_exp_f32 (and v_sqrt/v_rcp in the assignment
//     sweep); distances use the same fma form as K1.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kEmdWaves = 16;
constexpr int kEmdThreads = 64 * kEmdWaves;
constexpr int kEmdTile = 2048;  // other-cloud points staged per LDS pass (4 floats each = 32 KiB)

enum EmdMode { kRatioL = 0, kRatioR = 1, kMatch = 2, kGradOther = 3 };

struct EmdArgs {
  const float* own;      // [B, No, 3] owner cloud (one point per lane)
  const float* oth;      // [B, Nt, 3] cloud that is swept
  const float* oth_w;    // [B, Nt] weight of every swept point (remainR | ratioL | ratioR)
  float* own_remain;     // [B, No]  remainL (kRatioL: read, kMatch: updated) | remainR (kRatioR: updated)
  float* own_ratio;      // [B, No]  ratioL (kRatioL: written, kMatch: read) | ratioR (kRatioR: written; kGradOther: read)
  float* own_cost;       // [B, No]  kMatch: per-owner transport cost, accumulated over levels
  float* own_grad;       // [B, No, 3] or null: kMatch / kGradOther gradient accumulators
  int No, Nt;
  float level;
};

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

template <int MODE, bool GRAD>
__global__ __launch_bounds__(kEmdThreads) void emd_sweep_kernel(EmdArgs a) {
  __shared__ __attribute__((aligned(16))) float sx[kEmdTile], sy[kEmdTile], sz[kEmdTile], sw[kEmdTile];
  __shared__ float part[kEmdWaves][5][64];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = blockIdx.x * 64 + lane;
  const bool live = o < a.No;
  const int oc = live ? o : a.No - 1;
  const float* __restrict__ own = a.own + (size_t)b * a.No * 3;
  const float* __restrict__ oth = a.oth + (size_t)b * a.Nt * 3;
  const float* __restrict__ ow = a.oth_w + (size_t)b * a.Nt;
  const float px = own[3 * oc], py = own[3 * oc + 1], pz = own[3 * oc + 2];
  // factor applied to every pair weight of this owner (assignment sweeps only)
  float own_fac = 1.0f;
  if (MODE == kMatch || MODE == kGradOther) own_fac = a.own_ratio[(size_t)b * a.No + oc];

  float s = 0.0f, c = 0.0f, gx = 0.0f, gy = 0.0f, gz = 0.0f;
  for (int t0 = 0; t0 < a.Nt; t0 += kEmdTile) {
    if (t0) __syncthreads();
    const int cnt = (a.Nt - t0) < kEmdTile ? (a.Nt - t0) : kEmdTile;
    for (int e = tid; e < cnt; e += kEmdThreads) {
      sx[e] = oth[3 * (t0 + e)];
      sy[e] = oth[3 * (t0 + e) + 1];
      sz[e] = oth[3 * (t0 + e) + 2];
      sw[e] = ow[t0 + e];
    }
    __syncthreads();
    const int per = (cnt + kEmdWaves - 1) / kEmdWaves;
    const int lo = wave * per;
    const int hi = (lo + per) < cnt ? (lo + per) : cnt;
    for (int l = lo; l < hi; ++l) {
      const float dx = sx[l] - px, dy = sy[l] - py, dz = sz[l] - pz;   // LDS broadcast reads
      const float d2 = fma_rn(dz, dz, fma_rn(dy, dy, dx * dx));
      const float w = fast_exp(a.level * d2) * sw[l] * own_fac;
      s += w;
      if (MODE == kMatch || MODE == kGradOther) {
        const float dist = __builtin_sqrtf(d2);
        if (MODE == kMatch) c = fma_rn(w, dist, c);
        if (GRAD) {
          const float f = w / __builtin_fmaxf(dist, 1e-20f);
          gx = fma_rn(f, -dx, gx);   // owner - other
          gy = fma_rn(f, -dy, gy);
          gz = fma_rn(f, -dz, gz);
        }
      }
    }
  }
  part[wave][0][lane] = s;
  if (MODE == kMatch) part[wave][1][lane] = c;
  if (GRAD) { part[wave][2][lane] = gx; part[wave][3][lane] = gy; part[wave][4][lane] = gz; }
  __syncthreads();
  if (wave != 0 || !live) return;
  s = 0.0f; c = 0.0f; gx = 0.0f; gy = 0.0f; gz = 0.0f;
#pragma unroll
  for (int w = 0; w < kEmdWaves; ++w) {   // fixed order: deterministic
    s += part[w][0][lane];
    if (MODE == kMatch) c += part[w][1][lane];
    if (GRAD) { gx += part[w][2][lane]; gy += part[w][3][lane]; gz += part[w][4][lane]; }
  }
  const size_t oi = (size_t)b * a.No + o;
  if (MODE == kRatioL) {
    a.own_ratio[oi] = a.own_remain[oi] / (s + 1e-9f);
  } else if (MODE == kRatioR) {
    const float rem = a.own_remain[oi];
    const float sumr = s * rem;
    const float consumption = __builtin_fminf(rem / (sumr + 1e-9f), 1.0f);
    a.own_ratio[oi] = consumption * rem;
    a.own_remain[oi] = __builtin_fmaxf(0.0f, rem - sumr);
  } else if (MODE == kMatch) {
    a.own_cost[oi] += c;
    a.own_remain[oi] = __builtin_fmaxf(0.0f, a.own_remain[oi] - s);
  }
  if (GRAD && (MODE == kMatch || MODE == kGradOther)) {
    a.own_grad[3 * oi] += gx;
    a.own_grad[3 * oi + 1] += gy;
    a.own_grad[3 * oi + 2] += gz;
  }
}

__global__ void emd_init_kernel(float* remainL, float* costrow, float* remainR, float* g1, float* g2,
                                int B, int N, int M, float multiL, float multiR) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)B * N) { remainL[t] = multiL; costrow[t] = 0.0f; }
  if (t < (size_t)B * M) remainR[t] = multiR;
  if (g1 && t < (size_t)B * N * 3) g1[t] = 0.0f;
  if (g2 && t < (size_t)B * M * 3) g2[t] = 0.0f;
}

// cost[b] = sum_k costrow[b,k]: one workgroup per cloud, fixed-shape tree (deterministic)
__global__ __launch_bounds__(256) void emd_cost_kernel(const float* __restrict__ costrow, int N,
                                                       float* __restrict__ cost) {
  __shared__ float red[256];
  const float* r = costrow + (size_t)blockIdx.x * N;
  float acc = 0.0f;
  for (int k = threadIdx.x; k < N; k += 256) acc += r[k];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) cost[blockIdx.x] = red[0];
}

template <int MODE, bool GRAD>
int sweep(const EmdArgs& a, int B, hipStream_t s, const char* what) {
  dim3 grid((a.No + 63) / 64, B);
  hipLaunchKernelGGL((emd_sweep_kernel<MODE, GRAD>), grid, dim3(kEmdThreads), 0, s, a);
  return launch_status(what);
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_emd_workspace_floats(int B, int N, int M) {
  return (size_t)B * (3 * (size_t)N + 2 * (size_t)M);
}

extern "C" int fpsg_emd_approx(const float* xyz1, const float* xyz2, int B, int N, int M,
                               float* cost, float* gxyz1, float* gxyz2, float* ws,
                               fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE,
               "fpsg_emd_approx: B,N,M must be positive (got %d,%d,%d)", B, N, M);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_emd_approx: B=%d exceeds 65535", B);
  FPSG_REQUIRE_PTR(xyz1); FPSG_REQUIRE_PTR(xyz2); FPSG_REQUIRE_PTR(cost); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(!misaligned4(gxyz1) && !misaligned4(gxyz2), FPSG_E_ALIGN,
               "fpsg_emd_approx: gradient buffers must be 4-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* remainL = ws;
  float* ratioL = remainL + (size_t)B * N;
  float* costrow = ratioL + (size_t)B * N;
  float* remainR = costrow + (size_t)B * N;
  float* ratioR = remainR + (size_t)B * M;
  const float multiL = (M > N) ? (float)(M / N) : 1.0f;
  const float multiR = (N >= M) ? (float)(N / M) : 1.0f;
  {
    const size_t n = (size_t)B * 3 * (size_t)(N > M ? N : M);
    hipLaunchKernelGGL(emd_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, remainL,
                       costrow, remainR, gxyz1, gxyz2, B, N, M, multiL, multiR);
    int rc = launch_status("fpsg_emd_approx(init)");
    if (rc) return rc;
  }
  for (int j = 7; j >= -2; --j) {
    float level = -1.0f;
    for (int p = 0; p < (j < 0 ? -j : j); ++p) level = (j < 0) ? level * 0.25f : level * 4.0f;
    if (j == -2) level = 0.0f;   // -4^j for j = 7..-1, then 0
    EmdArgs a{};
    a.level = level;
    int rc;
    // row normalisers: owners = cloud 1, sweep cloud 2 weighted by remainR
    a.own = xyz1; a.oth = xyz2; a.oth_w = remainR; a.own_remain = remainL; a.own_ratio = ratioL;
    a.own_cost = nullptr; a.own_grad = nullptr; a.No = N; a.Nt = M;
    if ((rc = sweep<kRatioL, false>(a, B, s, "fpsg_emd_approx(ratioL)"))) return rc;
    // column consumption: owners = cloud 2, sweep cloud 1 weighted by ratioL
    a.own = xyz2; a.oth = xyz1; a.oth_w = ratioL; a.own_remain = remainR; a.own_ratio = ratioR;
    a.No = M; a.Nt = N;
    if ((rc = sweep<kRatioR, false>(a, B, s, "fpsg_emd_approx(ratioR)"))) return rc;
    // assignment: owners = cloud 1 (factor ratioL), sweep cloud 2 weighted by ratioR
    a.own = xyz1; a.oth = xyz2; a.oth_w = ratioR; a.own_remain = remainL; a.own_ratio = ratioL;
    a.own_cost = costrow; a.own_grad = gxyz1; a.No = N; a.Nt = M;
    rc = gxyz1 ? sweep<kMatch, true>(a, B, s, "fpsg_emd_approx(match+grad)")
               : sweep<kMatch, false>(a, B, s, "fpsg_emd_approx(match)");
    if (rc) return rc;
    if (gxyz2) {  // same weights seen from cloud 2
      a.own = xyz2; a.oth = xyz1; a.oth_w = ratioL; a.own_remain = nullptr; a.own_ratio = ratioR;
      a.own_cost = nullptr; a.own_grad = gxyz2; a.No = M; a.Nt = N;
      if ((rc = sweep<kGradOther, true>(a, B, s, "fpsg_emd_approx(grad2)"))) return rc;
    }
  }
  hipLaunchKernelGGL(emd_cost_kernel, dim3(B), dim3(256), 0, s, costrow, N, cost);
  return launch_status("fpsg_emd_approx(cost)");
}
