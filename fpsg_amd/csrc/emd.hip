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
//     column consumption, assignment); each sweep is ONE launch of small workgroups: a wave owns
//     two points and its lanes stride over the other cloud (see emd_sweep_kernel);
//   * the per-pair work is packed FP32 VALU + one v_exp_f32 (and v_sqrt/v_rcp in the assignment
//     sweep); distances use the same fma form as K1.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kEmdWaves = 4;     // waves per workgroup
constexpr int kEmdOwners = 2;    // owner points per wave
constexpr int kEmdThreads = 64 * kEmdWaves;

enum EmdMode { kRatioL = 0, kRatioR = 1, kMatch = 2, kGradOther = 3 };

struct EmdArgs {
  const float* own;      // [B, No, 3] owner cloud
  const float* oth;      // [B, Nt, 3] cloud that is swept
  const float* oth_w;    // [B, Nt] weight of every swept point (remainR | ratioL | ratioR)
  float* own_remain;     // [B, No]  remainL (kRatioL: read, kMatch: updated) | remainR (kRatioR: updated)
  float* own_ratio;      // [B, No]  ratioL (kRatioL: written, kMatch: read) | ratioR (kRatioR: written; kGradOther: read)
  float* own_cost;       // [B, No]  kMatch: per-owner transport cost, accumulated over levels
  float* own_grad;       // [B, No, 3] or null: kMatch / kGradOther gradient accumulators
  int No, Nt;
  float level;
};

// One wave = kEmdOwners owner points against the whole swept cloud: lane l takes the candidates
// l, l+64, l+128, ... straight from global memory (a wave reads contiguous 768-byte runs of the AoS
// cloud; the cloud stays in L2 -- no LDS staging, no barrier), two candidates per packed FP32
// instruction, and the per-owner sums are folded over the 64 lanes by a fixed DPP tree
// (deterministic, no float atomics).  A sweep of one 2048-point pair is 256 workgroups of 4 waves,
// so even a single pair fills the chip; per pair: distance (3 packed-half instructions) + v_exp_f32 +
// weights (+ v_sqrt / v_rcp and 4 FMAs in the assignment sweeps): VALU + transcendental bound.
template <int MODE, bool GRAD>
__global__ __launch_bounds__(kEmdThreads) void emd_sweep_kernel(EmdArgs a) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int o0 = (blockIdx.x * kEmdWaves + wave) * kEmdOwners;
  if (o0 >= a.No) return;                               // whole wave; no barrier in this kernel
  const float* __restrict__ own = a.own + (size_t)b * a.No * 3;
  const float* __restrict__ oth = a.oth + (size_t)b * a.Nt * 3;
  const float* __restrict__ ow = a.oth_w + (size_t)b * a.Nt;
  v2f px[kEmdOwners], py[kEmdOwners], pz[kEmdOwners];
  float fac[kEmdOwners];
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) {
    const int oc = (o0 + r) < a.No ? (o0 + r) : a.No - 1;
    const float x = own[3 * oc], y = own[3 * oc + 1], z = own[3 * oc + 2];
    px[r] = v2f{x, x}; py[r] = v2f{y, y}; pz[r] = v2f{z, z};
    // factor applied to every pair weight of this owner (assignment sweeps only)
    fac[r] = (MODE == kMatch || MODE == kGradOther) ? a.own_ratio[(size_t)b * a.No + oc] : 1.0f;
  }
  v2f s[kEmdOwners], c[kEmdOwners], gx[kEmdOwners], gy[kEmdOwners], gz[kEmdOwners];
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) { s[r] = v2f{0, 0}; c[r] = v2f{0, 0}; gx[r] = v2f{0, 0}; gy[r] = v2f{0, 0}; gz[r] = v2f{0, 0}; }
  const v2f lvl = {a.level, a.level};

#pragma unroll 4
  for (int l0 = lane; l0 < a.Nt; l0 += 128) {           // candidates l0 and l0 + 64 as one packed pair; 4 pairs of loads in flight
    const int l1 = l0 + 64;
    const bool in1 = l1 < a.Nt;
    const int l1c = in1 ? l1 : l0;
    const v2f cx = {oth[3 * l0], oth[3 * l1c]}, cy = {oth[3 * l0 + 1], oth[3 * l1c + 1]},
              cz = {oth[3 * l0 + 2], oth[3 * l1c + 2]};
    const v2f cw = {ow[l0], in1 ? ow[l1] : 0.0f};       // a missing candidate weighs nothing
#pragma unroll
    for (int r = 0; r < kEmdOwners; ++r) {
      const v2f dx = cx - px[r], dy = cy - py[r], dz = cz - pz[r];
      const v2f d2 = fma_rn(dz, dz, fma_rn(dy, dy, dx * dx));
      const v2f e = lvl * d2;
      const v2f f2 = {fac[r], fac[r]};
      const v2f w = v2f{__expf(e.x), __expf(e.y)} * cw * f2;
      s[r] += w;
      if (MODE == kMatch || MODE == kGradOther) {
        const v2f dist = {__builtin_sqrtf(d2.x), __builtin_sqrtf(d2.y)};
        if (MODE == kMatch) c[r] = fma_rn(w, dist, c[r]);
        if (GRAD) {
          const v2f f = {w.x / __builtin_fmaxf(dist.x, 1e-20f), w.y / __builtin_fmaxf(dist.y, 1e-20f)};
          gx[r] = fma_rn(f, -dx, gx[r]);   // owner - other
          gy[r] = fma_rn(f, -dy, gy[r]);
          gz[r] = fma_rn(f, -dz, gz[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) {
    const float st = wave_sum(s[r].x + s[r].y);
    float ct = 0.0f, gxt = 0.0f, gyt = 0.0f, gzt = 0.0f;
    if (MODE == kMatch) ct = wave_sum(c[r].x + c[r].y);
    if (GRAD) {
      gxt = wave_sum(gx[r].x + gx[r].y);
      gyt = wave_sum(gy[r].x + gy[r].y);
      gzt = wave_sum(gz[r].x + gz[r].y);
    }
    const int o = o0 + r;
    if (lane != 0 || o >= a.No) continue;
    const size_t oi = (size_t)b * a.No + o;
    if (MODE == kRatioL) {
      a.own_ratio[oi] = a.own_remain[oi] / (st + 1e-9f);
    } else if (MODE == kRatioR) {
      const float rem = a.own_remain[oi];
      const float sumr = st * rem;
      const float consumption = __builtin_fminf(rem / (sumr + 1e-9f), 1.0f);
      a.own_ratio[oi] = consumption * rem;
      a.own_remain[oi] = __builtin_fmaxf(0.0f, rem - sumr);
    } else if (MODE == kMatch) {
      a.own_cost[oi] += ct;
      a.own_remain[oi] = __builtin_fmaxf(0.0f, a.own_remain[oi] - st);
    }
    if (GRAD && (MODE == kMatch || MODE == kGradOther)) {
      a.own_grad[3 * oi] += gxt;
      a.own_grad[3 * oi + 1] += gyt;
      a.own_grad[3 * oi + 2] += gzt;
    }
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
  dim3 grid((a.No + kEmdWaves * kEmdOwners - 1) / (kEmdWaves * kEmdOwners), B);
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
