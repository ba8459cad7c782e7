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
//     sweep); distances use the same fma form as K1;
//   * round 4, forward-only calls (evaluation): a level's assignment sweep and the NEXT level's row-normaliser sweep
//     have the same owners and sweep the same cloud, so they are ONE launch (kMatchRatio*: 21 sweeps instead of 30
//     per call, the distance evaluated once for both); the levels are a factor 4 apart, so the assignment's
//     exp(4 l d^2) is the square of the square of the normaliser's exp(l d^2) -- two packed multiplications instead of
//     a second quarter-rate v_exp_f32 (relative difference ~4e-7 per weight; the cost is checked against the CPU
//     restatement, which calls expf per level, to 5e-5 -- tests/test_emd_gpu.py; measured 5e-7 at 2048 points and
//     2.6e-5 on a 101 x 67 pair, where the auction's clamps amplify the per-weight difference, so the automatic choice
//     keeps the separate sweeps (expf per level: <= 2e-6) for pairs below 256 x 256); a wave owns FOUR points (half the vector-memory instructions
//     per pair: the sweeps were close to the CU's 64 B/clk load path).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kEmdWaves = 4;     // waves per workgroup
constexpr int kEmdOwners = 2;    // owner points per wave (gradient sweeps; the forward-only sweeps take 4)
constexpr int kEmdThreads = 64 * kEmdWaves;

enum EmdMode { kRatioL = 0, kRatioR = 1, kMatch = 2, kGradOther = 3,
               kMatchRatioQ = 4,    // kMatch at `level` + the next level's kRatioL at `level2` = level / 4
               kMatchRatioZ = 5 };  // the same with level2 = 0 (the last level: exp(0 d^2) = 1)

inline int emd_pad4(int n) { return (n + 3) & ~3; }

struct EmdArgs {
  const float* own;      // [B, 3, Nop] owner cloud, coordinate-major (SoA copy in the workspace)
  const float* oth;      // [B, 3, Ntp] cloud that is swept, coordinate-major
  const float* oth_w;    // [B, Ntp] weight of every swept point (remainR | ratioL | ratioR); 0 in the padding
  float* own_remain;     // [B, Nop]  remainL (kRatioL: read, kMatch: updated) | remainR (kRatioR: updated)
  float* own_ratio;      // [B, Nop]  ratioL (kRatioL: written, kMatch: read) | ratioR (kRatioR: written; kGradOther: read)
  float* own_cost;       // [B, Nop]  kMatch: per-owner transport cost, accumulated over levels
  float* own_grad;       // [B, No, 3] or null: kMatch / kGradOther gradient accumulators (the caller's layout)
  int No, Nt;            // points; the workspace rows are padded to multiples of 4 (Nop, Ntp)
  float level;
  const float* oth_w2;   // kMatchRatio*: [B, Ntp] the swept points' weights of the normaliser half (remainR)
  float level2;          // kMatchRatioQ: level / 4
};

// One wave = kEmdOwners owner points against the whole swept cloud.  The clouds are read from coordinate-major copies
// (x[], y[], z[] rows, made once per call by emd_init_kernel): lane l takes the FOUR candidates 4(l + 64 t) .. +3 of
// trip t with one 16-byte load per coordinate and one for their weights -- 4 load instructions per 8 pairs (the
// point-major cloud needed 8 dword loads per 4 pairs, and their waits were the sweep's critical path) -- the next trip's
// vectors are on their way while this one is evaluated; two candidates per packed FP32 instruction; the per-owner
// sums are folded over the 64 lanes by the row_shr / row_bcast DPP tree (deterministic, no float atomics, no LDS).
// A sweep of one 2048-point pair is 256 workgroups of 4 waves, so even a single pair fills the chip; per pair: distance
// (3 packed-half instructions) + v_exp_f32 + weights (+ v_sqrt / v_rcp and 4 FMAs in the assignment sweeps).
template <int MODE, bool GRAD, int OWN>
__global__ __launch_bounds__(kEmdThreads) void emd_sweep_kernel(EmdArgs a) {
  constexpr int kEmdOwners = OWN;                       // (shadows the default: this instance's owners per wave)
  constexpr bool kMerged = MODE == kMatchRatioQ || MODE == kMatchRatioZ;
  constexpr bool kAssign = MODE == kMatch || MODE == kGradOther || kMerged;
  static_assert(!(GRAD && kMerged), "the merged sweeps are forward-only");
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int o0 = (blockIdx.x * kEmdWaves + wave) * kEmdOwners;
  if (o0 >= a.No) return;                               // whole wave; no barrier in this kernel
  const int Nop = (a.No + 3) & ~3, Ntp = (a.Nt + 3) & ~3;
  const float* __restrict__ own = a.own + (size_t)b * 3 * Nop;
  const float* __restrict__ othx = a.oth + (size_t)b * 3 * Ntp;
  const float* __restrict__ othy = othx + Ntp;
  const float* __restrict__ othz = othy + Ntp;
  const float* __restrict__ ow = a.oth_w + (size_t)b * Ntp;
  const float* __restrict__ ow2 = kMerged ? a.oth_w2 + (size_t)b * Ntp : ow;
  v2f px[kEmdOwners], py[kEmdOwners], pz[kEmdOwners];
  float fac[kEmdOwners];
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) {
    const int oc = (o0 + r) < a.No ? (o0 + r) : a.No - 1;
    const float x = own[oc], y = own[Nop + oc], z = own[2 * Nop + oc];
    px[r] = v2f{x, x}; py[r] = v2f{y, y}; pz[r] = v2f{z, z};
    // factor applied to every pair weight of this owner (assignment sweeps only)
    fac[r] = kAssign ? a.own_ratio[(size_t)b * Nop + oc] : 1.0f;
  }
  v2f s[kEmdOwners], c[kEmdOwners], gx[kEmdOwners], gy[kEmdOwners], gz[kEmdOwners], s2[kEmdOwners];
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) {
    s[r] = v2f{0, 0}; c[r] = v2f{0, 0}; gx[r] = v2f{0, 0}; gy[r] = v2f{0, 0}; gz[r] = v2f{0, 0}; s2[r] = v2f{0, 0};
  }
  // exp(level d^2) = 2^((level log2 e) d^2): v_exp_f32 itself (base 2), no range / denormal scaffolding around it -- with
  // -fno-fast-math __expf and sqrtf expand to ~15 compare / select / refine instructions per call, 120 of the merged
  // sweep's 250 loop instructions; a weight below 2^-126 or a distance below 1e-19 is nothing this sum can see
  constexpr float kLog2e = 1.44269504088896340736f;
  const v2f lvl = {a.level * kLog2e, a.level * kLog2e};
  const v2f lvl2 = {a.level2 * kLog2e, a.level2 * kLog2e};

  const int ngroups = Ntp >> 2;                         // groups of four candidates
  auto fetch = [&](int g, v4f& X, v4f& Y, v4f& Z, v4f& Wt, v4f& W2) {
    const bool in = g < ngroups;
    const int gc = in ? g : 0;                          // a lane beyond the cloud re-reads group 0 with weight 0
    X = reinterpret_cast<const v4f*>(othx)[gc];
    Y = reinterpret_cast<const v4f*>(othy)[gc];
    Z = reinterpret_cast<const v4f*>(othz)[gc];
    Wt = reinterpret_cast<const v4f*>(ow)[gc];
    if (kMerged) W2 = reinterpret_cast<const v4f*>(ow2)[gc];
    if (!in) { Wt = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; W2 = Wt; }
  };
  v4f X, Y, Z, Wt, W2 = {0.0f, 0.0f, 0.0f, 0.0f};
  fetch(lane, X, Y, Z, Wt, W2);
  for (int g = lane; g < ngroups; g += 64) {
    v4f Xn, Yn, Zn, Wn, W2n = {0.0f, 0.0f, 0.0f, 0.0f};
    fetch(g + 64, Xn, Yn, Zn, Wn, W2n);                 // (past the end: group 0, weight 0, never used)
#pragma unroll
    for (int h = 0; h < 2; ++h) {                       // candidates (4g, 4g+1) and (4g+2, 4g+3) as packed pairs
      const v2f cx = h ? v2f{X[2], X[3]} : v2f{X[0], X[1]};
      const v2f cy = h ? v2f{Y[2], Y[3]} : v2f{Y[0], Y[1]};
      const v2f cz = h ? v2f{Z[2], Z[3]} : v2f{Z[0], Z[1]};
      const v2f cw = h ? v2f{Wt[2], Wt[3]} : v2f{Wt[0], Wt[1]};
      const v2f cw2 = h ? v2f{W2[2], W2[3]} : v2f{W2[0], W2[1]};
#pragma unroll
      for (int r = 0; r < kEmdOwners; ++r) {
        const v2f dx = cx - px[r], dy = cy - py[r], dz = cz - pz[r];
        const v2f d2 = fma_rn(dz, dz, fma_rn(dy, dy, dx * dx));
        const v2f f2 = {fac[r], fac[r]};
        v2f ex;                                        // exp(level d^2)
        if (MODE == kMatchRatioQ) {
          const v2f e2 = lvl2 * d2;
          const v2f x1 = {__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)};  // the next level's exp(level/4 d^2) ...
          s2[r] = fma_rn(x1, cw2, s2[r]);
          const v2f x2 = x1 * x1;
          ex = x2 * x2;                                 // ... and its fourth power: this level's
        } else {
          const v2f e = lvl * d2;
          ex = v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          if (MODE == kMatchRatioZ) s2[r] += cw2;       // exp(0 d^2) = 1
        }
        const v2f w = ex * cw * f2;
        s[r] += w;
        if (kAssign) {
          const v2f dist = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
          if (MODE == kMatch || kMerged) c[r] = fma_rn(w, dist, c[r]);
          if (GRAD) {
            const v2f f = {w.x / __builtin_fmaxf(dist.x, 1e-20f), w.y / __builtin_fmaxf(dist.y, 1e-20f)};
            gx[r] = fma_rn(f, -dx, gx[r]);   // owner - other
            gy[r] = fma_rn(f, -dy, gy[r]);
            gz[r] = fma_rn(f, -dz, gz[r]);
          }
        }
      }
    }
    X = Xn; Y = Yn; Z = Zn; Wt = Wn; W2 = W2n;
  }
  static_assert(kEmdOwners == 2 || (kEmdOwners == 4 && !GRAD), "the reductions below fold four sums per DPP tree");
  float st[kEmdOwners], ct[kEmdOwners], s2t[kEmdOwners];
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) { st[r] = s[r].x + s[r].y; ct[r] = c[r].x + c[r].y; s2t[r] = s2[r].x + s2[r].y; }
  if constexpr (kEmdOwners == 2) {
    wave_sum4_to_last(st[0], st[1], ct[0], ct[1]);
    if (kMerged) { float p0 = 0.0f, p1 = 0.0f; wave_sum4_to_last(s2t[0], s2t[1], p0, p1); }
  } else {
    wave_sum4_to_last(st[0], st[1], st[2], st[3]);
    if (MODE == kMatch || kMerged) wave_sum4_to_last(ct[0], ct[1], ct[2], ct[3]);
    if (kMerged) wave_sum4_to_last(s2t[0], s2t[1], s2t[2], s2t[3]);
  }
  float gt[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  if constexpr (GRAD) {
#pragma unroll
    for (int r = 0; r < 2; ++r) { gt[r][0] = gx[r].x + gx[r].y; gt[r][1] = gy[r].x + gy[r].y; gt[r][2] = gz[r].x + gz[r].y; }
    float pad = 0.0f, pad2 = 0.0f;
    wave_sum4_to_last(gt[0][0], gt[0][1], gt[0][2], pad);
    wave_sum4_to_last(gt[1][0], gt[1][1], gt[1][2], pad2);
  }
  if (lane != 63) return;                                // the DPP tree delivers the totals in the last lane
#pragma unroll
  for (int r = 0; r < kEmdOwners; ++r) {
    const int o = o0 + r;
    if (o >= a.No) continue;
    const size_t oi = (size_t)b * Nop + o;
    if (MODE == kRatioL) {
      a.own_ratio[oi] = a.own_remain[oi] / (st[r] + 1e-9f);
    } else if (MODE == kRatioR) {
      const float rem = a.own_remain[oi];
      const float sumr = st[r] * rem;
      const float consumption = __builtin_fminf(rem / (sumr + 1e-9f), 1.0f);
      a.own_ratio[oi] = consumption * rem;
      a.own_remain[oi] = __builtin_fmaxf(0.0f, rem - sumr);
    } else if (MODE == kMatch) {
      a.own_cost[oi] += ct[r];
      a.own_remain[oi] = __builtin_fmaxf(0.0f, a.own_remain[oi] - st[r]);
    } else if (kMerged) {
      a.own_cost[oi] += ct[r];
      const float rem = __builtin_fmaxf(0.0f, a.own_remain[oi] - st[r]);
      a.own_remain[oi] = rem;
      a.own_ratio[oi] = rem / (s2t[r] + 1e-9f);         // the next level's row normaliser (kRatioL's epilogue)
    }
    if constexpr (GRAD) if (MODE == kMatch || MODE == kGradOther) {
      const size_t gi = ((size_t)b * a.No + o) * 3;
      a.own_grad[gi] += gt[r][0];
      a.own_grad[gi + 1] += gt[r][1];
      a.own_grad[gi + 2] += gt[r][2];
    }
  }
}

// Workspace set-up of a call: the per-point state (padding included: a padded point weighs nothing in every sweep), the
// coordinate-major copies of both clouds and zeroed gradient accumulators.
__global__ void emd_init_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, float* remainL,
                                float* ratioL, float* costrow, float* remainR, float* ratioR, float* soa1, float* soa2,
                                float* g1, float* g2, int B, int N, int M, int Np, int Mp, float multiL, float multiR) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)B * Np) {
    const int b = (int)(t / Np), i = (int)(t - (size_t)b * Np);
    const bool in = i < N;
    remainL[t] = in ? multiL : 0.0f;
    ratioL[t] = 0.0f;
    costrow[t] = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) soa1[((size_t)b * 3 + d) * Np + i] = in ? xyz1[((size_t)b * N + i) * 3 + d] : 0.0f;
  }
  if (t < (size_t)B * Mp) {
    const int b = (int)(t / Mp), i = (int)(t - (size_t)b * Mp);
    const bool in = i < M;
    remainR[t] = in ? multiR : 0.0f;
    ratioR[t] = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) soa2[((size_t)b * 3 + d) * Mp + i] = in ? xyz2[((size_t)b * M + i) * 3 + d] : 0.0f;
  }
  if (g1 && t < (size_t)B * N * 3) g1[t] = 0.0f;
  if (g2 && t < (size_t)B * M * 3) g2[t] = 0.0f;
}

// cost[b] = sum_k costrow[b,k]: one workgroup per cloud, fixed-shape tree (deterministic)
__global__ __launch_bounds__(256) void emd_cost_kernel(const float* __restrict__ costrow, int N, int Np,
                                                       float* __restrict__ cost) {
  __shared__ float red[256];
  const float* r = costrow + (size_t)blockIdx.x * Np;
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

template <int MODE, bool GRAD, int OWN = kEmdOwners>
int sweep(const EmdArgs& a, int B, hipStream_t s, const char* what) {
  dim3 grid((a.No + kEmdWaves * OWN - 1) / (kEmdWaves * OWN), B);
  hipLaunchKernelGGL((emd_sweep_kernel<MODE, GRAD, OWN>), grid, dim3(kEmdThreads), 0, s, a);
  return launch_status(what);
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_emd_workspace_floats(int B, int N, int M) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  const size_t Np = (size_t)fpsg::emd_pad4(N), Mp = (size_t)fpsg::emd_pad4(M);
  return (size_t)B * (6 * Np + 5 * Mp);      // remainL, ratioL, costrow, x1/y1/z1 | remainR, ratioR, x2/y2/z2
}

namespace fpsg {
namespace {
float emd_level(int j) {                      // -4^j for j = 7 .. -1, then 0
  if (j == -2) return 0.0f;
  float level = -1.0f;
  for (int p = 0; p < (j < 0 ? -j : j); ++p) level = (j < 0) ? level * 0.25f : level * 4.0f;
  return level;
}
}  // namespace
}  // namespace fpsg

// variant: -1 automatic; else bit 0 = a level's assignment sweep and the next level's row-normaliser sweep as one launch
// (forward-only calls), bit 1 = four owner points per wave in the forward-only sweeps (two otherwise).
extern "C" int fpsg_emd_approx_variant(const float* xyz1, const float* xyz2, int B, int N, int M,
                                       float* cost, float* gxyz1, float* gxyz2, float* ws, int variant,
                                       fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE,
               "fpsg_emd_approx: B,N,M must be positive (got %d,%d,%d)", B, N, M);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_emd_approx: B=%d exceeds 65535", B);
  FPSG_REQUIRE(variant >= -1 && variant <= 3, FPSG_E_SHAPE, "fpsg_emd_approx_variant: variant %d not in [-1, 3]", variant);
  FPSG_REQUIRE_PTR(xyz1); FPSG_REQUIRE_PTR(xyz2); FPSG_REQUIRE_PTR(cost); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(!misaligned4(gxyz1) && !misaligned4(gxyz2), FPSG_E_ALIGN,
               "fpsg_emd_approx: gradient buffers must be 4-byte aligned");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, FPSG_E_ALIGN, "fpsg_emd_approx: ws must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Np = emd_pad4(N), Mp = emd_pad4(M);
  float* remainL = ws;
  float* ratioL = remainL + (size_t)B * Np;
  float* costrow = ratioL + (size_t)B * Np;
  float* soa1 = costrow + (size_t)B * Np;
  float* remainR = soa1 + (size_t)B * 3 * Np;
  float* ratioR = remainR + (size_t)B * Mp;
  float* soa2 = ratioR + (size_t)B * Mp;
  const float multiL = (M > N) ? (float)(M / N) : 1.0f;
  const float multiR = (N >= M) ? (float)(N / M) : 1.0f;
  {
    const size_t n = (size_t)B * 3 * (size_t)(Np > Mp ? Np : Mp);
    hipLaunchKernelGGL(emd_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyz1, xyz2, remainL, ratioL,
                       costrow, remainR, ratioR, soa1, soa2, gxyz1, gxyz2, B, N, M, Np, Mp, multiL, multiR);
    int rc = launch_status("fpsg_emd_approx(init)");
    if (rc) return rc;
  }
  const bool forward_only = gxyz1 == nullptr && gxyz2 == nullptr;
  // measured on MI355X (profiles/r04/k2_variants.txt): the merged sweeps win at every size (B = 5: 372 -> 297 us); four
  // owners per wave only once the grid has waves to spare (B = 37: 1630 -> 1565 us; B = 5: 297 -> 348: 2.5 waves per SIMD)
  // small pairs keep the separate sweeps: the merged form's exp(l d^2)^4 deviates 25x more there and the
  // launches saved are a few microseconds
  if (variant < 0) variant = ((long)N * M >= 256L * 256L ? 1 : 0) | ((long)B * (N > M ? N : M) >= 32768 ? 2 : 0);
  const bool merged = forward_only && (variant & 1);
  const bool own4 = forward_only && (variant & 2);
  int rc;
  auto rows = [&](EmdArgs& a) {               // owners = cloud 1, sweeping cloud 2
    a.own = soa1; a.oth = soa2; a.own_remain = remainL; a.own_ratio = ratioL; a.No = N; a.Nt = M;
  };
  auto cols = [&](EmdArgs& a) {               // owners = cloud 2, sweeping cloud 1
    a.own = soa2; a.oth = soa1; a.own_remain = remainR; a.own_ratio = ratioR; a.No = M; a.Nt = N;
  };
  for (int j = 7; j >= -2; --j) {
    EmdArgs a{};
    a.level = emd_level(j);
    // row normalisers: sweep cloud 2 weighted by remainR (merged: done by the previous level's assignment launch)
    if (!merged || j == 7) {
      rows(a); a.oth_w = remainR;
      rc = own4 ? sweep<kRatioL, false, 4>(a, B, s, "fpsg_emd_approx(ratioL)")
                : sweep<kRatioL, false>(a, B, s, "fpsg_emd_approx(ratioL)");
      if (rc) return rc;
    }
    // column consumption: sweep cloud 1 weighted by ratioL
    cols(a); a.oth_w = ratioL;
    rc = own4 ? sweep<kRatioR, false, 4>(a, B, s, "fpsg_emd_approx(ratioR)")
              : sweep<kRatioR, false>(a, B, s, "fpsg_emd_approx(ratioR)");
    if (rc) return rc;
    // assignment: owners = cloud 1 (factor ratioL), sweep cloud 2 weighted by ratioR
    rows(a); a.oth_w = ratioR; a.own_cost = costrow; a.own_grad = gxyz1;
    if (merged && j > -2) {                   // + the next level's row normalisers (weights remainR)
      a.oth_w2 = remainR;
      a.level2 = emd_level(j - 1);
      if (j - 1 == -2)
        rc = own4 ? sweep<kMatchRatioZ, false, 4>(a, B, s, "fpsg_emd_approx(match+ratioL)")
                  : sweep<kMatchRatioZ, false>(a, B, s, "fpsg_emd_approx(match+ratioL)");
      else
        rc = own4 ? sweep<kMatchRatioQ, false, 4>(a, B, s, "fpsg_emd_approx(match+ratioL)")
                  : sweep<kMatchRatioQ, false>(a, B, s, "fpsg_emd_approx(match+ratioL)");
    } else if (gxyz1) {
      rc = sweep<kMatch, true>(a, B, s, "fpsg_emd_approx(match+grad)");
    } else {
      rc = own4 ? sweep<kMatch, false, 4>(a, B, s, "fpsg_emd_approx(match)")
                : sweep<kMatch, false>(a, B, s, "fpsg_emd_approx(match)");
    }
    if (rc) return rc;
    if (gxyz2) {  // same weights seen from cloud 2
      a.own = soa2; a.oth = soa1; a.oth_w = ratioL; a.own_remain = nullptr; a.own_ratio = ratioR;
      a.own_cost = nullptr; a.own_grad = gxyz2; a.No = M; a.Nt = N;
      if ((rc = sweep<kGradOther, true>(a, B, s, "fpsg_emd_approx(grad2)"))) return rc;
    }
  }
  hipLaunchKernelGGL(emd_cost_kernel, dim3(B), dim3(256), 0, s, costrow, N, Np, cost);
  return launch_status("fpsg_emd_approx(cost)");
}

extern "C" int fpsg_emd_approx(const float* xyz1, const float* xyz2, int B, int N, int M,
                               float* cost, float* gxyz1, float* gxyz2, float* ws,
                               fpsg_stream_t stream) {
  return fpsg_emd_approx_variant(xyz1, xyz2, B, N, M, cost, gxyz1, gxyz2, ws, -1, stream);
}
