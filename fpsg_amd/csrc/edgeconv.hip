// edgeconv.hip -- K4b: fused, non-materialising EdgeConv layer of DGCNN for gfx950.
//
// Reference (src/dgcnn/model.py:23-42,53-56,63-76): gather k neighbour rows, build
// cat(x_j - x_i, x_i) as a [B,2C,N,k] tensor (2.7 GB at C=128, B=64), 1x1 Conv2d, training
// mode BatchNorm2d over (B,N,k), LeakyReLU(0.2), max over k.
//
// Here the [B,2C,N,k] tensor (and the [B,Co,N,k] conv output) never exist:
//   W [x_j - x_i ; x_i] = W1 x_j + (W2 - W1) x_i  =  P[j] + Q[i]      (two N-row GEMMs, host side)
//   BatchNorm + LeakyReLU are monotone per channel, increasing if gamma >= 0 and decreasing
//   otherwise, so  max_j act(bn(y_j)) = act(bn(max_j y_j))  or  act(bn(min_j y_j)).
// Forward kernel: one wave per point, lanes across the Co output channels (16-B loads of
// whole P rows = coalesced, L2/MALL resident): for the k neighbours y = P[idx] + Q, track
// the selected extreme + its neighbour slot, sum y (S1) and sum y^2; per-workgroup partial
// BatchNorm sums go to a [blocks,2,Co] buffer that the host reduces in float64.
// Backward kernel: one wave per point m walks the REVERSE graph (in-edges of m, built by a
// stable sort on the host side, so the summation order is fixed => deterministic, no float
// atomics): dP[m] = sum_{(n,j)->m} [jsel[n]==j] dzs[n] - cnt (A + Bc (P[m]-mu)) - Bc sum Q[n],
// dQ[m] = dzs[m] - k A - Bc (S1[m] - k mu)   (A, Bc: the BatchNorm statistic terms).
#include "fpsg_common.h"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace fpsg {
namespace {

constexpr int kEcThreads = 256;  // 4 waves
#ifndef FPSG_EC_IN_FLIGHT
#define FPSG_EC_IN_FLIGHT 4
#endif
constexpr int kInFlight = FPSG_EC_IN_FLIGHT;   // in-edges whose rows a wave of the backward has in flight
constexpr int kFwdInFlight = 5;                // forward: G = 4 groups x 5 = the 20 neighbours of a point in one round of loads

template <int VEC>
struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef v2f type; };
template <> struct VecT<4> { typedef v4f type; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  typedef typename VecT<VEC>::type T;
  const T t = *reinterpret_cast<const T*>(p);
  if constexpr (VEC == 1) {
    v[0] = t;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = t[i];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  typedef typename VecT<VEC>::type T;
  T t;
  if constexpr (VEC == 1) {
    t = v[0];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) t[i] = v[i];
  }
  *reinterpret_cast<T*>(p) = t;
}

// Workgroup -> (cloud, block of points).  The neighbour gathers of a cloud touch its whole [N, Co]
// arrays (2 MB at Co = 256) about k times: they should come from L2, not HBM.  Workgroup ids are
// dealt round-robin to the 8 XCDs (id % 8), each with its own 4 MB L2, so cloud b is given to XCD
// b % 8 and an XCD works through its clouds one after the other (PMC before: 6.3 GB of HBM reads
// for 0.6 GB of operands in the Co = 256 backward).
struct CloudBlock { int b, bx; };
__device__ __forceinline__ CloudBlock cloud_block(int bpc) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  CloudBlock r;
  r.b = xcd + 8 * (slot / bpc);
  r.bx = slot % bpc;
  return r;
}

// PQ [B,N,2*Co] (P = first Co columns, Q = last Co), idx [B,N,k], sgn [Co] (+1: take max, -1:
// take min).  Outputs ysel [B,N,Co], jsel [B,N,Co] (uint8 neighbour slot), s1 [B,N,Co] or null,
// part [gridDim.y*gridDim.x][2][Co] or null.
template <int VEC>
__global__ __launch_bounds__(kEcThreads) void edgeconv_fwd_kernel(
    const float* __restrict__ PQ, const int32_t* __restrict__ idx, const float* __restrict__ sgn,
    int B, int N, int k, int bpc, int ppw, float* __restrict__ ysel, uint8_t* __restrict__ jsel, float* __restrict__ s1,
    float* __restrict__ part) {
  constexpr int Co = 64 * VEC;
  __shared__ float red[4][2][Co];
  const CloudBlock cb = cloud_block(bpc);
  if (cb.b >= B) return;
  const int b = cb.b;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = lane * VEC;
  const float* __restrict__ base = PQ + (size_t)b * N * 2 * Co;
  float sg[VEC], sum[VEC], sumsq[VEC];
  load_vec<VEC>(sgn + c0, sg);
#pragma unroll
  for (int v = 0; v < VEC; ++v) { sg[v] = sg[v] < 0.0f ? -1.0f : 1.0f; sum[v] = 0.0f; sumsq[v] = 0.0f; }   // only the sign counts

  const int n_first = (cb.bx * 4 + wave) * ppw;
  for (int pp = 0; pp < ppw; ++pp) {
    const int n = n_first + pp;
    if (n >= N) break;  // wave-uniform
    const size_t row = (size_t)b * N + n;
    int my = lane < k ? idx[row * k + lane] : 0;
    my = my < 0 ? 0 : (my >= N ? N - 1 : my);
    float q[VEC], best[VEC], tot[VEC];
    int bj[VEC];
    load_vec<VEC>(base + (size_t)n * 2 * Co + Co + c0, q);
#pragma unroll
    for (int v = 0; v < VEC; ++v) { best[v] = -__builtin_inff(); bj[v] = 0; tot[v] = 0.0f; }
    for (int j = 0; j < k; ++j) {
      const int m = __builtin_amdgcn_readlane(my, j);
      float p[VEC];
      load_vec<VEC>(base + (size_t)m * 2 * Co + c0, p);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float y = p[v] + q[v];
        tot[v] += y;
        sumsq[v] = fma_rn(y, y, sumsq[v]);
        const float t = y * sg[v];          // +-y: one code path for max and min
        const bool gt = t > best[v];
        bj[v] = gt ? j : bj[v];
        best[v] = gt ? t : best[v];
      }
    }
    float out[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { out[v] = best[v] * sg[v]; sum[v] += tot[v]; }
    store_vec<VEC>(ysel + row * Co + c0, out);
    if (s1) store_vec<VEC>(s1 + row * Co + c0, tot);
    if constexpr (VEC == 1) {
      jsel[row * Co + c0] = (uint8_t)bj[0];
    } else if constexpr (VEC == 2) {
      *reinterpret_cast<uint16_t*>(jsel + row * Co + c0) = (uint16_t)(bj[0] | (bj[1] << 8));
    } else {
      *reinterpret_cast<uint32_t*>(jsel + row * Co + c0) =
          (uint32_t)bj[0] | ((uint32_t)bj[1] << 8) | ((uint32_t)bj[2] << 16) | ((uint32_t)bj[3] << 24);
    }
  }
  if (!part) return;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { red[wave][0][c0 + v] = sum[v]; red[wave][1][c0 + v] = sumsq[v]; }
  __syncthreads();
  float* dst = part + ((size_t)b * bpc + cb.bx) * 2 * Co;
  for (int e = threadIdx.x; e < 2 * Co; e += kEcThreads) {
    const int which = e / Co, c = e - which * Co;
    dst[e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// The forward with SEVERAL neighbours per load instruction (round 3; Co = 64 and 128).  At Co = 64 a neighbour's P row
// is 256 bytes: one float per lane and 20 small loads per point.  With 4 channels per lane (16-byte loads) a row takes
// LPE = Co / 4 lanes and a wave instruction reads the rows of G = 64 / LPE neighbours.  Group g walks neighbours
// g, g + G, ... in order; the groups' partial results are then merged: sums pairwise in a fixed order (deterministic;
// last bits differ from the one-neighbour order), the selected extreme by (value, then lower slot) -- exactly the
// "first strictly better wins" rule of the sequential walk.
template <int LPE>
__global__ __launch_bounds__(kEcThreads) void edgeconv_fwd_grouped_kernel(
    const float* __restrict__ PQ, const int32_t* __restrict__ idx, const float* __restrict__ sgn,
    int B, int N, int k, int bpc, int ppw, float* __restrict__ ysel, uint8_t* __restrict__ jsel, float* __restrict__ s1,
    float* __restrict__ part) {
  constexpr int G = 64 / LPE;
  constexpr int Co = 4 * LPE;
  __shared__ float red[4][2][Co];
  const CloudBlock cb = cloud_block(bpc);
  if (cb.b >= B) return;
  const int b = cb.b;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane / LPE, cl = lane - grp * LPE;
  const int c0 = cl * 4;
  const float* __restrict__ base = PQ + (size_t)b * N * 2 * Co;
  float sg[4], sum[4], sumsq[4];
  load_vec<4>(sgn + c0, sg);
#pragma unroll
  for (int v = 0; v < 4; ++v) { sg[v] = sg[v] < 0.0f ? -1.0f : 1.0f; sum[v] = 0.0f; sumsq[v] = 0.0f; }

  auto merge = [&](auto xor_tag, float (&best)[4], int (&bj)[4], float (&tot)[4]) {
    constexpr int M = decltype(xor_tag)::value;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float ob = __uint_as_float(lane_xor<M>(__float_as_uint(best[v])));
      const int oj = (int)lane_xor<M>((unsigned)bj[v]);
      const bool take = ob > best[v] || (ob == best[v] && oj < bj[v]);
      bj[v] = take ? oj : bj[v];
      best[v] = take ? ob : best[v];
      tot[v] += __uint_as_float(lane_xor<M>(__float_as_uint(tot[v])));
    }
  };

  const int n_first = (cb.bx * 4 + wave) * ppw;
  for (int pp = 0; pp < ppw; ++pp) {
    const int n = n_first + pp;
    if (n >= N) break;  // wave-uniform
    const size_t row = (size_t)b * N + n;
    int my = lane < k ? idx[row * k + lane] : 0;
    my = my < 0 ? 0 : (my >= N ? N - 1 : my);
    float q[4], best[4], tot[4];
    int bj[4];
    load_vec<4>(base + (size_t)n * 2 * Co + Co + c0, q);
#pragma unroll
    for (int v = 0; v < 4; ++v) { best[v] = -__builtin_inff(); bj[v] = 0x7fffffff; tot[v] = 0.0f; }
    for (int j0 = 0; j0 < k; j0 += G * kFwdInFlight) {
      float p[kFwdInFlight][4];
      int jj[kFwdInFlight];
#pragma unroll
      for (int u = 0; u < kFwdInFlight; ++u) {
        jj[u] = j0 + u * G + grp;
        const int m = __shfl(my, jj[u] < k ? jj[u] : 0, 64);
        load_vec<4>(base + (size_t)m * 2 * Co + c0, p[u]);
      }
#pragma unroll
      for (int u = 0; u < kFwdInFlight; ++u) {
        if (jj[u] < k) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float y = p[u][v] + q[v];
            tot[v] += y;
            sumsq[v] = fma_rn(y, y, sumsq[v]);
            const float t = y * sg[v];          // +-y: one code path for max and min
            const bool gt = t > best[v];
            bj[v] = gt ? jj[u] : bj[v];
            best[v] = gt ? t : best[v];
          }
        }
      }
    }
    if constexpr (G == 4) merge(std::integral_constant<int, 16>{}, best, bj, tot);
    merge(std::integral_constant<int, 32>{}, best, bj, tot);
    if (grp == 0) {
      float out[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) { out[v] = best[v] * sg[v]; sum[v] += tot[v]; }
      store_vec<4>(ysel + row * Co + c0, out);
      if (s1) store_vec<4>(s1 + row * Co + c0, tot);
      *reinterpret_cast<uint32_t*>(jsel + row * Co + c0) =
          (uint32_t)(bj[0] & 0xff) | ((uint32_t)(bj[1] & 0xff) << 8) | ((uint32_t)(bj[2] & 0xff) << 16) | ((uint32_t)(bj[3] & 0xff) << 24);
    }
  }
  if (!part) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {            // sum of squares: every group holds a share
    if constexpr (G == 4) sumsq[v] += __uint_as_float(lane_xor<16>(__float_as_uint(sumsq[v])));
    sumsq[v] += __uint_as_float(lane_xor<32>(__float_as_uint(sumsq[v])));
  }
  if (grp == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) { red[wave][0][c0 + v] = sum[v]; red[wave][1][c0 + v] = sumsq[v]; }
  }
  __syncthreads();
  float* dst = part + ((size_t)b * bpc + cb.bx) * 2 * Co;
  for (int e = threadIdx.x; e < 2 * Co; e += kEcThreads) {
    const int which = e / Co, c = e - which * Co;
    dst[e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// dzs [B,N,Co] = dz*scale; jsel; PQ; s1 (or null when coef A=Bc=0); rev [B,N*k] edge ids
// (n*k+j) grouped by destination, ascending inside a group; off [B,N+1] group offsets;
// coef [3][Co] = A, Bc, mu.  Output dPQ [B,N,2*Co].
//
// A workgroup serves one CHANNEL SLICE of 64*VEC channels of a block of points (Co = nsl * 64 * VEC).
// The in-edge gathers of a cloud touch its whole dzs / jsel / Q arrays about k times; at Co = 256 those
// are 2 + 0.5 + 2 MB per cloud -- more than the 4 MB L2 of an XCD, and the PMC counters showed 5-6 GB
// of HBM reads per episode for 0.6 GB of operands.  A 128-channel slice is 2.25 MB per cloud: an XCD
// works through (cloud, slice) pairs one after the other and the gathers stay in its L2.
// A wave loads its point's in-edge list with one coalesced read (64 edges per pass, readlane per edge)
// and keeps the rows of two edges in flight; the sums run in edge order (deterministic, bit-identical
// to the one-edge-at-a-time form).
template <int VEC>
__global__ __launch_bounds__(kEcThreads) void edgeconv_bwd_kernel(
    const float* __restrict__ dzs, const uint8_t* __restrict__ jsel, const float* __restrict__ PQ,
    const float* __restrict__ s1, const int32_t* __restrict__ rev, const int32_t* __restrict__ off,
    const float* __restrict__ coef, int B, int N, int k, int bpc, int ppw, int nsl, int stats, float* __restrict__ dPQ) {
  const int Co = 64 * VEC * nsl;
  // (cloud, slice) pairs are dealt to the XCDs like clouds in the forward kernel
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int pair = xcd + 8 * (slot / bpc);            // = slice-major inside a cloud: pair = b * nsl + sl
  const int bx = slot % bpc;
  const int b = pair / nsl, sl = pair - b * nsl;
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = sl * 64 * VEC + lane * VEC;
  float A[VEC], Bc[VEC], mu[VEC];
  load_vec<VEC>(coef + c0, A);
  load_vec<VEC>(coef + Co + c0, Bc);
  load_vec<VEC>(coef + 2 * Co + c0, mu);
  const float* __restrict__ pq = PQ + (size_t)b * N * 2 * Co;
  const int32_t* __restrict__ revb = rev + (size_t)b * N * k;
  const int32_t* __restrict__ offb = off + (size_t)b * (N + 1);
  const int m_first = (bx * 4 + wave) * ppw;

  auto load_js = [&](size_t rown) -> unsigned {
    if constexpr (VEC == 1) return jsel[rown * Co + c0];
    else if constexpr (VEC == 2) return *reinterpret_cast<const uint16_t*>(jsel + rown * Co + c0);
    else return *reinterpret_cast<const uint32_t*>(jsel + rown * Co + c0);
  };

  for (int pp = 0; pp < ppw; ++pp) {
    const int m = m_first + pp;
    if (m >= N) break;
    const size_t rowm = (size_t)b * N + m;
    int e0 = offb[m], e1 = offb[m + 1];
    e0 = e0 < 0 ? 0 : e0;
    e1 = e1 > N * k ? N * k : e1;  // never walk outside the edge list
    float acc[VEC], accq[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[v] = 0.0f; accq[v] = 0.0f; }
    for (int tb = e0; tb < e1; tb += 64) {
      const int here = (e1 - tb) < 64 ? (e1 - tb) : 64;
      const int mine = lane < here ? revb[tb + lane] : 0;        // one coalesced read of up to 64 edge ids
      for (int t = 0; t < here; t += kInFlight) {
        int n[kInFlight], j[kInFlight];
        size_t rown[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
          const int e = __builtin_amdgcn_readlane(mine, (t + u) < here ? (t + u) : t);
          int nn = e / k;
          nn = nn < 0 ? 0 : (nn >= N ? N - 1 : nn);
          n[u] = nn;
          j[u] = e - nn * k;
          rown[u] = (size_t)b * N + nn;
        }
        float g[kInFlight][VEC], qn[kInFlight][VEC];
        unsigned js[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {                      // the rows of kInFlight edges in flight
          load_vec<VEC>(dzs + rown[u] * Co + c0, g[u]);
          js[u] = load_js(rown[u]);
          if (stats) load_vec<VEC>(pq + (size_t)n[u] * 2 * Co + Co + c0, qn[u]);
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {                      // summed in edge order
          if (t + u >= here) break;
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += ((int)((js[u] >> (8 * v)) & 0xffu) == j[u]) ? g[u][v] : 0.0f;
          if (stats) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) accq[v] += qn[u][v];
          }
        }
      }
    }
    float dp[VEC], dq[VEC];
    load_vec<VEC>(dzs + rowm * Co + c0, dq);
    if (stats) {
      float pm[VEC], sm[VEC];
      load_vec<VEC>(pq + (size_t)m * 2 * Co + c0, pm);
      load_vec<VEC>(s1 + rowm * Co + c0, sm);
      const float cnt = (float)(e1 - e0), kf = (float)k;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        dp[v] = acc[v] - cnt * (A[v] + Bc[v] * (pm[v] - mu[v])) - Bc[v] * accq[v];
        dq[v] = dq[v] - kf * A[v] - Bc[v] * (sm[v] - kf * mu[v]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) dp[v] = acc[v];
    }
    store_vec<VEC>(dPQ + rowm * 2 * Co + c0, dp);
    store_vec<VEC>(dPQ + rowm * 2 * Co + Co + c0, dq);
  }
}

// The same backward with SEVERAL in-edges per load instruction (round 3).  The one-edge form reads a 256-byte row per
// load at Co = 64 (one float per lane): 60 small loads per point, and the kernel is bound by the number of memory
// transactions, not by bytes.  Here a lane always holds 4 channels (16-byte loads), so a row of a 64 * 4 / G-channel
// slice takes LPE = 64 / G lanes and a wave instruction reads the rows of G in-edges: G = 4 at Co = 64, G = 2 at
// Co = 128 and for the 128-channel slices of Co = 256.  Group g sums in-edges g, g + G, ... in order; the groups are
// then added pairwise ((g0 + g1) + (g2 + g3)) with lane exchanges: a fixed order, so still deterministic (it differs
// from the one-edge order in the last bits).
template <int LPE>
__global__ __launch_bounds__(kEcThreads) void edgeconv_bwd_grouped_kernel(
    const float* __restrict__ dzs, const uint8_t* __restrict__ jsel, const float* __restrict__ PQ,
    const float* __restrict__ s1, const int32_t* __restrict__ rev, const int32_t* __restrict__ off,
    const float* __restrict__ coef, int B, int N, int k, int bpc, int ppw, int nsl, int stats, float* __restrict__ dPQ) {
  constexpr int G = 64 / LPE;                 // in-edges per load instruction
  constexpr int SLICE = 4 * LPE;              // channels of a workgroup's slice
  const int Co = SLICE * nsl;
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int pair = xcd + 8 * (slot / bpc);            // = slice-major inside a cloud: pair = b * nsl + sl
  const int bx = slot % bpc;
  const int b = pair / nsl, sl = pair - b * nsl;
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane / LPE, cl = lane - grp * LPE;
  const int c0 = sl * SLICE + cl * 4;
  float A[4], Bc[4], mu[4];
  load_vec<4>(coef + c0, A);
  load_vec<4>(coef + Co + c0, Bc);
  load_vec<4>(coef + 2 * Co + c0, mu);
  // per-cloud bases (wave-uniform) + 32-bit element offsets: the loads take the scalar-base form and the row
  // offsets of the edges in flight cost one register each instead of two
  const float* __restrict__ pq = PQ + (size_t)b * N * 2 * Co;
  const float* __restrict__ dzb = dzs + (size_t)b * N * Co;
  const uint8_t* __restrict__ jsb = jsel + (size_t)b * N * Co;
  const int32_t* __restrict__ revb = rev + (size_t)b * N * k;
  const int32_t* __restrict__ offb = off + (size_t)b * (N + 1);
  const int m_first = (bx * 4 + wave) * ppw;
  const size_t row0 = (size_t)b * N;

  for (int pp = 0; pp < ppw; ++pp) {
    const int m = m_first + pp;
    if (m >= N) break;
    const size_t rowm = row0 + m;
    int e0 = offb[m], e1 = offb[m + 1];
    e0 = e0 < 0 ? 0 : e0;
    e1 = e1 > N * k ? N * k : e1;  // never walk outside the edge list
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, accq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tb = e0; tb < e1; tb += 64) {
      const int here = (e1 - tb) < 64 ? (e1 - tb) : 64;
      const int mine = lane < here ? revb[tb + lane] : 0;        // one coalesced read of up to 64 edge ids
      for (int t = 0; t < here; t += G * kInFlight) {
        bool on[kInFlight];
        int j[kInFlight];
        unsigned rown[kInFlight];                                  // source point inside the cloud
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
          const int te = t + u * G + grp;                          // this lane group's edge of the round
          on[u] = te < here;
          const int e = __shfl(mine, on[u] ? te : 0, 64);
          int nn = e / k;
          nn = nn < 0 ? 0 : (nn >= N ? N - 1 : nn);
          j[u] = e - nn * k;
          rown[u] = (unsigned)nn;
        }
        float g[kInFlight][4], qn[kInFlight][4];
        unsigned js[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {                      // the rows of G * kInFlight in-edges in flight
          const unsigned o = rown[u] * (unsigned)Co + (unsigned)c0;
          load_vec<4>(dzb + o, g[u]);
          js[u] = *reinterpret_cast<const uint32_t*>(jsb + o);
          if (stats) load_vec<4>(pq + (2u * rown[u] * (unsigned)Co + (unsigned)(Co + c0)), qn[u]);
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {                      // a group's edges in order
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            acc[v] += (on[u] && (int)((js[u] >> (8 * v)) & 0xffu) == j[u]) ? g[u][v] : 0.0f;
            if (stats) accq[v] += on[u] ? qn[u][v] : 0.0f;
          }
        }
      }
    }
    // the groups' sums, pairwise
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if constexpr (G == 4) {
        acc[v] += __uint_as_float(lane_xor<16>(__float_as_uint(acc[v])));
        accq[v] += __uint_as_float(lane_xor<16>(__float_as_uint(accq[v])));
      }
      acc[v] += __uint_as_float(lane_xor<32>(__float_as_uint(acc[v])));
      accq[v] += __uint_as_float(lane_xor<32>(__float_as_uint(accq[v])));
    }
    if (grp == 0) {
      float dp[4], dq[4];
      load_vec<4>(dzs + rowm * Co + c0, dq);
      if (stats) {
        float pm[4], sm[4];
        load_vec<4>(pq + (size_t)m * 2 * Co + c0, pm);
        load_vec<4>(s1 + rowm * Co + c0, sm);
        const float cnt = (float)(e1 - e0), kf = (float)k;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          dp[v] = acc[v] - cnt * (A[v] + Bc[v] * (pm[v] - mu[v])) - Bc[v] * accq[v];
          dq[v] = dq[v] - kf * A[v] - Bc[v] * (sm[v] - kf * mu[v]);
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) dp[v] = acc[v];
      }
      store_vec<4>(dPQ + rowm * 2 * Co + c0, dp);
      store_vec<4>(dPQ + rowm * 2 * Co + Co + c0, dq);
    }
  }
}

// Points per wave.  An XCD keeps ~256 of these 4-wave workgroups resident; they should belong to as few
// clouds as fit its 4 MB L2 together (the k-fold gathers of a cloud touch 2 MB of P rows at Co = 256,
// 0.5 MB at Co = 64): one cloud = 256 workgroups at Co = 256, 128 at Co = 128, 64 at Co = 64 (N = 2048).
inline bool edgeconv_bwd_one_edge() {
  const char* e = getenv("FPSG_EDGECONV_BWD");
  return e != nullptr && strcmp(e, "one_edge") == 0;
}
inline int ec_points_per_wave(int Co) { return Co == 256 ? 2 : Co == 128 ? 4 : 8; }
inline int ec_blocks_per_cloud(int N, int Co) { return (N + 4 * ec_points_per_wave(Co) - 1) / (4 * ec_points_per_wave(Co)); }
inline dim3 ec_grid(int B, int N, int Co) {          // 8 * ceil(B/8) clouds' worth of workgroups, see cloud_block
  return dim3((unsigned)(8 * ((B + 7) / 8) * ec_blocks_per_cloud(N, Co)));
}


// ---- the elementwise halves around the two kernels above (point-major [rows, Co], rows = B*N) -------------------
// forward tail: out = LeakyReLU(fma(ysel, scale[c], shift[c])) -- BatchNorm's affine form on the selected
// neighbour sum + the activation, one pass (torch: addcmul + leaky_relu, two).
// The per-channel scalar work between the kernels, one launch each (as PyTorch ops: ~20 one-vector launches per
// layer in the forward, ~15 in the backward).  part [blocks][2][Co] -> fp64 sums in block order.
// forward: batch statistics of the edge activations (count = B*N*k) -> chan [4][Co] = (scale, shift, mean, rstd) with
// scale = gamma * rstd, shift = beta - mean * scale; running statistics updated as nn.BatchNorm2d does (unbiased
// variance).  training = 0: chan from the running statistics, part unused.
// fp64 sums over the blocks for the 4 channels of a workgroup: thread (row slice q = tid / 4, channel tid % 4) adds
// blocks q, q + slices, ... in order (slices = threads / 4), the slices are then added in order by the first 4
// threads (deterministic)
constexpr int kFinThreads = 1024, kFinBwdThreads = 256, kFinCh = 4;   // forward: 8,192 partial blocks per layer at 64 clouds
                                                                      // (32 loads per thread); backward: 512
template <int THREADS>
__device__ __forceinline__ bool part_sums(const float* __restrict__ part, int blocks, int Co, int& c, double& s0, double& s1) {
  constexpr int kSlices = THREADS / kFinCh;
  __shared__ double red[2][THREADS];
  const int q = threadIdx.x / kFinCh, cl = threadIdx.x % kFinCh;
  c = blockIdx.x * kFinCh + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < Co && part) {
#pragma unroll 4
    for (int b = q; b < blocks; b += kSlices) { a0 += part[((size_t)b * 2) * Co + c]; a1 += part[((size_t)b * 2 + 1) * Co + c]; }
  }
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  if (q != 0 || c >= Co) return false;
  s0 = 0.0; s1 = 0.0;
  for (int r = 0; r < kSlices; ++r) { s0 += red[0][r * kFinCh + cl]; s1 += red[1][r * kFinCh + cl]; }
  return true;
}

__global__ __launch_bounds__(kFinThreads) void edgeconv_stats_finalize_kernel(const float* __restrict__ part, int blocks, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ run_mean,
                                               float* __restrict__ run_var, float momentum, float eps, double count,
                                               int Co, int training, float* __restrict__ chan) {
  int c;
  double s0, s1;
  if (!part_sums<kFinThreads>(training ? part : nullptr, blocks, Co, c, s0, s1)) return;
  float mean, var;
  if (training) {
    const double m = s0 / count;
    double v = s1 / count - m * m;
    v = v > 0.0 ? v : 0.0;
    mean = (float)m;
    var = (float)v;
    if (run_mean) {
      const float unbiased = var * (float)(count / (count > 1.0 ? count - 1.0 : 1.0));
      run_mean[c] = fma_rn(momentum, mean, (1.0f - momentum) * run_mean[c]);
      run_var[c] = fma_rn(momentum, unbiased, (1.0f - momentum) * run_var[c]);
    }
  } else {
    mean = run_mean[c];
    var = run_var[c];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float scale = gamma[c] * rstd;
  chan[c] = scale;
  chan[Co + c] = beta[c] - mean * scale;
  chan[2 * Co + c] = mean;
  chan[3 * Co + c] = rstd;
}

// Two stages for the forward's 8,192-16,384 partial rows (round 4): the one-launch form above keeps Co / 4 workgroups busy with
// 32-64 strided 16-byte reads per thread and a 256-long serial fp64 sum (13 / 13 / 38 / 96 us for DGCNN's four layers at 64
// clouds).  Stage 1: a workgroup takes a slice of the rows and 128 of the 2 Co columns -- whole 512-byte row pieces,
// coalesced -- thread (row phase, column) adds its rows in ascending order in fp64, the two phases are joined in LDS
// (phase 0 first) -> slice sums [Z][2 Co] doubles.  Stage 2: one wave per channel, lane l adds slices l, l + 64, ... in
// ascending order, then the fixed shuffle tree (wave_sum2), then the same finalize arithmetic.  Deterministic.
constexpr int kS1Threads = 256, kS1Cols = 128;
__device__ __forceinline__ void wave_sum2(double& a, double& b) {      // lane 0 receives the sums; fixed tree
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
}
inline int stats_slices(int blocks, int Co) {          // so that stage 1 is ~256-512 workgroups and a slice >= 8 rows
  const int groups = (2 * Co + kS1Cols - 1) / kS1Cols;
  int Z = 512 / groups;
  while (Z > 1 && blocks / Z < 8) Z >>= 1;
  return Z < 1 ? 1 : Z;
}

__global__ __launch_bounds__(kS1Threads) void edgeconv_stats_slices_kernel(const float* __restrict__ part, int blocks, int Co,
                                                                          int Z, double* __restrict__ slices /*[Z][2 Co]*/) {
  __shared__ double red[kS1Cols];
  const int z = blockIdx.x, col = blockIdx.y * kS1Cols + (threadIdx.x & (kS1Cols - 1)), phase = threadIdx.x / kS1Cols;
  const int per = (blocks + Z - 1) / Z;
  const int b0 = z * per, b1 = (b0 + per < blocks) ? b0 + per : blocks;
  double a = 0.0;
  if (col < 2 * Co) {
    int b = b0 + phase;
    for (; b + 6 < b1; b += 8) {                       // four loads of a thread in flight
      const float v0 = part[(size_t)b * 2 * Co + col], v1 = part[(size_t)(b + 2) * 2 * Co + col];
      const float v2 = part[(size_t)(b + 4) * 2 * Co + col], v3 = part[(size_t)(b + 6) * 2 * Co + col];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; b < b1; b += 2) a += part[(size_t)b * 2 * Co + col];
  }
  if (phase == 1) red[threadIdx.x & (kS1Cols - 1)] = a;
  __syncthreads();
  if (phase == 0 && col < 2 * Co) slices[(size_t)z * 2 * Co + col] = a + red[threadIdx.x];
}

__global__ __launch_bounds__(256) void edgeconv_stats_finalize2_kernel(const double* __restrict__ slices, int Z, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ run_mean,
                                                float* __restrict__ run_var, float momentum, float eps, double count,
                                                int Co, float* __restrict__ chan) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= Co) return;                                  // whole wave
  double s0 = 0.0, s1 = 0.0;
  for (int z = lane; z < Z; z += 64) { s0 += slices[(size_t)z * 2 * Co + c]; s1 += slices[(size_t)z * 2 * Co + Co + c]; }
  wave_sum2(s0, s1);
  if (lane != 0) return;
  const double m = s0 / count;
  double v = s1 / count - m * m;
  v = v > 0.0 ? v : 0.0;
  const float mean = (float)m, var = (float)v;
  if (run_mean) {
    const float unbiased = var * (float)(count / (count > 1.0 ? count - 1.0 : 1.0));
    run_mean[c] = fma_rn(momentum, mean, (1.0f - momentum) * run_mean[c]);
    run_var[c] = fma_rn(momentum, unbiased, (1.0f - momentum) * run_var[c]);
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float scale = gamma[c] * rstd;
  chan[c] = scale;
  chan[Co + c] = beta[c] - mean * scale;
  chan[2 * Co + c] = mean;
  chan[3 * Co + c] = rstd;
}

// backward: the sums of dz and dz * ysel -> dbeta, dgamma = (sum dz*ysel - mean * sum dz) * rstd, and the coefficients
// coef [3][Co] = (scale * dbeta / count, scale * rstd * dgamma / count, mean) of fpsg_edgeconv_bwd (zeros in eval mode)
__global__ __launch_bounds__(kFinBwdThreads) void edgeconv_bwd_finalize_kernel(const float* __restrict__ part, int blocks, const float* __restrict__ chan,
                                             double count, int Co, int training, float* __restrict__ dgamma,
                                             float* __restrict__ dbeta, float* __restrict__ coef) {
  int c;
  double s0, s1;
  if (!part_sums<kFinBwdThreads>(part, blocks, Co, c, s0, s1)) return;
  const float scale = chan[c], mean = chan[2 * Co + c], rstd = chan[3 * Co + c];
  const float db = (float)s0;
  const float dg = (float)((s1 - (double)mean * s0) * (double)rstd);
  dbeta[c] = db;
  dgamma[c] = dg;
  coef[c] = training ? (float)((double)scale * db / count) : 0.0f;
  coef[Co + c] = training ? (float)((double)scale * rstd * dg / count) : 0.0f;
  coef[2 * Co + c] = training ? mean : 0.0f;
}

constexpr int kEpThreads = 256;
constexpr int kEpRows = 256;            // rows per workgroup

__global__ __launch_bounds__(kEpThreads) void edgeconv_act_kernel(const float* __restrict__ ysel,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float slope,
                                                                  long rows, int Co, float* __restrict__ out) {
  const int C4 = Co >> 2, c4 = threadIdx.x % C4, rg = threadIdx.x / C4, RG = kEpThreads / C4;
  const v4f sc = reinterpret_cast<const v4f*>(scale)[c4], sh = reinterpret_cast<const v4f*>(shift)[c4];
  const long r1 = ((long)blockIdx.x + 1) * kEpRows < rows ? ((long)blockIdx.x + 1) * kEpRows : rows;
  // four rows' vectors in flight per thread (one per trip was a chain of full round trips)
  for (long r0 = (long)blockIdx.x * kEpRows + rg; r0 < r1; r0 += 4 * RG) {
    v4f yq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = r0 + (long)j * RG;
      yq[j] = r < r1 ? reinterpret_cast<const v4f*>(ysel + r * Co)[c4] : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = r0 + (long)j * RG;
      if (r < r1) {
        v4f o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float z = fma_rn(yq[j][u], sc[u], sh[u]);
          o[u] = z > 0.0f ? z : z * slope;
        }
        reinterpret_cast<v4f*>(out + r * Co)[c4] = o;
      }
    }
  }
}

// backward head: z re-derived with the forward's arithmetic, dz = g * LeakyReLU'(z), dzs = dz * scale[c] written for
// the fused backward kernel, and per channel the partial sums of dz and dz * ysel (-> dbeta, dgamma and the BatchNorm
// coefficients) over the workgroup's rows -> part[blockIdx.x][2][Co] (a thread's rows in order, then the row groups in
// order: deterministic).  torch: addcmul, leaky_relu_backward, sum, mul, sum, mul = twelve passes; here g and ysel are
// read once and dzs written once.
__global__ __launch_bounds__(kEpThreads) void edgeconv_bwd_prep_kernel(const float* __restrict__ g,
                                                                       const float* __restrict__ ysel,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, float slope,
                                                                       long rows, int Co, float* __restrict__ dzs,
                                                                       float* __restrict__ part) {
  __shared__ float red[2][kEpThreads][4];
  const int C4 = Co >> 2, c4 = threadIdx.x % C4, rg = threadIdx.x / C4, RG = kEpThreads / C4;
  const v4f sc = reinterpret_cast<const v4f*>(scale)[c4], sh = reinterpret_cast<const v4f*>(shift)[c4];
  v4f s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const long r1 = ((long)blockIdx.x + 1) * kEpRows < rows ? ((long)blockIdx.x + 1) * kEpRows : rows;
  // four rows' vector pairs in flight per thread, consumed in row order (the sums are unchanged bit for bit)
  for (long r0 = (long)blockIdx.x * kEpRows + rg; r0 < r1; r0 += 4 * RG) {
    v4f yq[4], gq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = r0 + (long)j * RG;
      yq[j] = gq[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      if (r < r1) {
        yq[j] = reinterpret_cast<const v4f*>(ysel + r * Co)[c4];
        gq[j] = reinterpret_cast<const v4f*>(g + r * Co)[c4];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = r0 + (long)j * RG;
      if (r < r1) {
        v4f o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float z = fma_rn(yq[j][u], sc[u], sh[u]);
          const float dz = z > 0.0f ? gq[j][u] : gq[j][u] * slope;
          s0[u] += dz;
          s1[u] = fma_rn(dz, yq[j][u], s1[u]);
          o[u] = dz * sc[u];
        }
        reinterpret_cast<v4f*>(dzs + r * Co)[c4] = o;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) { red[0][threadIdx.x][u] = s0[u]; red[1][threadIdx.x][u] = s1[u]; }
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      v4f t = {red[w][c4][0], red[w][c4][1], red[w][c4][2], red[w][c4][3]};
      for (int q = 1; q < RG; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] += red[w][q * C4 + c4][u];
      reinterpret_cast<v4f*>(part + ((size_t)blockIdx.x * 2 + w) * Co)[c4] = t;
    }
  }
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_edgeconv_blocks(int B, int N, int Co) {
  return fpsg::ec_blocks_per_cloud(N, Co) * B;
}

extern "C" int fpsg_edgeconv_fwd(const float* PQ, const int32_t* idx, const float* sgn, int B, int N,
                                 int k, int Co, float* ysel, uint8_t* jsel, float* s1, float* part,
                                 fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_edgeconv_fwd: B,N,k must be positive (got %d,%d,%d)", B, N, k);
  FPSG_REQUIRE(Co == 64 || Co == 128 || Co == 256, FPSG_E_SHAPE,
               "fpsg_edgeconv_fwd: Co must be 64, 128 or 256 (got %d)", Co);
  FPSG_REQUIRE(k <= 64 && B <= 65535, FPSG_E_LIMIT, "fpsg_edgeconv_fwd: k=%d > 64 or B=%d > 65535", k, B);
  FPSG_REQUIRE_PTR(PQ); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(sgn); FPSG_REQUIRE_PTR(ysel);
  FPSG_REQUIRE(jsel != nullptr, FPSG_E_NULL, "fpsg_edgeconv_fwd: null pointer 'jsel'");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(PQ) & 15) == 0 && (reinterpret_cast<uintptr_t>(ysel) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(jsel) & 3) == 0 && (reinterpret_cast<uintptr_t>(s1) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_edgeconv_fwd: PQ/ysel/s1 must be 16-byte and jsel 4-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid = ec_grid(B, N, Co);
  const int bpc = ec_blocks_per_cloud(N, Co), ppw = ec_points_per_wave(Co);
  const bool one = edgeconv_bwd_one_edge();       // FPSG_EDGECONV_BWD=one_edge also selects the one-neighbour forward
  if (Co == 64 && !one) hipLaunchKernelGGL(edgeconv_fwd_grouped_kernel<16>, grid, dim3(kEcThreads), 0, s, PQ, idx, sgn, B, N, k, bpc, ppw, ysel, jsel, s1, part);
  else if (Co == 128 && !one) hipLaunchKernelGGL(edgeconv_fwd_grouped_kernel<32>, grid, dim3(kEcThreads), 0, s, PQ, idx, sgn, B, N, k, bpc, ppw, ysel, jsel, s1, part);
  else if (Co == 64) hipLaunchKernelGGL(edgeconv_fwd_kernel<1>, grid, dim3(kEcThreads), 0, s, PQ, idx, sgn, B, N, k, bpc, ppw, ysel, jsel, s1, part);
  else if (Co == 128) hipLaunchKernelGGL(edgeconv_fwd_kernel<2>, grid, dim3(kEcThreads), 0, s, PQ, idx, sgn, B, N, k, bpc, ppw, ysel, jsel, s1, part);
  else hipLaunchKernelGGL(edgeconv_fwd_kernel<4>, grid, dim3(kEcThreads), 0, s, PQ, idx, sgn, B, N, k, bpc, ppw, ysel, jsel, s1, part);
  return launch_status("fpsg_edgeconv_fwd");
}

extern "C" int fpsg_edgeconv_bwd(const float* dzs, const uint8_t* jsel, const float* PQ, const float* s1,
                                 const int32_t* rev, const int32_t* off, const float* coef, int B, int N,
                                 int k, int Co, float* dPQ, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_edgeconv_bwd: B,N,k must be positive (got %d,%d,%d)", B, N, k);
  FPSG_REQUIRE(Co == 64 || Co == 128 || Co == 256, FPSG_E_SHAPE,
               "fpsg_edgeconv_bwd: Co must be 64, 128 or 256 (got %d)", Co);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_edgeconv_bwd: B=%d > 65535", B);
  FPSG_REQUIRE_PTR(dzs); FPSG_REQUIRE_PTR(PQ); FPSG_REQUIRE_PTR(rev); FPSG_REQUIRE_PTR(off);
  FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(dPQ);
  FPSG_REQUIRE(jsel != nullptr, FPSG_E_NULL, "fpsg_edgeconv_bwd: null pointer 'jsel'");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(PQ) & 15) == 0 && (reinterpret_cast<uintptr_t>(dzs) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(dPQ) & 15) == 0 && (reinterpret_cast<uintptr_t>(s1) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(coef) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_edgeconv_bwd: float buffers must be 16-byte aligned");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(jsel) & 3) == 0, FPSG_E_ALIGN, "fpsg_edgeconv_bwd: jsel must be 4-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int stats = s1 != nullptr;
  // channel slices: 128 channels at Co = 256 (two slices), whole rows otherwise
  const int nsl = Co == 256 ? 2 : 1;
  // An XCD keeps ~256 of these workgroups resident.  They should all belong to ONE (cloud, slice) pair, or
  // the pairs' working sets (2.25 MB each at 128 channels) evict each other from the 4 MB L2: a pair is cut
  // into ~256 workgroups (2 points per wave at N = 2048; 4 at 64 channels, where two pairs fit).
  const int ppw = Co == 64 ? 4 : 2;
  const int bpc = (N + 4 * ppw - 1) / (4 * ppw);
  const dim3 grid((unsigned)(8 * ((B * nsl + 7) / 8) * bpc));
  if (edgeconv_bwd_one_edge()) {      // FPSG_EDGECONV_BWD=one_edge: round 2's form, one in-edge per load (A/B timing)
    if (Co == 64) hipLaunchKernelGGL(edgeconv_bwd_kernel<1>, grid, dim3(kEcThreads), 0, s, dzs, jsel, PQ, s1, rev, off, coef, B, N, k, bpc, ppw, nsl, stats, dPQ);
    else hipLaunchKernelGGL(edgeconv_bwd_kernel<2>, grid, dim3(kEcThreads), 0, s, dzs, jsel, PQ, s1, rev, off, coef, B, N, k, bpc, ppw, nsl, stats, dPQ);
  } else {
    if (Co == 64) hipLaunchKernelGGL(edgeconv_bwd_grouped_kernel<16>, grid, dim3(kEcThreads), 0, s, dzs, jsel, PQ, s1, rev, off, coef, B, N, k, bpc, ppw, nsl, stats, dPQ);
    else hipLaunchKernelGGL(edgeconv_bwd_grouped_kernel<32>, grid, dim3(kEcThreads), 0, s, dzs, jsel, PQ, s1, rev, off, coef, B, N, k, bpc, ppw, nsl, stats, dPQ);
  }
  return launch_status("fpsg_edgeconv_bwd");
}

extern "C" int fpsg_edgeconv_prep_blocks(long rows) {
  return rows > 0 ? (int)((rows + fpsg::kEpRows - 1) / fpsg::kEpRows) : 0;
}

static int edgeconv_ep_check(const char* fn, long rows, int Co) {
  using namespace fpsg;
  FPSG_REQUIRE(rows > 0 && (rows + kEpRows - 1) / kEpRows < (1L << 31), FPSG_E_SHAPE, "%s: rows must be positive (got %ld)", fn, rows);
  FPSG_REQUIRE(Co == 64 || Co == 128 || Co == 256, FPSG_E_SHAPE, "%s: Co must be 64, 128 or 256 (got %d)", fn, Co);
  return 0;
}

extern "C" int fpsg_edgeconv_act(const float* ysel, const float* scale, const float* shift, float slope, long rows,
                                 int Co, float* out, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = edgeconv_ep_check("fpsg_edgeconv_act", rows, Co);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(ysel); FPSG_REQUIRE_PTR(scale); FPSG_REQUIRE_PTR(shift); FPSG_REQUIRE_PTR(out);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(ysel) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                 reinterpret_cast<uintptr_t>(out)) & 15) == 0, FPSG_E_ALIGN, "fpsg_edgeconv_act: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(edgeconv_act_kernel, dim3((unsigned)((rows + kEpRows - 1) / kEpRows)), dim3(kEpThreads), 0,
                     static_cast<hipStream_t>(stream), ysel, scale, shift, slope, rows, Co, out);
  return launch_status("fpsg_edgeconv_act");
}

extern "C" int fpsg_edgeconv_bwd_prep(const float* g, const float* ysel, const float* scale, const float* shift,
                                      float slope, long rows, int Co, float* dzs, float* part, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = edgeconv_ep_check("fpsg_edgeconv_bwd_prep", rows, Co);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(g); FPSG_REQUIRE_PTR(ysel); FPSG_REQUIRE_PTR(scale); FPSG_REQUIRE_PTR(shift); FPSG_REQUIRE_PTR(dzs);
  FPSG_REQUIRE_PTR(part);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ysel) | reinterpret_cast<uintptr_t>(scale) |
                 reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(dzs) | reinterpret_cast<uintptr_t>(part)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_edgeconv_bwd_prep: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(edgeconv_bwd_prep_kernel, dim3((unsigned)((rows + kEpRows - 1) / kEpRows)), dim3(kEpThreads), 0,
                     static_cast<hipStream_t>(stream), g, ysel, scale, shift, slope, rows, Co, dzs, part);
  return launch_status("fpsg_edgeconv_bwd_prep");
}

extern "C" int fpsg_edgeconv_stats_finalize(const float* part, int blocks, const float* gamma, const float* beta,
                                            float* running_mean, float* running_var, float momentum, float eps,
                                            double count, int Co, int training, float* chan, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(Co > 0 && count > 0.0, FPSG_E_SHAPE, "fpsg_edgeconv_stats_finalize: Co and count must be positive (got %d, %g)", Co, count);
  FPSG_REQUIRE_PTR(gamma); FPSG_REQUIRE_PTR(beta); FPSG_REQUIRE_PTR(chan);
  if (training) { FPSG_REQUIRE_PTR(part); FPSG_REQUIRE(blocks > 0, FPSG_E_SHAPE, "fpsg_edgeconv_stats_finalize: blocks must be positive (got %d)", blocks); }
  else { FPSG_REQUIRE_PTR(running_mean); FPSG_REQUIRE_PTR(running_var); }
  hipLaunchKernelGGL(edgeconv_stats_finalize_kernel, dim3((unsigned)((Co + kFinCh - 1) / kFinCh)), dim3(kFinThreads), 0, static_cast<hipStream_t>(stream),
                     part, blocks, gamma, beta, running_mean, running_var, momentum, eps, count, Co, training, chan);
  return launch_status("fpsg_edgeconv_stats_finalize");
}

extern "C" size_t fpsg_edgeconv_stats_ws_floats(int blocks, int Co) {
  if (blocks <= 0 || Co <= 0) return 0;
  return (size_t)fpsg::stats_slices(blocks, Co) * 2 * Co * 2;          // [Z][2 Co] doubles
}

extern "C" int fpsg_edgeconv_stats_finalize_ws(const float* part, int blocks, const float* gamma, const float* beta,
                                               float* running_mean, float* running_var, float momentum, float eps,
                                               double count, int Co, int training, float* chan, float* ws,
                                               fpsg_stream_t stream) {
  using namespace fpsg;
  if (!training || ws == nullptr || blocks < 256)       // nothing to sum / few rows: the one-launch form
    return fpsg_edgeconv_stats_finalize(part, blocks, gamma, beta, running_mean, running_var, momentum, eps, count, Co,
                                        training, chan, stream);
  FPSG_REQUIRE(Co > 0 && count > 0.0, FPSG_E_SHAPE, "fpsg_edgeconv_stats_finalize_ws: Co and count must be positive (got %d, %g)", Co, count);
  FPSG_REQUIRE_PTR(part); FPSG_REQUIRE_PTR(gamma); FPSG_REQUIRE_PTR(beta); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7) == 0, FPSG_E_ALIGN, "fpsg_edgeconv_stats_finalize_ws: ws must be 8-byte aligned");
  const int Z = stats_slices(blocks, Co);
  double* slices = reinterpret_cast<double*>(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(edgeconv_stats_slices_kernel, dim3((unsigned)Z, (unsigned)((2 * Co + kS1Cols - 1) / kS1Cols)), dim3(kS1Threads), 0,
                     s, part, blocks, Co, Z, slices);
  int rc = launch_status("fpsg_edgeconv_stats_finalize_ws(slices)");
  if (rc) return rc;
  hipLaunchKernelGGL(edgeconv_stats_finalize2_kernel, dim3((unsigned)((Co + 3) / 4)), dim3(256), 0, s, slices, Z, gamma, beta,
                     running_mean, running_var, momentum, eps, count, Co, chan);
  return launch_status("fpsg_edgeconv_stats_finalize_ws");
}

extern "C" int fpsg_edgeconv_bwd_finalize(const float* part, int blocks, const float* chan, double count, int Co,
                                          int training, float* dgamma, float* dbeta, float* coef, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(Co > 0 && count > 0.0 && blocks > 0, FPSG_E_SHAPE,
               "fpsg_edgeconv_bwd_finalize: Co, blocks and count must be positive (got %d, %d, %g)", Co, blocks, count);
  FPSG_REQUIRE_PTR(part); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef);
  hipLaunchKernelGGL(edgeconv_bwd_finalize_kernel, dim3((unsigned)((Co + kFinCh - 1) / kFinCh)), dim3(kFinBwdThreads), 0, static_cast<hipStream_t>(stream),
                     part, blocks, chan, count, Co, training, dgamma, dbeta, coef);
  return launch_status("fpsg_edgeconv_bwd_finalize");
}

// ------------------------------------------------------------------------------------------------
// The in-edge lists the backward gathers through: edges e = n*k + j grouped by destination idx[n][j], ascending e
// inside a group (the summation order of the backward, and of the oracle).  The reference reaches this through
// autograd's index backward of get_graph_feature's gather (dgcnn/model.py:30-56), which scatters with atomics.
//
// One workgroup per cloud, a stable counting sort: wave w owns the contiguous edge range [w*per_wave, ...), counts its
// destinations in its own row of 16-bit counters (two per LDS word), the rows are scanned per destination (wave 0's
// edges first) and over the destinations, then every wave places its edges in order, 64 at a time.  Lanes of one group
// that share a destination are served lowest lane first: pending lanes race for a tag with ds_min on the lane id, the
// winner takes the next slot of the destination and releases the tag.  The tag table is hashed (256-1024 entries per wave):
// a lane that loses to another destination's lane just waits a round.  No result depends on timing.
namespace fpsg {
namespace {

constexpr int kRgWaves = 16;
constexpr int kRgThreads = 64 * kRgWaves;
constexpr int kRgMinTags = 256, kRgMaxTags = 1024;   // per-wave tag table: as large as the LDS allows (fewer lost rounds)
constexpr int kRgMaxPairs = 2;                 // destination pairs per thread in the scan: N <= 4096

inline size_t rg_lds_bytes(int N, int tags) {
  return ((size_t)(N + 1) + kRgWaves + (size_t)kRgWaves * tags + (size_t)kRgWaves * ((N + 1) / 2)) * 4;
}
inline int rg_tags(int N) {
  int tags = kRgMaxTags;
  while (tags > kRgMinTags && rg_lds_bytes(N, tags) > 160 * 1024) tags >>= 1;
  return tags;
}

__global__ __launch_bounds__(kRgThreads) void reverse_graph_kernel(const int32_t* __restrict__ idx, int N, int k,
                                                                   int per_wave, int n_tags,
                                                                   int32_t* __restrict__ rev,
                                                                   int32_t* __restrict__ off) {
  extern __shared__ __attribute__((aligned(16))) unsigned rg_lds[];
  const int E = N * k;
  const int Nh = (N + 1) >> 1;
  int* loff = reinterpret_cast<int*>(rg_lds);                          // [N + 1] first slot of a destination
  int* wsum = loff + (N + 1);                                          // [kRgWaves]
  unsigned* tags = reinterpret_cast<unsigned*>(wsum + kRgWaves);       // [kRgWaves][n_tags]
  unsigned* cur = tags + kRgWaves * n_tags;                             // [kRgWaves][Nh]: count, then cursor (u16 x 2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t* __restrict__ ix = idx + (size_t)blockIdx.x * E;
  int32_t* __restrict__ rv = rev + (size_t)blockIdx.x * E;
  int32_t* __restrict__ of = off + (size_t)blockIdx.x * (N + 1);

  for (int i = tid; i < kRgWaves * Nh; i += kRgThreads) cur[i] = 0;
  for (int i = tid; i < kRgWaves * n_tags; i += kRgThreads) tags[i] = 0xffffffffu;
  __syncthreads();

  // ---- counts of this wave's edge range
  unsigned* mycur = cur + wave * Nh;
  const int e0 = wave * per_wave;
  const int e1 = e0 + per_wave < E ? e0 + per_wave : E;
  for (int e = e0 + lane; e < e1; e += 256) {          // four loads in flight
    int d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = e + 64 * u < e1 ? ix[e + 64 * u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((unsigned)d[u] < (unsigned)N) atomicAdd(&mycur[d[u] >> 1], 1u << ((d[u] & 1) * 16));
  }
  __syncthreads();

  // ---- per destination: the waves' counts become each wave's first rank; in-degrees scanned over the destinations
  const int pp = (Nh + kRgThreads - 1) / kRgThreads;                  // pairs per thread (<= kRgMaxPairs)
  int deg[2 * kRgMaxPairs];
  int local = 0;
#pragma unroll
  for (int u = 0; u < kRgMaxPairs; ++u) {
    unsigned run = 0;                                                  // both halves at once: no carry below 65536
    const int pair = tid * pp + u;
    if (u < pp && pair < Nh) {
      for (int w = 0; w < kRgWaves; ++w) {
        const unsigned c = cur[w * Nh + pair];
        cur[w * Nh + pair] = run;
        run += c;
      }
    }
    deg[2 * u] = (int)(run & 0xffffu);
    deg[2 * u + 1] = (int)(run >> 16);
    local += deg[2 * u] + deg[2 * u + 1];
  }
  int incl = local;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int o = __shfl_up(incl, s, 64);
    if (lane >= s) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - local;
  int total = 0;
  for (int w = 0; w < kRgWaves; ++w) {
    const int t = wsum[w];
    base += w < wave ? t : 0;
    total += t;
  }
#pragma unroll
  for (int u = 0; u < kRgMaxPairs; ++u) {
    const int pair = tid * pp + u;
    if (u < pp && pair < Nh) {
      const int d = 2 * pair;
      loff[d] = base; of[d] = base;
      base += deg[2 * u];
      if (d + 1 < N) { loff[d + 1] = base; of[d + 1] = base; }
      base += deg[2 * u + 1];
    }
  }
  if (tid == 0) of[N] = total;
  __syncthreads();

  // ---- placement, this wave's edges in ascending order
  unsigned* mytag = tags + wave * n_tags;
  int d_next = e0 + lane < e1 ? ix[e0 + lane] : -1;
  for (int g = e0; g < e1; g += 64) {
    const int e = g + lane;
    const int d = d_next;
    d_next = e + 64 < e1 ? ix[e + 64] : -1;          // the next group's load flies under this group's rounds
    bool pending = (unsigned)d < (unsigned)N;
    const int h = d & (n_tags - 1);
    while (__ballot(pending) != 0ull) {
      if (pending) __hip_atomic_fetch_min(&mytag[h], (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const bool win = pending && __hip_atomic_load(&mytag[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == (unsigned)lane;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (win) {
        const int sh = (d & 1) * 16;
        const unsigned old = atomicAdd(&mycur[d >> 1], 1u << sh);
        rv[loff[d] + (int)((old >> sh) & 0xffffu)] = e;
        __hip_atomic_store(&mytag[h], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        pending = false;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_edgeconv_reverse_graph_fits(int N, int k) {
  return N > 0 && k > 0 && (long)N * k <= 65535 && (N + 1) / 2 <= fpsg::kRgThreads * fpsg::kRgMaxPairs &&
         fpsg::rg_lds_bytes(N, fpsg::kRgMinTags) <= 160 * 1024;
}

extern "C" int fpsg_edgeconv_reverse_graph(const int32_t* idx, int B, int N, int k, int32_t* rev, int32_t* off,
                                           fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && k > 0, FPSG_E_SHAPE, "fpsg_edgeconv_reverse_graph: B,N,k must be positive (got %d,%d,%d)", B, N, k);
  FPSG_REQUIRE(fpsg_edgeconv_reverse_graph_fits(N, k), FPSG_E_LIMIT,
               "fpsg_edgeconv_reverse_graph: N*k = %ld edges per cloud (limit 65535: 16-bit ranks) or N = %d too large for the LDS",
               (long)N * k, N);
  FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(rev); FPSG_REQUIRE_PTR(off);
  const int n_tags = rg_tags(N);
  const size_t lds = rg_lds_bytes(N, n_tags);
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(reverse_graph_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("fpsg_edgeconv_reverse_graph: %s", hipGetErrorString(e)); return (int)e; }
  }
  const int E = N * k;
  const int per_wave = ((E + kRgWaves - 1) / kRgWaves + 63) & ~63;
  hipLaunchKernelGGL(reverse_graph_kernel, dim3((unsigned)B), dim3(kRgThreads), lds, static_cast<hipStream_t>(stream),
                     idx, N, k, per_wave, n_tags, rev, off);
  return launch_status("fpsg_edgeconv_reverse_graph");
}
