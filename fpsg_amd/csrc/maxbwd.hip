// maxbwd.hip -- K5m: the backward of  conv1x1 -> training-mode BatchNorm (+ReLU) -> max over the points  (the tail of
// PointNet's shared MLPs, reference src/pointnet/model.py:35-37 and 222-224) WITHOUT the dense [B,C,L] gradient, gfx950.
//
// With x' = W a + pb  (a [B,K,L] the layer's input, W [C,K], pb [C] the convolution bias) the gradient of x' is
//     dx'[b,c,l] = k1_c dz[b,c] [l = sel(b,c)] + k2_c x'[b,c,l] + k3_c
// (fpsg_bn_act_max_bwd_coef: dz is the [B,C] gradient through the activation at each row's selected element, k1..k3 the
// BatchNorm-backward coefficients).  The library path materialises dx' (537 MB at B = 64, C = 1024, L = 2048) and runs two
// GEMMs over it (dW = dx' a^T, da = W^T dx': 34 GFLOP each).  Both reduce to K x K algebra plus one gather and one scatter
// of B*C columns:
//     dW = k1 (.) S + diag(k2) (W G + pb s^T) + k3 s^T          G = sum_{b,l} a a^T [K,K],  s = sum_{b,l} a [K],
//                                                                S[c,:] = sum_b dz[b,c] a[b,:,sel(b,c)]
//     da = (W^T diag(k2) W) a + v 1^T + scatter,                 v = W^T (k2 (.) pb + k3),
//          scatter: da[b,:,sel(b,c)] += k1_c dz[b,c] W[c,:]  for every (b,c)
// G, the K x K products and (W^T diag(k2) W) a are library GEMMs (8.6 GFLOP instead of 68); this file holds
//   max_bwd_gather_kernel : S   (sums over b in ascending order)
//   max_bwd_scatter_kernel: da += v + scatter   (a point's contributions in ascending channel order)
// Deterministic: fixed summation orders, no float atomics.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kMbThreads = 256;

// S[c,k] = sum over b (ascending) of dz[b,c] * a[b,k,idx[b,c]].  Workgroup (k, quarter of the channels): the rows
// a[b .. b+3, k, :] of four clouds (L floats each, read coalesced) are staged in LDS per step, two buffers, the next
// step's rows requested before this step's gathers; thread t gathers for its channel c = quarter * 256 + t from LDS.
// (Gathering straight from global memory touched a 64-byte sector per 4-byte element: 69 us at B = 64, K = 128,
// C = 1024, L = 2048; one cloud per step was a chain of 64 barriers: 80 us.)  4 L floats per buffer: L <= kGatherL.
constexpr int kGatherL = 2048;     // points of a row held in LDS (2 buffers x 4 clouds x 8 KB = 64 KB)
constexpr int kGatherNB = 4;       // clouds per step

__global__ __launch_bounds__(kMbThreads) void max_bwd_gather_kernel(const float* __restrict__ a, const float* __restrict__ dz,
                                                                   const int32_t* __restrict__ idx, int B, int K, int C,
                                                                   int L, float* __restrict__ S,
                                                                   float* __restrict__ spart /*[B][K] or null*/) {
  extern __shared__ __attribute__((aligned(16))) float rows[];        // [2][kGatherNB][L]
  constexpr int kPre = kGatherNB * kGatherL / 4 / kMbThreads;         // 16-byte vectors a thread moves per step (8)
  const int k = blockIdx.x;
  const int c = blockIdx.y * kMbThreads + threadIdx.x;
  const bool live = c < C;
  const int L4 = L >> 2;                                     // L % 4 == 0 (checked by the launcher)
  const int per_step = kGatherNB * L4;                       // vectors of a full step
  float acc = 0.0f;
  auto fetch = [&](int b0, v4f (&pre)[kPre]) {               // rows of clouds b0 .. b0+3 -> registers
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int e = threadIdx.x + u * kMbThreads;
      const int j = e / L4, q = e - j * L4;
      pre[u] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      if (e < per_step && b0 + j < B) pre[u] = reinterpret_cast<const v4f*>(a + ((size_t)(b0 + j) * K + k) * L)[q];
    }
  };
  auto park = [&](float* buf, const v4f (&pre)[kPre]) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int e = threadIdx.x + u * kMbThreads;
      if (e < per_step) reinterpret_cast<v4f*>(buf)[e] = pre[u];
    }
  };
  v4f pre[kPre];
  fetch(0, pre);
  park(rows, pre);
  __syncthreads();
  int step = 0;
  for (int b0 = 0; b0 < B; b0 += kGatherNB, ++step) {
    const float* cur = rows + (size_t)(step & 1) * kGatherNB * L;
    float* nxt = rows + (size_t)((step + 1) & 1) * kGatherNB * L;
    const bool more = b0 + kGatherNB < B;
    if (more) fetch(b0 + kGatherNB, pre);                    // in flight during this step's gathers
    if (spart && blockIdx.y == 0) {                          // the row sums of a (s = sum a): wave w sums cloud b0 + w's staged row
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
      if (w < kGatherNB && b0 + w < B) {
        float t = 0.0f;
        for (int l = lane; l < L; l += 64) t += cur[w * L + l];
        t = wave_sum(t);
        if (lane == 0) spart[(size_t)(b0 + w) * K + k] = t;
      }
    }
    if (live) {
      float d[kGatherNB];
      int sel[kGatherNB];
#pragma unroll
      for (int j = 0; j < kGatherNB; ++j) {
        const bool in = b0 + j < B;
        const size_t row = (size_t)(in ? b0 + j : b0) * C + c;
        d[j] = in ? dz[row] : 0.0f;
        sel[j] = in ? idx[row] : 0;
      }
#pragma unroll
      for (int j = 0; j < kGatherNB; ++j) acc = fma_rn(d[j], cur[j * L + sel[j]], acc);      // ascending b (a missing cloud adds +0)
    }
    if (more) park(nxt, pre);
    __syncthreads();
  }
  if (live) S[(size_t)c * K + k] = acc;
}

// ---- scatter: da[b,k,l] += v[k] + sum over the channels c with idx[b,c] = l, ascending c, of (k1[c] dz[b,c]) W[c,k] ----
// Two launches.  max_bwd_sort_kernel: one workgroup per cloud sorts its C (point, channel) pairs by (point, channel) --
// keys point << 16 | channel-in-pass, a bitonic network over 1024-key passes in LDS -- and leaves the sorted keys and
// the pairs' coefficients k1[c] dz[b,c] in the caller's scratch.  max_bwd_scatter_kernel: workgroup (tile of 128 points,
// cloud): the tile's pairs are a contiguous range of every pass's sorted keys (two binary searches); wave w sums the
// runs of the points w, w + 4, ... in ascending channel order into the LDS tile D[l][k] (lanes across k, up to eight
// rows of W in flight), and the tile is then added to da by rows of k (coalesced 512-byte segments; row stride K + 1:
// conflict-free LDS reads).
constexpr int kTL = 128;          // points per tile
constexpr int kCP = 1024;         // keys sorted at a time (channels per pass)

constexpr int kSortThreadsMb = 512;   // one compare-exchange per thread and stage

__global__ __launch_bounds__(kSortThreadsMb) void max_bwd_sort_kernel(const int32_t* __restrict__ idx, const float* __restrict__ k1,
                                                                     const float* __restrict__ dz, int C, int passes, int tiles,
                                                                     unsigned* __restrict__ skeys /*[B][passes][kCP]*/,
                                                                     float* __restrict__ scoef /*[B][passes][kCP]*/,
                                                                     int* __restrict__ toff /*[B][passes][tiles + 1]*/) {
  __shared__ unsigned keys[kCP];
  constexpr int kMbThreads = kSortThreadsMb;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int ps = 0; ps < passes; ++ps) {
    const int c0 = ps * kCP;
    for (int e = tid; e < kCP; e += kMbThreads) {
      const int c = c0 + e;
      keys[e] = c < C ? (((unsigned)idx[(size_t)b * C + c] << 16) | (unsigned)e) : 0xffffffffu;
    }
    __syncthreads();
    for (int span = 2; span <= kCP; span <<= 1) {           // bitonic sort, ascending
      for (int j = span >> 1; j > 0; j >>= 1) {
        for (int e = tid; e < kCP; e += kMbThreads) {
          const int p = e ^ j;
          if (p > e) {
            const unsigned x = keys[e], y = keys[p];
            const bool up = (e & span) == 0;
            if ((x > y) == up) { keys[e] = y; keys[p] = x; }
          }
        }
        __syncthreads();
      }
    }
    for (int e = tid; e < kCP; e += kMbThreads) {
      const unsigned key = keys[e];
      const size_t o = ((size_t)b * passes + ps) * kCP + e;
      skeys[o] = key;
      float cf = 0.0f;
      if (key != 0xffffffffu) {
        const int c = c0 + (int)(key & 0xffffu);
        cf = k1[c] * dz[(size_t)b * C + c];
      }
      scoef[o] = cf;
    }
    // first sorted position of every tile of kTL points (and the end): a tile's pairs are [toff[t], toff[t + 1])
    for (int t = tid; t <= tiles; t += kMbThreads) {
      const unsigned long long want = (unsigned long long)t * kTL << 16;
      int lo = 0, hi = kCP;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned long long)keys[mid] < want) lo = mid + 1; else hi = mid; }
      toff[((size_t)b * passes + ps) * (tiles + 1) + t] = lo;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kMbThreads) void max_bwd_scatter_kernel(float* __restrict__ da, const float* __restrict__ W,
                                                                    const unsigned* __restrict__ skeys,
                                                                    const float* __restrict__ scoef,
                                                                    const int* __restrict__ toff, const float* __restrict__ v,
                                                                    int K, int C, int L, int passes, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ld = K + 1;
  float* D = lds;                                           // [kTL][K + 1]
  unsigned* ek = reinterpret_cast<unsigned*>(lds + kTL * ld);   // [kCP] the tile's pairs of a pass
  float* ec = lds + kTL * ld + kCP;                         // [kCP] their coefficients
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * kTL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < kTL * ld; e += kMbThreads) D[e] = 0.0f;
  for (int ps = 0; ps < passes; ++ps) {                     // ascending passes keep a point's channel order
    const size_t base = ((size_t)b * passes + ps) * kCP;
    const int* tf = toff + ((size_t)b * passes + ps) * (tiles + 1) + blockIdx.x;
    const int first = tf[0], n = tf[1] - first;
    const int c0 = ps * kCP;
    __syncthreads();                                        // D zeroed / the previous pass's pairs consumed
    for (int e = tid; e < n; e += kMbThreads) { ek[e] = skeys[base + first + e]; ec[e] = scoef[base + first + e]; }
    __syncthreads();
    // every wave walks the tile's pairs; a pair belongs to the wave of its point (l % 4): runs are summed in order,
    // the rows of W of up to eight of the wave's pairs in flight
    int q = 0;
    while (q < n) {
      int ql[8];
      float cf[8], w0[8], w1[8];
      int m = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) { ql[u] = -1; cf[u] = 0.0f; w0[u] = 0.0f; w1[u] = 0.0f; }
      while (q < n && m < 8) {                              // wave-uniform walk over LDS
        const unsigned key = ek[q];
        const int l = (int)(key >> 16) - l0;
        if ((l & 3) == wave) {
          const int c = c0 + (int)(key & 0xffffu);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (u == m) {
              ql[u] = l;
              cf[u] = ec[q];
              w0[u] = lane < K ? W[(size_t)c * K + lane] : 0.0f;
              w1[u] = lane + 64 < K ? W[(size_t)c * K + lane + 64] : 0.0f;
            }
          }
          ++m;
        }
        ++q;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (ql[u] < 0) continue;                            // wave-uniform
        if (lane < K) D[ql[u] * ld + lane] = fma_rn(cf[u], w0[u], D[ql[u] * ld + lane]);
        if (lane + 64 < K) D[ql[u] * ld + lane + 64] = fma_rn(cf[u], w1[u], D[ql[u] * ld + lane + 64]);
      }
    }
  }
  __syncthreads();
  // da rows: wave w takes k = w, w + 4, ...; a lane moves the two adjacent points 2 lane, 2 lane + 1 of a row as one
  // 8-byte access (a row of the tile = 512 contiguous bytes per wave instruction); eight rows' loads in flight
  const bool pair_ok = (L & 1) == 0;                        // rows 8-byte aligned (L even): the vector form
  for (int k0 = wave; k0 < K; k0 += 8 * (kMbThreads / 64)) {
    v2f r[8];
    const int l = 2 * lane;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * (kMbThreads / 64);
      r[u] = (v2f){0.0f, 0.0f};
      if (k < K) {
        const float* row = da + ((size_t)b * K + k) * L + l0;
        if (pair_ok && l0 + l + 1 < L) r[u] = *reinterpret_cast<const v2f*>(row + l);
        else { if (l0 + l < L) r[u][0] = row[l]; if (l0 + l + 1 < L) r[u][1] = row[l + 1]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * (kMbThreads / 64);
      if (k >= K) continue;
      const float vk = v[k];
      float* row = da + ((size_t)b * K + k) * L + l0;
      const v2f o = {r[u][0] + (vk + D[l * ld + k]), r[u][1] + (vk + D[(l + 1) * ld + k])};
      if (pair_ok && l0 + l + 1 < L) *reinterpret_cast<v2f*>(row + l) = o;
      else { if (l0 + l < L) row[l] = o[0]; if (l0 + l + 1 < L) row[l + 1] = o[1]; }
    }
  }
}

// The per-channel scalings around the K x K products, one launch each instead of a dozen elementwise ones:
//   prep : Wk[c,k] = k2[c] W[c,k];  u[c] = k2[c] pb[c] + k3[c];  dpb[c] = k1[c] sum_b dz[b,c] + (k2[c] mean[c] + k3[c]) B L
//          s[k] = sum_b spart[b,k] (ascending b; spart = the row sums of a)
//   dw   : dw[c,k] = k1[c] S[c,k] + k2[c] (WG[c,k] + pb[c] s[k]) + k3[c] s[k]
__global__ __launch_bounds__(kMbThreads) void max_bwd_prep_kernel(const float* __restrict__ W, const float* __restrict__ coef,
                                                                 const float* __restrict__ pb, const float* __restrict__ mean,
                                                                 const float* __restrict__ dz, const float* __restrict__ spart,
                                                                 int B, int K, int C, float count, float* __restrict__ Wk,
                                                                 float* __restrict__ u, float* __restrict__ dpb,
                                                                 float* __restrict__ s_out) {
  const long e = (long)blockIdx.x * kMbThreads + threadIdx.x;
  if (e >= (long)C * K) return;
  const int c = (int)(e / K), k = (int)(e - (long)c * K);
  const float k1 = coef[c], k2 = coef[C + c], k3 = coef[2 * C + c];
  Wk[e] = k2 * W[e];
  if (k == 0) {
    const float b = pb ? pb[c] : 0.0f;
    u[c] = fma_rn(k2, b, k3);
    if (dpb) {
      float sum = 0.0f;
      int n = 0;
      for (; n + 8 <= B; n += 8) {                           // eight loads in flight, added in ascending order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = dz[(size_t)(n + j) * C + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += t[j];
      }
      for (; n < B; ++n) sum += dz[(size_t)n * C + c];
      dpb[c] = fma_rn(k1, sum, fma_rn(k2, mean[c], k3) * count);
    }
  }
  if (c == 0 && spart) {                                     // s[k] = sum over the clouds of a's row sums, ascending
    float sk = 0.0f;
    int n = 0;
    for (; n + 8 <= B; n += 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = spart[(size_t)(n + j) * K + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) sk += t[j];
    }
    for (; n < B; ++n) sk += spart[(size_t)n * K + k];
    s_out[k] = sk;
  }
}

__global__ __launch_bounds__(kMbThreads) void max_bwd_dw_kernel(const float* __restrict__ S, const float* __restrict__ WG,
                                                               const float* __restrict__ coef, const float* __restrict__ pb,
                                                               const float* __restrict__ spart, int B, int K, int C,
                                                               float* __restrict__ dw) {
  const long e = (long)blockIdx.x * kMbThreads + threadIdx.x;
  if (e >= (long)C * K) return;
  const int c = (int)(e / K), k = (int)(e - (long)c * K);
  const float k1 = coef[c], k2 = coef[C + c], k3 = coef[2 * C + c];
  float r = k1 * S[e];
  if (WG) {                                                 // training mode: the dense part
    const float sk = spart[k];                               // s[k] (fpsg_max_bwd_prep)
    const float b = pb ? pb[c] : 0.0f;
    r = fma_rn(k2, fma_rn(b, sk, WG[e]), r);
    r = fma_rn(k3, sk, r);
  }
  dw[e] = r;
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_max_bwd_gather(const float* a, const float* dz, const int32_t* idx, int B, int K, int C, int L,
                                   float* S, float* spart, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && K > 0 && C > 0 && L > 0, FPSG_E_SHAPE, "fpsg_max_bwd_gather: B,K,C,L must be positive (got %d,%d,%d,%d)",
               B, K, C, L);
  FPSG_REQUIRE(K <= 65535 && (C + 255) / 256 <= 65535, FPSG_E_LIMIT, "fpsg_max_bwd_gather: C=%d or K=%d beyond the grid", C, K);
  FPSG_REQUIRE_PTR(a); FPSG_REQUIRE_PTR(dz); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(S);
  FPSG_REQUIRE(L % 4 == 0 && L <= kGatherL, FPSG_E_LIMIT, "fpsg_max_bwd_gather: L=%d must be a multiple of 4, at most %d", L, kGatherL);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(a) & 15) == 0, FPSG_E_ALIGN, "fpsg_max_bwd_gather: a must be 16-byte aligned");
  dim3 grid(K, (C + kMbThreads - 1) / kMbThreads);
  const size_t lds_bytes = (size_t)2 * kGatherNB * L * sizeof(float);
  FPSG_REQUIRE(!misaligned4(spart), FPSG_E_ALIGN, "fpsg_max_bwd_gather: spart not 4-byte aligned");
  hipLaunchKernelGGL(max_bwd_gather_kernel, grid, dim3(kMbThreads), lds_bytes, static_cast<hipStream_t>(stream), a, dz, idx, B, K, C,
                     L, S, spart);
  return launch_status("fpsg_max_bwd_gather");
}

extern "C" size_t fpsg_max_bwd_scatter_workspace_floats(int B, int C, int L) {
  if (B <= 0 || C <= 0 || L <= 0) return 0;
  const size_t passes = (C + fpsg::kCP - 1) / fpsg::kCP, tiles = (L + fpsg::kTL - 1) / fpsg::kTL;
  return (size_t)B * passes * (2 * fpsg::kCP + tiles + 1);                   // sorted keys, coefficients, tile offsets
}

extern "C" int fpsg_max_bwd_scatter(float* da, const float* W, const float* k1, const float* dz, const int32_t* idx,
                                    const float* v, int B, int K, int C, int L, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && K > 0 && C > 0 && L > 0, FPSG_E_SHAPE, "fpsg_max_bwd_scatter: B,K,C,L must be positive (got %d,%d,%d,%d)",
               B, K, C, L);
  FPSG_REQUIRE(K <= 128, FPSG_E_LIMIT, "fpsg_max_bwd_scatter: K=%d beyond 128 input channels (two per lane)", K);
  FPSG_REQUIRE(L <= 65535 - kTL && B <= 65535, FPSG_E_LIMIT,
               "fpsg_max_bwd_scatter: B=%d or L=%d beyond the 16-bit point field of the sort keys / the grid", B, L);
  FPSG_REQUIRE_PTR(da); FPSG_REQUIRE_PTR(W); FPSG_REQUIRE_PTR(k1); FPSG_REQUIRE_PTR(dz); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(v);
  FPSG_REQUIRE_PTR(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int passes = (C + kCP - 1) / kCP, tiles = (L + kTL - 1) / kTL;
  unsigned* skeys = reinterpret_cast<unsigned*>(ws);
  float* scoef = ws + (size_t)B * passes * kCP;
  int* toff = reinterpret_cast<int*>(ws + (size_t)2 * B * passes * kCP);
  hipLaunchKernelGGL(max_bwd_sort_kernel, dim3(B), dim3(kSortThreadsMb), 0, s, idx, k1, dz, C, passes, tiles, skeys, scoef, toff);
  int rc = launch_status("fpsg_max_bwd_scatter(sort)");
  if (rc) return rc;
  const size_t lds_bytes = ((size_t)kTL * (K + 1) + 2 * kCP) * sizeof(float);
  if (lds_bytes > 65536) {                                  // dynamic LDS beyond 64 KiB has to be requested (no state kept)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(max_bwd_scatter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("fpsg_max_bwd_scatter: %s", hipGetErrorString(e)); return (int)e; }
  }
  dim3 grid((L + kTL - 1) / kTL, B);
  hipLaunchKernelGGL(max_bwd_scatter_kernel, grid, dim3(kMbThreads), lds_bytes, s, da, W, skeys, scoef, toff, v, K, C, L, passes, tiles);
  return launch_status("fpsg_max_bwd_scatter");
}

extern "C" int fpsg_max_bwd_prep(const float* W, const float* coef, const float* pre_bias, const float* mean, const float* dz,
                                 const float* spart, int B, int K, int C, int L, float* Wk, float* u, float* dpre_bias,
                                 float* s, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && K > 0 && C > 0 && L > 0, FPSG_E_SHAPE, "fpsg_max_bwd_prep: B,K,C,L must be positive (got %d,%d,%d,%d)", B, K, C, L);
  FPSG_REQUIRE_PTR(W); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(mean); FPSG_REQUIRE_PTR(dz); FPSG_REQUIRE_PTR(Wk); FPSG_REQUIRE_PTR(u);
  FPSG_REQUIRE(!misaligned4(pre_bias) && !misaligned4(dpre_bias) && !misaligned4(spart) && !misaligned4(s), FPSG_E_ALIGN,
               "fpsg_max_bwd_prep: pre_bias / dpre_bias / spart / s not 4-byte aligned");
  FPSG_REQUIRE((spart == nullptr) == (s == nullptr), FPSG_E_NULL, "fpsg_max_bwd_prep: spart and s go together");
  const long n = (long)C * K;
  hipLaunchKernelGGL(max_bwd_prep_kernel, dim3((unsigned)((n + kMbThreads - 1) / kMbThreads)), dim3(kMbThreads), 0,
                     static_cast<hipStream_t>(stream), W, coef, pre_bias, mean, dz, spart, B, K, C, (float)((double)B * (double)L), Wk,
                     u, dpre_bias, s);
  return launch_status("fpsg_max_bwd_prep");
}

extern "C" int fpsg_max_bwd_dw(const float* S, const float* WG, const float* coef, const float* pre_bias, const float* s,
                               int B, int K, int C, float* dw, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && K > 0 && C > 0, FPSG_E_SHAPE, "fpsg_max_bwd_dw: B,K,C must be positive (got %d,%d,%d)", B, K, C);
  FPSG_REQUIRE_PTR(S); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(dw);
  FPSG_REQUIRE(!misaligned4(WG) && !misaligned4(pre_bias) && !misaligned4(s), FPSG_E_ALIGN, "fpsg_max_bwd_dw: not 4-byte aligned");
  FPSG_REQUIRE(WG == nullptr || s != nullptr, FPSG_E_NULL, "fpsg_max_bwd_dw: s missing");
  const long n = (long)C * K;
  hipLaunchKernelGGL(max_bwd_dw_kernel, dim3((unsigned)((n + kMbThreads - 1) / kMbThreads)), dim3(kMbThreads), 0,
                     static_cast<hipStream_t>(stream), S, WG, coef, pre_bias, s, B, K, C, dw);
  return launch_status("fpsg_max_bwd_dw");
}
