// knn.hip -- K3: k-nearest-neighbour graph of DGCNN's EdgeConv, for gfx950.
// Replaces `knn` of reference src/dgcnn/model.py:13-20 (torch.matmul of x^T x into a
// [B,N,N] tensor + torch.topk): here the N x N matrix never reaches HBM.
//
// One workgroup (4 waves) owns 16 query points of one cloud:
//   phase A  the 16 x N block of  pd_ij = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  is produced with
//            fp32-input MFMA (v_mfma_f32_16x16x4_f32: exact k-ordered fma chain, same
//            64 FLOP/clk/SIMD as the VALU) -- A = the 16 queries (staged once in LDS),
//            B = 16 candidates straight from global/L2 (coalesced 64-B segments of the
//            channel-major x), accumulator carried over C/4 steps -- and written to a
//            16 x N fp32 tile in LDS (128 KiB at N = 2048);
//   phase B  each wave selects the k largest of 4 rows: every lane keeps N/64 values in
//            registers, k rounds of {lane-local best, wave-wide 64-bit (value, ~index)
//            max via cross-lane shuffles, retire the winner}.  Ties go to the lower index.
// Results are bit-identical to oracle_knn (same fma chains, same tie rule).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kQ = 16;            // query rows per workgroup (= MFMA M)
constexpr int kKnnThreads = 256;

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, int C, int N,
                                                     float* __restrict__ xx) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const float* xb = x + (size_t)b * C * N;
  float acc = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float v = xb[(size_t)c * N + j];
    acc = fma_rn(v, v, acc);
  }
  xx[(size_t)b * N + j] = acc;
}

__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// LDS: pd[kQ][ldp] floats, then qa[C4*4][16] floats (the 16 queries, channel-major, zero
// padded to a multiple of 4 channels).
template <int VPL>
__global__ __launch_bounds__(kKnnThreads) void knn_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ xx, int C,
                                                          int N, int k, int ldp,
                                                          int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* pd = lds;
  float* qa = lds + (size_t)kQ * ldp;

  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kQ;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C4 = (C + 3) >> 2;
  const float* __restrict__ xb = x + (size_t)b * C * N;
  const float* __restrict__ xxb = xx + (size_t)b * N;

  // ---- stage the 16 queries: qa[c][q] = x[c][i0+q] (0 beyond C or N)
  for (int e = tid; e < C4 * 4 * kQ; e += kKnnThreads) {
    const int c = e >> 4, q = e & 15;
    qa[e] = (c < C && i0 + q < N) ? xb[(size_t)c * N + i0 + q] : 0.0f;
  }
  __syncthreads();

  // ---- phase A: tiles of 16 candidates, interleaved over the 4 waves
  const int kk = lane >> 4;       // k index inside an MFMA step (0..3)
  const int col = lane & 15;      // candidate column of this lane / query row for A
  float xxq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = i0 + 4 * kk + r;
    xxq[r] = q < N ? xxb[q] : 0.0f;
  }
  const int n_tiles = (N + 15) >> 4;
  for (int t = wave; t < n_tiles; t += 4) {
    const int j = t * 16 + col;
    const bool jin = j < N;
    const float* __restrict__ bp = xb + (jin ? j : 0);
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c4 = 0; c4 < C4; ++c4) {
      const int c = 4 * c4 + kk;
      const float a = qa[c * kQ + col];
      const float bv = (jin && c < C) ? bp[(size_t)c * N] : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
    const float xxj = jin ? xxb[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // D layout: column = lane&15 (candidate), row = 4*(lane>>4) + r (query)
      const float v = fma_rn(2.0f, acc[r], -xxj) - xxq[r];
      pd[(4 * kk + r) * ldp + j] = jin ? v : -__builtin_inff();
    }
  }
  __syncthreads();

  // ---- phase B: wave w selects rows 4w .. 4w+3
  const int n_cols = n_tiles * 16;
  for (int rr = 0; rr < 4; ++rr) {
    const int q = 4 * wave + rr;
    const int i = i0 + q;
    if (i >= N) break;  // wave-uniform
    float v[VPL];
#pragma unroll
    for (int t = 0; t < VPL; ++t) {
      const int e = t * 64 + lane;
      v[t] = e < n_cols ? pd[q * ldp + e] : -__builtin_inff();
    }
    int mine = 0;
    int wt = -1;  // slot retired in this lane at the start of the next round
    for (int round = 0; round < k; ++round) {
      float bv = -__builtin_inff();
      int bt = 0;
#pragma unroll
      for (int t = 0; t < VPL; ++t) {
        const float cur = (t == wt) ? -__builtin_inff() : v[t];
        v[t] = cur;
        const bool gt = cur > bv || t == 0;
        bt = gt ? t : bt;
        bv = gt ? cur : bv;
      }
      const unsigned e = (unsigned)(bt * 64 + lane);
      unsigned long long key = ((unsigned long long)orderable(bv) << 32) | (unsigned)(~e);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
      }
      const unsigned win = ~(unsigned)(key & 0xffffffffull);
      wt = ((int)(win & 63u) == lane) ? (int)(win >> 6) : -1;
      mine = (lane == round) ? (int)win : mine;
    }
    if (lane < k) idx[((size_t)b * N + i) * k + lane] = mine;
  }
}

template <int VPL>
int launch_knn(const float* x, const float* xx, int B, int C, int N, int k, int32_t* idx,
               hipStream_t s) {
  const int n_tiles = (N + 15) / 16;
  const int ldp = n_tiles * 16 + 4;  // +4: the 4 query rows a lane group writes hit disjoint banks
  const int C4 = (C + 3) / 4;
  const size_t lds_bytes = ((size_t)kQ * ldp + (size_t)C4 * 4 * kQ) * sizeof(float);
  dim3 grid((N + kQ - 1) / kQ, B);
  auto kern = knn_kernel<VPL>;
  // one-time opt-in to the full 160 KiB of LDS for this instantiation (per process)
  static const hipError_t lds_optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_optin != hipSuccess && lds_bytes > 64 * 1024) {
    set_error("fpsg_knn: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(lds_optin));
    return (int)lds_optin;
  }
  hipLaunchKernelGGL(kern, grid, dim3(kKnnThreads), lds_bytes, s, x, xx, C, N, k, ldp, idx);
  return launch_status("fpsg_knn");
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_knn(const float* x, int B, int C, int N, int k, int32_t* idx,
                        float* sqnorm_ws, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && C > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_knn: B,C,N,k must be positive (got %d,%d,%d,%d)", B, C, N, k);
  FPSG_REQUIRE(k <= 64 && k <= N, FPSG_E_LIMIT, "fpsg_knn: need k <= min(64, N) (k=%d, N=%d)", k, N);
  {
    const size_t need = ((size_t)kQ * (((N + 15) / 16) * 16 + 4) + (size_t)((C + 3) / 4) * 4 * kQ) * 4;
    FPSG_REQUIRE(need <= 160 * 1024, FPSG_E_LIMIT,
                 "fpsg_knn: N=%d, C=%d need %zu B of LDS (16 x N distance tile + 16 x C queries); "
                 "limit is 163840 B (N <= 2048 at C <= 448)", N, C, need);
  }
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_knn: B=%d exceeds 65535", B);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(sqnorm_ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, x, C, N, sqnorm_ws);
  int rc = launch_status("fpsg_knn(sqnorm)");
  if (rc) return rc;
  const int vpl = (((N + 15) / 16) * 16 + 63) / 64;
  if (vpl <= 4) return launch_knn<4>(x, sqnorm_ws, B, C, N, k, idx, s);
  if (vpl <= 8) return launch_knn<8>(x, sqnorm_ws, B, C, N, k, idx, s);
  if (vpl <= 16) return launch_knn<16>(x, sqnorm_ws, B, C, N, k, idx, s);
  if (vpl <= 32) return launch_knn<32>(x, sqnorm_ws, B, C, N, k, idx, s);
  return launch_knn<40>(x, sqnorm_ws, B, C, N, k, idx, s);
}
