// edge.hip -- K4a: EdgeConv edge features (materialising form) for gfx950.
// Replaces `get_graph_feature` of reference src/dgcnn/model.py:23-42: the row gather by
// advanced indexing, the k-fold repeat and the concat/permute copies become one pass that
// writes out[b, 0:C, n, j] = x[b, :, idx[b,n,j]] - x[b, :, n] and out[b, C:2C, n, j] = x[b, :, n]
// directly in the [B, 2C, N, k] layout the following 1x1 convolution expects.
// (The fused, non-materialising EdgeConv used by DGCNNfeat lives in edgeconv.hip.)
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kEdgeThreads = 256;

// grid: (ceil(N*k / 256), B).  A thread owns one (n, j) edge for every channel: its
// neighbour index is read once; per channel the two stores of a wave are contiguous.
__global__ __launch_bounds__(kEdgeThreads) void edge_feature_fwd_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ idx, int C, int N, int k,
    float* __restrict__ out) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * kEdgeThreads + threadIdx.x;
  const int nk = N * k;
  if (e >= nk) return;
  const int n = e / k;
  int m = idx[(size_t)b * nk + e];
  m = m < 0 ? 0 : (m >= N ? N - 1 : m);  // never read outside the cloud
  const float* __restrict__ xb = x + (size_t)b * C * N;
  float* __restrict__ ob = out + (size_t)b * 2 * C * nk;
  for (int c = 0; c < C; ++c) {
    const float ctr = xb[(size_t)c * N + n];
    const float nb = xb[(size_t)c * N + m];
    ob[(size_t)c * nk + e] = nb - ctr;
    ob[(size_t)(C + c) * nk + e] = ctr;
  }
}

// centre terms: gx[b,c,n] = sum_j (g_centre[n,j] - g_diff[n,j]); one thread per (c, n)
__global__ __launch_bounds__(kEdgeThreads) void edge_feature_bwd_centre_kernel(
    const float* __restrict__ gout, int C, int N, int k, float* __restrict__ gx) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * kEdgeThreads + threadIdx.x;
  if (t >= C * N) return;
  const int c = t / N, n = t - c * N;
  const size_t nk = (size_t)N * k;
  const float* gd = gout + ((size_t)b * 2 * C + c) * nk + (size_t)n * k;
  const float* gc = gout + ((size_t)b * 2 * C + C + c) * nk + (size_t)n * k;
  float acc = 0.0f;
  for (int j = 0; j < k; ++j) acc += gc[j] - gd[j];
  gx[((size_t)b * C + c) * N + n] = acc;
}

// neighbour terms: gx[b,c,idx[n,j]] += g_diff[b,c,n,j]  (fp32 atomics: summation order, and
// therefore the last bits, may differ from run to run -- as with the reference's index_put)
__global__ __launch_bounds__(kEdgeThreads) void edge_feature_bwd_scatter_kernel(
    const float* __restrict__ gout, const int32_t* __restrict__ idx, int C, int N, int k,
    float* __restrict__ gx) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * kEdgeThreads + threadIdx.x;
  const int nk = N * k;
  if (e >= nk) return;
  int m = idx[(size_t)b * nk + e];
  m = m < 0 ? 0 : (m >= N ? N - 1 : m);
  const float* __restrict__ gb = gout + (size_t)b * 2 * C * nk;
  float* __restrict__ gxb = gx + (size_t)b * C * N;
  for (int c = 0; c < C; ++c) atomicAdd(&gxb[(size_t)c * N + m], gb[(size_t)c * nk + e]);
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_edge_feature_fwd(const float* x, const int32_t* idx, int B, int C, int N,
                                     int k, float* out, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && C > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_edge_feature_fwd: B,C,N,k must be positive (got %d,%d,%d,%d)", B, C, N, k);
  FPSG_REQUIRE(B <= 65535 && (long)N * k < (1L << 31), FPSG_E_LIMIT,
               "fpsg_edge_feature_fwd: B=%d or N*k=%ld too large", B, (long)N * k);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(out);
  dim3 grid((N * k + kEdgeThreads - 1) / kEdgeThreads, B);
  hipLaunchKernelGGL(edge_feature_fwd_kernel, grid, dim3(kEdgeThreads), 0,
                     static_cast<hipStream_t>(stream), x, idx, C, N, k, out);
  return launch_status("fpsg_edge_feature_fwd");
}

extern "C" int fpsg_edge_feature_bwd(const float* gout, const int32_t* idx, int B, int C, int N,
                                     int k, float* gx, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && C > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_edge_feature_bwd: B,C,N,k must be positive (got %d,%d,%d,%d)", B, C, N, k);
  FPSG_REQUIRE(B <= 65535 && (long)N * k < (1L << 31) && (long)C * N < (1L << 31), FPSG_E_LIMIT,
               "fpsg_edge_feature_bwd: B=%d, N*k=%ld or C*N=%ld too large", B, (long)N * k, (long)C * N);
  FPSG_REQUIRE_PTR(gout); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(gx);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(edge_feature_bwd_centre_kernel, dim3((C * N + kEdgeThreads - 1) / kEdgeThreads, B),
                     dim3(kEdgeThreads), 0, s, gout, C, N, k, gx);
  int rc = launch_status("fpsg_edge_feature_bwd(centre)");
  if (rc) return rc;
  hipLaunchKernelGGL(edge_feature_bwd_scatter_kernel, dim3((N * k + kEdgeThreads - 1) / kEdgeThreads, B),
                     dim3(kEdgeThreads), 0, s, gout, idx, C, N, k, gx);
  return launch_status("fpsg_edge_feature_bwd(scatter)");
}
