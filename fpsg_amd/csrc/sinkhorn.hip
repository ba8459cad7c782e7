// sinkhorn.hip -- K2b: the Sinkhorn divergence the reference's evaluation calls "EMD", for gfx950.
// neuralnet_pytorch.metrics.emd_loss(sinkhorn=True) (src/models/utils.py:12-13, used at
// src/models/few_shot.py:168) is geomloss.SamplesLoss(): a debiased Sinkhorn divergence with cost
// |x-y|^2/2 whose whole cost is the repeated evaluation of the soft-min
//     out[b,i] = -eps * log sum_j exp( h[b,j] - |x_i - y_j|^2 / (2 eps) ),   h = log-weight + dual / eps
// geomloss' "tensorized" backend materialises several [B,N,M] matrices (16 MB per 2048-point pair each);
// here nothing of size N x M exists.
//
//   softmin_core<R>: a workgroup = 16 waves sharing 64*R owner points (R per lane, VGPRs); the summed
//     cloud and its h are staged once in LDS as SoA and split 16 ways; every lane reads the same
//     address (broadcast ds_read_b128 = 4 candidates per coordinate per instruction), distances and
//     exponent arguments are evaluated two candidates at a time with packed FP32 (v_pk_*), the running
//     (max, sum) pair lives in base-2 exponent space (v_exp_f32 / v_log_f32 are base 2), one rescale per
//     8 candidates; the 16 partial (max, sum) pairs merge through LDS in a fixed order => deterministic.
//     VALU + transcendental bound: per pair 3.5 packed-half instructions + 1 v_exp_f32 (quarter rate).
//   sinkhorn_step_kernel: ONE launch per annealing step evaluates the step's four independent soft-mins
//     (blockIdx.y selects x<-y, y<-x, x<-x, y<-y), forms h = log-weight + dual / eps while staging and
//     applies the symmetric averaging new = (old + softmin) / 2 in its epilogue -- the loop of
//     fpsg_sinkhorn_divergence is ~12 launches per call instead of ~40 soft-mins + ~90 elementwise kernels.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kSmWaves = 16;
constexpr int kSmThreads = 64 * kSmWaves;
constexpr int kSmTile = 2048;    // summed-cloud points staged per LDS pass (4 floats each = 32 KiB)
constexpr int kSmChunk = 8;      // candidates per rescale of the running sum
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegHuge = -3.0e38f;   // "no finite term yet": finite, so that max/sub never make a NaN

struct SmLds {
  float sx[kSmTile], sy[kSmTile], sz[kSmTile], sh[kSmTile];
  float pm[kSmWaves][64 * 2], ps[kSmWaves][64 * 2];     // R <= 2 owners per lane
};

// Soft-min of 64*R owners (rows o_base + lane*R + r of xo [No,3]) over the cloud ys [Ns,3] with
// h_j = (pot ? logw + pot[j] * (1/eps) : logw)  -- or h_j = hin[j] when hin is given.  Result for owner
// (lane, r) is returned to the threads of wave 0 in res[r]; other waves return garbage.
template <int R>
__device__ __forceinline__ void softmin_core(SmLds& L, const float* __restrict__ xo, int No, int o_base,
                                             const float* __restrict__ ys, int Ns,
                                             const float* __restrict__ hin, const float* __restrict__ pot,
                                             float logw, float inv_eps, float eps, float k2, float (&res)[R]) {
  static_assert(R <= 2, "SmLds::pm / ps hold two owners per lane");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  v2f qx[R], qy[R], qz[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int o = o_base + lane * R + r;
    o = o < No ? o : No - 1;
    const float x = xo[3 * o], y = xo[3 * o + 1], z = xo[3 * o + 2];
    qx[r] = v2f{x, x}; qy[r] = v2f{y, y}; qz[r] = v2f{z, z};
  }
  float m[R], s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { m[r] = kNegHuge; s[r] = 0.0f; }
  const v2f nk2 = {-k2, -k2};

  for (int t0 = 0; t0 < Ns; t0 += kSmTile) {
    if (t0) __syncthreads();
    const int cnt = (Ns - t0) < kSmTile ? (Ns - t0) : kSmTile;
    const int padded = (cnt + kSmChunk - 1) / kSmChunk * kSmChunk;
    for (int e = tid; e < padded; e += kSmThreads) {
      const bool in = e < cnt;
      L.sx[e] = in ? ys[3 * (t0 + e)] : 0.0f;
      L.sy[e] = in ? ys[3 * (t0 + e) + 1] : 0.0f;
      L.sz[e] = in ? ys[3 * (t0 + e) + 2] : 0.0f;
      float h = -__builtin_inff();                  // padding never contributes
      if (in) h = hin ? hin[t0 + e] : (pot ? logw + pot[t0 + e] * inv_eps : logw);   // torch divides by a scalar this way
      L.sh[e] = h * kLog2e;
    }
    __syncthreads();
    const int chunks = padded / kSmChunk;
    const int per = (chunks + kSmWaves - 1) / kSmWaves;
    const int lo = wave * per;
    const int hi = (lo + per) < chunks ? (lo + per) : chunks;
    for (int c = lo; c < hi; ++c) {
      const v4f* px = reinterpret_cast<const v4f*>(L.sx + c * kSmChunk);
      const v4f* py = reinterpret_cast<const v4f*>(L.sy + c * kSmChunk);
      const v4f* pz = reinterpret_cast<const v4f*>(L.sz + c * kSmChunk);
      const v4f* ph = reinterpret_cast<const v4f*>(L.sh + c * kSmChunk);
      const v4f X0 = px[0], X1 = px[1], Y0 = py[0], Y1 = py[1], Z0 = pz[0], Z1 = pz[1], H0 = ph[0], H1 = ph[1];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v2f v[4];
        {
          v2f dx = X0.xy - qx[r], dy = Y0.xy - qy[r], dz = Z0.xy - qz[r];
          v[0] = fma_rn(nk2, fma_rn(dz, dz, fma_rn(dy, dy, dx * dx)), H0.xy);
          dx = X0.zw - qx[r]; dy = Y0.zw - qy[r]; dz = Z0.zw - qz[r];
          v[1] = fma_rn(nk2, fma_rn(dz, dz, fma_rn(dy, dy, dx * dx)), H0.zw);
          dx = X1.xy - qx[r]; dy = Y1.xy - qy[r]; dz = Z1.xy - qz[r];
          v[2] = fma_rn(nk2, fma_rn(dz, dz, fma_rn(dy, dy, dx * dx)), H1.xy);
          dx = X1.zw - qx[r]; dy = Y1.zw - qy[r]; dz = Z1.zw - qz[r];
          v[3] = fma_rn(nk2, fma_rn(dz, dz, fma_rn(dy, dy, dx * dx)), H1.zw);
        }
        float cmax = __builtin_fmaxf(__builtin_fmaxf(v[0].x, v[0].y), v[1].x);
        cmax = __builtin_fmaxf(__builtin_fmaxf(cmax, v[1].y), v[2].x);
        cmax = __builtin_fmaxf(__builtin_fmaxf(cmax, v[2].y), v[3].x);
        cmax = __builtin_fmaxf(cmax, v[3].y);
        const float mn = __builtin_fmaxf(m[r], cmax);
        const v2f vm = {mn, mn};
        float acc = s[r] * __builtin_amdgcn_exp2f(m[r] - mn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {                  // fixed order: candidate index ascending
          const v2f t = v[u] - vm;
          acc += __builtin_amdgcn_exp2f(t.x);
          acc += __builtin_amdgcn_exp2f(t.y);
        }
        s[r] = acc;
        m[r] = mn;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    L.pm[wave][lane * R + r] = m[r];
    L.ps[wave][lane * R + r] = s[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float mm = L.pm[0][lane * R + r];
#pragma unroll
    for (int w = 1; w < kSmWaves; ++w) mm = __builtin_fmaxf(mm, L.pm[w][lane * R + r]);
    float ss = 0.0f;
#pragma unroll
    for (int w = 0; w < kSmWaves; ++w)      // fixed order
      ss += L.ps[w][lane * R + r] * __builtin_amdgcn_exp2f(L.pm[w][lane * R + r] - mm);
    res[r] = -eps * kLn2 * (mm + __builtin_amdgcn_logf(ss));
  }
}

template <int R>
__global__ __launch_bounds__(kSmThreads) void softmin_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ h, int N, int M,
                                                             float k2, float eps,
                                                             float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) SmLds L;
  const int b = blockIdx.y;
  const int o_base = blockIdx.x * (64 * R);
  float res[R];
  softmin_core<R>(L, x + (size_t)b * N * 3, N, o_base, y + (size_t)b * M * 3, M, h + (size_t)b * M, nullptr,
                  0.0f, 0.0f, eps, k2, res);
  if (threadIdx.x >= 64) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int o = o_base + (int)threadIdx.x * R + r;
    if (o < N) out[(size_t)b * N + o] = res[r];
  }
}

// The four soft-mins of one Sinkhorn step (geomloss sinkhorn_loop; notation of oracle/__init__.py):
//   op 0: a_y[M] <- softmin over x of (a_log + b_x / eps)      (owners y, summed cloud x)
//   op 1: b_x[N] <- softmin over y of (b_log + a_y / eps)      (owners x, summed cloud y)
//   op 2: a_x[N] <- softmin over x of (a_log + a_x / eps)      (owners x, summed cloud x)
//   op 3: b_y[M] <- softmin over y of (b_log + b_y / eps)      (owners y, summed cloud y)
// mode 0: initialisation (h = log-weight only), mode 1: new = (old + softmin) / 2, mode 2: final
// extrapolation (new = softmin).  Duals are read from `in` and written to `out` (ping-pong), layout
// per batch item [a_x N | b_x N | a_y M | b_y M].
struct StepArgs {
  const float* x;
  const float* y;
  const float* in;
  float* out;
  int N, M, mode;
  float eps, inv_eps, k2, a_log, b_log;
};

template <int R>
__global__ __launch_bounds__(kSmThreads) void sinkhorn_step_kernel(StepArgs a) {
  __shared__ __attribute__((aligned(16))) SmLds L;
  const int op = blockIdx.y;
  const int b = blockIdx.z;
  const bool own_is_x = (op == 1 || op == 2);
  const bool sum_is_x = (op == 0 || op == 2);
  const int No = own_is_x ? a.N : a.M;
  const int Ns = sum_is_x ? a.N : a.M;
  const int o_base = blockIdx.x * (64 * R);
  if (o_base >= No) return;                              // the grid is sized for max(N, M)
  const size_t stride = 2 * (size_t)a.N + 2 * (size_t)a.M;
  const float* in = a.in + (size_t)b * stride;
  float* out = a.out + (size_t)b * stride;
  const int off_ax = 0, off_bx = a.N, off_ay = 2 * a.N, off_by = 2 * a.N + a.M;
  const int off_pot = op == 0 ? off_bx : op == 1 ? off_ay : op == 2 ? off_ax : off_by;   // dual inside h
  const int off_out = op == 0 ? off_ay : op == 1 ? off_bx : op == 2 ? off_ax : off_by;   // dual updated
  const float* xo = (own_is_x ? a.x + (size_t)b * a.N * 3 : a.y + (size_t)b * a.M * 3);
  const float* ys = (sum_is_x ? a.x + (size_t)b * a.N * 3 : a.y + (size_t)b * a.M * 3);
  float res[R];
  softmin_core<R>(L, xo, No, o_base, ys, Ns, nullptr, a.mode == 0 ? nullptr : in + off_pot,
                  sum_is_x ? a.a_log : a.b_log, a.inv_eps, a.eps, a.k2, res);
  if (threadIdx.x >= 64) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int o = o_base + (int)threadIdx.x * R + r;
    if (o >= No) continue;
    out[off_out + o] = a.mode == 1 ? 0.5f * (in[off_out + o] + res[r]) : res[r];
  }
}

// S[b] = mean_i (b_x - a_x) + mean_j (a_y - b_y): one workgroup per item, fixed-shape tree
__global__ __launch_bounds__(256) void sinkhorn_cost_kernel(const float* __restrict__ duals, int N, int M,
                                                            float* __restrict__ out) {
  __shared__ float red[2][256];
  const float* d = duals + (size_t)blockIdx.x * (2 * (size_t)N + 2 * (size_t)M);
  float sx = 0.0f, sy = 0.0f;
  for (int i = threadIdx.x; i < N; i += 256) sx += d[N + i] - d[i];
  for (int j = threadIdx.x; j < M; j += 256) sy += d[2 * N + j] - d[2 * N + M + j];
  red[0][threadIdx.x] = sx;
  red[1][threadIdx.x] = sy;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0][0] / (float)N + red[1][0] / (float)M;
}

// Owners per lane.  A workgroup of 16 waves stages the whole summed cloud; two of them fit a CU.  With two owners per
// lane a B = 5 evaluation call is 320 workgroups for 512 slots, and the CUs that got two take twice as long as the
// others; with one owner per lane (640 smaller workgroups) the step drops from 34 to 26 us (0.41 -> 0.31 ms per call;
// identical results: an owner's sum is split over the waves the same way).  Two owners per lane once the grid fills
// the chip several times over (less staging per owner).
constexpr int kCUs = 256;
inline int step_owners_per_lane(long workgroups_at_two) { return workgroups_at_two >= 4L * kCUs ? 2 : 1; }

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
  if (step_owners_per_lane((long)((N + 127) / 128) * B) == 2)
    hipLaunchKernelGGL((softmin_kernel<2>), dim3((N + 127) / 128, B), dim3(kSmThreads), 0, static_cast<hipStream_t>(stream), x, y, h,
                       N, M, k2, eps, out);
  else
    hipLaunchKernelGGL((softmin_kernel<1>), dim3((N + 63) / 64, B), dim3(kSmThreads), 0, static_cast<hipStream_t>(stream), x, y, h,
                       N, M, k2, eps, out);
  return launch_status("fpsg_softmin");
}

extern "C" size_t fpsg_sinkhorn_workspace_floats(int B, int N, int M) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  return 2 * (size_t)B * (2 * (size_t)N + 2 * (size_t)M);
}

extern "C" int fpsg_sinkhorn_divergence(const float* x, const float* y, int B, int N, int M,
                                        const float* eps_host, int n_eps, float* out, float* ws,
                                        fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE,
               "fpsg_sinkhorn_divergence: B,N,M must be positive (got %d,%d,%d)", B, N, M);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_sinkhorn_divergence: B=%d exceeds 65535", B);
  FPSG_REQUIRE(eps_host != nullptr && n_eps >= 1 && n_eps <= 4096, FPSG_E_SHAPE,
               "fpsg_sinkhorn_divergence: an epsilon schedule of 1..4096 host floats is required");
  for (int i = 0; i < n_eps; ++i)
    FPSG_REQUIRE(eps_host[i] > 0.0f, FPSG_E_SHAPE, "fpsg_sinkhorn_divergence: eps[%d] = %g is not positive", i,
                 (double)eps_host[i]);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(out); FPSG_REQUIRE_PTR(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t set = (size_t)B * (2 * (size_t)N + 2 * (size_t)M);
  float* buf[2] = {ws, ws + set};
  const int nmax = N > M ? N : M;
  const int R = step_owners_per_lane((long)((nmax + 127) / 128) * 4 * B);
  const dim3 grid((nmax + 64 * R - 1) / (64 * R), 4, B);
  StepArgs a{};
  a.x = x; a.y = y; a.N = N; a.M = M;
  a.a_log = -logf((float)N);
  a.b_log = -logf((float)M);
  int cur = 0;                                   // buf[cur] holds the current duals (after the first launch)
  auto launch = [&](int mode, float eps) -> int {
    a.mode = mode;
    a.eps = eps;
    a.inv_eps = 1.0f / eps;
    a.k2 = 0.5f / eps * 1.4426950408889634f;
    a.in = buf[cur];
    a.out = buf[cur ^ 1];
    if (R == 2) hipLaunchKernelGGL((sinkhorn_step_kernel<2>), grid, dim3(kSmThreads), 0, s, a);
    else hipLaunchKernelGGL((sinkhorn_step_kernel<1>), grid, dim3(kSmThreads), 0, s, a);
    cur ^= 1;
    return launch_status("fpsg_sinkhorn_divergence");
  };
  int rc = launch(0, eps_host[0]);               // duals at the first (largest) epsilon
  for (int i = 0; rc == 0 && i < n_eps; ++i) rc = launch(1, eps_host[i]);   // annealing with symmetric averaging
  if (rc == 0) rc = launch(2, eps_host[n_eps - 1]);                           // final extrapolation
  if (rc) return rc;
  hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3(B), dim3(256), 0, s, buf[cur], N, M, out);
  return launch_status("fpsg_sinkhorn_divergence(cost)");
}
