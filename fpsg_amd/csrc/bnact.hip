// bnact.hip -- K5: training-mode BatchNorm fused with its activation, for gfx950.
//
// Every BatchNorm of the hot path is followed by ReLU / LeakyReLU (VGG16-BN trunk
// src/models/image_net.py:14, PointNet src/pointnet/model.py:30-44,220-233, decoder
// src/models/point_cloud_net.py:52-54,76-79).  The library path runs them as separate passes
// over activation tensors of up to 475 MB: BN forward (2 reads + 1 write), ReLU (1R+1W), ReLU
// backward (2R+1W), BN backward (4R+1W).  Here the pair is 2R+1W forward and 4R+1W backward:
//   forward : per-(row segment) partial sums -> per-channel mean/rstd (fp64 finalize) ->
//             y = act(x*scale + shift)
//   backward: the activation mask is RE-DERIVED from x (same arithmetic as the forward), so only
//             x is saved -- not the BN output nor the activation output;
//             partial sums of dz and dz*xhat -> per-channel coefficients ->
//             dx = k1*dz + k2*x + k3.
// Optional per-channel PRE-BIAS pb (the bias of the convolution in front of the BatchNorm,
// image_net.py:14 / pointnet/model.py:30-34): the op is then act(BN(x + pb[c])) with x the
// bias-free convolution output -- fl(x + pb) is formed in registers in every pass, exactly the
// value the library's separate broadcast-add kernel would have stored, and the backward also
// returns dpb[c] = sum over (n, l) of dx (what autograd's bias reduction over dx would give),
// from per-block partial sums of the dx pass.  That removes one full read + write (the add)
// and one full read (the bias-gradient reduction) of the activation tensor per layer.
// Layout: x is [N, C, L] contiguous (L = H*W or points), statistics per channel over (N, L).
// All loops are 16-byte (float4) streams over contiguous row segments; work is split so that a
// 64-channel, 37x224x224 tensor still fills the chip (the per-channel workgroup of the library
// kernel cannot).  Deterministic: fixed partial layout, no atomics.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kBnThreads = 256;
constexpr int kBnSeg = 4096;      // floats per work item (one contiguous row segment)
constexpr int kBnSlices = 64;     // partial sums per channel
// Vector loads a thread has in flight in the streaming loops: its whole share of a segment (4096 floats / 256 threads
// = 4 vectors).  With one load per trip the loop was a chain of full round trips (`s_waitcnt vmcnt(0)` per vector, which
// on gfx9 also waits for the previous trip's store): 32 KB in flight per CU, ~5 TB/s at best.  Arithmetic and its order
// per thread are unchanged.
constexpr int kBnU = kBnSeg / 4 / kBnThreads;

enum BnAct { kActNone = 0, kActRelu = 1, kActLeaky = 2 };

template <int ACT>
__device__ __forceinline__ float act_fwd(float z, float slope) {
  if (ACT == kActRelu) return z > 0.0f ? z : 0.0f;
  if (ACT == kActLeaky) return z > 0.0f ? z : z * slope;
  return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z, float slope) {
  if (ACT == kActRelu) return z > 0.0f ? 1.0f : 0.0f;
  if (ACT == kActLeaky) return z > 0.0f ? 1.0f : slope;
  return 1.0f;
}

__device__ __forceinline__ void block_reduce2(float& a, float& b, float* red /*[2][4]*/) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave] = a; red[4 + wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// FOLD (round 5; opt-in by FPSG_BN_FINALIZE_FOLD=1): the forward finalize runs in the LAST-ARRIVING workgroup of
// each channel instead of in a launch of its own.  The partial sums leave the CU as agent-scope (sc1) stores, the wave
// drains them (vmcnt 0) and takes a ticket from the channel's counter; the workgroup that draws ticket S - 1 reads all S
// partials back with agent-scope loads (per-XCD L2s are not coherent: plain loads could be stale), sums them in the
// finalize kernel's own order -- so the result does not depend on which workgroup came last -- and puts the counter
// back to zero.  Counters live in the library (zero at load, every use leaves them zero), kBnFoldBanks banks handed out
// round robin so that calls in flight on different streams do not share one.
constexpr int kBnFoldBanks = 16, kBnFoldChannels = 4096;
__device__ unsigned g_bn_fold_counters[kBnFoldBanks * kBnFoldChannels];
struct BnFoldArgs {
  const float* gamma; const float* beta;
  double count; float eps, momentum;
  float* chan; float* batch_mean; float* batch_var_unbiased; float* run_mean; float* run_var;
  unsigned* counters;
};
__device__ __forceinline__ void wave_sum2(double& a, double& b);
__device__ __forceinline__ void bn_fwd_finalize_channel(double s0, double s1, int c, int C, const float* gamma, const float* beta,
                                                        double count, float eps, float* chan, float* batch_mean,
                                                        float* batch_var_unbiased, float* run_mean, float* run_var, float momentum);

// Work items of channel c: (n, seg) pairs, seg over ceil(L / kBnSeg) segments of a row.
// Block (s, c) takes items s, s + S, ...
// MODE 0: sums of x and x^2.   MODE 1: sums of dz and dz*xhat (dz = dy * act'(x*scale+shift)).
template <int MODE, int ACT, bool NT = false, bool FOLD = false>
__global__ __launch_bounds__(kBnThreads) void bn_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ chan /*[4][C]: scale, shift, mean, rstd*/,
    const float* __restrict__ pb, int N, int C, int L, int S, float slope, float* __restrict__ part /*[C][S][2]*/,
    const BnFoldArgs f = BnFoldArgs{}) {
  __shared__ float red[8];
  const int c = blockIdx.y, s = blockIdx.x;
  const float b = pb ? pb[c] : 0.0f;
  const int segs = (L + kBnSeg - 1) / kBnSeg;
  const int items = N * segs;
  float sc = 0.0f, sh = 0.0f, mu = 0.0f, rs = 0.0f;
  if (MODE == 1) { sc = chan[c]; sh = chan[C + c]; mu = chan[2 * C + c]; rs = chan[3 * C + c]; }
  float a0 = 0.0f, a1 = 0.0f;
  const bool vec = (L & 3) == 0;
  for (int it = s; it < items; it += S) {
    const int n = it / segs, seg = it - n * segs;
    const size_t base = ((size_t)n * C + c) * L + (size_t)seg * kBnSeg;
    const int len = (L - seg * kBnSeg) < kBnSeg ? (L - seg * kBnSeg) : kBnSeg;
    if (vec) {
      const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
      const v4f* __restrict__ gp = reinterpret_cast<const v4f*>(MODE == 1 ? dy + base : x + base);
      const int nvec = len / 4;
      for (int e0 = threadIdx.x; e0 < nvec; e0 += kBnU * kBnThreads) {
        v4f xq[kBnU], gq[kBnU];
#pragma unroll
        for (int j = 0; j < kBnU; ++j) {
          const int e = e0 + j * kBnThreads;
          xq[j] = gq[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
          if (e < nvec) {
            xq[j] = ld_stream<NT>(xp + e);
            if (MODE == 1) gq[j] = ld_stream<NT>(gp + e);
          }
        }
#pragma unroll
        for (int j = 0; j < kBnU; ++j) {
          if (e0 + j * kBnThreads < nvec) {
            v4f xv = xq[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] += b;
            if (MODE == 0) {
#pragma unroll
              for (int u = 0; u < 4; ++u) { a0 += xv[u]; a1 = fma_rn(xv[u], xv[u], a1); }
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float dz = gq[j][u] * act_grad<ACT>(fma_rn(xv[u], sc, sh), slope);
                a0 += dz;
                a1 = fma_rn(dz, (xv[u] - mu) * rs, a1);
              }
            }
          }
        }
      }
    } else {
      for (int e = threadIdx.x; e < len; e += kBnThreads) {
        const float xv = x[base + e] + b;
        if (MODE == 0) {
          a0 += xv; a1 = fma_rn(xv, xv, a1);
        } else {
          const float dz = dy[base + e] * act_grad<ACT>(fma_rn(xv, sc, sh), slope);
          a0 += dz;
          a1 = fma_rn(dz, (xv - mu) * rs, a1);
        }
      }
    }
  }
  block_reduce2(a0, a1, red);
  if constexpr (FOLD) {
    __shared__ int last;
    if (threadIdx.x == 0) {
      __hip_atomic_store(&part[((size_t)c * S + s) * 2 + 0], a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[((size_t)c * S + s) * 2 + 1], a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(&f.counters[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = t == (unsigned)S - 1u;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int sl = lane; sl < S; sl += 64) {
      s0 += __hip_atomic_load(&part[((size_t)c * S + sl) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s1 += __hip_atomic_load(&part[((size_t)c * S + sl) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_sum2(s0, s1);
    if (lane != 0) return;
    bn_fwd_finalize_channel(s0, s1, c, C, f.gamma, f.beta, f.count, f.eps, f.chan, f.batch_mean, f.batch_var_unbiased,
                            f.run_mean, f.run_var, f.momentum);
    __hip_atomic_store(&f.counters[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x == 0) {
    part[((size_t)c * S + s) * 2 + 0] = a0;
    part[((size_t)c * S + s) * 2 + 1] = a1;
  }
}

// forward finalize: mean / biased var -> chan = (scale, shift, mean, rstd); optional outputs of
// the batch mean and UNBIASED variance (what running statistics are updated with).
// One wave per channel: lane s sums slices s, s + 64, ... of the channel's partial sums in fp64 (K5's own passes
// use S <= 64; a convolution's epilogue delivers one slice per workgroup), then a fixed shuffle tree.
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
}

__global__ __launch_bounds__(64) void bn_fwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, int C, int S, double count, float eps,
                                       float* __restrict__ chan, float* __restrict__ batch_mean,
                                       float* __restrict__ batch_var_unbiased, float* __restrict__ run_mean,
                                       float* __restrict__ run_var, float momentum) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int sl = lane; sl < S; sl += 64) { s0 += part[((size_t)c * S + sl) * 2]; s1 += part[((size_t)c * S + sl) * 2 + 1]; }
  wave_sum2(s0, s1);
  if (lane != 0) return;
  bn_fwd_finalize_channel(s0, s1, c, C, gamma, beta, count, eps, chan, batch_mean, batch_var_unbiased, run_mean, run_var, momentum);
}

__device__ __forceinline__ void bn_fwd_finalize_channel(double s0, double s1, int c, int C, const float* gamma, const float* beta,
                                                        double count, float eps, float* chan, float* batch_mean,
                                                        float* batch_var_unbiased, float* run_mean, float* run_var, float momentum) {
  const double mean = s0 / count;
  double var = s1 / count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  const float scale = g * rstd;
  chan[c] = scale;
  chan[C + c] = b - (float)mean * scale;
  chan[2 * C + c] = (float)mean;
  chan[3 * C + c] = rstd;
  const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
  if (batch_mean) batch_mean[c] = (float)mean;
  if (batch_var_unbiased) batch_var_unbiased[c] = unbiased;
  if (run_mean && momentum >= 0.0f) {       // running <- (1 - m) * running + m * batch (torch's rule)
    run_mean[c] = fma_rn(momentum, (float)mean, (1.0f - momentum) * run_mean[c]);
    run_var[c] = fma_rn(momentum, unbiased, (1.0f - momentum) * run_var[c]);
  }
}

// eval mode: chan from running statistics
__global__ void bn_eval_chan_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                    float eps, float* __restrict__ chan) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(rvar[c] + eps);
  const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  chan[c] = g * rstd;
  chan[C + c] = b - rmean[c] * g * rstd;
  chan[2 * C + c] = rmean[c];
  chan[3 * C + c] = rstd;
}

// backward finalize: dgamma, dbeta and the coefficients of dx = k1*dz + k2*x + k3
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ chan, int C,
                                       int S, double count, int training, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ coef /*[3][C]*/) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int q = lane; q < S; q += 64) {          // S <= 64 for K5's own pass; a convolution epilogue may deliver more
    s0 += part[((size_t)c * S + q) * 2];
    s1 += part[((size_t)c * S + q) * 2 + 1];
  }
  wave_sum2(s0, s1);
  if (lane != 0) return;
  dbeta[c] = (float)s0;
  dgamma[c] = (float)s1;
  const double scale = chan[c], mean = chan[2 * C + c], rstd = chan[3 * C + c];
  coef[c] = (float)scale;
  if (training) {
    const double k2 = -scale * s1 * rstd / count;
    coef[C + c] = (float)k2;
    coef[2 * C + c] = (float)(-scale * s0 / count - k2 * mean);
  } else {
    coef[C + c] = 0.0f;
    coef[2 * C + c] = 0.0f;
  }
}

// MODE 0: y = act(x*scale + shift).   MODE 1: dx = k1*dz + k2*x + k3.
// grid (N*C rows, segments of the row)
// NT: the tensors are larger than the Infinity Cache (fpsg_common.h: beyond_cache)
template <int MODE, int ACT, bool NT = false>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ chan,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ pb, int C, int L,
                                                              float slope, float* __restrict__ out,
                                                              float* __restrict__ dxpart /*[N*C][segs] or null*/) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int c = row % C;
  const int seg = blockIdx.y;
  const float b = pb ? pb[c] : 0.0f;
  float acc = 0.0f, unused = 0.0f;
  const size_t base = (size_t)row * L + (size_t)seg * kBnSeg;
  const int len = (L - seg * kBnSeg) < kBnSeg ? (L - seg * kBnSeg) : kBnSeg;
  const float sc = chan[c], sh = chan[C + c];
  float k1 = 0.0f, k2 = 0.0f, k3 = 0.0f;
  if (MODE == 1) { k1 = coef[c]; k2 = coef[C + c]; k3 = coef[2 * C + c]; }
  if ((L & 3) == 0) {
    const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
    const v4f* __restrict__ gp = reinterpret_cast<const v4f*>(MODE == 1 ? dy + base : x + base);
    v4f* __restrict__ op = reinterpret_cast<v4f*>(out + base);
    const int nvec = len / 4;
    for (int e0 = threadIdx.x; e0 < nvec; e0 += kBnU * kBnThreads) {
      v4f xq[kBnU], gq[kBnU];
#pragma unroll
      for (int j = 0; j < kBnU; ++j) {
        const int e = e0 + j * kBnThreads;
        xq[j] = gq[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        if (e < nvec) {
          xq[j] = ld_stream<NT>(xp + e);
          if (MODE == 1) gq[j] = ld_stream<NT>(gp + e);
        }
      }
#pragma unroll
      for (int j = 0; j < kBnU; ++j) {
        const int e = e0 + j * kBnThreads;
        if (e < nvec) {
          v4f xv = xq[j];
#pragma unroll
          for (int u = 0; u < 4; ++u) xv[u] += b;
          v4f r;
          if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = act_fwd<ACT>(fma_rn(xv[u], sc, sh), slope);
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float dz = gq[j][u] * act_grad<ACT>(fma_rn(xv[u], sc, sh), slope);
              r[u] = fma_rn(k1, dz, fma_rn(k2, xv[u], k3));
            }
            acc += (r[0] + r[1]) + (r[2] + r[3]);
          }
          st_stream<NT>(op + e, r);
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < len; e += kBnThreads) {
      const float xv = x[base + e] + b;
      if (MODE == 0) {
        out[base + e] = act_fwd<ACT>(fma_rn(xv, sc, sh), slope);
      } else {
        const float dz = dy[base + e] * act_grad<ACT>(fma_rn(xv, sc, sh), slope);
        const float r = fma_rn(k1, dz, fma_rn(k2, xv, k3));
        out[base + e] = r;
        acc += r;
      }
    }
  }
  if (MODE == 1 && dxpart) {      // uniform: per-(row, segment) partial of sum(dx)
    block_reduce2(acc, unused, red);
    if (threadIdx.x == 0) dxpart[(size_t)row * gridDim.y + seg] = acc;
  }
}

// dpb[c] = sum over rows n and segments of the dx partials, one wave per channel, fixed order
__global__ __launch_bounds__(64) void bn_dxsum_kernel(const float* __restrict__ dxpart, int N, int C, int segs,
                                                      float* __restrict__ dpb) {
  const int c = blockIdx.x;
  const int items = N * segs;
  double a = 0.0;
  for (int it = threadIdx.x; it < items; it += 64) {
    const int n = it / segs, seg = it - n * segs;
    a += (double)dxpart[((size_t)n * C + c) * segs + seg];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
  if (threadIdx.x == 0) dpb[c] = (float)a;
}

// Small problems (N*L values per channel fit a few loops of one workgroup): statistics,
// finalize and apply in ONE launch, one workgroup per channel, the second sweep served by L2.
// MODE 0 forward, MODE 1 backward.
constexpr int kBnSmallMax = 16384;

// Column segments of the rows (blockIdx.y): segment s covers the elements off[s] .. off[s] + len[s] - 1 of every row and
// is a BatchNorm call of its own -- its statistics and backward sums go to the s-th block of the per-channel outputs
// (chan [4][C], batch_mean / batch_var [C] each, dgamma / dbeta / dpb [C] each).  The ordinary call is one segment
// that is the whole row (ld == len[0]).  The decoder's two decodes of an episode share their GEMMs this way.
constexpr int kBnMaxSegs = 4;
struct BnSegs {
  int ld;                  // distance between consecutive rows, in elements
  int off[kBnMaxSegs], len[kBnMaxSegs];
  int staged[kBnMaxSegs];  // the first sweep's values are kept in LDS (dynamic: N*L floats forward, 2*N*L backward)
  int tpr_log2[kBnMaxSegs];  // threads per row: the smallest power of two >= the row's vectors (elements), at most the workgroup
  int stat_stride;         // distance between the segments' blocks of batch_mean / batch_var_unbiased, in elements
};

template <int MODE, int ACT>
__global__ __launch_bounds__(kBnThreads) void bn_small_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ pb, int N, int C, BnSegs segs, int training, float eps,
    float slope, float* __restrict__ out, float* __restrict__ chan, float* __restrict__ batch_mean,
    float* __restrict__ batch_var_unbiased, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dpb, float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  const int sgi = blockIdx.y;
  const int L = segs.len[sgi], ld = segs.ld, staged = segs.staged[sgi], tpr_log2 = segs.tpr_log2[sgi];
  x += segs.off[sgi];
  out += segs.off[sgi];
  if (MODE == 1) dy += segs.off[sgi];
  chan += (size_t)sgi * 4 * C;
  if (batch_mean) batch_mean += (size_t)sgi * segs.stat_stride;
  if (batch_var_unbiased) batch_var_unbiased += (size_t)sgi * segs.stat_stride;
  if (MODE == 1) {
    dgamma += (size_t)sgi * C;
    dbeta += (size_t)sgi * C;
    if (dpb) dpb += (size_t)sgi * C;
  }
  // The second sweep re-reads what the first one read.  With tens of thousands of 16-64 KB channels
  // in flight (the decoder's grouped BatchNorm: 24,624 channels) the 4 MB L2 of an XCD does not hold
  // them -- PMC: 3.7 reads per write in the backward instead of 2 -- so each thread parks its own
  // elements in LDS (its own slots: no barrier needed) when they fit 64 KB.
  extern __shared__ __attribute__((aligned(16))) float stage[];
  __shared__ float red[8];
  __shared__ float bc[4];
  const int c = blockIdx.x;
  const float b = pb ? pb[c] : 0.0f;
  const double count = (double)N * (double)L;
  const bool vec = (L & 3) == 0;
  const int L4 = L >> 2;
  v4f* __restrict__ sx = reinterpret_cast<v4f*>(stage);                   // x + b
  v4f* __restrict__ sd = reinterpret_cast<v4f*>(stage) + (size_t)N * L4;  // dz (backward)
  float sc = 0.0f, sh = 0.0f, mu = 0.0f, rs = 0.0f;
  if (MODE == 1 || !training) { sc = chan[c]; sh = chan[C + c]; mu = chan[2 * C + c]; rs = chan[3 * C + c]; }
  const bool two_sweeps = MODE == 1 || training;
  const bool keep = staged && vec && two_sweeps;
  // Rows shorter than the workgroup are taken several at a time (a 14 x 14 map is 49 vectors, an FC layer's row one
  // element: with one row per pass most threads idle through N dependent load round trips).
  const int tpr = 1 << tpr_log2, rpi = kBnThreads >> tpr_log2;
  const int e0 = threadIdx.x & (tpr - 1), n0 = threadIdx.x >> tpr_log2;
  if (two_sweeps) {
    float a0 = 0.0f, a1 = 0.0f;
    int n_first = n0;
    if (vec && L4 <= tpr) {
      // a row is at most one vector per thread (the decoder's 128-point rows, 14 x 14 maps): the thread's vectors of
      // kBnU row groups are requested together -- one row group per trip made the sweep N / rpi dependent round
      // trips to HBM per thread -- and consumed in row order (same sums, bit for bit)
      const bool has = e0 < L4;
      for (; n_first + (kBnU - 1) * rpi < N; n_first += kBnU * rpi) {
        v4f xq[kBnU], gq[kBnU];
#pragma unroll
        for (int j = 0; j < kBnU; ++j) {
          xq[j] = gq[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
          if (has) {
            const size_t base = ((size_t)(n_first + j * rpi) * C + c) * ld;
            xq[j] = reinterpret_cast<const v4f*>(x + base)[e0];
            if (MODE == 1) gq[j] = reinterpret_cast<const v4f*>(dy + base)[e0];
          }
        }
        if (has) {
#pragma unroll
          for (int j = 0; j < kBnU; ++j) {
            const int n = n_first + j * rpi;
            v4f xv = xq[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] += b;
            if (MODE == 0) {
#pragma unroll
              for (int u = 0; u < 4; ++u) { a0 += xv[u]; a1 = fma_rn(xv[u], xv[u], a1); }
            } else {
              v4f dz;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                dz[u] = gq[j][u] * act_grad<ACT>(fma_rn(xv[u], sc, sh), slope);
                a0 += dz[u];
                a1 = fma_rn(dz[u], (xv[u] - mu) * rs, a1);
              }
              if (keep) sd[(size_t)n * L4 + e0] = dz;
            }
            if (keep) sx[(size_t)n * L4 + e0] = xv;
          }
        }
      }
    }
    for (int n = n_first; n < N; n += rpi) {
      const size_t base = ((size_t)n * C + c) * ld;
      if (vec) {
        const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
        const v4f* __restrict__ gp = reinterpret_cast<const v4f*>(MODE == 1 ? dy + base : x + base);
        for (int e = e0; e < L4; e += tpr) {
          v4f xv = xp[e];
#pragma unroll
          for (int u = 0; u < 4; ++u) xv[u] += b;
          if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += xv[u]; a1 = fma_rn(xv[u], xv[u], a1); }
          } else {
            const v4f gv = gp[e];
            v4f dz;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              dz[u] = gv[u] * act_grad<ACT>(fma_rn(xv[u], sc, sh), slope);
              a0 += dz[u];
              a1 = fma_rn(dz[u], (xv[u] - mu) * rs, a1);
            }
            if (keep) sd[(size_t)n * L4 + e] = dz;
          }
          if (keep) sx[(size_t)n * L4 + e] = xv;
        }
      } else {
        for (int e = e0; e < L; e += tpr) {
          const float xv = x[base + e] + b;
          if (MODE == 0) {
            a0 += xv; a1 = fma_rn(xv, xv, a1);
          } else {
            const float dz = dy[base + e] * act_grad<ACT>(fma_rn(xv, sc, sh), slope);
            a0 += dz; a1 = fma_rn(dz, (xv - mu) * rs, a1);
          }
        }
      }
    }
    block_reduce2(a0, a1, red);
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        const double mean = a0 / count;
        double var = a1 / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.0f, bb = beta ? beta[c] : 0.0f;
        bc[0] = g * rstd; bc[1] = bb - (float)mean * g * rstd; bc[2] = (float)mean; bc[3] = rstd;
        chan[c] = bc[0]; chan[C + c] = bc[1]; chan[2 * C + c] = bc[2]; chan[3 * C + c] = bc[3];
        const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
        if (batch_mean) batch_mean[c] = (float)mean;
        if (batch_var_unbiased) batch_var_unbiased[c] = unbiased;
        if (run_mean && momentum >= 0.0f) {
          run_mean[c] = fma_rn(momentum, (float)mean, (1.0f - momentum) * run_mean[c]);
          run_var[c] = fma_rn(momentum, unbiased, (1.0f - momentum) * run_var[c]);
        }
      } else {
        dbeta[c] = a0; dgamma[c] = a1;
        const double k2 = training ? -(double)sc * a1 * rs / count : 0.0;
        bc[0] = sc; bc[1] = (float)k2;
        bc[2] = training ? (float)(-(double)sc * a0 / count - k2 * mu) : 0.0f;
      }
    }
    __syncthreads();
    if (MODE == 0) { sc = bc[0]; sh = bc[1]; }
  } else {
    if (threadIdx.x == 0) { bc[0] = sc; bc[1] = 0.0f; bc[2] = 0.0f; }
    __syncthreads();
  }
  const float k1 = bc[0], k2 = bc[1], k3 = bc[2];
  float acc = 0.0f, unused = 0.0f;
  for (int n = n0; n < N; n += rpi) {
    const size_t base = ((size_t)n * C + c) * ld;
    if (vec) {
      const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
      const v4f* __restrict__ gp = reinterpret_cast<const v4f*>(MODE == 1 ? dy + base : x + base);
      v4f* __restrict__ op = reinterpret_cast<v4f*>(out + base);
      for (int e = e0; e < L4; e += tpr) {
        v4f xv;
        if (keep) {
          xv = sx[(size_t)n * L4 + e];
        } else {
          xv = xp[e];
#pragma unroll
          for (int u = 0; u < 4; ++u) xv[u] += b;
        }
        v4f r;
        if (MODE == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] = act_fwd<ACT>(fma_rn(xv[u], sc, sh), slope);
        } else {
          v4f dz;
          if (keep) {
            dz = sd[(size_t)n * L4 + e];
          } else {
            const v4f gv = gp[e];
#pragma unroll
            for (int u = 0; u < 4; ++u) dz[u] = gv[u] * act_grad<ACT>(fma_rn(xv[u], sc, sh), slope);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] = fma_rn(k1, dz[u], fma_rn(k2, xv[u], k3));
          acc += (r[0] + r[1]) + (r[2] + r[3]);
        }
        op[e] = r;
      }
    } else {
      for (int e = e0; e < L; e += tpr) {
        const float xv = x[base + e] + b;
        if (MODE == 0) {
          out[base + e] = act_fwd<ACT>(fma_rn(xv, sc, sh), slope);
        } else {
          const float dz = dy[base + e] * act_grad<ACT>(fma_rn(xv, sc, sh), slope);
          const float r = fma_rn(k1, dz, fma_rn(k2, xv, k3));
          out[base + e] = r;
          acc += r;
        }
      }
    }
  }
  if (MODE == 1 && dpb) {
    __syncthreads();              // red[] is reused
    block_reduce2(acc, unused, red);
    if (threadIdx.x == 0) dpb[c] = acc;
  }
}

inline void small_segment(BnSegs& g, int i, int N, int off, int L, int training, bool backward) {
  g.off[i] = off;
  g.len[i] = L;
  // first-sweep values parked in LDS when they fit the default 64 KB of dynamic LDS
  const size_t need = (size_t)N * L * sizeof(float) * (backward ? 2 : 1);
  g.staged[i] = ((L & 3) == 0 && need <= 64 * 1024 && (backward || training)) ? 1 : 0;
  const int row_items = (L & 3) == 0 ? L / 4 : L;
  int t = 0;
  while ((1 << t) < row_items && (1 << t) < kBnThreads) ++t;
  g.tpr_log2[i] = t;
}

template <int MODE>
void launch_small_segs(int act, const float* x, const float* dy, const float* gamma, const float* beta, const float* pb,
                       int N, int C, const BnSegs& g, int nseg, int training, float eps, float slope, float* out,
                       float* chan, float* bm, float* bv, float* dgamma, float* dbeta, float* dpb, float* rmean,
                       float* rvar, float momentum, hipStream_t s) {
  dim3 grid(C, nseg);
  size_t lds = 0;
  for (int i = 0; i < nseg; ++i)
    if (g.staged[i]) lds = std::max(lds, (size_t)N * g.len[i] * sizeof(float) * (MODE == 1 ? 2 : 1));
#define FPSG_SMALL(A) hipLaunchKernelGGL((bn_small_kernel<MODE, A>), grid, dim3(kBnThreads), lds, s, x, dy, gamma, beta, \
                                         pb, N, C, g, training, eps, slope, out, chan, bm, bv, dgamma, dbeta, dpb, rmean, \
                                         rvar, momentum)
  if (act == kActRelu) FPSG_SMALL(kActRelu);
  else if (act == kActLeaky) FPSG_SMALL(kActLeaky);
  else FPSG_SMALL(kActNone);
#undef FPSG_SMALL
}

template <int MODE>
void launch_small(int act, const float* x, const float* dy, const float* gamma, const float* beta, const float* pb,
                  int N, int C, int L, int training, float eps, float slope, float* out, float* chan, float* bm,
                  float* bv, float* dgamma, float* dbeta, float* dpb, float* rmean, float* rvar, float momentum,
                  hipStream_t s) {
  BnSegs g{};
  g.ld = L;
  small_segment(g, 0, N, 0, L, training, MODE == 1);
  launch_small_segs<MODE>(act, x, dy, gamma, beta, pb, N, C, g, 1, training, eps, slope, out, chan, bm, bv, dgamma,
                          dbeta, dpb, rmean, rvar, momentum, s);
}

int slices_for(int N, int L) {
  const int items = N * ((L + kBnSeg - 1) / kBnSeg);
  return items < kBnSlices ? items : kBnSlices;
}

template <int MODE>
void launch_reduce(int act, const float* x, const float* dy, const float* chan, const float* pb, int N, int C, int L,
                   int S, float slope, float* part, hipStream_t s) {
  dim3 grid(S, C);
#define FPSG_RED(A, T) hipLaunchKernelGGL((bn_reduce_kernel<MODE, A, T>), grid, dim3(kBnThreads), 0, s, x, dy, chan, pb, N, C, L, S, slope, part)
  if (beyond_cache((size_t)N * C * L * sizeof(float))) {
    if (act == kActRelu) FPSG_RED(kActRelu, true); else if (act == kActLeaky) FPSG_RED(kActLeaky, true); else FPSG_RED(kActNone, true);
  } else {
    if (act == kActRelu) FPSG_RED(kActRelu, false); else if (act == kActLeaky) FPSG_RED(kActLeaky, false); else FPSG_RED(kActNone, false);
  }
#undef FPSG_RED
}

template <int MODE>
void launch_apply(int act, const float* x, const float* dy, const float* chan, const float* coef, const float* pb,
                  int N, int C, int L, float slope, float* out, float* dxpart, hipStream_t s) {
  dim3 grid((unsigned)((size_t)N * C), (L + kBnSeg - 1) / kBnSeg);
#define FPSG_APP(A, T) hipLaunchKernelGGL((bn_apply_kernel<MODE, A, T>), grid, dim3(kBnThreads), 0, s, x, dy, chan, coef, pb, C, L, slope, out, dxpart)
  if (beyond_cache((size_t)N * C * L * sizeof(float))) {
    if (act == kActRelu) FPSG_APP(kActRelu, true); else if (act == kActLeaky) FPSG_APP(kActLeaky, true); else FPSG_APP(kActNone, true);
  } else {
    if (act == kActRelu) FPSG_APP(kActRelu, false); else if (act == kActLeaky) FPSG_APP(kActLeaky, false); else FPSG_APP(kActNone, false);
  }
#undef FPSG_APP
}

// ---- pooled variants: act(BN(x + pb)) followed by MaxPool2d(kernel 2, stride 2) -------------
// The VGG16-BN trunk ends 5 of its 13 conv+BN+ReLU groups with a 2x2 max-pool
// (image_net.py:14).  Unfused, the full-resolution activation is written, read back by the
// pool (which also stores an int64 index per window), and in the backward the pool scatters
// a full-resolution, mostly-zero gradient that BatchNorm's backward then reads twice.  Here
// the forward writes ONLY the pooled tensor and the backward re-derives, per 2x2 window, the
// activation values and their arg-max from x (scan order (h, w), first strictly greater or
// NaN wins -- the rule of at::native::max_pool_forward_nchw), so that neither the
// full-resolution output, nor the indices, nor the scattered gradient ever exist in HBM.
// x [N,C,H,W] with H, W even; pooled tensors [N,C,H/2,W/2].
// Work item = RP consecutive row pairs of one (n, c) plane (about kBnSeg floats).
// MODE 0: pooled forward.  MODE 1: backward sums of dz, dz*xhat.  MODE 2: backward dx.
// VW = windows per thread (2: 16-byte loads, needs W % 4 == 0; 1: 8-byte loads).
template <int VW> struct PoolVec;
template <> struct PoolVec<2> { typedef v4f in_t; typedef v2f out_t; };
template <> struct PoolVec<1> { typedef v2f in_t; typedef float out_t; };

template <int ACT>
__device__ __forceinline__ int window_argmax(const float (&z)[4], float slope, float& ymax) {
  float m = -INFINITY;
  int sel = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float y = act_fwd<ACT>(z[k], slope);
    if (y > m || y != y) { m = y; sel = k; }
  }
  ymax = m;
  return sel;
}

template <int MODE, int ACT, int VW, bool NT>
__device__ __forceinline__ void pool_item(const float* __restrict__ xpl, const float* __restrict__ gpl,
                                          float* __restrict__ opl, int W, int r0, int r1, float b, float sc,
                                          float sh, float mu, float rs, float k1, float k2, float k3, float slope,
                                          float& a0, float& a1) {
  typedef typename PoolVec<VW>::in_t in_t;
  typedef typename PoolVec<VW>::out_t out_t;
  const int Wp = W >> 1;
  const int vpr = Wp / VW;                 // vectors per pooled row
  const int nvec = (r1 - r0) * vpr;
  // two trips' loads (two row pairs of vectors and their pooled gradients) in flight before the first is used
  constexpr int PU = 2;
  for (int e0 = threadIdx.x; e0 < nvec; e0 += PU * kBnThreads) {
   in_t q0[PU], q1[PU];
   float gq[PU][VW];
#pragma unroll
   for (int j = 0; j < PU; ++j) {
    const int e = e0 + j * kBnThreads;
    if (e < nvec) {
      const int rr = e / vpr, v = e - rr * vpr;
      const size_t o0 = (size_t)(2 * (r0 + rr)) * W + 2 * VW * v;
      q0[j] = ld_stream<NT>(reinterpret_cast<const in_t*>(xpl + o0));
      q1[j] = ld_stream<NT>(reinterpret_cast<const in_t*>(xpl + o0 + W));
      if (MODE != 0) {
#pragma unroll
        for (int w = 0; w < VW; ++w) gq[j][w] = gpl[(size_t)(r0 + rr) * Wp + VW * v + w];
      }
    }
   }
#pragma unroll
   for (int j = 0; j < PU; ++j) {
    const int e = e0 + j * kBnThreads;
    if (e >= nvec) break;
    const int rr = e / vpr, v = e - rr * vpr;
    const int r = r0 + rr;
    const size_t o0 = (size_t)(2 * r) * W + 2 * VW * v;
    const in_t t0 = q0[j], t1 = q1[j];
    float x0[2 * VW], x1[2 * VW];
#pragma unroll
    for (int u = 0; u < 2 * VW; ++u) { x0[u] = t0[u] + b; x1[u] = t1[u] + b; }
    float po[VW];
    float d0[2 * VW], d1[2 * VW];
#pragma unroll
    for (int w = 0; w < VW; ++w) {
      const float xb[4] = {x0[2 * w], x0[2 * w + 1], x1[2 * w], x1[2 * w + 1]};
      const float z[4] = {fma_rn(xb[0], sc, sh), fma_rn(xb[1], sc, sh), fma_rn(xb[2], sc, sh), fma_rn(xb[3], sc, sh)};
      float ymax;
      const int sel = window_argmax<ACT>(z, slope, ymax);
      if (MODE == 0) {
        po[w] = ymax;
      } else {
        const float g = gq[j][w];
        const float zs = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
        const float dz = g * act_grad<ACT>(zs, slope);
        if (MODE == 1) {
          const float xs = sel == 0 ? xb[0] : sel == 1 ? xb[1] : sel == 2 ? xb[2] : xb[3];
          a0 += dz;
          a1 = fma_rn(dz, (xs - mu) * rs, a1);
        } else {
          float rk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) rk[k] = fma_rn(k1, (k == sel ? dz : 0.0f), fma_rn(k2, xb[k], k3));
          d0[2 * w] = rk[0]; d0[2 * w + 1] = rk[1]; d1[2 * w] = rk[2]; d1[2 * w + 1] = rk[3];
          a0 += (rk[0] + rk[1]) + (rk[2] + rk[3]);
        }
      }
    }
    if (MODE == 0) {
      out_t ov;
      if constexpr (VW == 2) { ov[0] = po[0]; ov[1] = po[1]; } else { ov = po[0]; }
      *reinterpret_cast<out_t*>(opl + (size_t)r * Wp + VW * v) = ov;
    } else if (MODE == 2) {
      in_t o0v, o1v;
#pragma unroll
      for (int u = 0; u < 2 * VW; ++u) { o0v[u] = d0[u]; o1v[u] = d1[u]; }
      st_stream<NT>(reinterpret_cast<in_t*>(opl + o0), o0v);
      st_stream<NT>(reinterpret_cast<in_t*>(opl + o0 + W), o1v);
    }
   }
  }
}

// MODE 0 / 2: grid (N*C planes, items of a plane).  `out`: pooled y (MODE 0) or dx (MODE 2).
template <int MODE, int ACT, int VW, bool NT = false>
__global__ __launch_bounds__(kBnThreads) void bn_pool_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dyp, const float* __restrict__ chan,
    const float* __restrict__ coef, const float* __restrict__ pb, int C, int H, int W, int RP, float slope,
    float* __restrict__ out, float* __restrict__ dxpart) {
  __shared__ float red[8];
  const int plane = blockIdx.x, c = plane % C, it = blockIdx.y;
  const int Hp = H >> 1, Wp = W >> 1;
  const int r0 = it * RP, r1 = (r0 + RP) < Hp ? (r0 + RP) : Hp;
  const float b = pb ? pb[c] : 0.0f;
  const float sc = chan[c], sh = chan[C + c];
  float k1 = 0.0f, k2 = 0.0f, k3 = 0.0f;
  if (MODE == 2) { k1 = coef[c]; k2 = coef[C + c]; k3 = coef[2 * C + c]; }
  float acc = 0.0f, unused = 0.0f;
  const float* xpl = x + (size_t)plane * H * W;
  const float* gpl = MODE == 2 ? dyp + (size_t)plane * Hp * Wp : nullptr;
  float* opl = MODE == 0 ? out + (size_t)plane * Hp * Wp : out + (size_t)plane * H * W;
  pool_item<MODE, ACT, VW, NT>(xpl, gpl, opl, W, r0, r1, b, sc, sh, 0.0f, 0.0f, k1, k2, k3, slope, acc, unused);
  if (MODE == 2 && dxpart) {
    block_reduce2(acc, unused, red);
    if (threadIdx.x == 0) dxpart[(size_t)plane * gridDim.y + it] = acc;
  }
}

// backward sums: grid (S slices, C channels); block (s, c) takes items s, s + S, ... of channel c
template <int ACT, int VW, bool NT = false>
__global__ __launch_bounds__(kBnThreads) void bn_pool_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dyp, const float* __restrict__ chan,
    const float* __restrict__ pb, int N, int C, int H, int W, int RP, int S, float slope,
    float* __restrict__ part /*[C][S][2]*/) {
  __shared__ float red[8];
  const int c = blockIdx.y, s = blockIdx.x;
  const int Hp = H >> 1, Wp = W >> 1;
  const int per_plane = (Hp + RP - 1) / RP;
  const int items = N * per_plane;
  const float b = pb ? pb[c] : 0.0f;
  const float sc = chan[c], sh = chan[C + c], mu = chan[2 * C + c], rs = chan[3 * C + c];
  float a0 = 0.0f, a1 = 0.0f;
  for (int item = s; item < items; item += S) {
    const int n = item / per_plane, it = item - n * per_plane;
    const int r0 = it * RP, r1 = (r0 + RP) < Hp ? (r0 + RP) : Hp;
    const size_t plane = (size_t)n * C + c;
    pool_item<1, ACT, VW, NT>(x + plane * H * W, dyp + plane * Hp * Wp, nullptr, W, r0, r1, b, sc, sh, mu, rs, 0.0f, 0.0f,
                          0.0f, slope, a0, a1);
  }
  block_reduce2(a0, a1, red);
  if (threadIdx.x == 0) {
    part[((size_t)c * S + s) * 2 + 0] = a0;
    part[((size_t)c * S + s) * 2 + 1] = a1;
  }
}

int pool_rows_per_item(int W) {
  const int rp = kBnSeg / (2 * W);
  return rp < 1 ? 1 : rp;
}

template <int MODE>
void launch_pool_apply(int act, const float* x, const float* dyp, const float* chan, const float* coef,
                       const float* pb, int N, int C, int H, int W, float slope, float* out, float* dxpart,
                       hipStream_t s) {
  const int RP = pool_rows_per_item(W);
  dim3 grid((unsigned)((size_t)N * C), (H / 2 + RP - 1) / RP);
#define FPSG_PA(A, V) hipLaunchKernelGGL((bn_pool_apply_kernel<MODE, A, V>), grid, dim3(kBnThreads), 0, s, x, dyp, chan, \
                                         coef, pb, C, H, W, RP, slope, out, dxpart)
  if ((W & 3) == 0 && act == kActRelu && beyond_cache((size_t)N * C * H * W * sizeof(float))) {      // VGG's first stage
    hipLaunchKernelGGL((bn_pool_apply_kernel<MODE, kActRelu, 2, true>), grid, dim3(kBnThreads), 0, s, x, dyp, chan, coef, pb,
                       C, H, W, RP, slope, out, dxpart);
  } else if ((W & 3) == 0) {
    if (act == kActRelu) FPSG_PA(kActRelu, 2); else if (act == kActLeaky) FPSG_PA(kActLeaky, 2); else FPSG_PA(kActNone, 2);
  } else {
    if (act == kActRelu) FPSG_PA(kActRelu, 1); else if (act == kActLeaky) FPSG_PA(kActLeaky, 1); else FPSG_PA(kActNone, 1);
  }
#undef FPSG_PA
}

void launch_pool_reduce(int act, const float* x, const float* dyp, const float* chan, const float* pb, int N, int C,
                        int H, int W, int S, float slope, float* part, hipStream_t s) {
  const int RP = pool_rows_per_item(W);
  dim3 grid(S, C);
#define FPSG_PR(A, V) hipLaunchKernelGGL((bn_pool_reduce_kernel<A, V>), grid, dim3(kBnThreads), 0, s, x, dyp, chan, pb, \
                                         N, C, H, W, RP, S, slope, part)
  if ((W & 3) == 0 && act == kActRelu && beyond_cache((size_t)N * C * H * W * sizeof(float))) {
    hipLaunchKernelGGL((bn_pool_reduce_kernel<kActRelu, 2, true>), grid, dim3(kBnThreads), 0, s, x, dyp, chan, pb, N, C, H, W,
                       RP, S, slope, part);
  } else if ((W & 3) == 0) {
    if (act == kActRelu) FPSG_PR(kActRelu, 2); else if (act == kActLeaky) FPSG_PR(kActLeaky, 2); else FPSG_PR(kActNone, 2);
  } else {
    if (act == kActRelu) FPSG_PR(kActRelu, 1); else if (act == kActLeaky) FPSG_PR(kActLeaky, 1); else FPSG_PR(kActNone, 1);
  }
#undef FPSG_PR
}

int check_pool_dims(const char* fn, int N, int C, int H, int W, int act) {
  FPSG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0, FPSG_E_SHAPE,
               "%s: N,C positive and H,W positive and even (got %d,%d,%d,%d)", fn, N, C, H, W);
  FPSG_REQUIRE(act >= 0 && act <= 2, FPSG_E_SHAPE, "%s: act must be 0 (none), 1 (relu) or 2 (leaky)", fn);
  FPSG_REQUIRE(C <= 65535 && (long)N * C < (1L << 31) && (long)H * W < (1L << 30) && H / 2 <= 65535, FPSG_E_LIMIT,
               "%s: C=%d, H*W=%ld or N*C=%ld beyond the grid limits", fn, C, (long)H * W, (long)N * C);
  return 0;
}

// ---- max-over-the-row variants: max_l act(BN(x + pb))[n, c, l] ---------------------------------
// PointNet ends its shared MLP with BatchNorm (+ReLU in the T-Net) and a max over the N points
// (src/pointnet/model.py:35-37, 222-224) on a [B,1024,N] tensor (537 MB at B=64, N=2048).
// z -> act(z) and x -> x*scale+shift are monotone, so
//     max_l act(BN(x_l)) = act(BN(x_sel)),  x_sel = max_l x_l if scale >= 0 else min_l x_l,
// bit-equal to the max of the individually normalised values.  The statistics pass therefore also
// tracks each row's extremes (first index on ties) and the normalised tensor is never written:
// forward 1 read (library chain: BN 2R+1W, ReLU, max 1R); backward: the incoming gradient is
// [N, C] and lands on one element per row, so the sums of dz and dz*xhat are over N values per
// channel and dx = k1*dz*[l = sel] + k2*x + k3 is 1 read + 1 write (library chain: scatter 1W, ReLU
// backward, BN backward 4R+1W).  The gradient goes to the FIRST arg-extreme of x in the row; the
// library's max picks among equal OUTPUTS, which only differs where distinct inputs round to the
// same output (same gradient value on a numerically equal neighbour) or are clipped by the ReLU
// (zero gradient either way).
struct RowExt { float vmax, vmin; int imax, imin; };

__device__ __forceinline__ void ext_merge(float& vmax, int& imax, float& vmin, int& imin, float omax, int oimax,
                                          float omin, int oimin) {
  if (omax > vmax || (omax == vmax && oimax < imax)) { vmax = omax; imax = oimax; }
  if (omin < vmin || (omin == vmin && oimin < imin)) { vmin = omin; imin = oimin; }
}

// grid (C, S), channel fastest: workgroups that run together read neighbouring rows of one image (rows of one channel
// are C*L floats apart; with the slice index fastest the chip streamed 8 KB pieces 8 MB apart: 2.2 TB/s).
// Wave w of block (c, s) takes the (n, seg) items 4s + w, 4s + w + 4S, ... of channel c
// (a 2048-point row is 8 KB: a wave streams it with 8 vector loads per lane and reduces the
// extremes with shuffles, no LDS, no barrier); the block's sums go to part[c][s].
// STATS = 0 (eval mode): extremes only.
template <int STATS, bool NT = false>
__global__ __launch_bounds__(kBnThreads) void bn_reduce_ext_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ pb, int N, int C, int L,
                                                                   int S, float* __restrict__ part /*[C][S][2]*/,
                                                                   RowExt* __restrict__ ext /*[N*C][segs]*/) {
  __shared__ float red[8];
  const int c = blockIdx.x, s = blockIdx.y;
  const int segs = (L + kBnSeg - 1) / kBnSeg;
  const int items = N * segs;
  const float b = pb ? pb[c] : 0.0f;
  float a0 = 0.0f, a1 = 0.0f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int it = 4 * s + wave; it < items; it += 4 * S) {
    const int n = it / segs, seg = it - n * segs;
    const size_t base = ((size_t)n * C + c) * L + (size_t)seg * kBnSeg;
    const int len = (L - seg * kBnSeg) < kBnSeg ? (L - seg * kBnSeg) : kBnSeg;
    float vmax = -INFINITY, vmin = INFINITY;
    int imax = 0x7fffffff, imin = 0x7fffffff;
    // ascending positions per lane: strict compares keep the first occurrence
    if ((L & 3) == 0) {
      const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
      // eight vector loads in flight per lane: a whole 2048-point row (one load per trip leaves it as eight serial round trips)
      for (int e0 = lane; e0 < len / 4; e0 += 8 * 64) {
        v4f q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = e0 + 64 * j;
          q[j] = e < len / 4 ? ld_stream<NT>(xp + e) : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int e = e0 + 64 * j;
          if (e < len / 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float xv = q[j][u] + b;
              if (STATS) { a0 += xv; a1 = fma_rn(xv, xv, a1); }
              // selects, not branches: as `if` bodies the two updates became ~60 exec-mask branches per row
              const int pos = seg * kBnSeg + 4 * e + u;
              const bool gt = xv > vmax, lt = xv < vmin;
              vmax = gt ? xv : vmax;
              imax = gt ? pos : imax;
              vmin = lt ? xv : vmin;
              imin = lt ? pos : imin;
            }
          }
        }
      }
    } else {
      for (int e = lane; e < len; e += 64) {
        const float xv = x[base + e] + b;
        if (STATS) { a0 += xv; a1 = fma_rn(xv, xv, a1); }
        if (xv > vmax) { vmax = xv; imax = seg * kBnSeg + e; }
        if (xv < vmin) { vmin = xv; imin = seg * kBnSeg + e; }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      ext_merge(vmax, imax, vmin, imin, __shfl_down(vmax, off, 64), __shfl_down(imax, off, 64),
                __shfl_down(vmin, off, 64), __shfl_down(imin, off, 64));
    if (lane == 0) {
      RowExt r; r.vmax = vmax; r.vmin = vmin; r.imax = imax; r.imin = imin;
      ext[((size_t)n * C + c) * segs + seg] = r;
    }
  }
  if (STATS) {
    block_reduce2(a0, a1, red);
    if (threadIdx.x == 0) {
      part[((size_t)c * S + s) * 2 + 0] = a0;
      part[((size_t)c * S + s) * 2 + 1] = a1;
    }
  }
}

// one thread per (n, c) row: combine the segments' extremes, pick by the sign of scale, emit
// out = act(x_sel*scale + shift) and the selected index
template <int ACT>
__global__ void bn_max_out_kernel(const RowExt* __restrict__ ext, const float* __restrict__ chan, int rows, int C,
                                  int segs, float slope, float* __restrict__ out, int* __restrict__ idx) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int c = row % C;
  RowExt r = ext[(size_t)row * segs];
  for (int sgi = 1; sgi < segs; ++sgi) {
    const RowExt o = ext[(size_t)row * segs + sgi];
    ext_merge(r.vmax, r.imax, r.vmin, r.imin, o.vmax, o.imax, o.vmin, o.imin);
  }
  const float sc = chan[c], sh = chan[C + c];
  const bool up = sc >= 0.0f;
  out[row] = act_fwd<ACT>(fma_rn(up ? r.vmax : r.vmin, sc, sh), slope);
  idx[row] = sc == 0.0f ? 0 : (up ? r.imax : r.imin);
}

// backward, per channel (one wave): dz[n] = g[n,c] * act'(z_sel), sums over n in a fixed order,
// dgamma / dbeta / coefficients as bn_bwd_finalize_kernel, dz stored for the dx pass
template <int ACT>
__global__ __launch_bounds__(64) void bn_max_bwd_coef_kernel(const float* __restrict__ x, const float* __restrict__ pb,
                                                             const float* __restrict__ g, const int* __restrict__ idx,
                                                             const float* __restrict__ chan, int N, int C, int L,
                                                             int training, float slope, float* __restrict__ dz,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ coef) {
  const int c = blockIdx.x;
  const float b = pb ? pb[c] : 0.0f;
  const float sc = chan[c], sh = chan[C + c], mu = chan[2 * C + c], rs = chan[3 * C + c];
  double s0 = 0.0, s1 = 0.0;
  for (int n = threadIdx.x; n < N; n += 64) {
    const size_t row = (size_t)n * C + c;
    const float xs = x[row * L + idx[row]] + b;
    const float d = g[row] * act_grad<ACT>(fma_rn(xs, sc, sh), slope);
    dz[row] = d;
    s0 += (double)d;
    s1 += (double)(d * ((xs - mu) * rs));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); }
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    const double count = (double)N * (double)L;
    coef[c] = sc;
    if (training) {
      const double k2 = -(double)sc * s1 * (double)rs / count;
      coef[C + c] = (float)k2;
      coef[2 * C + c] = (float)(-(double)sc * s0 / count - k2 * (double)mu);
    } else {
      coef[C + c] = 0.0f;
      coef[2 * C + c] = 0.0f;
    }
  }
}

// dx = k1*dz*[l == sel] + k2*(x+b) + k3 ; grid (N*C rows, segments)
template <bool NT>
__global__ __launch_bounds__(kBnThreads) void bn_max_apply_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ pb,
                                                                  const float* __restrict__ dz,
                                                                  const int* __restrict__ idx,
                                                                  const float* __restrict__ coef, int C, int L,
                                                                  float* __restrict__ dx, float* __restrict__ dxpart) {
  __shared__ float red[8];
  const int row = blockIdx.x, c = row % C, seg = blockIdx.y;
  const size_t base = (size_t)row * L + (size_t)seg * kBnSeg;
  const int len = (L - seg * kBnSeg) < kBnSeg ? (L - seg * kBnSeg) : kBnSeg;
  const float b = pb ? pb[c] : 0.0f;
  const float k1 = coef[c], k2 = coef[C + c], k3 = coef[2 * C + c];
  const float d = dz[row];
  const int sel = idx[row] - seg * kBnSeg;
  float acc = 0.0f, unused = 0.0f;
  if ((L & 3) == 0) {
    const v4f* __restrict__ xp = reinterpret_cast<const v4f*>(x + base);
    v4f* __restrict__ op = reinterpret_cast<v4f*>(dx + base);
    const int nvec = len / 4;
    for (int e0 = threadIdx.x; e0 < nvec; e0 += kBnU * kBnThreads) {
      v4f xq[kBnU];
#pragma unroll
      for (int j = 0; j < kBnU; ++j) {
        const int e = e0 + j * kBnThreads;
        xq[j] = e < nvec ? ld_stream<NT>(xp + e) : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int j = 0; j < kBnU; ++j) {
        const int e = e0 + j * kBnThreads;
        if (e < nvec) {
          v4f r;
#pragma unroll
          for (int u = 0; u < 4; ++u) r[u] = fma_rn(k1, (4 * e + u == sel) ? d : 0.0f, fma_rn(k2, xq[j][u] + b, k3));
          acc += (r[0] + r[1]) + (r[2] + r[3]);
          st_stream<NT>(op + e, r);
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < len; e += kBnThreads) {
      const float r = fma_rn(k1, (e == sel) ? d : 0.0f, fma_rn(k2, x[base + e] + b, k3));
      acc += r;
      dx[base + e] = r;
    }
  }
  if (dxpart) {
    block_reduce2(acc, unused, red);
    if (threadIdx.x == 0) dxpart[(size_t)row * gridDim.y + seg] = acc;
  }
}

int check_dims(const char* fn, int N, int C, int L, int act) {
  FPSG_REQUIRE(N > 0 && C > 0 && L > 0, FPSG_E_SHAPE, "%s: N,C,L must be positive (got %d,%d,%d)", fn, N, C, L);
  FPSG_REQUIRE(act >= 0 && act <= 2, FPSG_E_SHAPE, "%s: act must be 0 (none), 1 (relu) or 2 (leaky)", fn);
  FPSG_REQUIRE(C <= 65535 && (L + kBnSeg - 1) / kBnSeg <= 65535 && (long)N * C < (1L << 31), FPSG_E_LIMIT,
               "%s: C=%d, L=%d or N*C=%ld beyond the grid limits", fn, C, L, (long)N * C);
  return 0;
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_bn_workspace_floats(int N, int C, int L) {
  if (N <= 0 || C <= 0 || L <= 0) return 0;
  const size_t segs = ((size_t)L + fpsg::kBnSeg - 1) / fpsg::kBnSeg;
  return (size_t)C * fpsg::kBnSlices * 2 + (size_t)N * C * segs;     // channel partials + sum(dx) partials
}

extern "C" int fpsg_bn_act_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, int N, int C, int L,
                               int training, float eps, int act, float slope, float* y, float* chan,
                               float* batch_mean, float* batch_var_unbiased, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_fwd", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_bn_act_fwd: x and y must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // training == 2: evaluation mode with `chan` already holding the channel coefficients (the caller computed them once
  // for a block of calls during which the running statistics do not change: fpsg_bn_stats(training = 0))
  const bool chan_given = training == 2;
  if (chan_given) training = 0;
  if (!training && !chan_given) { FPSG_REQUIRE_PTR(running_mean); FPSG_REQUIRE_PTR(running_var); }
  if ((long)N * L <= kBnSmallMax) {
    if (!training && !chan_given) {
      hipLaunchKernelGGL(bn_eval_chan_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var,
                         gamma, beta, C, eps, chan);
      if ((rc = launch_status("fpsg_bn_act_fwd(eval)"))) return rc;
    }
    launch_small<0>(act, x, nullptr, gamma, beta, pre_bias, N, C, L, training, eps, slope, y, chan, batch_mean,
                    batch_var_unbiased, nullptr, nullptr, nullptr, running_mean, running_var, momentum, s);
    return launch_status("fpsg_bn_act_fwd(small)");
  }
  if (training) {
    FPSG_REQUIRE_PTR(ws);
    const int S = slices_for(N, L);
    const char* fold = getenv("FPSG_BN_FINALIZE_FOLD");
    if (fold && fold[0] == '1' && C <= kBnFoldChannels) {      // opt-in: the finalize in the last-arriving workgroup
      static std::atomic<unsigned> next_bank{0};
      unsigned* counters = nullptr;
      FPSG_REQUIRE(hipGetSymbolAddress(reinterpret_cast<void**>(&counters), HIP_SYMBOL(g_bn_fold_counters)) == hipSuccess &&
                       counters, FPSG_E_LIMIT, "fpsg_bn_act_fwd: the fold counters are not addressable");
      BnFoldArgs f{gamma, beta, (double)N * (double)L, eps, momentum, chan, batch_mean, batch_var_unbiased, running_mean,
                   running_var, counters + (size_t)(next_bank.fetch_add(1) % kBnFoldBanks) * kBnFoldChannels};
      const dim3 grid(S, C);
      if (beyond_cache((size_t)N * C * L * sizeof(float)))
        hipLaunchKernelGGL((bn_reduce_kernel<0, kActNone, true, true>), grid, dim3(kBnThreads), 0, s, x, nullptr, nullptr, pre_bias, N, C, L, S, 0.0f, ws, f);
      else
        hipLaunchKernelGGL((bn_reduce_kernel<0, kActNone, false, true>), grid, dim3(kBnThreads), 0, s, x, nullptr, nullptr, pre_bias, N, C, L, S, 0.0f, ws, f);
      if ((rc = launch_status("fpsg_bn_act_fwd(stats + finalize)"))) return rc;
    } else {
      launch_reduce<0>(kActNone, x, nullptr, nullptr, pre_bias, N, C, L, S, 0.0f, ws, s);
      if ((rc = launch_status("fpsg_bn_act_fwd(stats)"))) return rc;
      hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(64), 0, s, ws, gamma, beta, C, S,
                         (double)N * (double)L, eps, chan, batch_mean, batch_var_unbiased, running_mean, running_var,
                         momentum);
      if ((rc = launch_status("fpsg_bn_act_fwd(finalize)"))) return rc;
    }
  } else if (!chan_given) {
    hipLaunchKernelGGL(bn_eval_chan_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var,
                       gamma, beta, C, eps, chan);
    if ((rc = launch_status("fpsg_bn_act_fwd(eval)"))) return rc;
  }
  launch_apply<0>(act, x, nullptr, chan, nullptr, pre_bias, N, C, L, slope, y, nullptr, s);
  return launch_status("fpsg_bn_act_fwd(apply)");
}

namespace fpsg {
namespace {
int rows_segments(const char* fn, int ld, const int* seg_off, const int* seg_len, int nseg, int C, int act, bool backward,
                  BnSegs& g) {
  FPSG_REQUIRE(C > 0 && C <= 65535 * 64 && ld > 0, FPSG_E_SHAPE, "%s: C, ld must be positive (got %d, %d)", fn, C, ld);
  FPSG_REQUIRE(act >= kActNone && act <= kActLeaky, FPSG_E_SHAPE, "%s: unknown activation %d", fn, act);
  FPSG_REQUIRE(nseg >= 1 && nseg <= kBnMaxSegs, FPSG_E_LIMIT, "%s: %d segments (1..%d)", fn, nseg, kBnMaxSegs);
  FPSG_REQUIRE_PTR(seg_off); FPSG_REQUIRE_PTR(seg_len);
  FPSG_REQUIRE((ld & 3) == 0, FPSG_E_ALIGN, "%s: ld = %d must be a multiple of 4", fn, ld);
  g = BnSegs{};
  g.ld = ld;
  for (int i = 0; i < nseg; ++i) {
    FPSG_REQUIRE(seg_len[i] > 0 && seg_len[i] <= kBnSmallMax && seg_off[i] >= 0 && seg_off[i] + seg_len[i] <= ld,
                 FPSG_E_SHAPE, "%s: segment %d = [%d, +%d) does not fit a row of %d (a segment holds at most %d)", fn, i,
                 seg_off[i], seg_len[i], ld, kBnSmallMax);
    FPSG_REQUIRE((seg_off[i] & 3) == 0, FPSG_E_ALIGN, "%s: segment offset %d must be a multiple of 4", fn, seg_off[i]);
    for (int j = 0; j < i; ++j)
      FPSG_REQUIRE(seg_off[i] >= seg_off[j] + seg_len[j] || seg_off[j] >= seg_off[i] + seg_len[i], FPSG_E_SHAPE,
                   "%s: segments %d and %d overlap", fn, j, i);
    small_segment(g, i, 1, seg_off[i], seg_len[i], 1, backward);
  }
  return 0;
}
}  // namespace
}  // namespace fpsg

extern "C" int fpsg_bn_act_rows_fwd(const float* x, int ld, const int* seg_off, const int* seg_len, int nseg,
                                    const float* pre_bias, const float* gamma, const float* beta, int C, float eps,
                                    int act, float slope, float* y, float* chan, float* stats,
                                    fpsg_stream_t stream) {
  using namespace fpsg;
  BnSegs g;
  int rc = rows_segments("fpsg_bn_act_rows_fwd", ld, seg_off, seg_len, nseg, C, act, false, g);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(y); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_bn_act_rows_fwd: x and y must be 16-byte aligned");
  g.stat_stride = 2 * C;
  launch_small_segs<0>(act, x, nullptr, gamma, beta, pre_bias, 1, C, g, nseg, 1, eps, slope, y, chan, stats,
                       stats ? stats + C : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, -1.0f,
                       static_cast<hipStream_t>(stream));
  return launch_status("fpsg_bn_act_rows_fwd");
}

extern "C" int fpsg_bn_act_rows_bwd(const float* x, int ld, const int* seg_off, const int* seg_len, int nseg,
                                    const float* pre_bias, const float* dy, const float* chan, int C, int act,
                                    float slope, float* dx, float* dgamma, float* dbeta, float* dpre_bias,
                                    fpsg_stream_t stream) {
  using namespace fpsg;
  BnSegs g;
  int rc = rows_segments("fpsg_bn_act_rows_bwd", ld, seg_off, seg_len, nseg, C, act, true, g);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dx);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_bn_act_rows_bwd: x, dy and dx must be 16-byte aligned");
  launch_small_segs<1>(act, x, dy, nullptr, nullptr, pre_bias, 1, C, g, nseg, 1, 0.0f, slope, dx,
                       const_cast<float*>(chan), nullptr, nullptr, dgamma, dbeta, dpre_bias, nullptr, nullptr, -1.0f,
                       static_cast<hipStream_t>(stream));
  return launch_status("fpsg_bn_act_rows_bwd");
}

extern "C" int fpsg_bn_stats(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, int N, int C, int L,
                             int training, float eps, float* chan, float* batch_mean, float* batch_var_unbiased,
                             float* ws, const float* parts, int n_parts, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_stats", N, C, L, kActNone);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, FPSG_E_ALIGN, "fpsg_bn_stats: x must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (training) {
    int S = n_parts;
    if (parts) {
      FPSG_REQUIRE(n_parts > 0, FPSG_E_SHAPE, "fpsg_bn_stats: n_parts must be positive with parts (got %d)", n_parts);
    } else {
      FPSG_REQUIRE_PTR(ws);
      S = slices_for(N, L);
      launch_reduce<0>(kActNone, x, nullptr, nullptr, pre_bias, N, C, L, S, 0.0f, ws, s);
      if ((rc = launch_status("fpsg_bn_stats(stats)"))) return rc;
    }
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(64), 0, s, parts ? parts : ws, gamma, beta, C, S,
                       (double)N * (double)L, eps, chan, batch_mean, batch_var_unbiased, running_mean, running_var,
                       momentum);
    return launch_status("fpsg_bn_stats(finalize)");
  }
  FPSG_REQUIRE_PTR(running_mean); FPSG_REQUIRE_PTR(running_var);
  hipLaunchKernelGGL(bn_eval_chan_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var,
                     gamma, beta, C, eps, chan);
  return launch_status("fpsg_bn_stats(eval)");
}

extern "C" int fpsg_bn_act_bwd(const float* x, const float* pre_bias, const float* dy, const float* chan, int N,
                               int C, int L, int training, int act, float slope, float* dx, float* dgamma,
                               float* dbeta, float* dpre_bias, float* coef, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_bwd", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dx);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_bn_act_bwd: x, dy and dx must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((long)N * L <= kBnSmallMax) {
    launch_small<1>(act, x, dy, nullptr, nullptr, pre_bias, N, C, L, training, 0.0f, slope, dx,
                    const_cast<float*>(chan), nullptr, nullptr, dgamma, dbeta, dpre_bias, nullptr, nullptr, -1.0f, s);
    return launch_status("fpsg_bn_act_bwd(small)");
  }
  const int S = slices_for(N, L);
  launch_reduce<1>(act, x, dy, chan, pre_bias, N, C, L, S, slope, ws, s);
  if ((rc = launch_status("fpsg_bn_act_bwd(reduce)"))) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, ws, chan, C, S,
                     (double)N * (double)L, training, dgamma, dbeta, coef);
  if ((rc = launch_status("fpsg_bn_act_bwd(finalize)"))) return rc;
  float* dxpart = dpre_bias ? ws + (size_t)C * kBnSlices * 2 : nullptr;
  launch_apply<1>(act, x, dy, chan, coef, pre_bias, N, C, L, slope, dx, dxpart, s);
  if ((rc = launch_status("fpsg_bn_act_bwd(apply)"))) return rc;
  if (dpre_bias) {
    hipLaunchKernelGGL(bn_dxsum_kernel, dim3(C), dim3(64), 0, s, dxpart, N, C, (L + kBnSeg - 1) / kBnSeg, dpre_bias);
    return launch_status("fpsg_bn_act_bwd(dpre_bias)");
  }
  return 0;
}

// The sums + coefficient half of fpsg_bn_act_bwd alone (no dx pass): dgamma, dbeta and coef [3,C] of
// dx = k1 dz + k2 (x + pre_bias) + k3 for a consumer that forms dx itself while it reads x and dy
// (fpsg_conv_first_dw_fold).  The tensors of the large path only (N * L > the small-tensor limit).
extern "C" int fpsg_bn_act_bwd_coef(const float* x, const float* pre_bias, const float* dy, const float* chan, int N,
                                    int C, int L, int training, int act, float slope, float* dgamma, float* dbeta,
                                    float* coef, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_bwd_coef", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_bn_act_bwd_coef: x and dy must be 16-byte aligned");
  FPSG_REQUIRE((long)N * L > kBnSmallMax, FPSG_E_LIMIT, "fpsg_bn_act_bwd_coef: N*L = %ld is a small tensor (<= %d): use fpsg_bn_act_bwd",
               (long)N * L, kBnSmallMax);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int S = slices_for(N, L);
  launch_reduce<1>(act, x, dy, chan, pre_bias, N, C, L, S, slope, ws, s);
  if ((rc = launch_status("fpsg_bn_act_bwd_coef(reduce)"))) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, ws, chan, C, S,
                     (double)N * (double)L, training, dgamma, dbeta, coef);
  return launch_status("fpsg_bn_act_bwd_coef(finalize)");
}

extern "C" int fpsg_bn_act_bwd_parts(const float* x, const float* pre_bias, const float* dy, const float* chan, int N,
                                     int C, int L, int training, int act, float slope, float* dx, float* dgamma,
                                     float* dbeta, float* dpre_bias, float* coef, float* ws, const float* parts,
                                     int n_parts, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_bwd_parts", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dx); FPSG_REQUIRE_PTR(parts);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef);
  FPSG_REQUIRE(n_parts > 0, FPSG_E_SHAPE, "fpsg_bn_act_bwd_parts: n_parts must be positive (got %d)", n_parts);
  FPSG_REQUIRE(dpre_bias == nullptr || ws != nullptr, FPSG_E_NULL, "fpsg_bn_act_bwd_parts: dpre_bias needs the workspace");
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_bn_act_bwd_parts: x, dy and dx must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, parts, chan, C, n_parts,
                     (double)N * (double)L, training, dgamma, dbeta, coef);
  if ((rc = launch_status("fpsg_bn_act_bwd_parts(finalize)"))) return rc;
  float* dxpart = dpre_bias ? ws + (size_t)C * kBnSlices * 2 : nullptr;
  launch_apply<1>(act, x, dy, chan, coef, pre_bias, N, C, L, slope, dx, dxpart, s);
  if ((rc = launch_status("fpsg_bn_act_bwd_parts(apply)"))) return rc;
  if (dpre_bias) {
    hipLaunchKernelGGL(bn_dxsum_kernel, dim3(C), dim3(64), 0, s, dxpart, N, C, (L + kBnSeg - 1) / kBnSeg, dpre_bias);
    return launch_status("fpsg_bn_act_bwd_parts(dpre_bias)");
  }
  return 0;
}

extern "C" size_t fpsg_bn_pool_workspace_floats(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 1 || W <= 1) return 0;
  const int RP = fpsg::pool_rows_per_item(W);
  const size_t items = (size_t)(H / 2 + RP - 1) / RP;
  return (size_t)C * fpsg::kBnSlices * 2 + (size_t)N * C * items;
}

extern "C" int fpsg_bn_act_pool_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, int N, int C, int H, int W,
                                    int training, float eps, int act, float slope, float* y_pooled, float* chan,
                                    float* batch_mean, float* batch_var_unbiased, float* ws, const float* parts,
                                    int n_parts, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_pool_dims("fpsg_bn_act_pool_fwd", N, C, H, W, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(y_pooled); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_pooled)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_bn_act_pool_fwd: x and y_pooled must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = H * W;
  const bool chan_given = training == 2;        // as in fpsg_bn_act_fwd
  if (chan_given) training = 0;
  if (training) {
    int S = n_parts;
    if (parts) {
      FPSG_REQUIRE(n_parts > 0, FPSG_E_SHAPE, "fpsg_bn_act_pool_fwd: n_parts must be positive with parts (got %d)", n_parts);
    } else {
      FPSG_REQUIRE_PTR(ws);
      S = slices_for(N, L);
      launch_reduce<0>(kActNone, x, nullptr, nullptr, pre_bias, N, C, L, S, 0.0f, ws, s);
      if ((rc = launch_status("fpsg_bn_act_pool_fwd(stats)"))) return rc;
    }
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(64), 0, s, parts ? parts : ws, gamma, beta, C, S,
                       (double)N * (double)L, eps, chan, batch_mean, batch_var_unbiased, running_mean, running_var,
                       momentum);
    if ((rc = launch_status("fpsg_bn_act_pool_fwd(finalize)"))) return rc;
  } else if (!chan_given) {
    FPSG_REQUIRE_PTR(running_mean); FPSG_REQUIRE_PTR(running_var);
    hipLaunchKernelGGL(bn_eval_chan_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var,
                       gamma, beta, C, eps, chan);
    if ((rc = launch_status("fpsg_bn_act_pool_fwd(eval)"))) return rc;
  }
  launch_pool_apply<0>(act, x, nullptr, chan, nullptr, pre_bias, N, C, H, W, slope, y_pooled, nullptr, s);
  return launch_status("fpsg_bn_act_pool_fwd(apply)");
}

extern "C" int fpsg_bn_act_pool_bwd(const float* x, const float* pre_bias, const float* dy_pooled, const float* chan,
                                    int N, int C, int H, int W, int training, int act, float slope, float* dx,
                                    float* dgamma, float* dbeta, float* dpre_bias, float* coef, float* ws,
                                    fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_pool_dims("fpsg_bn_act_pool_bwd", N, C, H, W, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(dy_pooled); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dx);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy_pooled) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_bn_act_pool_bwd: x, dy_pooled and dx must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int RP = pool_rows_per_item(W);
  const int per_plane = (H / 2 + RP - 1) / RP;
  const long items = (long)N * per_plane;
  const int S = items < kBnSlices ? (int)items : kBnSlices;
  launch_pool_reduce(act, x, dy_pooled, chan, pre_bias, N, C, H, W, S, slope, ws, s);
  if ((rc = launch_status("fpsg_bn_act_pool_bwd(reduce)"))) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, ws, chan, C, S,
                     (double)N * (double)H * (double)W, training, dgamma, dbeta, coef);
  if ((rc = launch_status("fpsg_bn_act_pool_bwd(finalize)"))) return rc;
  float* dxpart = dpre_bias ? ws + (size_t)C * kBnSlices * 2 : nullptr;
  launch_pool_apply<2>(act, x, dy_pooled, chan, coef, pre_bias, N, C, H, W, slope, dx, dxpart, s);
  if ((rc = launch_status("fpsg_bn_act_pool_bwd(apply)"))) return rc;
  if (dpre_bias) {
    hipLaunchKernelGGL(bn_dxsum_kernel, dim3(C), dim3(64), 0, s, dxpart, N, C, per_plane, dpre_bias);
    return launch_status("fpsg_bn_act_pool_bwd(dpre_bias)");
  }
  return 0;
}

extern "C" size_t fpsg_bn_max_workspace_floats(int N, int C, int L) {
  if (N <= 0 || C <= 0 || L <= 0) return 0;
  const size_t segs = ((size_t)L + fpsg::kBnSeg - 1) / fpsg::kBnSeg;
  // channel partials + per-(row, segment) extremes (4 words) / sum(dx) partials + dz [N*C]
  return (size_t)C * fpsg::kBnSlices * 2 + (size_t)N * C * segs * 4 + (size_t)N * C;
}

extern "C" size_t fpsg_bn_max_dz_offset(int N, int C, int L) {
  if (N <= 0 || C <= 0 || L <= 0) return 0;
  const size_t segs = ((size_t)L + fpsg::kBnSeg - 1) / fpsg::kBnSeg;
  return (size_t)C * fpsg::kBnSlices * 2 + (size_t)N * C * segs * 4;     // dz [N*C] is the workspace's last block
}

extern "C" int fpsg_bn_act_max_fwd(const float* x, const float* pre_bias, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, int N, int C, int L,
                                   int training, float eps, int act, float slope, float* out, int32_t* idx,
                                   float* chan, float* batch_mean, float* batch_var_unbiased, float* ws,
                                   fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_max_fwd", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(out); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int segs = (L + kBnSeg - 1) / kBnSeg;
  // one item per WAVE and pass: S workgroups of 4 waves per channel
  const int items_max = N * segs;
  const int S = items_max >= 4 * kBnSlices ? kBnSlices : (items_max + 3) / 4;
  RowExt* ext = reinterpret_cast<RowExt*>(ws + (size_t)C * kBnSlices * 2);
  dim3 grid(C, S);
  const bool chan_given = training == 2;        // as in fpsg_bn_act_fwd
  if (chan_given) training = 0;
  if (training) {
    if (beyond_cache((size_t)N * C * L * sizeof(float))) hipLaunchKernelGGL((bn_reduce_ext_kernel<1, true>), grid, dim3(kBnThreads), 0, s, x, pre_bias, N, C, L, S, ws, ext);
    else hipLaunchKernelGGL(bn_reduce_ext_kernel<1>, grid, dim3(kBnThreads), 0, s, x, pre_bias, N, C, L, S, ws, ext);
    if ((rc = launch_status("fpsg_bn_act_max_fwd(stats)"))) return rc;
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(64), 0, s, ws, gamma, beta, C, S,
                       (double)N * (double)L, eps, chan, batch_mean, batch_var_unbiased, running_mean, running_var,
                       momentum);
    if ((rc = launch_status("fpsg_bn_act_max_fwd(finalize)"))) return rc;
  } else {
    if (!chan_given) { FPSG_REQUIRE_PTR(running_mean); FPSG_REQUIRE_PTR(running_var); }
    hipLaunchKernelGGL(bn_reduce_ext_kernel<0>, grid, dim3(kBnThreads), 0, s, x, pre_bias, N, C, L, S, ws, ext);
    if ((rc = launch_status("fpsg_bn_act_max_fwd(extremes)"))) return rc;
    if (!chan_given) {
      hipLaunchKernelGGL(bn_eval_chan_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var,
                         gamma, beta, C, eps, chan);
      if ((rc = launch_status("fpsg_bn_act_max_fwd(eval)"))) return rc;
    }
  }
  const int rows = N * C;
  dim3 og((rows + 255) / 256);
  if (act == kActRelu) hipLaunchKernelGGL(bn_max_out_kernel<kActRelu>, og, dim3(256), 0, s, ext, chan, rows, C, segs, slope, out, idx);
  else if (act == kActLeaky) hipLaunchKernelGGL(bn_max_out_kernel<kActLeaky>, og, dim3(256), 0, s, ext, chan, rows, C, segs, slope, out, idx);
  else hipLaunchKernelGGL(bn_max_out_kernel<kActNone>, og, dim3(256), 0, s, ext, chan, rows, C, segs, slope, out, idx);
  return launch_status("fpsg_bn_act_max_fwd(out)");
}

// The coefficient pass of fpsg_bn_act_max_bwd alone: dz [N,C] (left in the workspace at the offset that function uses: after
// the channel partials and the dx partials), dgamma, dbeta and coef [3][C] = k1, k2, k3 of
//   dx'[n,c,l] = k1_c dz[n,c] [l = idx[n,c]] + k2_c (x[n,c,l] + pre_bias_c) + k3_c .
// For callers that do not form the dense dx: with x = W a the two products over it reduce to 128 x 128 Gram-matrix
// algebra plus a gather and a scatter of N*C columns (fpsg_amd/fused_bn.py: _ConvBNActMax).
extern "C" int fpsg_bn_act_max_bwd_coef(const float* x, const float* pre_bias, const float* gout, const int32_t* idx,
                                        const float* chan, int N, int C, int L, int training, int act, float slope,
                                        float* dgamma, float* dbeta, float* coef, float* ws, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_max_bwd_coef", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(gout); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int segs = (L + kBnSeg - 1) / kBnSeg;
  float* dz = ws + (size_t)C * kBnSlices * 2 + (size_t)N * C * segs * 4;
#define FPSG_MAXB(A) hipLaunchKernelGGL(bn_max_bwd_coef_kernel<A>, dim3(C), dim3(64), 0, s, x, pre_bias, gout, idx, chan, \
                                        N, C, L, training, slope, dz, dgamma, dbeta, coef)
  if (act == kActRelu) FPSG_MAXB(kActRelu); else if (act == kActLeaky) FPSG_MAXB(kActLeaky); else FPSG_MAXB(kActNone);
#undef FPSG_MAXB
  return launch_status("fpsg_bn_act_max_bwd_coef");
}

extern "C" int fpsg_bn_act_max_bwd(const float* x, const float* pre_bias, const float* gout, const int32_t* idx,
                                   const float* chan, int N, int C, int L, int training, int act, float slope,
                                   float* dx, float* dgamma, float* dbeta, float* dpre_bias, float* coef, float* ws,
                                   fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = check_dims("fpsg_bn_act_max_bwd", N, C, L, act);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(gout); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(chan); FPSG_REQUIRE_PTR(dx);
  FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta); FPSG_REQUIRE_PTR(coef); FPSG_REQUIRE_PTR(ws);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0, FPSG_E_ALIGN,
               "fpsg_bn_act_max_bwd: x and dx must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int segs = (L + kBnSeg - 1) / kBnSeg;
  float* dxpart = ws + (size_t)C * kBnSlices * 2;
  float* dz = dxpart + (size_t)N * C * segs * 4;
#define FPSG_MAXB(A) hipLaunchKernelGGL(bn_max_bwd_coef_kernel<A>, dim3(C), dim3(64), 0, s, x, pre_bias, gout, idx, chan, \
                                        N, C, L, training, slope, dz, dgamma, dbeta, coef)
  if (act == kActRelu) FPSG_MAXB(kActRelu); else if (act == kActLeaky) FPSG_MAXB(kActLeaky); else FPSG_MAXB(kActNone);
#undef FPSG_MAXB
  if ((rc = launch_status("fpsg_bn_act_max_bwd(coef)"))) return rc;
  dim3 grid((unsigned)((size_t)N * C), segs);
  hipLaunchKernelGGL(beyond_cache((size_t)N * C * L * sizeof(float)) ? bn_max_apply_kernel<true> : bn_max_apply_kernel<false>, grid, dim3(kBnThreads), 0, s, x, pre_bias, dz, idx, coef, C, L, dx,
                     dpre_bias ? dxpart : nullptr);
  if ((rc = launch_status("fpsg_bn_act_max_bwd(apply)"))) return rc;
  if (dpre_bias) {
    hipLaunchKernelGGL(bn_dxsum_kernel, dim3(C), dim3(64), 0, s, dxpart, N, C, segs, dpre_bias);
    return launch_status("fpsg_bn_act_max_bwd(dpre_bias)");
  }
  return 0;
}
