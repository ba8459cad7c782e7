// adam.hip -- K7: the optimizer step of the train loop as ONE HBM stream, for gfx950.
//
// Reference: torch.optim.Adam(lr, betas=(.9,.999)) stepped once per episode, src/trainNetwork.py:
// 118-123 and 144 ("optimizer over 77 M params -- elementwise, HBM-bound").
// The library's fused Adam is a multi-tensor launch over the model's ~600 separate parameter
// tensors (10 chunked launches, 2.2 TB/s); with parameters, gradients and both moments each living
// in one flat fp32 buffer (fpsg_amd/optim.py) the step is a single float4 grid-stride stream:
//   g  = grad * grad_scale            (the mean over the step's episodes, folded in: no separate
//                                      pass over the 310 MB gradient buffer)
//   m  = m + (1-b1) * (g - m)         (torch: exp_avg.lerp_(grad, 1-b1))
//   v  = b2 * v + (1-b2) * g * g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// 4 reads + 3 writes of the buffer = 28 B per parameter (2.17 GB for the 77.4 M of the full model).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kAdamThreads = 256;

// Non-temporal accesses to the four streams (each larger than the Infinity Cache, touched once per step) were
// measured: 450 vs 457 us, no difference -- plain accesses (FPSG_ADAM_NT=1 builds the other form).
#ifndef FPSG_ADAM_NT
#define FPSG_ADAM_NT 0
#endif
__device__ __forceinline__ v4f ld4(const float* p, size_t i) {
#if FPSG_ADAM_NT
  return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + i);
#else
  return reinterpret_cast<const v4f*>(p)[i];
#endif
}
__device__ __forceinline__ void st4(float* p, size_t i, v4f v) {
#if FPSG_ADAM_NT
  __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p) + i);
#else
  reinterpret_cast<v4f*>(p)[i] = v;
#endif
}

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                            size_t n, float step_size, float b1, float b2, float eps,
                                                            float inv_sqrt_bc2, float gscale) {
  const size_t stride = (size_t)gridDim.x * kAdamThreads;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  auto update = [&](v4f& pv, const v4f& gv, v4f& mv, v4f& vv) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float gg = gv[u] * gscale;
      mv[u] = fma_rn(omb1, gg - mv[u], mv[u]);
      vv[u] = fma_rn(omb2 * gg, gg, b2 * vv[u]);
      const float denom = fma_rn(__fsqrt_rn(vv[u]), inv_sqrt_bc2, eps);
      pv[u] = fma_rn(-step_size, mv[u] / denom, pv[u]);
    }
  };
  // two vectors of each stream per pass: eight 16-byte loads in flight per thread before the first use
  size_t i = (size_t)blockIdx.x * kAdamThreads + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const size_t j = i + stride;
    v4f pa = ld4(p, i), pb = ld4(p, j);
    const v4f ga = ld4(g, i), gb = ld4(g, j);
    v4f ma = ld4(m, i), mb = ld4(m, j);
    v4f va = ld4(v, i), vb = ld4(v, j);
    update(pa, ga, ma, va);
    update(pb, gb, mb, vb);
    st4(p, i, pa); st4(m, i, ma); st4(v, i, va);
    st4(p, j, pb); st4(m, j, mb); st4(v, j, vb);
  }
  if (i < n4) {
    v4f pv = ld4(p, i);
    const v4f gv = ld4(g, i);
    v4f mv = ld4(m, i);
    v4f vv = ld4(v, i);
    update(pv, gv, mv, vv);
    st4(p, i, pv);
    st4(m, i, mv);
    st4(v, i, vv);
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += kAdamThreads) {
      const float gg = g[i] * gscale;
      const float mm = fma_rn(omb1, gg - m[i], m[i]);
      const float vq = fma_rn(omb2 * gg, gg, b2 * v[i]);
      const float denom = fma_rn(__fsqrt_rn(vq), inv_sqrt_bc2, eps);
      p[i] = fma_rn(-step_size, mm / denom, p[i]);
      m[i] = mm;
      v[i] = vq;
    }
  }
}

// The same update with the gradient read from the per-parameter tensors autograd produced instead
// of a flat copy of them: segment s of the flat buffers is [seg_off[s], seg_off[s+1]) and its
// gradient starts at gtab[s] (null: no gradient = zero).  A step of ONE episode on one rank then
// needs no 310 MB gather.  A vector of four elements looks its segment up by bisection (10 steps
// over ~600 cached offsets, free next to 28 B of traffic per element); a vector that straddles
// two segments is done element by element.
__device__ __forceinline__ int segment_of(const long long* seg_off, int nseg, long long i) {
  int lo = 0, hi = nseg;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

constexpr int kSegLds = 2048;       // segment offsets kept in LDS by flat_accumulate_kernel (16 KB)

// the (up to) four gradient elements of flat positions i .. i+cnt-1 from the segment table (0 where a segment has none)
__device__ __forceinline__ bool gather_grad4(const float* const* __restrict__ gtab, const long long* seg_off,
                                             int nseg, long long i, int cnt, float (&gg)[4]) {
  int sg = segment_of(seg_off, nseg, i);
  bool any = false;
  gg[0] = gg[1] = gg[2] = gg[3] = 0.0f;
  if (i + cnt <= seg_off[sg + 1]) {
    const float* g = gtab[sg];
    if (g) {
      any = true;
      const long long base = i - seg_off[sg];
      // a gradient tensor usually starts 256-B aligned, but one that autograd took over from a view
      // (e.g. a slice of a stacked tensor's gradient) may start anywhere: test the address itself
      if (cnt == 4 && ((reinterpret_cast<uintptr_t>(g + base) & 15) == 0)) {
        const v4f q = *reinterpret_cast<const v4f*>(g + base);
        gg[0] = q[0]; gg[1] = q[1]; gg[2] = q[2]; gg[3] = q[3];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < cnt) gg[u] = g[base + u];
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u >= cnt) break;
      while (sg + 1 < nseg && i + u >= seg_off[sg + 1]) ++sg;      // never past the table, whatever seg_off[nseg] says
      const float* g = gtab[sg];
      any = any || g != nullptr;
      gg[u] = g ? g[i + u - seg_off[sg]] : 0.0f;
    }
  }
  return any;
}

// The same for a WHOLE vector when the lanes of the wave hold consecutive vectors (grid-stride loops): one table search
// for the wave's first vector -- on wave-uniform values, so the ten dependent reads are scalar loads -- and a range
// test per lane; only a wave that straddles a segment boundary (~600 of ~300,000) searches per lane.  Same values.
__device__ __forceinline__ bool gather_grad4_wave(const float* const* __restrict__ gtab, const long long* __restrict__ seg_off,
                                                  int nseg, long long i, float (&gg)[4]) {
  const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)i);
  const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)i >> 32));
  const long long i0 = (long long)(((unsigned long long)hi32 << 32) | lo32);
  const int sg0 = segment_of(seg_off, nseg, i0);
  const long long lo = seg_off[sg0], hi = seg_off[sg0 + 1];
  const bool inside = i >= lo && i + 4 <= hi;
  if (__builtin_amdgcn_ballot_w64(!inside) != 0ull) return gather_grad4(gtab, seg_off, nseg, i, 4, gg);
  const float* g = gtab[sg0];
  gg[0] = gg[1] = gg[2] = gg[3] = 0.0f;
  if (!g) return false;
  const long long base = i - lo;
  if ((reinterpret_cast<uintptr_t>(g + base) & 15) == 0) {
    const v4f q = *reinterpret_cast<const v4f*>(g + base);
    gg[0] = q[0]; gg[1] = q[1]; gg[2] = q[2]; gg[3] = q[3];
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) gg[u] = g[base + u];
  }
  return true;
}

// flat[i] (+)= the segment table's gradient: the episode's fresh gradient tensors added into (or, first episode of a
// step, copied over) the step's flat gradient buffer in ONE stream at HBM rate -- the multi-tensor add it replaces
// runs ~25 chunked launches at half of it.  accumulate = 0 writes zeros where a parameter has no gradient.
__global__ __launch_bounds__(kAdamThreads) void flat_accumulate_kernel(float* __restrict__ flat,
                                                                       const float* const* __restrict__ gtab,
                                                                       const long long* __restrict__ seg_off, int nseg,
                                                                       size_t n4, size_t n, int accumulate) {
  // the segment table in LDS (when it fits): the bisection is ten dependent reads per vector -- from L2 they bound the
  // kernel at 3.2 TB/s of its 12 B per element
  __shared__ long long soff[kSegLds];
  const long long* so = seg_off;
  if (nseg + 1 <= kSegLds) {
    for (int e = threadIdx.x; e <= nseg; e += kAdamThreads) soff[e] = seg_off[e];
    __syncthreads();
    so = soff;
  }
  const size_t stride = (size_t)gridDim.x * kAdamThreads;
  size_t j = (size_t)blockIdx.x * kAdamThreads + threadIdx.x;
  // whole vectors, kAccU per trip: the flat loads go out first, then the table searches, then the gradient loads --
  // each group in flight together (one vector per trip was two dependent round trips per 16 bytes)
  constexpr int kAccU = 4;
  for (; j + (kAccU - 1) * stride < n4; j += kAccU * stride) {
    v4f f[kAccU];
#pragma unroll
    for (int u = 0; u < kAccU; ++u) {
      f[u] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      if (accumulate) f[u] = reinterpret_cast<const v4f*>(flat)[j + u * stride];
    }
    float gg[kAccU][4];
    bool any[kAccU];
#pragma unroll
    for (int u = 0; u < kAccU; ++u) any[u] = gather_grad4(gtab, so, nseg, (long long)(4 * (j + u * stride)), 4, gg[u]);
#pragma unroll
    for (int u = 0; u < kAccU; ++u) {
      if (accumulate && !any[u]) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) f[u][e] += gg[u][e];
      reinterpret_cast<v4f*>(flat)[j + u * stride] = f[u];
    }
  }
  for (; j < n4 + 1; j += stride) {
    const long long i = (long long)(4 * j);
    const int cnt = j < n4 ? 4 : (int)(n - 4 * n4);
    if (cnt == 0) break;
    v4f f = {0.0f, 0.0f, 0.0f, 0.0f};
    if (accumulate && cnt == 4) f = reinterpret_cast<const v4f*>(flat)[j];
    float gg[4];
    const bool any = gather_grad4(gtab, so, nseg, i, cnt, gg);
    if (accumulate && !any) continue;
    if (cnt == 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) f[u] += gg[u];
      reinterpret_cast<v4f*>(flat)[j] = f;
    } else {
      for (int u = 0; u < cnt; ++u) flat[(size_t)i + u] = (accumulate ? flat[(size_t)i + u] : 0.0f) + gg[u];
    }
  }
}

// The same for up to kAccTables gradient tables at once: flat (+)= g_0 + g_1 + ... added in table order, i.e. the
// fp32 sums of that many consecutive flat_accumulate_kernel launches, bit for bit -- but flat is read and written once
// instead of once per table (8 episodes of a step: 2.8 GB of traffic instead of 7.4).  gtabs [ntab][nseg].
constexpr int kAccTables = 8;

__global__ __launch_bounds__(kAdamThreads) void flat_accumulate_tables_kernel(float* __restrict__ flat,
                                                                              const float* const* __restrict__ gtabs,
                                                                              const long long* __restrict__ seg_off,
                                                                              int nseg, int ntab, size_t n4, size_t n,
                                                                              int accumulate) {
  __shared__ long long soff[kSegLds];
  const long long* so = seg_off;
  if (nseg + 1 <= kSegLds) {
    for (int e = threadIdx.x; e <= nseg; e += kAdamThreads) soff[e] = seg_off[e];
    __syncthreads();
    so = soff;
  }
  const size_t stride = (size_t)gridDim.x * kAdamThreads;
  size_t j = (size_t)blockIdx.x * kAdamThreads + threadIdx.x;
  // whole vectors, two per trip: the flat loads, one table search per vector (the tables share the layout), then the
  // vectors of all tables in flight together; the adds run in table order
  constexpr int kU = 2;
  for (; j + (kU - 1) * stride < n4; j += kU * stride) {
    v4f f[kU];
    int sg[kU];
    bool fast[kU];
    long long base[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      f[u] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
      if (accumulate) f[u] = reinterpret_cast<const v4f*>(flat)[j + u * stride];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long i = (long long)(4 * (j + u * stride));
      sg[u] = segment_of(so, nseg, i);
      fast[u] = i + 4 <= so[sg[u] + 1];
      base[u] = i - so[sg[u]];
    }
    v4f g[kU][kAccTables];
    bool any[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      any[u] = false;
#pragma unroll
      for (int t = 0; t < kAccTables; ++t) {
        g[u][t] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        if (t >= ntab) continue;
        const float* const* tab = gtabs + (size_t)t * nseg;
        if (fast[u]) {
          const float* gp = tab[sg[u]];
          if (gp) {
            any[u] = true;
            if ((reinterpret_cast<uintptr_t>(gp + base[u]) & 15) == 0) {
              g[u][t] = *reinterpret_cast<const v4f*>(gp + base[u]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) g[u][t][e] = gp[base[u] + e];
            }
          }
        } else {
          float gg[4];
          any[u] = gather_grad4(tab, so, nseg, (long long)(4 * (j + u * stride)), 4, gg) || any[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) g[u][t][e] = gg[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (accumulate && !any[u]) continue;
#pragma unroll
      for (int t = 0; t < kAccTables; ++t) {
        if (t >= ntab) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[u][e] += g[u][t][e];
      }
      reinterpret_cast<v4f*>(flat)[j + u * stride] = f[u];
    }
  }
  for (; j < n4 + 1; j += stride) {
    const long long i = (long long)(4 * j);
    const int cnt = j < n4 ? 4 : (int)(n - 4 * n4);
    if (cnt == 0) break;
    float f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (accumulate) {
      for (int u = 0; u < cnt; ++u) f[u] = flat[(size_t)i + u];
    }
    bool any = false;
    for (int t = 0; t < ntab; ++t) {
      float gg[4];
      any = gather_grad4(gtabs + (size_t)t * nseg, so, nseg, i, cnt, gg) || any;
      for (int u = 0; u < cnt; ++u) f[u] += gg[u];
    }
    if (accumulate && !any) continue;
    for (int u = 0; u < cnt; ++u) flat[(size_t)i + u] = f[u];
  }
}

__global__ __launch_bounds__(kAdamThreads) void adam_ptr_kernel(float* __restrict__ p, const float* const* __restrict__ gtab,
                                                                const long long* __restrict__ seg_off, int nseg,
                                                                float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                                size_t n, float step_size, float b1, float b2,
                                                                float eps, float inv_sqrt_bc2, float gscale) {
  const size_t stride = (size_t)gridDim.x * kAdamThreads;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  size_t j = (size_t)blockIdx.x * kAdamThreads + threadIdx.x;
  // whole vectors, two per trip: six flat loads out, two table searches, two gradient loads, then the arithmetic
  // (one vector per trip: the search and the gradient load sat between the flat loads and their use every 16 bytes)
  for (; j + stride < n4; j += 2 * stride) {
    v4f pv[2], mv[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pv[u] = reinterpret_cast<const v4f*>(p)[j + u * stride];
      mv[u] = reinterpret_cast<const v4f*>(m)[j + u * stride];
      vv[u] = reinterpret_cast<const v4f*>(v)[j + u * stride];
    }
    float gg[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) gather_grad4_wave(gtab, seg_off, nseg, (long long)(4 * (j + u * stride)), gg[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gq = gg[u][e] * gscale;
        mv[u][e] = fma_rn(omb1, gq - mv[u][e], mv[u][e]);
        vv[u][e] = fma_rn(omb2 * gq, gq, b2 * vv[u][e]);
        const float denom = fma_rn(__fsqrt_rn(vv[u][e]), inv_sqrt_bc2, eps);
        pv[u][e] = fma_rn(-step_size, mv[u][e] / denom, pv[u][e]);
      }
      reinterpret_cast<v4f*>(p)[j + u * stride] = pv[u];
      reinterpret_cast<v4f*>(m)[j + u * stride] = mv[u];
      reinterpret_cast<v4f*>(v)[j + u * stride] = vv[u];
    }
  }
  for (; j < n4 + 1; j += stride) {
    const long long i = (long long)(4 * j);
    const int cnt = j < n4 ? 4 : (int)(n - 4 * n4);          // the last "vector" is the tail
    if (cnt == 0) break;
    // the flat streams first: their loads fly while the table is searched and the gradient fetched
    v4f pv = {0.0f, 0.0f, 0.0f, 0.0f}, mv = pv, vv = pv;
    if (cnt == 4) {
      pv = reinterpret_cast<const v4f*>(p)[j];
      mv = reinterpret_cast<const v4f*>(m)[j];
      vv = reinterpret_cast<const v4f*>(v)[j];
    }
    float gg[4];
    gather_grad4(gtab, seg_off, nseg, i, cnt, gg);
    if (cnt == 4) {                                            // flat buffers: aligned vectors
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float gq = gg[u] * gscale;
        mv[u] = fma_rn(omb1, gq - mv[u], mv[u]);
        vv[u] = fma_rn(omb2 * gq, gq, b2 * vv[u]);
        const float denom = fma_rn(__fsqrt_rn(vv[u]), inv_sqrt_bc2, eps);
        pv[u] = fma_rn(-step_size, mv[u] / denom, pv[u]);
      }
      reinterpret_cast<v4f*>(p)[j] = pv;
      reinterpret_cast<v4f*>(m)[j] = mv;
      reinterpret_cast<v4f*>(v)[j] = vv;
    } else {
      for (int u = 0; u < cnt; ++u) {
        const size_t e = (size_t)i + u;
        const float gq = gg[u] * gscale;
        const float mm = fma_rn(omb1, gq - m[e], m[e]);
        const float vq = fma_rn(omb2 * gq, gq, b2 * v[e]);
        const float denom = fma_rn(__fsqrt_rn(vq), inv_sqrt_bc2, eps);
        p[e] = fma_rn(-step_size, mm / denom, p[e]);
        m[e] = mm;
        v[e] = vq;
      }
    }
  }
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_adam_step_segments(float* param, const float* const* grad_ptrs, const long long* seg_off, int nseg,
                                       float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                                       float eps, int step, float grad_scale, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(n > 0 && step >= 1 && nseg > 0, FPSG_E_SHAPE,
               "fpsg_adam_step_segments: n, nseg must be positive and step >= 1 (got %zu, %d, %d)", n, nseg, step);
  FPSG_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, FPSG_E_SHAPE,
               "fpsg_adam_step_segments: betas must lie in [0,1) and eps be non-negative");
  FPSG_REQUIRE_PTR(param); FPSG_REQUIRE_PTR(exp_avg); FPSG_REQUIRE_PTR(exp_avg_sq);
  FPSG_REQUIRE(grad_ptrs != nullptr && seg_off != nullptr, FPSG_E_NULL, "fpsg_adam_step_segments: null table");
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) |
                 reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_adam_step_segments: param, exp_avg and exp_avg_sq must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 1 + kAdamThreads - 1) / kAdamThreads;
  // no grid-stride cap, as adam_kernel (c3 same box, three alternating pairs: 43.99-44.01 -> 44.01-44.14 episodes/s)
  hipLaunchKernelGGL(adam_ptr_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, static_cast<hipStream_t>(stream),
                     param, grad_ptrs, seg_off, nseg, exp_avg, exp_avg_sq, n4, n, step_size, beta1, beta2, eps,
                     inv_sqrt_bc2, grad_scale);
  return launch_status("fpsg_adam_step_segments");
}

extern "C" int fpsg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, int step, float grad_scale, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(n > 0 && step >= 1, FPSG_E_SHAPE, "fpsg_adam_step: n must be positive and step >= 1 (got %zu, %d)", n, step);
  FPSG_REQUIRE(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, FPSG_E_SHAPE,
               "fpsg_adam_step: betas must lie in [0,1) and eps be non-negative");
  FPSG_REQUIRE_PTR(param); FPSG_REQUIRE_PTR(grad); FPSG_REQUIRE_PTR(exp_avg); FPSG_REQUIRE_PTR(exp_avg_sq);
  FPSG_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                 reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_adam_step: the four buffers must be 16-byte aligned");
  // bias corrections in double, as torch does for python-scalar steps
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const size_t n4 = n / 4;
  size_t blocks = (n4 + kAdamThreads - 1) / kAdamThreads;
  // One vector per thread, no grid-stride cap (round 4): rounds 2-3 launched 16 workgroups per CU walking the buffers with a
  // 16 MB stride, two vectors per trip; alternating on one box (tools/bench_adam.py, 5 pairs) that form took 392-438 us,
  // this one 356-389 us for the model's 77.4 M parameters (0.59 -> 0.64-0.75 of the HBM peak) -- a workgroup's 4 KB pieces of
  // the seven streams stay next to its neighbours' in time, whatever the residency.  (2 / 4 vectors per thread: 361-408 us.)
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, static_cast<hipStream_t>(stream), param,
                     grad, exp_avg, exp_avg_sq, n4, n, step_size, beta1, beta2, eps, inv_sqrt_bc2, grad_scale);
  return launch_status("fpsg_adam_step");
}

extern "C" int fpsg_flat_accumulate_segments(float* flat, const float* const* grad_ptrs, const long long* seg_off, int nseg,
                                             size_t n, int accumulate, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(n > 0 && nseg > 0, FPSG_E_SHAPE, "fpsg_flat_accumulate_segments: n, nseg must be positive (got %zu, %d)", n, nseg);
  FPSG_REQUIRE_PTR(flat);
  FPSG_REQUIRE(grad_ptrs != nullptr && seg_off != nullptr, FPSG_E_NULL, "fpsg_flat_accumulate_segments: null table");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 15) == 0, FPSG_E_ALIGN, "fpsg_flat_accumulate_segments: flat must be 16-byte aligned");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 1 + kAdamThreads - 1) / kAdamThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(flat_accumulate_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, static_cast<hipStream_t>(stream),
                     flat, grad_ptrs, seg_off, nseg, n4, n, accumulate);
  return launch_status("fpsg_flat_accumulate_segments");
}

extern "C" int fpsg_flat_accumulate_tables(float* flat, const float* const* grad_ptrs, const long long* seg_off, int nseg,
                                           int ntab, size_t n, int accumulate, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(n > 0 && nseg > 0, FPSG_E_SHAPE, "fpsg_flat_accumulate_tables: n, nseg must be positive (got %zu, %d)", n, nseg);
  FPSG_REQUIRE(ntab >= 1 && ntab <= kAccTables, FPSG_E_LIMIT, "fpsg_flat_accumulate_tables: %d tables (1..%d)", ntab, kAccTables);
  FPSG_REQUIRE_PTR(flat);
  FPSG_REQUIRE(grad_ptrs != nullptr && seg_off != nullptr, FPSG_E_NULL, "fpsg_flat_accumulate_tables: null table");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 15) == 0, FPSG_E_ALIGN, "fpsg_flat_accumulate_tables: flat must be 16-byte aligned");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 1 + kAdamThreads - 1) / kAdamThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(flat_accumulate_tables_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0,
                     static_cast<hipStream_t>(stream), flat, grad_ptrs, seg_off, nseg, ntab, n4, n, accumulate);
  return launch_status("fpsg_flat_accumulate_tables");
}
