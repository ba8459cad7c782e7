// decoder.hip -- K9: the first layer of the decoder's patch MLPs in one pass, for gfx950.
//
// Reference (src/models/point_cloud_net.py:97-112 + :57-80): every one of the 16 PrimitiveNodes gets
// cat(x.repeat(1,1,128), patch_points) [B,1539,128] and runs conv1 (1539 -> 1539, 1x1) + BatchNorm1d +
// ReLU on it.  The 1536 latent channels are constant over a patch's points, so
//     conv1(cat(x_rep, p))[d, b, q] = (W[d,:1536] x_b + bias_d)  +  W[d,1536:] p[b,q]
//                                   =        hlat[d, b]           +  w0 px + w1 py + w2 pz .
// The host computes hlat with ONE [D x L] x [L x B] GEMM per patch (library, MFMA); the rest of the
// layer is memory-bound and lives here.  As separate library passes it is a K = 3 batched GEMM that
// writes [G,D,B*P] (403 MB at 32 clouds), a broadcast add (read + write), BatchNorm (2 reads + 1 write)
// and, in the backward, ~8 more passes.  K9 never materialises the pre-BatchNorm tensor:
//
//   dec1_fwd_kernel: a workgroup owns 32 channels of one patch; the patch's deformed points pts[g]
//     ([3, B*P], 48 KB at 32 clouds) sit in LDS; a wave takes 4 channels, one at a time: sweep 1
//     evaluates h for the channel's B*P elements and sums h, h^2 (fp32 per lane, float64 across the
//     wave), sweep 2 re-evaluates h and writes relu(h * scale + shift) as aligned 16-byte stores.
//     HBM traffic: the output, once.
//   dec1_bwd_kernel: same ownership.  Sweep 1 re-derives h, the ReLU mask and the normalised value from
//     the same arithmetic and reduces sum(dz), sum(dz * hhat) (= dbeta, dgamma); sweep 2 forms
//     dh = scale (dz - mean(dz) - hhat mean(dz hhat)) in registers and reduces it three ways without
//     storing it: over a patch's points (-> dhlat[d,b], the gradient of the latent GEMM's output), against
//     the points (-> dW[d,1536:]), and over the workgroup's channels (-> a partial of dpts, summed over
//     the 49 channel tiles by the caller).  HBM traffic: the upstream gradient, read once (second sweep
//     from L2 / Infinity Cache).
// Deterministic: every sum has a fixed order (lane-strided partials, DPP trees, fixed wave order).
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kD1Waves = 8;
constexpr int kD1Threads = 64 * kD1Waves;
constexpr int kD1ChPerWave = 4;
constexpr int kD1Tile = kD1Waves * kD1ChPerWave;       // channels per workgroup
constexpr int kD1MaxBP = 8192;                         // points of a patch over the batch that fit LDS (96 KB)

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    lo = (unsigned)__shfl_xor((int)lo, m, 64);
    hi = (unsigned)__shfl_xor((int)hi, m, 64);
    v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  return v;
}

// sum over groups of `grp` consecutive lanes (grp a power of two <= 64); every lane of a group gets it
__device__ __forceinline__ float group_sum(float v, int grp) {
  if (grp > 1) v += __uint_as_float(lane_xor<1>(__float_as_uint(v)));
  if (grp > 2) v += __uint_as_float(lane_xor<2>(__float_as_uint(v)));
  if (grp > 4) v += __uint_as_float(lane_xor<4>(__float_as_uint(v)));
  if (grp > 8) v += __uint_as_float(lane_xor<8>(__float_as_uint(v)));
  if (grp > 16) v += __uint_as_float(lane_xor<16>(__float_as_uint(v)));
  if (grp > 32) v += __uint_as_float(lane_xor<32>(__float_as_uint(v)));
  return v;
}

struct Dec1Args {
  const float* hlat;    // [G, D, B]   W[:, :L] x + bias
  const float* w;       // [G, D, ldw] the stacked first-layer weight; columns wofs .. wofs+2 multiply the points
  const float* pts;     // [G, 3, B*P] deformed patch points
  const float* gamma;   // [G*D]
  const float* beta;    // [G*D]
  const float* rmean;   // [G*D] running statistics (eval mode) or null
  const float* rvar;
  int G, D, B, P, ldw, wofs, training;
  float eps;
  int ld_pts;           // distance between the rows of pts, in elements (B*P when the call owns the whole rows)
  int ld_out;           // the same for out (forward) / dout (backward)
  int accumulate;       // backward: dw's point columns, dgamma and dbeta are added to what the buffers hold
  int ld_hlat;          // distance between the rows of hlat / dhlat (B when the call owns them)
};

// LDS: pts [3][BP] | hl [kD1Tile][B]
template <bool NT>
__global__ __launch_bounds__(kD1Threads) void dec1_fwd_kernel(Dec1Args a, float* __restrict__ out,
                                                             float* __restrict__ chan, float* __restrict__ bmean,
                                                             float* __restrict__ bvar) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int g = blockIdx.y;
  const int d0 = blockIdx.x * kD1Tile;
  const int BP = a.B * a.P;
  float* sp = lds;                       // [3][BP]
  float* hl = lds + 3 * BP;              // [kD1Tile][B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    v4f* dst = reinterpret_cast<v4f*>(sp);
    for (int r = 0; r < 3; ++r) {
      const v4f* src = reinterpret_cast<const v4f*>(a.pts + ((size_t)g * 3 + r) * a.ld_pts);
      for (int e = tid; e < BP / 4; e += kD1Threads) dst[r * (BP / 4) + e] = src[e];
    }
    for (int e = tid; e < kD1Tile * a.B; e += kD1Threads) {
      const int c = e / a.B, b = e - c * a.B;
      hl[e] = (d0 + c < a.D) ? a.hlat[((size_t)g * a.D + d0 + c) * a.ld_hlat + b] : 0.0f;
    }
  }
  __syncthreads();
  const int pshift4 = a.P / 4;           // lanes (of 4 elements) per cloud
  for (int cc = 0; cc < kD1ChPerWave; ++cc) {
    const int c = wave * kD1ChPerWave + cc;
    const int d = d0 + c;
    if (d >= a.D) break;                 // wave-uniform
    const size_t gd = (size_t)g * a.D + d;
    const float* wr = a.w + gd * a.ldw + a.wofs;
    const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
    const float* hc = hl + c * a.B;
    float mean, rstd;
    if (a.training) {
      // The sums run over h - K, K = the mean of the latent part over the clouds: the variance does not depend on K, and
      // E[h^2] - E[h]^2 on the raw values cancels when h is a constant plus a small variation -- with ONE cloud (1-shot
      // episodes, BASELINE configs[1]) h = hlat[d] + w.p with |hlat| ~ 60 sigma: 2e-4 relative in the variance, 3.5e-4
      // in the episode's loss against the reference arithmetic (profiles/r04/episode_parity_deviation.jsonl).
      float ksum = 0.0f;
      for (int b = lane; b < a.B; b += 64) ksum += hc[b];
      const float K = wave_sum(ksum) / (float)a.B;
      float s = 0.0f, ss = 0.0f;
      for (int e4 = lane; e4 < BP / 4; e4 += 64) {
        const v4f X = reinterpret_cast<const v4f*>(sp)[e4], Y = reinterpret_cast<const v4f*>(sp + BP)[e4],
                  Z = reinterpret_cast<const v4f*>(sp + 2 * BP)[e4];
        const float hb = hc[e4 / pshift4] - K;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float h = fma_rn(w2, Z[u], fma_rn(w1, Y[u], w0 * X[u])) + hb;
          s += h;
          ss = fma_rn(h, h, ss);
        }
      }
      const double S = wave_sum_f64((double)s), SS = wave_sum_f64((double)ss);
      const double mk = S / BP;
      double var = SS / BP - mk * mk;
      var = var < 0.0 ? 0.0 : var;
      const double mu = (double)K + mk;
      mean = (float)mu;
      rstd = (float)(1.0 / sqrt(var + (double)a.eps));
      if (lane == 0) {
        if (bmean) bmean[gd] = mean;
        if (bvar) bvar[gd] = (float)(var * ((double)BP / (double)(BP > 1 ? BP - 1 : 1)));
      }
    } else {
      mean = a.rmean[gd];
      rstd = 1.0f / __builtin_sqrtf(a.rvar[gd] + a.eps);
    }
    const float scale = a.gamma[gd] * rstd;
    const float shift = a.beta[gd] - mean * scale;
    if (lane == 0) {
      const size_t C = (size_t)a.G * a.D;
      chan[gd] = scale; chan[C + gd] = shift; chan[2 * C + gd] = mean; chan[3 * C + gd] = rstd;
    }
    v4f* orow = reinterpret_cast<v4f*>(out + gd * a.ld_out);
    for (int e4 = lane; e4 < BP / 4; e4 += 64) {
      const v4f X = reinterpret_cast<const v4f*>(sp)[e4], Y = reinterpret_cast<const v4f*>(sp + BP)[e4],
                Z = reinterpret_cast<const v4f*>(sp + 2 * BP)[e4];
      const float hb = hc[e4 / pshift4];
      v4f o;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float h = fma_rn(w2, Z[u], fma_rn(w1, Y[u], w0 * X[u])) + hb;
        o[u] = __builtin_fmaxf(fma_rn(h, scale, shift), 0.0f);
      }
      st_stream<NT>(orow + e4, o);
    }
  }
}

// LDS: pts [3][BP] | hl [kD1Tile][B] | red [kD1Waves][64][12]
template <bool NT>
__global__ __launch_bounds__(kD1Threads) void dec1_bwd_kernel(Dec1Args a, const float* __restrict__ dout,
                                                             const float* __restrict__ chan,
                                                             float* __restrict__ dhlat, float* __restrict__ dw,
                                                             float* __restrict__ dpts_part,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int g = blockIdx.y;
  const int tile = blockIdx.x;
  const int d0 = tile * kD1Tile;
  const int BP = a.B * a.P;
  float* sp = lds;
  float* hl = lds + 3 * BP;
  float* red = hl + kD1Tile * a.B;       // [kD1Waves][64][12]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    v4f* dst = reinterpret_cast<v4f*>(sp);
    for (int r = 0; r < 3; ++r) {
      const v4f* src = reinterpret_cast<const v4f*>(a.pts + ((size_t)g * 3 + r) * a.ld_pts);
      for (int e = tid; e < BP / 4; e += kD1Threads) dst[r * (BP / 4) + e] = src[e];
    }
    for (int e = tid; e < kD1Tile * a.B; e += kD1Threads) {
      const int c = e / a.B, b = e - c * a.B;
      hl[e] = (d0 + c < a.D) ? a.hlat[((size_t)g * a.D + d0 + c) * a.ld_hlat + b] : 0.0f;
    }
  }
  __syncthreads();
  const size_t C = (size_t)a.G * a.D;
  const int pshift4 = a.P / 4;
  // per-channel constants of this wave's channels
  float w0[kD1ChPerWave], w1[kD1ChPerWave], w2[kD1ChPerWave], scale[kD1ChPerWave], shift[kD1ChPerWave],
      mean[kD1ChPerWave], rstd[kD1ChPerWave], c1[kD1ChPerWave], c2[kD1ChPerWave];
  bool live[kD1ChPerWave];
#pragma unroll
  for (int cc = 0; cc < kD1ChPerWave; ++cc) {
    const int d = d0 + wave * kD1ChPerWave + cc;
    live[cc] = d < a.D;
    const size_t gd = (size_t)g * a.D + (live[cc] ? d : a.D - 1);
    const float* wr = a.w + gd * a.ldw + a.wofs;
    w0[cc] = wr[0]; w1[cc] = wr[1]; w2[cc] = wr[2];
    scale[cc] = chan[gd]; shift[cc] = chan[C + gd]; mean[cc] = chan[2 * C + gd]; rstd[cc] = chan[3 * C + gd];
    c1[cc] = 0.0f; c2[cc] = 0.0f;
  }
  // ---- sweep 1: dbeta = sum dz, dgamma = sum dz * hhat (dz = dout where the activation was positive)
  // The wave's four channels advance together, two element blocks per trip: eight 16-byte loads of dout in flight per
  // lane (one channel at a time with one load per trip, a wave walked 4 x BP/256 dependent round trips to HBM with
  // 1 KB in flight).  Per channel and lane the sums run over the same elements in the same order as before.
  {
    const v4f* drow[kD1ChPerWave];
    const float* hc[kD1ChPerWave];
    float s[kD1ChPerWave], sh[kD1ChPerWave];
#pragma unroll
    for (int cc = 0; cc < kD1ChPerWave; ++cc) {
      const size_t gd = (size_t)g * a.D + (live[cc] ? d0 + wave * kD1ChPerWave + cc : a.D - 1);
      drow[cc] = reinterpret_cast<const v4f*>(dout + gd * a.ld_out);
      hc[cc] = hl + (wave * kD1ChPerWave + cc) * a.B;
      s[cc] = 0.0f; sh[cc] = 0.0f;
    }
    const int nv = BP / 4;
    for (int e4a = lane; e4a < nv; e4a += 128) {
      const int e4b = e4a + 64;
      const bool inb = e4b < nv;
      const int e4bc = inb ? e4b : e4a;
      v4f dya[kD1ChPerWave], dyb[kD1ChPerWave];
#pragma unroll
      for (int cc = 0; cc < kD1ChPerWave; ++cc) {
        dya[cc] = drow[cc][e4a];
        dyb[cc] = drow[cc][e4bc];
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && !inb) break;
        const int e4 = half ? e4b : e4a;
        const v4f X = reinterpret_cast<const v4f*>(sp)[e4], Y = reinterpret_cast<const v4f*>(sp + BP)[e4],
                  Z = reinterpret_cast<const v4f*>(sp + 2 * BP)[e4];
#pragma unroll
        for (int cc = 0; cc < kD1ChPerWave; ++cc) {
          if (!live[cc]) continue;           // wave-uniform
          const v4f dy = half ? dyb[cc] : dya[cc];
          const float hb = hc[cc][e4 / pshift4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float h = fma_rn(w2[cc], Z[u], fma_rn(w1[cc], Y[u], w0[cc] * X[u])) + hb;
            const float dz = fma_rn(h, scale[cc], shift[cc]) > 0.0f ? dy[u] : 0.0f;
            s[cc] += dz;
            sh[cc] = fma_rn(dz, (h - mean[cc]) * rstd[cc], sh[cc]);
          }
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < kD1ChPerWave; ++cc) {
      if (!live[cc]) continue;
      const size_t gd = (size_t)g * a.D + d0 + wave * kD1ChPerWave + cc;
      const double S = wave_sum_f64((double)s[cc]), SH = wave_sum_f64((double)sh[cc]);
      if (lane == 0) {
        dbeta[gd] = a.accumulate ? dbeta[gd] + (float)S : (float)S;
        dgamma[gd] = a.accumulate ? dgamma[gd] + (float)SH : (float)SH;
      }
      if (a.training) { c1[cc] = (float)(S / BP); c2[cc] = (float)(SH / BP); }
    }
  }
  // ---- sweep 2: dh in registers; three reductions
  float aw0[kD1ChPerWave], aw1[kD1ChPerWave], aw2[kD1ChPerWave];
#pragma unroll
  for (int cc = 0; cc < kD1ChPerWave; ++cc) { aw0[cc] = 0.0f; aw1[cc] = 0.0f; aw2[cc] = 0.0f; }
  float* mine = red + (wave * 64 + lane) * 12;
  float* pp = dpts_part + ((size_t)g * gridDim.x + tile) * 3 * BP;
  // the upstream gradient of the NEXT element block is requested before this block's arithmetic and barriers
  v4f dyn[kD1ChPerWave];
  const v4f* drow2[kD1ChPerWave];
#pragma unroll
  for (int cc = 0; cc < kD1ChPerWave; ++cc) {
    const size_t gd = (size_t)g * a.D + (live[cc] ? d0 + wave * kD1ChPerWave + cc : a.D - 1);
    drow2[cc] = reinterpret_cast<const v4f*>(dout + gd * a.ld_out);
    dyn[cc] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    if (lane < BP / 4) dyn[cc] = ld_stream<NT>(drow2[cc] + lane);
  }
  for (int e4b = 0; e4b < BP / 4; e4b += 64) {          // all waves walk the element blocks together
    const int e4 = e4b + lane;
    const bool in = e4 < BP / 4;
    const int e4c = in ? e4 : 0;
    v4f dyc[kD1ChPerWave];
#pragma unroll
    for (int cc = 0; cc < kD1ChPerWave; ++cc) dyc[cc] = dyn[cc];
    if (e4 + 64 < BP / 4) {
#pragma unroll
      for (int cc = 0; cc < kD1ChPerWave; ++cc) dyn[cc] = ld_stream<NT>(drow2[cc] + e4 + 64);
    }
    const v4f X = reinterpret_cast<const v4f*>(sp)[e4c], Y = reinterpret_cast<const v4f*>(sp + BP)[e4c],
              Z = reinterpret_cast<const v4f*>(sp + 2 * BP)[e4c];
    const int b = e4c / pshift4;
    float ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0}, az[4] = {0, 0, 0, 0};
#pragma unroll
    for (int cc = 0; cc < kD1ChPerWave; ++cc) {
      if (!live[cc]) continue;
      const size_t gd = (size_t)g * a.D + d0 + wave * kD1ChPerWave + cc;
      const float hb = hl[(wave * kD1ChPerWave + cc) * a.B + b];
      v4f dy = {0, 0, 0, 0};
      if (in) dy = dyc[cc];
      float part = 0.0f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float h = fma_rn(w2[cc], Z[u], fma_rn(w1[cc], Y[u], w0[cc] * X[u])) + hb;
        const float dz = fma_rn(h, scale[cc], shift[cc]) > 0.0f ? dy[u] : 0.0f;
        const float hh = (h - mean[cc]) * rstd[cc];
        float dh = scale[cc] * (dz - c1[cc] - hh * c2[cc]);
        dh = in ? dh : 0.0f;
        part += dh;
        aw0[cc] = fma_rn(dh, X[u], aw0[cc]);
        aw1[cc] = fma_rn(dh, Y[u], aw1[cc]);
        aw2[cc] = fma_rn(dh, Z[u], aw2[cc]);
        ax[u] = fma_rn(w0[cc], dh, ax[u]);
        ay[u] = fma_rn(w1[cc], dh, ay[u]);
        az[u] = fma_rn(w2[cc], dh, az[u]);
      }
      // gradient of hlat[d, b]: sum over the P points of cloud b = P/4 consecutive lanes of this block
      const float tot = group_sum(part, pshift4);
      if (in && (lane & (pshift4 - 1)) == 0) dhlat[gd * a.ld_hlat + b] = tot;
    }
    // dpts of this element block: sum over the workgroup's channels, fixed wave order
#pragma unroll
    for (int u = 0; u < 4; ++u) { mine[u] = ax[u]; mine[4 + u] = ay[u]; mine[8 + u] = az[u]; }
    __syncthreads();
    for (int t = tid; t < 64 * 12; t += kD1Threads) {
      const int l = t / 12, q = t - l * 12;              // q = coordinate * 4 + element
      float sum = 0.0f;
#pragma unroll
      for (int wv = 0; wv < kD1Waves; ++wv) sum += red[(wv * 64 + l) * 12 + q];
      const int e = 4 * (e4b + l) + (q & 3);
      if (e < BP) pp[(q >> 2) * BP + e] = sum;
    }
    __syncthreads();
  }
#pragma unroll
  for (int cc = 0; cc < kD1ChPerWave; ++cc) {
    if (!live[cc]) continue;
    const size_t gd = (size_t)g * a.D + d0 + wave * kD1ChPerWave + cc;
    const float t0 = wave_sum(aw0[cc]), t1 = wave_sum(aw1[cc]), t2 = wave_sum(aw2[cc]);
    if (lane == 0) {
      float* dst = dw + gd * a.ldw + a.wofs;
      if (a.accumulate) { dst[0] += t0; dst[1] += t1; dst[2] += t2; }
      else { dst[0] = t0; dst[1] = t1; dst[2] = t2; }
    }
  }
}

inline size_t dec1_lds_bytes(int B, int P, bool bwd) {
  return ((size_t)3 * B * P + (size_t)kD1Tile * B + (bwd ? (size_t)kD1Waves * 64 * 12 : 0)) * sizeof(float);
}

inline int dec1_check(const char* fn, int G, int D, int B, int P, int ldw, int wofs) {
  FPSG_REQUIRE(G > 0 && D > 0 && B > 0 && P > 0, FPSG_E_SHAPE, "%s: G,D,B,P must be positive (got %d,%d,%d,%d)", fn, G,
               D, B, P);
  FPSG_REQUIRE(P % 4 == 0 && P <= 256 && ((P / 4) & (P / 4 - 1)) == 0, FPSG_E_SHAPE,
               "%s: P = %d must be 4 x a power of two, at most 256 (lanes of 4 points, a cloud = one lane group)", fn, P);
  FPSG_REQUIRE((long)B * P <= kD1MaxBP, FPSG_E_LIMIT, "%s: B*P = %ld exceeds %d (the patch's points live in LDS)", fn,
               (long)B * P, kD1MaxBP);
  FPSG_REQUIRE(G <= 65535 && wofs >= 0 && wofs + 3 <= ldw, FPSG_E_SHAPE, "%s: bad G / weight columns", fn);
  return 0;
}

}  // namespace
}  // namespace fpsg

extern "C" int fpsg_dec1_tiles(int D) { return (D + fpsg::kD1Tile - 1) / fpsg::kD1Tile; }

extern "C" int fpsg_dec1_fwd_ld(const float* hlat, int ld_hlat, const float* w, int ldw, int wofs, const float* pts, int ld_pts,
                                const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, int G, int D, int B, int P, int training, float eps,
                                float* out, int ld_out, float* chan, float* batch_mean, float* batch_var_unbiased,
                                fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = dec1_check("fpsg_dec1_fwd", G, D, B, P, ldw, wofs);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(hlat); FPSG_REQUIRE_PTR(w); FPSG_REQUIRE_PTR(pts); FPSG_REQUIRE_PTR(gamma); FPSG_REQUIRE_PTR(beta);
  FPSG_REQUIRE_PTR(out); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE(training || (running_mean && running_var), FPSG_E_NULL,
               "fpsg_dec1_fwd: eval mode needs the running statistics");
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(pts) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_dec1_fwd: pts / out must be 16-byte aligned");
  FPSG_REQUIRE(ld_pts >= B * P && ld_out >= B * P && (ld_pts & 3) == 0 && (ld_out & 3) == 0 && ld_hlat >= B, FPSG_E_SHAPE,
               "fpsg_dec1_fwd: row strides %d / %d must be multiples of 4, at least B*P = %d (hlat: %d >= %d)", ld_pts,
               ld_out, B * P, ld_hlat, B);
  Dec1Args a{hlat, w, pts, gamma, beta, running_mean, running_var, G, D, B, P, ldw, wofs, training, eps, ld_pts, ld_out, 0,
             ld_hlat};
  const size_t lds_bytes = dec1_lds_bytes(B, P, false);
  // the output (403 MB at 32 clouds) is written once: past the Infinity Cache when it exceeds it
  auto kern = beyond_cache((size_t)G * D * B * P * sizeof(float)) ? dec1_fwd_kernel<true> : dec1_fwd_kernel<false>;
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("fpsg_dec1_fwd: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)fpsg_dec1_tiles(D), (unsigned)G), dim3(kD1Threads), lds_bytes,
                     static_cast<hipStream_t>(stream), a, out, chan, batch_mean, batch_var_unbiased);
  return launch_status("fpsg_dec1_fwd");
}

extern "C" int fpsg_dec1_fwd(const float* hlat, const float* w, int ldw, int wofs, const float* pts,
                             const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, int G, int D, int B, int P, int training, float eps,
                             float* out, float* chan, float* batch_mean, float* batch_var_unbiased,
                             fpsg_stream_t stream) {
  return fpsg_dec1_fwd_ld(hlat, B, w, ldw, wofs, pts, B * P, gamma, beta, running_mean, running_var, G, D, B, P, training,
                          eps, out, B * P, chan, batch_mean, batch_var_unbiased, stream);
}

extern "C" int fpsg_dec1_bwd_ld(const float* dout, int ld_dout, const float* hlat, int ld_hlat, const float* w, int ldw, int wofs,
                                const float* pts, int ld_pts, const float* chan, int G, int D, int B, int P,
                                int training, int accumulate, float* dhlat, float* dw, float* dpts_part, float* dgamma,
                                float* dbeta, fpsg_stream_t stream) {
  using namespace fpsg;
  int rc = dec1_check("fpsg_dec1_bwd", G, D, B, P, ldw, wofs);
  if (rc) return rc;
  FPSG_REQUIRE_PTR(dout); FPSG_REQUIRE_PTR(hlat); FPSG_REQUIRE_PTR(w); FPSG_REQUIRE_PTR(pts); FPSG_REQUIRE_PTR(chan);
  FPSG_REQUIRE_PTR(dhlat); FPSG_REQUIRE_PTR(dw); FPSG_REQUIRE_PTR(dpts_part); FPSG_REQUIRE_PTR(dgamma); FPSG_REQUIRE_PTR(dbeta);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(pts) & 15) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_dec1_bwd: pts / dout must be 16-byte aligned");
  FPSG_REQUIRE(ld_pts >= B * P && ld_dout >= B * P && (ld_pts & 3) == 0 && (ld_dout & 3) == 0 && ld_hlat >= B, FPSG_E_SHAPE,
               "fpsg_dec1_bwd: row strides %d / %d must be multiples of 4, at least B*P = %d (hlat: %d >= %d)", ld_pts,
               ld_dout, B * P, ld_hlat, B);
  Dec1Args a{hlat, w, pts, nullptr, nullptr, nullptr, nullptr, G, D, B, P, ldw, wofs, training, 0.0f, ld_pts, ld_dout,
             accumulate ? 1 : 0, ld_hlat};
  const size_t lds_bytes = dec1_lds_bytes(B, P, true);
  // the output gradient's second (last) read of a row
  auto kern = beyond_cache((size_t)G * D * B * P * sizeof(float)) ? dec1_bwd_kernel<true> : dec1_bwd_kernel<false>;
  if (lds_bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("fpsg_dec1_bwd: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)fpsg_dec1_tiles(D), (unsigned)G), dim3(kD1Threads), lds_bytes,
                     static_cast<hipStream_t>(stream), a, dout, chan, dhlat, dw, dpts_part, dgamma, dbeta);
  return launch_status("fpsg_dec1_bwd");
}

extern "C" int fpsg_dec1_bwd(const float* dout, const float* hlat, const float* w, int ldw, int wofs,
                             const float* pts, const float* chan, int G, int D, int B, int P, int training,
                             float* dhlat, float* dw, float* dpts_part, float* dgamma, float* dbeta,
                             fpsg_stream_t stream) {
  return fpsg_dec1_bwd_ld(dout, B * P, hlat, B, w, ldw, wofs, pts, B * P, chan, G, D, B, P, training, 0, dhlat, dw,
                          dpts_part, dgamma, dbeta, stream);
}
