// knn.hip -- K3: k-nearest-neighbour graph of DGCNN's EdgeConv, for gfx950.
// Replaces `knn` of reference src/dgcnn/model.py:13-20 (torch.matmul of x^T x into a
// [B,N,N] tensor + torch.topk): here the N x N matrix never reaches HBM.
//
// One workgroup (16 waves for C <= 64, else 8) owns 16 query points of one cloud:
//   phase A  the 16 x N block of  pd_ij = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  is produced with
//            fp32-input MFMA (v_mfma_f32_16x16x4_f32: exact k-ordered fma chain, same
//            64 FLOP/clk/SIMD as the VALU).  A = the 16 queries, held in registers for the
//            whole workgroup lifetime; B = 16 candidates per tile, 64-B segments of the
//            channel-major rows served from L2, prefetched TWO tiles ahead into a rotating
//            set of three register buffers so that the ~1 us L2 latency hides under the MFMA
//            chains of the tiles in flight; the accumulator is carried over C/4 steps.
//            Scores go to a 16 x min(N,2048) fp32 tile in LDS (128 KiB), never to HBM; longer
//            clouds are processed in 2048-column chunks whose top-k lists are merged.
//   phase B  each wave selects the k largest of its 1-2 rows.  A lane
//            holds N/64 scores of each row in registers, in groups of 8 with cached group
//            maxima.  A round = best of the lane's group maxima, wave-wide argmax with DPP
//            row operations (value max, then lowest index among the lanes that hold it: ties
//            go to the lower index), then only the winner's group -- wave-uniform, so a
//            scalar branch -- is rescanned.  No LDS traffic inside the rounds.
// Results are bit-identical to oracle_knn (same fma chains, same tie rule).
#include "knn_internal.h"

namespace fpsg {
namespace {

constexpr int kQ = 16;            // query rows per workgroup (= MFMA M)
// waves per workgroup NW (template parameter): 16 (4 per SIMD, 128 VGPRs) when the operands
// fit, else 8; the selection rounds are latency-bound, so more resident waves = more rows in flight
constexpr int kGroup = 8;         // scores per cached-maximum group in phase B

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

// Squared norms AND a point-major, k-interleaved copy of the features for the MFMA B operands:
//   xk[b][n][kk * (C/4) + c4] = x[b][4*c4 + kk][n]      (C a multiple of 16)
// so that lane group kk of the 16x16x4 MFMA fetches its operands of FOUR consecutive channel steps
// (channels 4*c4 + kk, c4 = 4m .. 4m+3) with one 16-byte load.  The channel-major rows give a lane 4 bytes
// per load (16 loads per 16-candidate tile at C = 64): phase A was load-issue bound at 3x its MFMA time.
// One workgroup = 64 points: coalesced reads of the [C][64] slab, transposed through LDS, 16-byte stores.
constexpr int kPrepPts = 64;
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ x, int C, int N,
                                                       float* __restrict__ xx, float* __restrict__ xk) {
  extern __shared__ __attribute__((aligned(16))) float tile[];      // [64][C + 4]
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * kPrepPts;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ld = C + 4;
  const int C4 = C >> 2;
  const float* xb = x + (size_t)b * C * N;
  const int n = n0 + lane;
  for (int c = wave; c < C; c += 4) {
    const float v = n < N ? xb[(size_t)c * N + n] : 0.0f;
    tile[lane * ld + (c & 3) * C4 + (c >> 2)] = v;
  }
  __syncthreads();
  if (wave == 0 && n < N) {                 // squared norm in channel order (the oracle's fma chain)
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float v = tile[lane * ld + (c & 3) * C4 + (c >> 2)];
      acc = fma_rn(v, v, acc);
    }
    xx[(size_t)b * N + n] = acc;
  }
  float* dst = xk + ((size_t)b * N + n0) * C;
  for (int e = threadIdx.x; e < kPrepPts * C4; e += 256) {
    const int p = e / C4, q = e - p * C4;
    if (n0 + p < N) *reinterpret_cast<v4f*>(dst + (size_t)p * C + 4 * q) = *reinterpret_cast<const v4f*>(tile + p * ld + 4 * q);
  }
}

__device__ __forceinline__ unsigned orderable(float f) { return knn_orderable(f); }

// ---- wave-wide reductions with DPP row operations (gfx9 encodings) ----------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int self) {
  return __builtin_amdgcn_update_dpp(self, self, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_max_f32(float x) {
#define FPSG_STEP(CTRL, RM) x = __builtin_fmaxf(x, __int_as_float(dpp_i<CTRL, RM>(__float_as_int(x))))
  FPSG_STEP(0xB1, 0xF);    // quad_perm [1,0,3,2]
  FPSG_STEP(0x4E, 0xF);    // quad_perm [2,3,0,1]
  FPSG_STEP(0x141, 0xF);   // row_half_mirror
  FPSG_STEP(0x140, 0xF);   // row_mirror          -> every row of 16 lanes is uniform
  FPSG_STEP(0x142, 0xA);   // row_bcast15 into rows 1,3
  FPSG_STEP(0x143, 0xC);   // row_bcast31 into rows 2,3 -> lane 63 holds the wave maximum
#undef FPSG_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
#define FPSG_STEP(CTRL, RM) { const unsigned o = (unsigned)dpp_i<CTRL, RM>((int)x); x = o < x ? o : x; }
  FPSG_STEP(0xB1, 0xF);
  FPSG_STEP(0x4E, 0xF);
  FPSG_STEP(0x141, 0xF);
  FPSG_STEP(0x140, 0xF);
  FPSG_STEP(0x142, 0xA);
  FPSG_STEP(0x143, 0xC);
#undef FPSG_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// ---- phase A helpers --------------------------------------------------------------------
template <int C4T>
__device__ __forceinline__ void load_b(const float* __restrict__ xb, int C, int N, int tile, int kk,
                                       int col, float (&b)[C4T]) {
  const int j = tile * 16 + col;
  const bool jin = j < N;   // also false for tiles past the end: no loads are issued
#pragma unroll
  for (int c4 = 0; c4 < C4T; ++c4) {
    const int c = 4 * c4 + kk;
    b[c4] = (jin && c < C) ? xb[(size_t)c * N + j] : 0.0f;
  }
}

// the same operands from the k-interleaved point-major copy: C4T/4 16-byte loads
template <int C4T>
__device__ __forceinline__ void load_b_pm(const float* __restrict__ xkb, int C, int N, int tile, int kk,
                                          int col, float (&b)[C4T]) {
  const int j = tile * 16 + col;
  const bool jin = j < N;   // also false for tiles past the end: no loads are issued
  const v4f* src = reinterpret_cast<const v4f*>(xkb + (size_t)(jin ? j : 0) * C + kk * (C >> 2));
#pragma unroll
  for (int m = 0; m < C4T / 4; ++m) {
    v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (jin && 16 * m < C) v = src[m];
    b[4 * m] = v.x; b[4 * m + 1] = v.y; b[4 * m + 2] = v.z; b[4 * m + 3] = v.w;
  }
}

template <int C4T>
__device__ __forceinline__ void score_tile(const float (&a)[C4T], const float (&b)[C4T], int tile,
                                           int N, int kk, int col, const float* __restrict__ xxb,
                                           const float (&xxq)[4], float* __restrict__ pd, int ldp,
                                           int c0) {
  v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int c4 = 0; c4 < C4T; ++c4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c4], b[c4], acc, 0, 0, 0);
  const int j = tile * 16 + col;
  const bool jin = j < N;
  const float xxj = jin ? xxb[j] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // D layout: column = lane&15 (candidate), row = 4*(lane>>4) + r (query)
    const float v = fma_rn(2.0f, acc[r], -xxj) - xxq[r];
    pd[(4 * kk + r) * ldp + (j - c0)] = jin ? v : -__builtin_inff();
  }
}

// ---- phase B, threshold form --------------------------------------------------------------
// The k best of a row's 64*VPL scores without k rounds of wave-wide argmax.  A threshold T0 that k lane maxima reach
// is a lower bound of the k-th best score, so the k best are among the scores >= T0 -- ~30 of 2048 for k = 20 on
// ordinary data.  They are compacted (64-bit keys: orderable score << 32 | ~index,
// the order of the round form: score descending, then index ascending; -0 counts as +0) into the wave's LDS strip,
// ranked by counting, and the first k land in lanes 0..k-1.  Returns false (nothing written) when more than 64
// scores pass -- many equal scores, or fewer than k lanes with a finite maximum: the caller then runs the rounds.
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long x, int l) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(x >> 32), l) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)x, l);
}

// number of lanes of this lane's row of 16 whose key is larger (keys distinct inside a row): 15 DPP row rotations
template <int N>
__device__ __forceinline__ int row_rank_step(unsigned key, int rank) {
  if constexpr (N < 16) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x120 + N, 0xF, 0xF, false);   // row_ror:N
    return row_rank_step<N + 1>(key, rank + (o > key ? 1 : 0));
  } else {
    return rank;
  }
}

template <int VPL>
__device__ __forceinline__ bool select_by_threshold(const float (&v)[VPL], float lane_max, int k, int c0, int lane,
                                                    unsigned long long* __restrict__ strip, int& mine, float& mval) {
  // T0: in each of the four rows of 16 lanes the ceil(k/4)-th largest lane maximum, then the smallest of the four:
  // k lanes (at least) hold a score >= T0.  Ranks inside a row come from integer keys made distinct by the lane
  // number in the low 4 bits; a near-tie may then pick a neighbouring rank -- T0 is still one of the lane maxima, and
  // the count checks below send the row to the rounds when fewer than k or more than 64 scores pass.
  const int need = (k + 3) >> 2;
  const unsigned rk = (orderable(lane_max + 0.0f) & ~15u) | (unsigned)(15 - (lane & 15));
  const int rank = row_rank_step<1>(rk, 0);
  const unsigned long long pick = __ballot(rank == need - 1);         // one lane per row (need <= 16)
  float T0 = __builtin_inff();
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const unsigned bits = (unsigned)(pick >> (16 * row)) & 0xffffu;
    const int src = 16 * row + (bits ? __builtin_ctz(bits) : 0);
    T0 = __builtin_fminf(T0, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_max), src)));
  }
  // compaction: per score slot one ballot; empty slots cost a compare and a scalar branch
  int total = 0;                                                      // wave-uniform
#pragma unroll
  for (int t = 0; t < VPL; ++t) {
    const bool in = v[t] >= T0;
    const unsigned long long m = __ballot(in);
    if (m != 0ull) {
      const int n = __builtin_popcountll(m);
      if (total + n > 64) return false;
      if (in) {
        const int pos = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        strip[pos] = ((unsigned long long)orderable(v[t] + 0.0f) << 32) | (unsigned)~(unsigned)(c0 + t * 64 + lane);
      }
      total += n;
    }
  }
  if (total < k) return false;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long key = lane < total ? strip[lane] : 0ull;     // one wave's LDS operations execute in order
  int r = 0;
  for (int l = 0; l < total; ++l) r += readlane_u64(key, l) > key ? 1 : 0;
  __builtin_amdgcn_wave_barrier();
  if (lane < total && r < k) strip[r] = key;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long mk = lane < k ? strip[lane] : 0ull;
  __builtin_amdgcn_wave_barrier();
  const unsigned ov = (unsigned)(mk >> 32);
  mval = lane < k ? __uint_as_float((ov & 0x80000000u) ? (ov & 0x7fffffffu) : ~ov) : -__builtin_inff();
  mine = lane < k ? (int)~(unsigned)mk : 0;
  return true;
}

// LDS: pd[kQ][ldp] floats (+ qa[C4*4][16] floats for the generic-C path).
// C4T > 0: compile-time channel steps (C <= 4*C4T), register-resident operands.
// C4T == 0: any C, operands re-read per step (slow path for unusual channel counts).
template <int VPL, int C4T, int NW, bool PM>
__global__ __launch_bounds__(64 * NW) void knn_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ xx,
                                                          const float* __restrict__ xk, int C,
                                                          int N, int k, int ldp,
                                                          int32_t* __restrict__ idx) {
  constexpr int kKnnWaves = NW;
  constexpr int kKnnThreads = 64 * NW;
  constexpr int kRowsPerWave = kQ / NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* pd = lds;
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * kQ;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* __restrict__ xb = x + (size_t)b * C * N;
  const float* __restrict__ xxb = xx + (size_t)b * N;
  const float* __restrict__ xkb = PM ? xk + (size_t)b * N * C : nullptr;
  auto ldb = [&](int tile, auto& dst) {
    if constexpr (PM) load_b_pm<(C4T > 0 ? C4T : 4)>(xkb, C, N, tile, lane >> 4, lane & 15, dst);
    else load_b<(C4T > 0 ? C4T : 1)>(xb, C, N, tile, lane >> 4, lane & 15, dst);
  };
  const int kk = lane >> 4;       // k index inside an MFMA step (0..3)
  const int col = lane & 15;      // candidate column of this lane / query row for A
  const int n_tiles = (N + 15) >> 4;
  float xxq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = i0 + 4 * kk + r;
    xxq[r] = q < N ? xxb[q] : 0.0f;
  }

  // candidates are processed in chunks of at most 64*VPL columns (one LDS tile); the running
  // top-k of every row lives in lanes 0..k-1 (lane r = r-th best) and is merged per chunk
  const int n_cols = n_tiles * 16;
  const int CW = (64 * VPL) < n_cols ? (64 * VPL) : n_cols;   // multiple of 16
  float* qa = lds + (size_t)kQ * ldp;                          // generic-C path only
  unsigned long long* scratch =
      reinterpret_cast<unsigned long long*>(lds + (size_t)kQ * ldp + (C4T == 0 ? ((C + 3) >> 2) * 4 * kQ : 0)) +
      wave * 64;
  float a[C4T > 0 ? C4T : 1];
  if constexpr (C4T > 0) {
    const int q = i0 + col;
#pragma unroll
    for (int c4 = 0; c4 < C4T; ++c4) {
      const int c = 4 * c4 + kk;
      a[c4] = (c < C && q < N) ? xb[(size_t)c * N + q] : 0.0f;
    }
  } else {
    const int C4 = (C + 3) >> 2;
    for (int e = tid; e < C4 * 4 * kQ; e += kKnnThreads) {
      const int c = e >> 4, q = e & 15;
      qa[e] = (c < C && i0 + q < N) ? xb[(size_t)c * N + i0 + q] : 0.0f;
    }
  }
  float run_val[kRowsPerWave];
  int run_idx[kRowsPerWave];
#pragma unroll
  for (int rr = 0; rr < kRowsPerWave; ++rr) { run_val[rr] = -__builtin_inff(); run_idx[rr] = 0; }

  for (int c0 = 0; c0 < n_cols; c0 += CW) {
    __syncthreads();   // previous chunk's tile fully consumed (and qa staged, first time round)
    const int t_first = c0 >> 4;
    const int t_last = ((c0 + CW) >> 4) < n_tiles ? ((c0 + CW) >> 4) : n_tiles;
    // ---------------------------------------------------------------- phase A
    if constexpr (C4T > 0) {
      float b0[C4T], b1[C4T];
      const int tw = t_first + wave;
      if constexpr (NW >= 16) {
        // 4 waves per SIMD (128 VGPRs each): prefetch one tile ahead, two register sets
        ldb(tw < t_last ? tw : n_tiles, b0);
        for (int t = tw; t < t_last; t += 2 * NW) {
          ldb(t + NW < t_last ? t + NW : n_tiles, b1);
          score_tile<C4T>(a, b0, t, N, kk, col, xxb, xxq, pd, ldp, c0);
          if (t + NW < t_last) {
            ldb(t + 2 * NW < t_last ? t + 2 * NW : n_tiles, b0);
            score_tile<C4T>(a, b1, t + NW, N, kk, col, xxb, xxq, pd, ldp, c0);
          }
        }
      } else {
        float b2[C4T];
        ldb(tw < t_last ? tw : n_tiles, b0);
        ldb(tw + NW < t_last ? tw + NW : n_tiles, b1);
        for (int t = tw; t < t_last; t += 3 * NW) {
          ldb(t + 2 * NW < t_last ? t + 2 * NW : n_tiles, b2);
          score_tile<C4T>(a, b0, t, N, kk, col, xxb, xxq, pd, ldp, c0);
          if (t + NW < t_last) {
            ldb(t + 3 * NW < t_last ? t + 3 * NW : n_tiles, b0);
            score_tile<C4T>(a, b1, t + NW, N, kk, col, xxb, xxq, pd, ldp, c0);
          }
          if (t + 2 * NW < t_last) {
            ldb(t + 4 * NW < t_last ? t + 4 * NW : n_tiles, b1);
            score_tile<C4T>(a, b2, t + 2 * NW, N, kk, col, xxb, xxq, pd, ldp, c0);
          }
        }
      }
    } else {
      const int C4 = (C + 3) >> 2;
      for (int t = t_first + wave; t < t_last; t += kKnnWaves) {
        const int j = t * 16 + col;
        const bool jin = j < N;
        const float* __restrict__ bp = xb + (jin ? j : 0);
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c4 = 0; c4 < C4; ++c4) {
          const int c = 4 * c4 + kk;
          const float av = qa[c * kQ + col];
          const float bv = (jin && c < C) ? bp[(size_t)c * N] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        const float xxj = jin ? xxb[j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fma_rn(2.0f, acc[r], -xxj) - xxq[r];
          pd[(4 * kk + r) * ldp + (j - c0)] = jin ? v : -__builtin_inff();
        }
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase B
    // wave w selects its kRowsPerWave rows together; VPL scores per lane and row, groups of 8
    constexpr int NG = VPL / kGroup;
    const int cw = (n_cols - c0) < CW ? (n_cols - c0) : CW;
    float v[kRowsPerWave][VPL];
    float gmax[kRowsPerWave][NG];
    int gt[kRowsPerWave][NG];
    int mine[kRowsPerWave];
    float mval[kRowsPerWave];
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
      const int q = kRowsPerWave * wave + rr;
      mine[rr] = 0;
      mval[rr] = -__builtin_inff();
#pragma unroll
      for (int t = 0; t < VPL; ++t) {
        const int e = t * 64 + lane;
        v[rr][t] = e < cw ? pd[q * ldp + e] : -__builtin_inff();
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float m = v[rr][g * kGroup];
        int mt = g * kGroup;
#pragma unroll
        for (int u = 1; u < kGroup; ++u) {
          const bool gtr = v[rr][g * kGroup + u] > m;   // strict: lower slot wins ties
          mt = gtr ? g * kGroup + u : mt;
          m = gtr ? v[rr][g * kGroup + u] : m;
        }
        gmax[rr][g] = m;
        gt[rr][g] = mt;
      }
    }
    const int rounds = k < cw ? k : cw;   // a short last chunk may hold fewer than k columns
    bool by_threshold = rounds == k && k <= 32;     // larger k: too many scores pass for the 64-entry strip
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
      if (by_threshold) {
        float lm = gmax[rr][0];
#pragma unroll
        for (int g = 1; g < NG; ++g) lm = __builtin_fmaxf(lm, gmax[rr][g]);
        by_threshold = select_by_threshold<VPL>(v[rr], lm, k, c0, lane, scratch, mine[rr], mval[rr]);
      }
    }
    for (int round = 0; round < (by_threshold ? 0 : rounds); ++round) {
#pragma unroll
      for (int rr = 0; rr < kRowsPerWave; ++rr) {
        float bv = gmax[rr][0];
        int bt = gt[rr][0];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          const bool gtr = gmax[rr][g] > bv;
          bt = gtr ? gt[rr][g] : bt;
          bv = gtr ? gmax[rr][g] : bv;
        }
        const float m = wave_max_f32(bv);
        const unsigned e = (unsigned)(bt * 64 + lane);
        const unsigned win = wave_min_u32(bv == m ? e : 0xffffffffu);   // lowest index among maxima
        mine[rr] = (lane == round) ? (int)win + c0 : mine[rr];
        mval[rr] = (lane == round) ? m : mval[rr];
        // retire the winner: only its group (wave-uniform) is rescanned
        const int wl = (int)(win & 63u), wt = (int)(win >> 6);
        const int wg = __builtin_amdgcn_readfirstlane(wt / kGroup);
        const bool me = lane == wl;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g == wg) {
            float mm = -__builtin_inff();
            int mt = g * kGroup;
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
              const int t = g * kGroup + u;
              const float cur = (me && t == wt) ? -__builtin_inff() : v[rr][t];
              v[rr][t] = cur;
              const bool gtr = cur > mm || u == 0;
              mt = gtr ? t : mt;
              mm = gtr ? cur : mm;
            }
            gmax[rr][g] = mm;
            gt[rr][g] = mt;
          }
        }
      }
    }
    // ---------------------------------------------------------------- merge with earlier chunks
    if (c0 == 0) {
#pragma unroll
      for (int rr = 0; rr < kRowsPerWave; ++rr) { run_val[rr] = mval[rr]; run_idx[rr] = mine[rr]; }
    } else {
      // two lists sorted by (score desc, index asc) in lanes 0..k-1: an element's place in the
      // merged order is its own rank plus the number of elements of the other list that
      // precede it; the first k places are written to a per-wave LDS strip and read back
#pragma unroll
      for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const bool have_a = lane < k, have_b = lane < rounds;
        const unsigned long long ka =
            have_a ? (((unsigned long long)orderable(run_val[rr] + 0.0f) << 32) | (unsigned)~run_idx[rr]) : 0ull;
        const unsigned long long kb =
            have_b ? (((unsigned long long)orderable(mval[rr] + 0.0f) << 32) | (unsigned)~mine[rr]) : 0ull;
        int rank_a = lane, rank_b = lane;
        for (int s2 = 0; s2 < k; ++s2) {
          const unsigned long long oa = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(ka >> 32), s2) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)ka, s2);
          const unsigned long long ob = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(kb >> 32), s2) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)kb, s2);
          rank_a += ob > ka ? 1 : 0;
          rank_b += oa > kb ? 1 : 0;
        }
        if (have_a && rank_a < k) scratch[rank_a] = ka;
        if (have_b && rank_b < k) scratch[rank_b] = kb;
        // same-wave LDS write -> read: ordered by the in-order LDS queue
        const unsigned long long mk = lane < k ? scratch[lane] : 0ull;
        const unsigned ov = (unsigned)(mk >> 32);
        run_val[rr] = __uint_as_float((ov & 0x80000000u) ? (ov & 0x7fffffffu) : ~ov);
        run_idx[rr] = (int)~(unsigned)mk;
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < kRowsPerWave; ++rr) {
    const int i = i0 + kRowsPerWave * wave + rr;
    if (i < N && lane < k) idx[((size_t)b * N + i) * k + lane] = run_idx[rr];
  }
}

template <int VPL, int C4T, int NW, bool PM = false>
int launch_knn(const float* x, const float* xx, const float* xk, int B, int C, int N, int k, int32_t* idx,
               hipStream_t s) {
  const int n_cols = ((N + 15) / 16) * 16;
  const int cw = n_cols < 64 * VPL ? n_cols : 64 * VPL;
  const int ldp = cw + 4;  // +4: the 4 query rows a lane group writes hit disjoint banks
  const int C4 = (C + 3) / 4;
  const size_t lds_bytes = ((size_t)kQ * ldp + (C4T == 0 ? (size_t)C4 * 4 * kQ : 0)) * sizeof(float) +
                           NW * 64 * sizeof(unsigned long long);
  dim3 grid((N + kQ - 1) / kQ, B);
  auto kern = knn_kernel<VPL, C4T, NW, PM>;
  // opt-in to the full 160 KiB of LDS for this instantiation: per call (the attribute belongs to the current
  // device's copy of the function, so a process driving several devices needs it on each; the call is a host-side
  // table update, ~1 us)
  const hipError_t lds_optin = hipFuncSetAttribute(
      reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (lds_optin != hipSuccess && lds_bytes > 64 * 1024) {
    set_error("fpsg_knn: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(lds_optin));
    return (int)lds_optin;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds_bytes, s, x, xx, xk, C, N, k, ldp, idx);
  return launch_status("fpsg_knn");
}

template <int VPL>
int launch_knn_c(const float* x, const float* xx, const float* xk, int B, int C, int N, int k, int32_t* idx,
                 hipStream_t s) {
  if (C <= 4) return launch_knn<VPL, 1, 16>(x, xx, nullptr, B, C, N, k, idx, s);
  if (xk && C > 32 && C <= 64) return launch_knn<VPL, 16, 16, true>(x, xx, xk, B, C, N, k, idx, s);
  if (xk && C > 64 && C <= 128) return launch_knn<VPL, 32, 8, true>(x, xx, xk, B, C, N, k, idx, s);
  if (C > 32 && C <= 64) return launch_knn<VPL, 16, 16>(x, xx, nullptr, B, C, N, k, idx, s);
  if (C > 64 && C <= 128) return launch_knn<VPL, 32, 8>(x, xx, nullptr, B, C, N, k, idx, s);
  return launch_knn<VPL, 0, 8>(x, xx, nullptr, B, C, N, k, idx, s);
}

inline bool knn_uses_pm(int C) { return C % 16 == 0 && C > 32 && C <= 128; }

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_knn_workspace_floats(int B, int C, int N) {
  if (B <= 0 || C <= 0 || N <= 0) return 0;
  const size_t bn = (((size_t)B * N + 3) / 4) * 4;      // the feature copy starts 16-byte aligned
  const size_t tile_copy = fpsg::knn_uses_pm(C) ? (size_t)B * N * C : 0;
  const size_t stream_copy = (size_t)B * N * fpsg::knn_stream_cpad(C);
  return bn + (tile_copy > stream_copy ? tile_copy : stream_copy);
}

namespace fpsg {
namespace {
// the score-tile kernel (any C <= 440, k <= 64): x channel-major
int knn_tile(const float* x, int B, int C, int N, int k, int32_t* idx, float* ws, hipStream_t s) {
  float* xx = ws;
  float* xk = nullptr;
  // the point-major copy needs 16-byte aligned rows; otherwise the channel-major operands are read as before
  if (knn_uses_pm(C) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
    xk = ws + (((size_t)B * N + 3) / 4) * 4;
    hipLaunchKernelGGL(knn_prep_kernel, dim3((N + kPrepPts - 1) / kPrepPts, B), dim3(256),
                       (size_t)kPrepPts * (C + 4) * sizeof(float), s, x, C, N, xx, xk);
  } else {
    hipLaunchKernelGGL(sqnorm_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, x, C, N, xx);
  }
  int rc = launch_status("fpsg_knn(prepare)");
  if (rc) return rc;
  const int vpl = (((N + 15) / 16) * 16 + 63) / 64;
  if (vpl <= 8) return launch_knn_c<8>(x, xx, xk, B, C, N, k, idx, s);
  if (vpl <= 16) return launch_knn_c<16>(x, xx, xk, B, C, N, k, idx, s);
  return launch_knn_c<32>(x, xx, xk, B, C, N, k, idx, s);   // N > 2048: chunks of 2048 columns
}
}  // namespace
}  // namespace fpsg

extern "C" int fpsg_knn_ex(const float* x, int layout, int B, int C, int N, int k, int32_t* idx, float* ws,
                           int flags, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(B > 0 && C > 0 && N > 0 && k > 0, FPSG_E_SHAPE,
               "fpsg_knn: B,C,N,k must be positive (got %d,%d,%d,%d)", B, C, N, k);
  FPSG_REQUIRE(k <= 64 && k <= N, FPSG_E_LIMIT, "fpsg_knn: need k <= min(64, N) (k=%d, N=%d)", k, N);
  FPSG_REQUIRE(C <= 440 && (long)N <= (1L << 24), FPSG_E_LIMIT,
               "fpsg_knn: C=%d exceeds 440 (16 x C query tile beside the 128 KiB score tile) or N=%d > 2^24", C, N);
  FPSG_REQUIRE(B <= 65535, FPSG_E_LIMIT, "fpsg_knn: B=%d exceeds 65535", B);
  FPSG_REQUIRE(layout == FPSG_KNN_CHANNEL_MAJOR || layout == FPSG_KNN_POINT_MAJOR, FPSG_E_SHAPE,
               "fpsg_knn: layout %d is neither FPSG_KNN_CHANNEL_MAJOR nor FPSG_KNN_POINT_MAJOR", layout);
  FPSG_REQUIRE_PTR(x); FPSG_REQUIRE_PTR(idx); FPSG_REQUIRE_PTR(ws);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool stream_ok = knn_stream_serves(C, k) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
                         !(flags & FPSG_KNN_FORCE_TILE);
  if (stream_ok) {
    float* xx = ws;
    float* xk = ws + (((size_t)B * N + 3) / 4) * 4;
    int rc = knn_stream_prepare(x, layout == FPSG_KNN_POINT_MAJOR, B, C, N, xx, xk, s);
    if (rc) return rc;
    return knn_stream_launch(xk, xx, B, C, N, k, (flags & FPSG_KNN_FORCE_SLOW) ? 1 : 0, idx, s);
  }
  FPSG_REQUIRE(layout == FPSG_KNN_CHANNEL_MAJOR, FPSG_E_LIMIT,
               "fpsg_knn: point-major features are served for C <= 128 and k <= 24 with a 16-byte aligned workspace "
               "(C=%d, k=%d)", C, k);
  return knn_tile(x, B, C, N, k, idx, ws, s);
}

extern "C" int fpsg_knn(const float* x, int B, int C, int N, int k, int32_t* idx,
                        float* ws, fpsg_stream_t stream) {
  return fpsg_knn_ex(x, FPSG_KNN_CHANNEL_MAJOR, B, C, N, k, idx, ws, 0, stream);
}
