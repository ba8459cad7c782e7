// knn_stream.hip -- K3, streaming form: the k-nearest-neighbour graph of DGCNN's EdgeConv for gfx950.
// Replaces `knn` of reference src/dgcnn/model.py:13-20 (torch.matmul of x^T x into a [B,N,N] tensor + torch.topk).
//
// A workgroup (8 waves) owns 128 query points of one cloud, a wave 16 of them for the whole sweep.  The cloud's
// candidates stream ONCE per workgroup through LDS in stages (a 16 KB tile of k-interleaved point-major features, two
// buffers, one barrier per stage); every wave reads its MFMA B operands from there (conflict-free 16-byte reads: the
// 16-byte units of a candidate's channel segment are XOR-swizzled by the candidate number) and keeps its A operands --
// its 16 queries -- in registers.  The 16 x 16 block of scores  pd_ij = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  of a tile
// is produced by v_mfma_f32_16x16x4_f32 (exact channel-ordered fma chain = the oracle's), and NOTHING of size N is
// kept per row: the selection works on the accumulator layout itself (lane (g, c) holds the scores of rows 4g..4g+3
// against candidate c, so a register holds 4 rows x 16 candidates and every vector instruction serves 4 rows):
//   * a score that reaches its row's threshold T is appended (score bits, index) to the row's 120-entry buffer in
//     LDS, its slot taken with an LDS atomic;
//   * T is a lower bound of the row's k-th best score, raised by events: the 16 lanes of a row split its buffered
//     entries, each finds the two best of its share (v_max / v_med3); if at least ceil(k/2) lanes hold a second-best
//     >= w, then k buffered scores are >= w.  The largest such w comes from 15 DPP row rotations, and the same pass
//     drops the entries below the new T (4 rows at a time).  Events run at fixed tiles (x2.25 in the number of
//     candidates seen: five per 2048-point cloud) and whenever a buffer passes its watermark.  Between two checks
//     (2 tiles) a row receives at most 32 entries, and a check leaves at most 88: a slot index cannot pass the
//     buffer's end;
//   * after the sweep a row's buffer holds every score >= its final T (about 1.4 k of them): they are ranked by
//     counting on 64-bit keys (orderable score << 32 | ~index: score descending, then index ascending -- the order of
//     the oracle's rounds) and the first k are written.
// If a compaction cannot bring a buffer under the watermark (a loose early threshold, hundreds of equal scores), the
// row's entries are ranked on the spot, its k best stay and the k-th score becomes its threshold.  A second, independent
// selection is kept for tests (FPSG_KNN_FORCE_SLOW): k masked arg-max sweeps over the cloud straight from global
// memory.  Results are bit-identical to oracle_knn either way.
#include "knn_internal.h"

namespace fpsg {
namespace {

constexpr int kSW = 8;                 // waves per workgroup
constexpr int kSRows = 16 * kSW;       // query rows per workgroup
constexpr int kCap = 120;              // entries per row buffer
constexpr int kWM = kCap - 32;         // a check leaves at most this many entries in a buffer
constexpr int kFilterIters = (kCap + 15) / 16;

template <int C4T> struct StreamCfg;
template <> struct StreamCfg<1> { static constexpr int TC = 512; };    //  8 KB per stage
template <> struct StreamCfg<16> { static constexpr int TC = 64; };    // 16 KB
template <> struct StreamCfg<32> { static constexpr int TC = 32; };    // 16 KB

template <int C4T>
constexpr size_t stream_lds_bytes() {
  constexpr int TC = StreamCfg<C4T>::TC;
  return (size_t)2 * TC * 4 * C4T * 4 + (size_t)2 * TC * 4 + (size_t)kSRows * 4 + (size_t)kSW * 4 +
         (size_t)kSRows * kCap * 8;
}

// ---- prepare: squared norms + the k-interleaved, zero-padded point-major copy ------------------------------------
//   xk[b][n][kk * C4T + c4] = x[b][4 c4 + kk][n]  (0 beyond C)      xx[b][n] = fma chain over c ascending
// One workgroup = 64 points; the slab is transposed through LDS so that both sides are coalesced.
template <bool PM>
__global__ __launch_bounds__(256) void knn_stream_prep_kernel(const float* __restrict__ x, int C, int N, int CP,
                                                              float* __restrict__ xx, float* __restrict__ xk) {
  extern __shared__ __attribute__((aligned(16))) float tile[];      // [64][CP + 4]
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ld = CP + 4;
  const int C4T = CP >> 2;
  for (int e = threadIdx.x; e < 64 * ld; e += 256) tile[e] = 0.0f;
  __syncthreads();
  if (PM) {
    const float* xb = x + ((size_t)b * N + n0) * C;
    const int total = (N - n0 < 64 ? N - n0 : 64) * C;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int p = e / C, c = e - p * C;
      tile[p * ld + (c & 3) * C4T + (c >> 2)] = xb[e];
    }
  } else {
    const float* xb = x + (size_t)b * C * N;
    const int n = n0 + lane;
    for (int c = wave; c < C; c += 4)
      if (n < N) tile[lane * ld + (c & 3) * C4T + (c >> 2)] = xb[(size_t)c * N + n];
  }
  __syncthreads();
  if (wave == 0 && n0 + lane < N) {                 // squared norm in channel order (the oracle's fma chain)
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float v = tile[lane * ld + (c & 3) * C4T + (c >> 2)];
      acc = fma_rn(v, v, acc);
    }
    xx[(size_t)b * N + n0 + lane] = acc;
  }
  float* dst = xk + ((size_t)b * N + n0) * CP;
  for (int e = threadIdx.x; e < 64 * C4T; e += 256) {
    const int p = e / C4T, q = e - p * C4T;
    if (n0 + p < N) *reinterpret_cast<v4f*>(dst + (size_t)p * CP + 4 * q) = *reinterpret_cast<const v4f*>(tile + p * ld + 4 * q);
  }
}

// ---- lane exchanges inside a row of 16 lanes ------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i0(int v) {           // lanes without a source receive 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float row16_max(float v) {     // every lane of a row of 16 receives the row's maximum
  v = __builtin_fmaxf(v, dpp_f<0xB1>(v));     // quad_perm [1,0,3,2]
  v = __builtin_fmaxf(v, dpp_f<0x4E>(v));     // quad_perm [2,3,0,1]
  v = __builtin_fmaxf(v, dpp_f<0x141>(v));    // row_half_mirror
  v = __builtin_fmaxf(v, dpp_f<0x140>(v));    // row_mirror
  return v;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long v) {
#define FPSG_STEP(M)                                                                                      \
  {                                                                                                       \
    const unsigned long long o = ((unsigned long long)lane_xor<M>((unsigned)(v >> 32)) << 32) | lane_xor<M>((unsigned)v); \
    v = o > v ? o : v;                                                                                    \
  }
  FPSG_STEP(1) FPSG_STEP(2) FPSG_STEP(4) FPSG_STEP(8)
#undef FPSG_STEP
  return v;
}
// number of lanes of the row (this one included) whose value is >= this lane's
template <int N>
__device__ __forceinline__ int row16_count_ge(float w, int acc) {
  if constexpr (N < 16) {
    return row16_count_ge<N + 1>(w, acc + (dpp_f<0x120 + N>(w) >= w ? 1 : 0));      // row_ror:N
  } else {
    return acc;
  }
}

// v_max_f32 as it is: the builtin adds two canonicalising moves per call, and NaN scores are outside the contract
__device__ __forceinline__ float raw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ unsigned long long knn_key(float score, unsigned j) {
  return ((unsigned long long)knn_orderable(score + 0.0f) << 32) | (unsigned)~j;
}

struct Entry { unsigned s, j; };       // score bits, candidate index

// ---- the kernel -----------------------------------------------------------------------------------------------------
// flags bit 0: every wave takes the slow exact path (tests).
template <int C4T>
__global__ __launch_bounds__(64 * kSW) void knn_stream_kernel(const float* __restrict__ xk,
                                                              const float* __restrict__ xx, int B, int N, int k,
                                                              int flags, int32_t* __restrict__ idx) {
  constexpr int TC = StreamCfg<C4T>::TC;
  constexpr int CP = 4 * C4T;
  constexpr int UPS = C4T >= 4 ? C4T / 4 : 1;                 // 16-byte units per (candidate, k) segment
  constexpr int NU = (TC * CP / 4 + 64 * kSW - 1) / (64 * kSW);   // 16-byte units per thread and stage
  constexpr int kThreads = 64 * kSW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* stage = lds;                                           // [2][TC * CP]
  float* sxx = stage + 2 * TC * CP;                             // [2][TC]
  unsigned* cnt = reinterpret_cast<unsigned*>(sxx + 2 * TC);    // [kSRows]
  unsigned* tsl = cnt + kSRows;                                 // [kSW]: a wave's word for the exact compaction
  Entry* buf = reinterpret_cast<Entry*>(tsl + kSW);             // [kSRows][kCap]

  // clouds -> XCDs: workgroup ids go round-robin over the 8 XCDs, so the row blocks of one cloud take ids of one
  // residue class and its features are served by one L2
  const int nblk = (N + kSRows - 1) / kSRows;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / nblk) * 8 + xcd;
  const int blk = slot - (slot / nblk) * nblk;
  if (b >= B) return;                                           // whole workgroup, before any barrier

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 4, col = lane & 15;
  const float* __restrict__ xkb = xk + (size_t)b * N * CP;
  const float* __restrict__ xxb = xx + (size_t)b * N;
  const int i0 = blk * kSRows + wave * 16;                      // the wave's first query row
  const int n_tiles = (N + 15) >> 4;
  const int n_stages = (N + TC - 1) / TC;
  const float NEG = -__builtin_inff();

  // A operands: lane (kk, col) holds channels 4 c4 + kk of query i0 + col
  float a[C4T];
  {
    const int q = i0 + col;
    if constexpr (C4T >= 4) {
      const v4f* src = reinterpret_cast<const v4f*>(xkb + (size_t)(q < N ? q : 0) * CP + kk * C4T);
#pragma unroll
      for (int m = 0; m < C4T / 4; ++m) {
        v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (q < N) v = src[m];
        a[4 * m] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
      }
    } else {
      a[0] = q < N ? xkb[(size_t)q * CP + kk] : 0.0f;
    }
  }
  float xxq[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int q = i0 + 4 * kk + rr;
    xxq[rr] = q < N ? xxb[q] : 0.0f;
  }
  const int row0 = wave * 16 + 4 * kk;                          // this lane's rows: row0 + rr
  if (lane < 16) cnt[wave * 16 + lane] = 0u;

  float T[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) T[rr] = NEG;
  const int jsel = (k + 1) >> 1;
  const bool failed = (flags & 1) != 0;                         // wave-uniform: the slow exact path (tests, A/B)

  // ---- stage transfer --------------------------------------------------------------------------------------------
  v4f pre[NU];
  float prexx = 0.0f;
  auto load_stage = [&](int s) {
    const int j0 = s * TC;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int g = tid + u * kThreads;
      const int jl = g / (CP / 4), q = g - jl * (CP / 4);
      v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (g < TC * CP / 4 && j0 + jl < N) v = *reinterpret_cast<const v4f*>(xkb + (size_t)(j0 + jl) * CP + 4 * q);
      pre[u] = v;
    }
    if (tid < TC) prexx = (j0 + tid < N) ? xxb[j0 + tid] : __builtin_inff();   // padding scores -inf
  };
  auto store_stage = [&](int sel) {
    float* st = stage + sel * TC * CP;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int g = tid + u * kThreads;
      if (g < TC * CP / 4) {
        const int jl = g / (CP / 4), q = g - jl * (CP / 4);
        int unit;
        if constexpr (C4T >= 4) {
          const int k2 = q / UPS, uu = q - k2 * UPS;
          unit = k2 * (TC * UPS) + jl * UPS + (uu ^ ((jl / (16 / UPS)) % UPS));
        } else {
          unit = jl;
        }
        *reinterpret_cast<v4f*>(st + 4 * unit) = pre[u];
      }
    }
    if (tid < TC) sxx[sel * TC + tid] = prexx;
  };

  // ---- selection pieces ------------------------------------------------------------------------------------------
  // An event raises the rows' thresholds and drops the buffered entries below them: 16 lanes per row, the 4 rows of a
  // register index at a time.  Lane c of a row holds its entries c, c + 16, ... (distinct candidates, all >= the old
  // T) and finds the two best of them; if at least ceil(k/2) lanes hold a second-best >= w, then k buffered scores are
  // >= w, so w is a lower bound of the row's k-th best score.  The largest such w comes from 15 DPP row rotations.
  auto raise_and_filter = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = row0 + rr;
      const int c = (int)cnt[row];
      int cmax = __builtin_amdgcn_readlane(c, 0);                    // the longest of the four rows: wave-uniform
      cmax = max(cmax, __builtin_amdgcn_readlane(c, 16));
      cmax = max(cmax, __builtin_amdgcn_readlane(c, 32));
      cmax = max(cmax, __builtin_amdgcn_readlane(c, 48));
      Entry* rb = buf + row * kCap;
      Entry e[kFilterIters];
      float sc[kFilterIters];
      float a1 = NEG, a2 = NEG;
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) {
        sc[i] = NEG;
        if (16 * i < cmax) {
          const int p = col + 16 * i;
          const bool valid = p < c;
          e[i] = rb[valid ? p : 0];
          sc[i] = valid ? __uint_as_float(e[i].s) : NEG;
          a2 = __builtin_amdgcn_fmed3f(a1, a2, sc[i]);
          a1 = raw_max(a1, sc[i]);
        }
      }
      const int ge = row16_count_ge<1>(a2, 1);
      T[rr] = __builtin_fmaxf(T[rr], row16_max(ge >= jsel ? a2 : NEG));
      int keep[kFilterIters];
      int n = 0;
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) {
        keep[i] = 0;
        if (16 * i < cmax) {
          keep[i] = (col + 16 * i < c && sc[i] >= T[rr]) ? 1 : 0;
          n += keep[i];
        }
      }
      int inc = n;                                   // inclusive scan over the row's 16 lanes
      inc += dpp_i0<0x111>(inc);                     // row_shr:1
      inc += dpp_i0<0x112>(inc);                     // row_shr:2
      inc += dpp_i0<0x114>(inc);                     // row_shr:4
      inc += dpp_i0<0x118>(inc);                     // row_shr:8
      int pos = inc - n;
      __builtin_amdgcn_wave_barrier();               // every read of this row precedes the writes (in-order LDS queue)
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) {
        if (16 * i < cmax) {
          if (keep[i]) rb[pos] = e[i];
          pos += keep[i];
        }
      }
      if (col == 15) cnt[row] = (unsigned)inc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // the c buffered entries of a row -> 64-bit keys in place and every entry's rank among them (wave-level; lane l
  // holds entries l and l + 64)
  auto rank_row = [&](int row, int c, Entry& e0, Entry& e1, unsigned long long& k0, unsigned long long& k1, int& r0,
                      int& r1) {
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(buf) + (size_t)row * kCap;
    const bool h0 = lane < c, h1 = lane + 64 < c;
    e0 = buf[row * kCap + (h0 ? lane : 0)];
    e1 = buf[row * kCap + (h1 ? lane + 64 : 0)];
    k0 = h0 ? knn_key(__uint_as_float(e0.s), e0.j) : 0ull;
    k1 = h1 ? knn_key(__uint_as_float(e1.s), e1.j) : 0ull;
    __builtin_amdgcn_wave_barrier();
    if (h0) rk[lane] = k0;                           // a lane rewrites only its own slots
    if (h1) rk[lane + 64] = k1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    r0 = 0;
    r1 = 0;
    if (c <= 64) {
#pragma unroll 4
      for (int l = 0; l < c; ++l) r0 += rk[l] > k0 ? 1 : 0;
    } else {
#pragma unroll 2
      for (int l = 0; l < c; ++l) {
        const unsigned long long o = rk[l];
        r0 += o > k0 ? 1 : 0;
        r1 += o > k1 ? 1 : 0;
      }
    }
    __builtin_amdgcn_wave_barrier();                 // the keys have been read: the slots may be rewritten
  };
  // a row the filter left over the watermark (a loose early threshold, many equal scores): keep exactly its k best
  // entries and take the k-th score as the row's threshold -- nothing below the k best seen so far can be selected
  auto exact_compact = [&]() {
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 16 + r;
      const int c = __builtin_amdgcn_readfirstlane((int)cnt[row]);
      if (c <= kWM) continue;
      Entry e0, e1;
      unsigned long long k0, k1;
      int r0, r1;
      rank_row(row, c, e0, e1, k0, k1, r0, r1);
      const bool h0 = lane < c, h1 = lane + 64 < c;
      Entry* rb = buf + row * kCap;
      if (h0 && r0 < k) rb[r0] = e0;
      if (h1 && r1 < k) rb[r1] = e1;
      if (h0 && r0 == k - 1) tsl[wave] = e0.s;
      if (h1 && r1 == k - 1) tsl[wave] = e1.s;
      if (lane == 0) cnt[row] = (unsigned)k;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float tn = __uint_as_float(tsl[wave]);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        T[rr] = (kk == (r >> 2) && rr == (r & 3)) ? __builtin_fmaxf(T[rr], tn) : T[rr];
    }
  };
  auto over_watermark = [&]() -> bool {
    const unsigned c = cnt[wave * 16 + col];
    return __builtin_amdgcn_ballot_w64(c > (unsigned)kWM) != 0ull;
  };
  auto event = [&]() {
    raise_and_filter();
    if (over_watermark()) exact_compact();
  };
  // the scores of tiles t, t + 1 (accumulator layout) -> threshold test, append.  A lane's rows are the same in both
  // tiles, so one LDS atomic per register index takes the slots of both; the four atomics of a pair are issued
  // together (one round trip).  Returns whether a row passed its watermark: the lane that took a row's last slot knows
  // the row's new count.  (Skipping the append of a register index none of whose 128 scores passes was measured: the
  // four ballots and scalar branches per pair cost more than they save, 324 -> 372 us at C = 3.)
  auto select_pair = [&](const v4f& acc0, const v4f& acc1, int t, float xx0, float xx1) -> bool {
    const unsigned j0 = (unsigned)(16 * t + col);
    float v0[4], v1[4];
    unsigned pos[4];
    bool p0[4], p1[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      v0[rr] = fma_rn(2.0f, acc0[rr], -xx0) - xxq[rr];
      v1[rr] = fma_rn(2.0f, acc1[rr], -xx1) - xxq[rr];
      p0[rr] = v0[rr] >= T[rr];
      p1[rr] = v1[rr] >= T[rr];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      pos[rr] = 0u;
      if (p0[rr] || p1[rr])
        pos[rr] = __hip_atomic_fetch_add(&cnt[row0 + rr], (p0[rr] ? 1u : 0u) + (p1[rr] ? 1u : 0u), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    unsigned top = 0u;                         // the largest new row count this lane knows of
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      Entry* rb = buf + (row0 + rr) * kCap;
      const unsigned q1 = pos[rr] + (p0[rr] ? 1u : 0u);
      if (p0[rr]) { Entry en; en.s = __float_as_uint(v0[rr]); en.j = j0; rb[pos[rr]] = en; }
      if (p1[rr]) { Entry en; en.s = __float_as_uint(v1[rr]); en.j = j0 + 16u; rb[q1] = en; }
      const unsigned endc = q1 + (p1[rr] ? 1u : 0u);      // pos = 0 for a lane that appended nothing
      top = endc > top ? endc : top;
    }
    return __builtin_amdgcn_ballot_w64(top > (unsigned)kWM) != 0ull;
  };

  // ---- the sweep --------------------------------------------------------------------------------------------------
  load_stage(0);
  store_stage(0);
  __syncthreads();
  int next_evt = 1 << 30;          // set by the first watermark event
  for (int s = 0; s < n_stages; ++s) {
    const int sel = s & 1;
    if (s + 1 < n_stages) load_stage(s + 1);
    const float* st = stage + sel * TC * CP;
    const int t0 = s * (TC / 16);
#pragma unroll 1
    for (int tt = 0; tt < TC / 16 && t0 + tt < n_tiles; tt += 2) {
      v4f acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (C4T >= 4) {
        const int sw = (col / (16 / UPS)) % UPS;      // the swizzle of candidate 16 tt + col (16 tt drops out)
        const float* p0 = st + kk * (TC * C4T) + (16 * tt + col) * C4T;
        const float* p1 = p0 + 16 * C4T;
#pragma unroll
        for (int m = 0; m < UPS; ++m) {
          const v4f b0 = *reinterpret_cast<const v4f*>(p0 + 4 * (m ^ sw));
          const v4f b1 = *reinterpret_cast<const v4f*>(p1 + 4 * (m ^ sw));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + e], b0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + e], b1[e], acc1, 0, 0, 0);
          }
        }
      } else {
        const float b0 = st[(16 * tt + col) * 4 + kk];
        const float b1 = st[(16 * tt + 16 + col) * 4 + kk];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1, acc1, 0, 0, 0);
      }
      const float xx0 = sxx[sel * TC + 16 * tt + col];
      const float xx1 = sxx[sel * TC + 16 * tt + 16 + col];
      if (!failed) {
        const bool over = select_pair(acc0, acc1, t0 + tt, xx0, xx1);   // a tile past the cloud's end scores -inf
        const int t_end = t0 + tt + 1;
        if (over || t_end >= next_evt) {
          event();
          next_evt = ((t_end + 1) * 9 / 4) | 1;
        }
      }
    }
    if (s + 1 < n_stages) store_stage(sel ^ 1);
    __syncthreads();
  }

  if (!failed) {
    event();
    // ---- rank the survivors of every row: wave-level, one row at a time ------------------------------------------
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 16 + r;
      const int i = i0 + r;
      if (i >= N) break;
      const int c = __builtin_amdgcn_readfirstlane((int)cnt[row]);
      Entry e0, e1;
      unsigned long long k0, k1;
      int r0, r1;
      rank_row(row, c, e0, e1, k0, k1, r0, r1);
      int32_t* out = idx + ((size_t)b * N + i) * k;
      if (lane < c && r0 < k) out[r0] = (int32_t)~(unsigned)k0;
      if (lane + 64 < c && r1 < k) out[r1] = (int32_t)~(unsigned)k1;
    }
    return;
  }

  // ---- slow exact path: k masked arg-max sweeps, operands straight from global memory ------------------------------
  unsigned long long prev[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) prev[rr] = ~0ull;
  for (int round = 0; round < k; ++round) {
    unsigned long long best[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
      const int j = 16 * t + col;
      const bool jin = j < N;
      v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (C4T >= 4) {
        const v4f* src = reinterpret_cast<const v4f*>(xkb + (size_t)(jin ? j : 0) * CP + kk * C4T);
#pragma unroll
        for (int m = 0; m < C4T / 4; ++m) {
          v4f bv = {0.0f, 0.0f, 0.0f, 0.0f};
          if (jin) bv = src[m];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * m + e], bv[e], acc, 0, 0, 0);
        }
      } else {
        const float bv = jin ? xkb[(size_t)j * CP + kk] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], bv, acc, 0, 0, 0);
      }
      const float xxj = jin ? xxb[j] : __builtin_inff();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float v = fma_rn(2.0f, acc[rr], -xxj) - xxq[rr];
        const unsigned long long key = knn_key(v, (unsigned)j);
        const bool take = key < prev[rr] && key > best[rr];
        best[rr] = take ? key : best[rr];
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const unsigned long long w = row16_max_u64(best[rr]);
      prev[rr] = w;
      const int i = i0 + 4 * kk + rr;
      if (col == 0 && i < N) idx[((size_t)b * N + i) * k + round] = (int32_t)~(unsigned)w;
    }
  }
}

template <int C4T>
int launch_stream(const float* xk, const float* xx, int B, int N, int k, int flags, int32_t* idx, hipStream_t s) {
  constexpr size_t lds_bytes = stream_lds_bytes<C4T>();
  static_assert(lds_bytes <= 160 * 1024, "stage + row buffers exceed the CU's LDS");
  auto kern = knn_stream_kernel<C4T>;
  const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (optin != hipSuccess) {
    set_error("fpsg_knn: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(optin));
    return (int)optin;
  }
  const int nblk = (N + kSRows - 1) / kSRows;
  const int grid = 8 * nblk * ((B + 7) / 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kSW), lds_bytes, s, xk, xx, B, N, k, flags, idx);
  return launch_status("fpsg_knn(stream)");
}

}  // namespace

int knn_stream_prepare(const float* x, bool point_major, int B, int C, int N, float* xx, float* xk, hipStream_t s) {
  const int CP = knn_stream_cpad(C);
  const dim3 grid((N + 63) / 64, B);
  const size_t lds = (size_t)64 * (CP + 4) * sizeof(float);
  if (point_major) hipLaunchKernelGGL(knn_stream_prep_kernel<true>, grid, dim3(256), lds, s, x, C, N, CP, xx, xk);
  else hipLaunchKernelGGL(knn_stream_prep_kernel<false>, grid, dim3(256), lds, s, x, C, N, CP, xx, xk);
  return launch_status("fpsg_knn(prepare)");
}

int knn_stream_launch(const float* xk, const float* xx, int B, int C, int N, int k, int flags, int32_t* idx,
                      hipStream_t s) {
  switch (knn_stream_cpad(C)) {
    case 4: return launch_stream<1>(xk, xx, B, N, k, flags, idx, s);
    case 64: return launch_stream<16>(xk, xx, B, N, k, flags, idx, s);
    case 128: return launch_stream<32>(xk, xx, B, N, k, flags, idx, s);
    default: set_error("fpsg_knn(stream): C=%d is not served", C); return FPSG_E_LIMIT;
  }
}

}  // namespace fpsg
