// knn_stream.hip -- K3, streaming form: the k-nearest-neighbour graph of DGCNN's EdgeConv for gfx950.
// Replaces `knn` of reference src/dgcnn/model.py:13-20 (torch.matmul of x^T x into a [B,N,N] tensor + torch.topk).
//
// A workgroup (8 waves) owns 128 query points of one cloud, a wave 16 of them for the whole sweep.  The cloud's
// candidates stream ONCE per workgroup through LDS in stages (a 16 KB tile of k-interleaved point-major features, two
// buffers, one barrier per stage); every wave reads its candidate operands from there (conflict-free 16-byte reads:
// the 16-byte units of a candidate's channel segment are XOR-swizzled by the candidate number) and keeps its 16
// queries in registers.  The 16 x 16 block of scores  pd_ij = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  of a tile is produced
// by v_mfma_f32_16x16x4_f32 (exact channel-ordered fma chain = the oracle's) with the CANDIDATES as the A operand:
// lane (g, c) then holds the scores of candidates 4g..4g+3 against query c -- a lane serves ONE row, so a row's
// threshold, slot counter and write position are per-lane values and one LDS atomic serves a lane's 8 scores of a
// tile pair.  Nothing of size N is kept per row:
//   * a score that reaches its row's threshold T is appended (score bits, index) to the row's 124-entry buffer in
//     LDS;
//   * T is a lower bound of the row's k-th best score, raised by events: the 16 lanes of a DPP row split a row's
//     buffered entries, each finds the two best of its share (v_max / v_med3); if at least ceil(k/2) lanes hold a
//     second-best >= w, then k buffered scores are >= w.  The largest such w comes from 15 DPP row rotations on
//     31-bit keys (subtract + v_alignbit collect the comparison bits without touching a scalar register), and the
//     same pass drops the entries below the new T (4 rows at a time).  Events run at fixed tiles (x2.25 in the number
//     of candidates seen: five per 2048-point cloud) and whenever a buffer passes its watermark.  Between two checks
//     (2 tiles) a row receives at most 32 entries, and a check leaves at most 92: a slot index cannot pass the
//     buffer's end;
//   * after the sweep a row's buffer holds every score >= its final T (about 1.4 k of them): they are ranked by
//     counting on 64-bit keys (orderable score << 32 | ~index: score descending, then index ascending -- the order of
//     the oracle's rounds), four rows at a time (16 lanes and two entries per lane for a row), and the first k are
//     written.
// If a compaction cannot bring a buffer under the watermark (a loose early threshold, hundreds of equal scores), the
// row's entries are ranked on the spot, its k best stay and the k-th score becomes its threshold.  A second, independent
// selection is kept for tests (FPSG_KNN_FORCE_SLOW): k masked arg-max sweeps over the cloud straight from global
// memory.  Results are bit-identical to oracle_knn either way.
#include "knn_internal.h"

namespace fpsg {
namespace {

constexpr int kSW = 8;                 // waves per workgroup
constexpr int kSRows = 16 * kSW;       // query rows per workgroup
constexpr int kCap = 124;              // entries per row buffer (row stride 248 dwords: the filter reads of two DPP rows tile the 64 banks)
constexpr int kWM = kCap - 32;         // a check leaves at most this many entries in a buffer
constexpr int kFilterIters = (kCap + 15) / 16;

template <int C4T> struct StreamCfg;
template <> struct StreamCfg<1> { static constexpr int TC = 512; };    //  8 KB per stage
template <> struct StreamCfg<16> { static constexpr int TC = 64; };    // 16 KB
template <> struct StreamCfg<32> { static constexpr int TC = 32; };    // 16 KB

template <int C4T>
constexpr size_t stream_lds_bytes() {
  constexpr int TC = StreamCfg<C4T>::TC;
  return (size_t)2 * TC * 4 * C4T * 4 + (size_t)2 * TC * 4 + (size_t)kSRows * 4 + (size_t)kSRows * 4 +
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
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i0(int v) {           // lanes without a source receive 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {     // every lane of a row of 16 receives the maximum
  v = max(v, dpp_u<0xB1>(v));     // quad_perm [1,0,3,2]
  v = max(v, dpp_u<0x4E>(v));     // quad_perm [2,3,0,1]
  v = max(v, dpp_u<0x141>(v));    // row_half_mirror
  v = max(v, dpp_u<0x140>(v));    // row_mirror
  return v;
}
// over the four lanes c, c + 16, c + 32, c + 48 (one per row of 16): every one receives the maximum
__device__ __forceinline__ unsigned long long col4_max_u64(unsigned long long v) {
#define FPSG_STEP(M)                                                                                      \
  {                                                                                                       \
    const unsigned long long o = ((unsigned long long)lane_xor<M>((unsigned)(v >> 32)) << 32) | lane_xor<M>((unsigned)v); \
    v = o > v ? o : v;                                                                                    \
  }
  FPSG_STEP(16) FPSG_STEP(32)
#undef FPSG_STEP
  return v;
}
// Bit n-1 of the result (n = 1..15): the key n lanes to the right in this row of 16 (row_ror:n) is SMALLER than this
// lane's.  Keys below 2^31, so the sign of the difference decides; v_alignbit shifts it into the collection: two vector
// instructions per rotation and no scalar register in the chain (a compare + add-with-carry costs wait states between
// the two on gfx950).
template <int N>
__device__ __forceinline__ unsigned row16_less_bits(unsigned key, unsigned acc) {
  if constexpr (N < 16) {
    const unsigned d = dpp_u<0x120 + N>(key) - key;                                   // row_ror:N
    return row16_less_bits<N + 1>(key, __builtin_amdgcn_alignbit(acc, d, 31));        // (acc << 1) | (d >> 31)
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
__device__ __forceinline__ float knn_unorderable(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
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
  unsigned* cnt = reinterpret_cast<unsigned*>(sxx + 2 * TC);    // [kSRows]: entries in a row's buffer
  float* trow = reinterpret_cast<float*>(cnt + kSRows);         // [kSRows]: a row's threshold
  Entry* buf = reinterpret_cast<Entry*>(trow + kSRows);         // [kSRows][kCap]

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

  // query operands (the MFMA's B side): lane (kk, col) holds channels 4 c4 + kk of query i0 + col
  float a[C4T];
  const int qrow = i0 + col;
  {
    if constexpr (C4T >= 4) {
      const v4f* src = reinterpret_cast<const v4f*>(xkb + (size_t)(qrow < N ? qrow : 0) * CP + kk * C4T);
#pragma unroll
      for (int m = 0; m < C4T / 4; ++m) {
        v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (qrow < N) v = src[m];
        a[4 * m] = v.x; a[4 * m + 1] = v.y; a[4 * m + 2] = v.z; a[4 * m + 3] = v.w;
      }
    } else {
      a[0] = qrow < N ? xkb[(size_t)qrow * CP + kk] : 0.0f;
    }
  }
  const float xxq = qrow < N ? xxb[qrow] : 0.0f;
  const int lrow = wave * 16 + col;                             // the row this lane's accumulators belong to
  const int erow0 = wave * 16 + 4 * kk;                         // the rows this lane's DPP row of 16 handles in events
  if (lane < 16) { cnt[wave * 16 + lane] = 0u; trow[wave * 16 + lane] = NEG; }
  unsigned* const my_cnt = cnt + lrow;
  Entry* const my_buf = buf + lrow * kCap;

  float T = NEG;
  const unsigned jsel = (unsigned)((k + 1) >> 1);
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

  auto lds_sync_wave = [&]() {           // one wave's LDS operations execute in order; this orders them for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // ---- selection pieces ------------------------------------------------------------------------------------------
  // An event raises the rows' thresholds and drops the buffered entries below them.  Lane (kk, col) works on rows
  // erow0 + rr: the 16 lanes of its DPP row split a row's entries (lane col holds entries col, col + 16, ...).
  auto raise_and_filter = [&]() {
    lds_sync_wave();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = erow0 + rr;
      const int c = (int)cnt[row];
      const float told = trow[row];
      Entry* rb = buf + row * kCap;
      Entry e[kFilterIters];
      float sc[kFilterIters];
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) e[i] = rb[col + 16 * i < c ? col + 16 * i : 0];
      float a1 = NEG, a2 = NEG;                     // the two best of this lane's share
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) {
        sc[i] = col + 16 * i < c ? __uint_as_float(e[i].s) : NEG;
        a2 = __builtin_amdgcn_fmed3f(a1, a2, sc[i]);
        a1 = raw_max(a1, sc[i]);
      }
      // 31-bit key of the second best (rounded down: a bound that is a hair lower is still a bound)
      const unsigned key = knn_orderable(a2 + 0.0f) >> 1;
      const unsigned less = row16_less_bits<1>(key, 0u);
      const unsigned ge = 16u - (unsigned)__builtin_popcount(less);         // lanes of the row with key >= mine
      const unsigned best = row16_max_u32(ge >= jsel ? key : 0u);
      // (the key of -inf, rounded down, would decode to a NaN: a row whose qualifying lanes hold less than two entries
      // keeps its threshold)
      const float tnew = raw_max(told, best > 0x003fffffu ? knn_unorderable(best << 1) : NEG);
      int keep[kFilterIters];
      int n = 0;
#pragma unroll
      for (int i = 0; i < kFilterIters; ++i) {
        keep[i] = sc[i] >= tnew ? 1 : 0;            // sc = -inf where the lane has no entry; tnew > -inf then
        keep[i] = (col + 16 * i < c) ? keep[i] : 0;
        n += keep[i];
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
        if (keep[i]) rb[pos] = e[i];
        pos += keep[i];
      }
      if (col == 15) { cnt[row] = (unsigned)inc; trow[row] = tnew; }
    }
    lds_sync_wave();
  };
  // The buffered entries of four rows -> 64-bit keys in place and every entry's rank among its row's: the 16 lanes of
  // a DPP row take row rbase + 4 kk + (their register index), lane col entries col and col + 16 (c <= 32).
  // A row with more entries is ranked by the whole wave (rank_row_wide).
  auto rank_row_wide = [&](int row, int c, Entry& e0, Entry& e1, unsigned long long& k0, unsigned long long& k1,
                           int& r0, int& r1) {
    unsigned long long* rk = reinterpret_cast<unsigned long long*>(buf) + (size_t)row * kCap;
    const bool h0 = lane < c, h1 = lane + 64 < c;
    e0 = buf[row * kCap + (h0 ? lane : 0)];
    e1 = buf[row * kCap + (h1 ? lane + 64 : 0)];
    k0 = h0 ? knn_key(__uint_as_float(e0.s), e0.j) : 0ull;
    k1 = h1 ? knn_key(__uint_as_float(e1.s), e1.j) : 0ull;
    __builtin_amdgcn_wave_barrier();
    if (h0) rk[lane] = k0;                           // a lane rewrites only its own slots
    if (h1) rk[lane + 64] = k1;
    lds_sync_wave();
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
      rank_row_wide(row, c, e0, e1, k0, k1, r0, r1);
      const bool h0 = lane < c, h1 = lane + 64 < c;
      Entry* rb = buf + row * kCap;
      if (h0 && r0 < k) rb[r0] = e0;
      if (h1 && r1 < k) rb[r1] = e1;
      if (h0 && r0 == k - 1) trow[row] = raw_max(trow[row], __uint_as_float(e0.s));
      if (h1 && r1 == k - 1) trow[row] = raw_max(trow[row], __uint_as_float(e1.s));
      if (lane == 0) cnt[row] = (unsigned)k;
      lds_sync_wave();
    }
  };
  auto over_watermark = [&]() -> bool {
    const unsigned c = cnt[wave * 16 + col];
    return __builtin_amdgcn_ballot_w64(c > (unsigned)kWM) != 0ull;
  };
  auto event = [&]() {
    raise_and_filter();
    if (over_watermark()) exact_compact();
    T = trow[lrow];
  };
  // The scores of tiles t, t + 1: this lane's 8 scores of ITS row (candidates 16 t + 4 kk + r and + 16) -> threshold
  // test and append.  One LDS atomic takes the lane's slots.  Returns whether a row passed its watermark: the lane that
  // took a row's last slot knows the row's new count.
  auto select_pair = [&](const v4f& acc0, const v4f& acc1, int t, const v4f& xx0, const v4f& xx1) -> bool {
    const unsigned j0 = (unsigned)(16 * t + 4 * kk);
    float v[8];
    bool p[8];
    {   // two scores per packed instruction (v_pk_fma_f32 / v_pk_add_f32): the same fma and subtraction per element
      const v2f two = {2.0f, 2.0f}, nq = {-xxq, -xxq};
      const v2f a = fma_rn(two, (v2f){acc0[0], acc0[1]}, -(v2f){xx0[0], xx0[1]}) + nq;
      const v2f b2 = fma_rn(two, (v2f){acc0[2], acc0[3]}, -(v2f){xx0[2], xx0[3]}) + nq;
      const v2f c = fma_rn(two, (v2f){acc1[0], acc1[1]}, -(v2f){xx1[0], xx1[1]}) + nq;
      const v2f d = fma_rn(two, (v2f){acc1[2], acc1[3]}, -(v2f){xx1[2], xx1[3]}) + nq;
      v[0] = a.x; v[1] = a.y; v[2] = b2.x; v[3] = b2.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
    }
    unsigned n = 0u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      p[r] = v[r] >= T;
      n += p[r] ? 1u : 0u;
    }
    unsigned pos = 0u;
    if (n != 0u) pos = __hip_atomic_fetch_add(my_cnt, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const bool over = n != 0u && pos + n > (unsigned)kWM;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (p[r]) {
        Entry en;
        en.s = __float_as_uint(v[r]);
        en.j = j0 + (unsigned)(r < 4 ? r : r + 12);
        my_buf[pos] = en;
      }
      pos += p[r] ? 1u : 0u;
    }
    return __builtin_amdgcn_ballot_w64(over) != 0ull;
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
          for (int e = 0; e < 4; ++e) {             // candidates = A (rows of D), queries = B (columns of D)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[e], a[4 * m + e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[e], a[4 * m + e], acc1, 0, 0, 0);
          }
        }
      } else {
        const float b0 = st[(16 * tt + col) * 4 + kk];
        const float b1 = st[(16 * tt + 16 + col) * 4 + kk];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, a[0], acc1, 0, 0, 0);
      }
      if (!failed) {
        const v4f xx0 = *reinterpret_cast<const v4f*>(sxx + sel * TC + 16 * tt + 4 * kk);
        const v4f xx1 = *reinterpret_cast<const v4f*>(sxx + sel * TC + 16 * tt + 16 + 4 * kk);
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
    // ---- rank the survivors: four rows at a time, 16 lanes and two entries per lane for a row ---------------------
    unsigned long long* kb = reinterpret_cast<unsigned long long*>(buf);
    for (int rr = 0; rr < 4; ++rr) {
      const int row = erow0 + rr;
      const int i = i0 + 4 * kk + rr;
      const int c = (int)cnt[row];
      const bool wide = __builtin_amdgcn_ballot_w64(c > 32) != 0ull;        // some row of the four: the wave-wide form
      if (!wide) {
        unsigned long long* rk = kb + (size_t)row * kCap;
        const bool h0 = col < c, h1 = col + 16 < c;
        const Entry e0 = buf[row * kCap + (h0 ? col : 0)];
        const Entry e1 = buf[row * kCap + (h1 ? col + 16 : 0)];
        const unsigned long long k0 = h0 ? knn_key(__uint_as_float(e0.s), e0.j) : 0ull;
        const unsigned long long k1 = h1 ? knn_key(__uint_as_float(e1.s), e1.j) : 0ull;
        __builtin_amdgcn_wave_barrier();
        if (h0) rk[col] = k0;
        if (h1) rk[col + 16] = k1;
        lds_sync_wave();
        int cmax = __builtin_amdgcn_readlane(c, 0);
        cmax = max(cmax, __builtin_amdgcn_readlane(c, 16));
        cmax = max(cmax, __builtin_amdgcn_readlane(c, 32));
        cmax = max(cmax, __builtin_amdgcn_readlane(c, 48));
        int r0 = 0, r1 = 0;
#pragma unroll 4
        for (int l = 0; l < cmax; ++l) {
          const unsigned long long o = l < c ? rk[l] : 0ull;      // a shorter row compares with key 0: no rank change
          r0 += o > k0 ? 1 : 0;
          r1 += o > k1 ? 1 : 0;
        }
        if (i < N) {
          int32_t* out = idx + ((size_t)b * N + i) * k;
          if (h0 && r0 < k) out[r0] = (int32_t)~(unsigned)k0;
          if (h1 && r1 < k) out[r1] = (int32_t)~(unsigned)k1;
        }
      } else {
        for (int g = 0; g < 4; ++g) {
          const int wrow = wave * 16 + 4 * g + rr;
          const int wi = i0 + 4 * g + rr;
          if (wi >= N) continue;
          const int wc = __builtin_amdgcn_readfirstlane((int)cnt[wrow]);
          Entry e0, e1;
          unsigned long long k0, k1;
          int r0, r1;
          rank_row_wide(wrow, wc, e0, e1, k0, k1, r0, r1);
          int32_t* out = idx + ((size_t)b * N + wi) * k;
          if (lane < wc && r0 < k) out[r0] = (int32_t)~(unsigned)k0;
          if (lane + 64 < wc && r1 < k) out[r1] = (int32_t)~(unsigned)k1;
        }
      }
    }
    return;
  }

  // ---- slow exact path: k masked arg-max sweeps, operands straight from global memory ------------------------------
  unsigned long long prev = ~0ull;
  for (int round = 0; round < k; ++round) {
    unsigned long long best = 0ull;
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
      const int j = 16 * t + col;                    // the candidate whose features this lane feeds to the MFMA
      const bool jin = j < N;
      v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (C4T >= 4) {
        const v4f* src = reinterpret_cast<const v4f*>(xkb + (size_t)(jin ? j : 0) * CP + kk * C4T);
#pragma unroll
        for (int m = 0; m < C4T / 4; ++m) {
          v4f bv = {0.0f, 0.0f, 0.0f, 0.0f};
          if (jin) bv = src[m];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[e], a[4 * m + e], acc, 0, 0, 0);
        }
      } else {
        const float bv = jin ? xkb[(size_t)j * CP + kk] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, a[0], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jc = 16 * t + 4 * kk + r;          // the candidate of accumulator r
        const float xxj = jc < N ? xxb[jc] : __builtin_inff();
        const float v = fma_rn(2.0f, acc[r], -xxj) - xxq;
        const unsigned long long key = knn_key(v, (unsigned)jc);
        const bool take = key < prev && key > best;
        best = take ? key : best;
      }
    }
    const unsigned long long w = col4_max_u64(best);
    prev = w;
    if (kk == 0 && qrow < N) idx[((size_t)b * N + qrow) * k + round] = (int32_t)~(unsigned)w;
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
