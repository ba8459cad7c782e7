// gemm_split.hip -- K10: batched fp32 GEMM on the bf16 matrix pipe with three-way split operands (opt-in, round 5).
//
//   C[b] [M x N] = A[b] [M x K] * B[b] [K x N]        (transB = 0: the Winograd-domain products U[xi] V[xi] of K6's
//                                                       forward / data gradient, B's rows N-contiguous)
//   C[b] [M x N] = A[b] [M x K] * B[b]^T, B [N x K]   (transB = 1: the weight gradient's dM[xi] V[xi]^T, reduction
//                                                       along the contiguous dimension of both operands, split over
//                                                       workgroups with a fixed-order reduction of the partial slabs)
//
// Arithmetic.  The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate.  Every fp32 operand is written
// as x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to nearest even; the
// subtractions are exact, the three pieces carry 3 x 9 >= 24 significant bits, so the split is exact but for overflow of
// bf16(x) beyond 3.39e38).  The six products of order <= 2^-18 -- a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1, bf16 x bf16 is
// exact in fp32 -- go through v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator, smallest first inside a k-step; the
// dropped a2 b3 + a3 b2 + a3 b3 are below 2^-26 of |a b|, a quarter of the fp32 product's own rounding.  The
// accumulation is fp32 with one rounding per MFMA (16 products) instead of one per product.  Measured error against
// float64 products: tools/bench_gemm_split.py, profiles/r05/.
//
// Data path.  The operands are split on the way from global memory into LDS (no bf16 planes in HBM): a thread loads 8
// consecutive k of one row (A, and B when transB) or 8 rows of one column (B, N-contiguous: eight coalesced dword
// loads), forms the three bf16x8 pieces in registers (5.5 vector instructions per element, v_cvt_pk_bf16_f32) and stores
// them as three 16-byte LDS writes: image [piece][k-group of 8][row] x 16 B, plane stride padded so that the writes of
// a group of 8 lanes fall into 8 different 16-byte bank slots; a lane's MFMA fragment is one ds_read_b128 and 32 lanes
// read 512 contiguous bytes (conflict-free).  Two stages in LDS, one barrier per k-step: global loads of step k+1 are
// issued before the MFMAs of step k, converted and written after them.
//
// Work split: 512 threads = 8 waves as 2 x 4 (rows x columns), one 32x32 accumulator block per (32 rows, 32 columns) of
// the wave's share; workgroup tile 256 x 256 x 16 (variant 0) or 256 x 128 x 32 (variant 1).  Workgroups that share an
// XCD (blockIdx mod 8) get consecutive tiles of one batch (they share A[b] through that XCD's L2).
#include "fpsg_common.h"

namespace fpsg {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


struct GemmArgs {
  const float* A;
  const float* B;
  float* C;                  // the output, or the partial slabs [split][batch][M][N] when splits > 1
  int M, N, K;
  int lda, ldb, ldc;
  long sA, sB, sC;           // batch strides (floats)
  int tiles_m, tiles_n, splits, k_per_split;
  long s_split;              // floats between two splits' slabs
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE): low half = a
}

// x[0..7] -> three bf16x8 pieces (element j in 16-bit slot j)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    const unsigned h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const unsigned m = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    p1[q] = h;
    p2[q] = m;
    p3[q] = cvt_pk_bf16(sa, sb);
  }
}

// s_setprio takes an immediate: a wave-uniform switch
__device__ __forceinline__ void set_wave_priority(int p) {
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

constexpr unsigned kOut = 0x7fffffffu;     // a lane offset beyond every buffer: the load returns 0, the store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// One operand tile of R rows x BK k, K-contiguous in memory ([rows][K]): item = (row, k-group), 8 consecutive k per
// item as two 16-byte buffer loads: descriptor + the lane's fixed byte offset + the step's scalar offset -- no address
// arithmetic in the loop; a row beyond the matrix has an offset beyond the buffer and reads zeros.
template <int R, int BK, int kThreads>
struct ContigStage {
  static constexpr int KG = BK / 8;
  static constexpr int PS = R * 16 + 128 / KG;          // plane stride (bytes)
  static constexpr int BYTES = 3 * KG * PS;
  static constexpr int ITEMS = (R * KG + kThreads - 1) / kThreads;
  static constexpr bool PARTIAL = R * KG % kThreads != 0;     // fewer items than threads: the upper threads carry none
  static_assert(!PARTIAL || ITEMS == 1, "tile / thread mismatch");
  float v[ITEMS][8];
  unsigned off[ITEMS];                                   // byte offset of the item's first element at k = 0

  __device__ __forceinline__ void init(int ld, int row0, int rows, int tid) {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      const int row = row0 + it / KG;
      off[n] = (row < rows && it < R * KG) ? ((unsigned)row * (unsigned)ld + (it % KG) * 8) * 4u : kOut;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int k0) {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const v4f q = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off[n] + 16 * hh, k0 * 4, 0));
        v[n][4 * hh] = q.x; v[n][4 * hh + 1] = q.y; v[n][4 * hh + 2] = q.z; v[n][4 * hh + 3] = q.w;
      }
  }
  // the last step of a reduction that is no multiple of BK: what was read beyond k_end (the next row's elements, or
  // zeros beyond the buffer) is zeroed in the registers -- a branch around selects, never around a load
  __device__ __forceinline__ void mask_tail(int k0, int k_end, int tid) {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int k = k0 + ((tid + n * kThreads) % KG) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[n][j] = k + j < k_end ? v[n][j] : 0.f;
    }
  }
  __device__ __forceinline__ void split_only() {        // ablation builds: the arithmetic kept alive, nothing stored
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      u32x4 p1, p2, p3;
      split8(v[n], p1, p2, p3);
      asm volatile("" ::"v"(p1), "v"(p2), "v"(p3));
      v[n][0] = __uint_as_float(p3[0] ^ p2[1]);
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      if (PARTIAL && it >= R * KG) break;
      const int row = it / KG, g = it % KG;
      u32x4 p1, p2, p3;
      split8(v[n], p1, p2, p3);
      unsigned char* d = lds + g * PS + row * 16;
      *reinterpret_cast<u32x4*>(d) = p1;
      *reinterpret_cast<u32x4*>(d + KG * PS) = p2;
      *reinterpret_cast<u32x4*>(d + 2 * KG * PS) = p3;
    }
  }
};

// One operand tile of BK k x R columns, column-contiguous in memory ([K][N]): item = (column, k-group), a lane reads its
// column of 8 consecutive rows (each wave-instruction is one coalesced 256-byte row segment; the row enters as the
// scalar offset).  Rows beyond K lie beyond the buffer (zeros); columns beyond N get the out-of-buffer lane offset.
template <int R, int BK, int kThreads>
struct StridedStage {
  static constexpr int KG = BK / 8;
  static constexpr int PS = R * 16 + 128 / KG;
  static constexpr int BYTES = 3 * KG * PS;
  static constexpr int ITEMS = (R * KG + kThreads - 1) / kThreads;
  static constexpr bool PARTIAL = R * KG % kThreads != 0;
  static_assert(!PARTIAL || ITEMS == 1, "tile / thread mismatch");
  float v[ITEMS][8];
  unsigned off[ITEMS];
  unsigned ld4;

  __device__ __forceinline__ void init(int ld, int col0, int cols, int tid) {
    ld4 = (unsigned)ld * 4u;
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      const int col = col0 + it % R;
      off[n] = (col < cols && it < R * KG) ? (unsigned)col * 4u + (unsigned)(it / R) * 8u * ld4 : kOut;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int k0) {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[n][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off[n], (unsigned)(k0 + j) * ld4, 0));
  }
  __device__ __forceinline__ void mask_tail(int, int, int) {}      // rows beyond K lie beyond the buffer: zeros already
  __device__ __forceinline__ void split_only() {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      u32x4 p1, p2, p3;
      split8(v[n], p1, p2, p3);
      asm volatile("" ::"v"(p1), "v"(p2), "v"(p3));
      v[n][0] = __uint_as_float(p3[0] ^ p2[1]);
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      if (PARTIAL && it >= R * KG) break;
      const int col = it % R, g = it / R;
      u32x4 p1, p2, p3;
      split8(v[n], p1, p2, p3);
      unsigned char* d = lds + g * PS + col * 16;
      *reinterpret_cast<u32x4*>(d) = p1;
      *reinterpret_cast<u32x4*>(d + KG * PS) = p2;
      *reinterpret_cast<u32x4*>(d + 2 * KG * PS) = p3;
    }
  }
};

// WR x WC waves (rows x columns of the tile); MINW: waves per SIMD the register allocation must leave room for
// ABL (measurements only, results wrong): 1 = no global loads in the loop, 2 = nor LDS stores, 3 = nor the split
template <int BM, int BN, int BK, int WR, int WC, int MINW, bool TRANSB, bool PRIO, bool PIPE = false, int ABL = 0>
__global__ __launch_bounds__(64 * WR * WC, MINW) void gemm_split_kernel(const GemmArgs g) {
  constexpr int kThreads = 64 * WR * WC;
  constexpr int WM = BM / WR, WN = BN / WC, TI = WM / 32, TJ = WN / 32;
  constexpr int KG = BK / 8;
  using StageA = ContigStage<BM, BK, kThreads>;
  using StageB = typename std::conditional<TRANSB, ContigStage<BN, BK, kThreads>, StridedStage<BN, BK, kThreads>>::type;
  constexpr int PSA = StageA::PS, PSB = StageB::PS;
  constexpr int STAGE = StageA::BYTES + StageB::BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  // consecutive tiles to the workgroups of one XCD (blockIdx mod 8 names the XCD's workgroups)
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb / 8, r8 = nb % 8, x8 = bid % 8;
  int id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + bid / 8;
  const int tm = id % g.tiles_m; id /= g.tiles_m;
  const int tn = id % g.tiles_n; id /= g.tiles_n;
  const int split = id % g.splits;
  const int batch = id / g.splits;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too (scalar offsets, no waterfall)
  // Co-resident waves of one SIMD run the same program; under the default (round-robin) arbitration they fall into
  // lock-step -- all in their MFMA block, then all in their conversion block with the matrix pipe idle (measured:
  // SQ_WAIT_INST_ANY = 3x the MFMA busy time, pipe 46 % busy).  Distinct static priorities make the arbitration strict:
  // the higher wave runs its MFMA block at full rate and the others fill the pipe while it converts.
  if (PRIO) set_wave_priority(WR * WC >= 8 ? (2 * (bid & 1) + (wave >= WR * WC / 2 ? 1 : 0)) : bid % 3);
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A + batch * g.sA, (long)(g.M - 1) * g.lda + g.K);
  const __amdgpu_buffer_rsrc_t rsB =
      make_rsrc(g.B + batch * g.sB, TRANSB ? (long)(g.N - 1) * g.ldb + g.K : (long)(g.K - 1) * g.ldb + g.N);

  StageA ra;
  StageB rb;
  ra.init(g.lda, m0, g.M, tid);
  rb.init(g.ldb, n0, g.N, tid);
  auto load = [&](int kt) {                             // stage kt -> registers
    const int k0 = k_begin + kt * BK;
    if (ABL >= 1 && kt > 1) return;
    ra.load(rsA, k0);
    rb.load(rsB, k0);
  };
  auto store = [&](int kt) {                            // registers -> split -> LDS stage kt & 1
    const int k0 = k_begin + kt * BK;
    if (ABL >= 3 && kt > 1) return;
    if (k0 + BK > k_end) {                              // workgroup-uniform
      ra.mask_tail(k0, k_end, tid);
      rb.mask_tail(k0, k_end, tid);
    }
    unsigned char* d = lds + (kt & 1) * STAGE;
    if (ABL == 2 && kt > 1) {                           // the split's arithmetic without its LDS stores
      ra.split_only();
      rb.split_only();
      return;
    }
    ra.store(d, tid);
    rb.store(d + StageA::BYTES, tid);
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int wm0 = (wave / WC) * WM, wn0 = (wave % WC) * WN;
  const int a_off = h * PSA + (wm0 + r) * 16;          // + (t*KG + 2s) * PSA + i * 512
  const int b_off = StageA::BYTES + h * PSB + (wn0 + r) * 16;
  auto compute = [&](int kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      // The B fragments of the k-step stay in registers; the A fragments come per block of 32 rows, the next block's
      // three reads issued before this block's MFMAs.  sched_barrier(0) pins that order: left alone, the scheduler
      // hoists every fragment read of the step to its top (3 x (TI + TJ) fragments live: spills beyond 256 registers
      // at 128 x 64 per wave).
      bf16x8 fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, bf16x8 (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + a_off + (t * KG + 2 * s) * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + b_off + (t * KG + 2 * s) * PSB + j * 512));
      read_a(0, fa[0]);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (i + 1 < TI) read_a(i + 1, fa[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // Stage kt+1 is split and written (into the buffer step kt-1 read: the barrier behind it has passed) around the
  // products of stage kt; its global loads were issued one step earlier, those of the next stage follow into the freed
  // registers.  With two waves per SIMD (w and w+4) the halves are staggered: waves 4-7 convert behind their MFMAs
  // (before the barrier), waves 0-3 behind the barrier -- i.e. in front of the next step's MFMAs -- so that one wave's
  // vector work runs beside its partner's MFMAs instead of both idling the matrix pipe together.  One copy of the
  // product code: the two orders differ only in where the (small) conversion block sits.
  if constexpr (PIPE) {
    // One wave's stream carries everything at once: the products of stage kt with the split of stage kt+1 (5.5 vector
    // instructions per element), its LDS stores and the global loads of stage kt+2 placed in the shadows of the MFMAs
    // (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles).  Measured before: with the conversion as a block
    // behind the MFMA block, co-resident waves fell into step and the matrix pipe idled half the time
    // (profiles/r05/pmc_gemm_split_v3.txt: SQ_WAIT_INST_ANY 3x the MFMA time, pipe 46 % busy) whatever the occupancy.
    // The main loop body is one basic block (no workgroup-uniform branches: the last two steps are peeled, a partial
    // last stage is masked there); sched_group_barrier lays out the order.
    static_assert(BK == 16, "pipelined form: one 16-deep k-step per barrier");
    auto step = [&](int kt, auto store_c, auto load_c, auto mask_c) {
      constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value, MASK = decltype(mask_c)::value;
      const unsigned char* cur = lds + (kt & 1) * STAGE;
      unsigned char* nxt = lds + ((kt + 1) & 1) * STAGE;
      bf16x8 fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, bf16x8 (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + a_off + t * KG * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + b_off + t * KG * PSB + j * 512));
      read_a(0, fa[0]);
      if (TI > 1) read_a(1, fa[1]);
      if constexpr (STORE && MASK) {
        const int k0 = k_begin + (kt + 1) * BK;
        ra.mask_tail(k0, k_end, tid);
        rb.mask_tail(k0, k_end, tid);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const bf16x8 (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
        if (i + 2 < TI) read_a(i + 2, fa[i & 1]);
      }
      if constexpr (STORE) {
        ra.store(nxt, tid);
        rb.store(nxt + StageA::BYTES, tid);
      }
      if constexpr (LOAD) {
        const int k0 = k_begin + (kt + 2) * BK;
        ra.load(rsA, k0);
        rb.load(rsB, k0);
      }
      // the order: fragment reads of the first two row blocks, then per row block its 6 * TJ MFMAs, each followed by
      // vector instructions of the split; the later row blocks' reads ride behind their predecessors; LDS stores and
      // global loads (whose registers the split has just released) in the last row block's shadows
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * TJ + (TI > 1 ? 6 : 3), 0);
      constexpr int PER = 6 * TJ;                          // MFMAs per row block
      constexpr int VAL = STORE ? (TI >= 4 ? 3 : 4) : 0;   // vector instructions behind each MFMA of the early blocks
#pragma unroll
      for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i + 1 < TI || TI == 1) {
            if (VAL) __builtin_amdgcn_sched_group_barrier(0x002, VAL, 0);
          } else {
            if (VAL) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            if (STORE && q % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (LOAD && q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          }
        }
        if (i + 2 < TI) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    load(0);
    store(0);
    if (nsteps > 1) load(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nsteps; ++kt) {
      step(kt, T{}, T{}, F{});
      __syncthreads();
    }
    if (kt + 1 < nsteps) {                                 // stage nsteps-1 may be partial; nothing left to load
      step(kt, T{}, F{}, T{});
      __syncthreads();
      ++kt;
    }
    step(kt, F{}, F{}, F{});
  } else {
  const bool early = WR * WC >= 8 && wave < WR * WC / 2;       // converts one step ahead, behind the barrier
  load(0);
  store(0);
  if (nsteps > 1) load(1);
  __syncthreads();
  if (early) {
    if (nsteps > 1) store(1);
    if (nsteps > 2) load(2);
  }
  for (int kt = 0; kt < nsteps; ++kt) {
    compute(kt);
    if (!early) {
      if (kt + 1 < nsteps) store(kt + 1);
      if (kt + 2 < nsteps) load(kt + 2);
    }
    __syncthreads();
    if (early) {
      if (kt + 2 < nsteps) store(kt + 2);
      if (kt + 3 < nsteps) load(kt + 3);
    }
  }
  }

  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); buffer stores:
  // a row beyond M lies beyond the buffer, a column beyond N gets the out-of-buffer lane offset (dropped)
  const __amdgpu_buffer_rsrc_t rsC =
      make_rsrc(g.C + split * g.s_split + batch * g.sC, (long)(g.M - 1) * g.ldc + g.N);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn0 + 32 * j + r;
    const unsigned voff = col < g.N ? ((unsigned)col + (unsigned)(4 * h) * g.ldc) * 4u : kOut;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][e]), rsC, voff, (unsigned)row * g.ldc * 4u, 0);
      }
  }
}

// ---- forward / data-gradient form with the A operand (the transformed filters: small, constant over an optimizer step)
// split ONCE into three bf16 planes that have the layout of the LDS image: [batch][row tile][stage][piece][k-group]
// [BM rows] x 16 B, zero-padded in rows and k.  The GEMM then brings A in by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KB
// per wave instruction, fully coalesced, no vector instruction, no LDS store) and only the streamed B operand (the
// transformed activations, read once from HBM) is split on the way into LDS.  Half the conversion work, LDS stores and
// vector-memory instructions of the generic kernel, whose units were all 60-90 % busy.
template <int BM, int BK>
__global__ __launch_bounds__(256) void gemm_split_pack_a_kernel(const float* __restrict__ A, int M, int K, int lda, long sA,
                                                                int tiles_m, int stages, u32x4* __restrict__ out, long items) {
  constexpr int KG = BK / 8;
  const long it = (long)blockIdx.x * 256 + threadIdx.x;       // one item = 8 consecutive k of one padded row
  if (it >= items) return;
  const int row_in = (int)(it % BM);
  long r = it / BM;
  const int g = (int)(r % KG); r /= KG;
  const int stage = (int)(r % stages); r /= stages;
  const int tm = (int)(r % tiles_m);
  const long batch = r / tiles_m;
  const int row = tm * BM + row_in, k0 = stage * BK + g * 8;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (row < M && k0 + j < K) ? A[batch * sA + (long)row * lda + k0 + j] : 0.f;
  u32x4 p1, p2, p3;
  split8(x, p1, p2, p3);
  const long base = (((batch * tiles_m + tm) * stages + stage) * 3) * (long)(KG * BM) + (long)g * BM + row_in;
  out[base] = p1;
  out[base + (long)KG * BM] = p2;
  out[base + 2L * KG * BM] = p3;
}

struct PackedArgs {
  const u32x4* Ap;
  const float* B;
  float* C;
  int M, N, K;
  int ldb, ldc;
  long sB, sC;
  int tiles_m, tiles_n, stages;
};

template <int BM, int BN, int BK, int WR, int WC, int MINW>
__global__ __launch_bounds__(64 * WR * WC, MINW) void gemm_split_pa_kernel(const PackedArgs g) {
  constexpr int NW = WR * WC;
  constexpr int WM = BM / WR, WN = BN / WC, TI = WM / 32, TJ = WN / 32;
  constexpr int KG = BK / 8;
  constexpr int PSA = BM * 16 + 128 / KG;                    // LDS plane strides as in the generic kernel
  constexpr int A_BYTES = 3 * KG * PSA;
  constexpr int kBThreads = BN * KG;                         // one item per thread of the converting waves
  static_assert(kBThreads % 64 == 0 && kBThreads <= 64 * NW, "B tile / wave mismatch");
  constexpr int NWB = kBThreads / 64;                        // waves 0 .. NWB-1 convert B, the others issue the DMAs
  using StageB = StridedStage<BN, BK, kBThreads>;
  constexpr int PSB = StageB::PS;
  constexpr int STAGE = A_BYTES + StageB::BYTES;
  constexpr int CHUNKS = 3 * KG * BM / 64;                   // 1 KB pieces of a packed A stage
  constexpr int NWD = NW - NWB > 0 ? NW - NWB : NW;          // waves that issue DMAs (all, when every wave converts)
  constexpr int DMA_PER_WAVE = (CHUNKS + NWD - 1) / NWD;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q8 = nb / 8, r8 = nb % 8, x8 = bid % 8;
  int id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + bid / 8;
  const int tm = id % g.tiles_m; id /= g.tiles_m;
  const int tn = id % g.tiles_n;
  const int batch = id / g.tiles_n;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  set_wave_priority(NW >= 8 ? (2 * (bid & 1) + (wave >= NW / 2 ? 1 : 0)) : bid % 3);   // see gemm_split_kernel
  const int m0 = tm * BM, n0 = tn * BN;
  const int nsteps = g.stages;
  const long a_tile_u4 = (long)nsteps * 3 * KG * BM;          // 16-byte units of one (batch, row tile)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(g.Ap + ((long)batch * g.tiles_m + tm) * a_tile_u4), 0, (int)(a_tile_u4 * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(g.B + batch * g.sB, (long)(g.K - 1) * g.ldb + g.N);

  const bool converts = wave < NWB;
  const bool dmas = NWB == NW || wave >= NWB;
  const int dwave = NWB == NW ? wave : wave - NWB;
  StageB rb;
  rb.init(g.ldb, n0, g.N, tid);
  auto dma_a = [&](int kt) {                                // packed stage kt -> LDS stage kt & 1, chunk by chunk
    unsigned char* d = lds + (kt & 1) * STAGE;
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) {
      const int c = dwave + q * NWD;                        // wave-uniform
      if (CHUNKS % NWD == 0 || c < CHUNKS) {
        const int plane = c / (BM / 64), r64 = c % (BM / 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(d + plane * PSA + r64 * 1024), 16,
                                                 lane * 16, (kt * CHUNKS + c) * 1024, 0, 0);
      }
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int wm0 = (wave / WC) * WM, wn0 = (wave % WC) * WN;
  const int a_off = h * PSA + (wm0 + r) * 16;
  const int b_off = A_BYTES + h * PSB + (wn0 + r) * 16;
  auto compute = [&](int kt) {
    const unsigned char* cur = lds + (kt & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, bf16x8 (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + a_off + (t * KG + 2 * s) * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + b_off + (t * KG + 2 * s) * PSB + j * 512));
      read_a(0, fa[0]);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (i + 1 < TI) read_a(i + 1, fa[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], fb[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // A(kt+1) is on its way by DMA while stage kt is multiplied; B(kt+1) -- loaded into registers behind the previous
  // barrier -- is split and stored behind the products; the wait in front of the barrier only ever sees the DMA (long
  // landed) and the LDS stores: the register loads of B(kt+2) are issued BEHIND the barrier and have a whole step.
  if (dmas) dma_a(0);
  if (converts) {
    rb.load(rsB, 0);
    rb.store(lds + A_BYTES, tid);
    if (nsteps > 1) rb.load(rsB, BK);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kt = 0; kt < nsteps; ++kt) {
    if (dmas && kt + 1 < nsteps) dma_a(kt + 1);
    compute(kt);
    if (converts && kt + 1 < nsteps) rb.store(lds + ((kt + 1) & 1) * STAGE + A_BYTES, tid);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (converts && kt + 2 < nsteps) rb.load(rsB, (kt + 2) * BK);
  }

  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(g.C + batch * g.sC, (long)(g.M - 1) * g.ldc + g.N);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn0 + 32 * j + r;
    const unsigned voff = col < g.N ? ((unsigned)col + (unsigned)(4 * h) * g.ldc) * 4u : kOut;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][e]), rsC, voff, (unsigned)row * g.ldc * 4u, 0);
      }
  }
}

// out[i] = slab 0 + slab 1 + ... (fixed order), n floats per slab
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long n,
                                                                long s_split, int splits) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = ws[i];
    for (int k = 1; k < splits; ++k) s += ws[i + k * s_split];
    out[i] = s;
  }
}

struct Tile {
  int bm, bn, bk, threads, per_cu;
};
// tile ids of `variant` (rows x columns x k-step, waves as rows x columns, workgroups per CU by LDS / registers)
constexpr Tile kTiles[] = {
    {256, 256, 16, 512, 1},   // 0: 2 x 4 waves of 128 x 64
    {256, 128, 32, 512, 1},   // 1: 2 x 4 waves of 128 x 32
    {128, 128, 16, 256, 3},   // 2: 2 x 2 waves of 64 x 64, 49 KB of LDS: three workgroups per CU overlap each other's
                              //    prologue, conversion and epilogue
    {128, 256, 16, 256, 2},   // 3: 1 x 4 waves of 128 x 64, 74 KB
    {256, 256, 16, 512, 1},   // 4: tile 0 with the split, LDS stores and loads placed among the MFMAs of one stream
    {128, 128, 16, 256, 3},   // 5: tile 2, the same
    {128, 256, 16, 256, 2},   // 6: tile 3, the same
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

struct Plan {
  int tile, bm, bn, bk, tiles_m, tiles_n, splits, k_per_split;
  bool prio;
  int abl;
};

// variant: -1 automatic; else tile + 10 * splits (splits 0 = automatic)
bool plan_for(int batch, int M, int N, int K, int transB, int variant, Plan* p) {
  int tile = variant < 0 ? -1 : variant % 10;
  int splits = variant < 0 ? 0 : (variant / 10) % 100;
  p->prio = variant < 0 || variant < 1000 || variant >= 2000;   // + 1000: without the static wave priorities (A/B)
  p->abl = variant >= 2000 ? variant / 1000 - 1 : 0;             // + 2000 / 3000 / 4000: ablation builds (tiles 0, 2; NN)
  if (tile >= kNumTiles) return false;
  const long cus = 256;
  if (tile < 0) tile = transB ? 3 : 2;                    // measured: tools/bench_gemm_split.py, profiles/r05/
  const Tile& t = kTiles[tile];
  p->tile = tile;
  p->bm = t.bm; p->bn = t.bn; p->bk = t.bk;
  p->tiles_m = (M + p->bm - 1) / p->bm;
  p->tiles_n = (N + p->bn - 1) / p->bn;
  const long tiles = (long)batch * p->tiles_m * p->tiles_n;
  const long slots = cus * t.per_cu;
  if (splits <= 0) {
    splits = 1;
    if (transB && tiles < slots) {                        // long reduction, few tiles: fill the chip with K ranges
      splits = (int)(slots / tiles);
      const int max_splits = (K + 8 * p->bk - 1) / (8 * p->bk);   // at least 8 k-steps per range
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
  }
  const int steps = (K + p->bk - 1) / p->bk;
  const int steps_per = (steps + splits - 1) / splits;
  p->k_per_split = steps_per * p->bk;
  p->splits = (steps + steps_per - 1) / steps_per;         // no empty range
  return true;
}

}  // namespace
}  // namespace fpsg

namespace fpsg {
namespace {
// packed-A tiles: variant 0 = 256 x 128 x 16 (2 x 4 waves of 128 x 32, two workgroups per CU), 1 = 256 x 256 x 16
// (2 x 4 waves of 128 x 64, one per CU), 2 = 128 x 128 x 16 (2 x 2 waves of 64 x 64, three per CU)
struct PaTile { int bm, bn, bk, threads; };
constexpr PaTile kPaTiles[] = {{256, 128, 16, 512}, {256, 256, 16, 512}, {128, 128, 16, 256}};
constexpr int kNumPaTiles = sizeof(kPaTiles) / sizeof(kPaTiles[0]);
inline int pa_tile(int variant) { return variant < 0 ? 0 : variant; }
}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_gemm_split_packed_a_bytes(int batch, int M, int K, int variant) {
  const int t = fpsg::pa_tile(variant);
  if (batch <= 0 || M <= 0 || K <= 0 || t >= fpsg::kNumPaTiles) return 0;
  const fpsg::PaTile& p = fpsg::kPaTiles[t];
  const long tiles_m = (M + p.bm - 1) / p.bm, stages = (K + p.bk - 1) / p.bk;
  return (size_t)batch * tiles_m * stages * 3 * (p.bk / 8) * p.bm * 16;
}

extern "C" int fpsg_gemm_split_pack_a(const float* A, int batch, int M, int K, int lda, long sA, int variant, void* Ap,
                                      fpsg_stream_t stream) {
  using namespace fpsg;
  const int t = pa_tile(variant);
  FPSG_REQUIRE(batch > 0 && M > 0 && K > 0 && lda >= K, FPSG_E_SHAPE, "fpsg_gemm_split_pack_a: bad shape");
  FPSG_REQUIRE(t < kNumPaTiles, FPSG_E_SHAPE, "fpsg_gemm_split_pack_a: unknown variant %d", variant);
  FPSG_REQUIRE_PTR(A);
  FPSG_REQUIRE_PTR(Ap);
  FPSG_REQUIRE((reinterpret_cast<uintptr_t>(Ap) & 15) == 0, FPSG_E_ALIGN, "fpsg_gemm_split_pack_a: Ap must be 16-byte aligned");
  const PaTile& p = kPaTiles[t];
  const int tiles_m = (M + p.bm - 1) / p.bm, stages = (K + p.bk - 1) / p.bk;
  const long items = (long)batch * tiles_m * stages * (p.bk / 8) * p.bm;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((items + 255) / 256)), block(256);
  if (p.bm == 256) hipLaunchKernelGGL((gemm_split_pack_a_kernel<256, 16>), grid, block, 0, s, A, M, K, lda, sA, tiles_m, stages,
                                      static_cast<u32x4*>(Ap), items);
  else hipLaunchKernelGGL((gemm_split_pack_a_kernel<128, 16>), grid, block, 0, s, A, M, K, lda, sA, tiles_m, stages,
                          static_cast<u32x4*>(Ap), items);
  return launch_status("fpsg_gemm_split_pack_a");
}

extern "C" int fpsg_gemm_split_nn_packed(const void* Ap, const float* B, float* C, int batch, int M, int N, int K, int ldb,
                                         int ldc, long sB, long sC, int variant, fpsg_stream_t stream) {
  using namespace fpsg;
  const int t = pa_tile(variant);
  FPSG_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && ldb >= N && ldc >= N, FPSG_E_SHAPE, "fpsg_gemm_split_nn_packed: bad shape");
  FPSG_REQUIRE(t < kNumPaTiles, FPSG_E_SHAPE, "fpsg_gemm_split_nn_packed: unknown variant %d", variant);
  FPSG_REQUIRE_PTR(Ap);
  FPSG_REQUIRE_PTR(B);
  FPSG_REQUIRE_PTR(C);
  FPSG_REQUIRE((long)K * ldb < (1L << 29) && (long)M * ldc < (1L << 29), FPSG_E_LIMIT,
               "fpsg_gemm_split_nn_packed: a matrix of one batch entry must stay below 2 GiB (32-bit buffer offsets)");
  const PaTile& p = kPaTiles[t];
  PackedArgs g;
  g.Ap = static_cast<const u32x4*>(Ap); g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldb = ldb; g.ldc = ldc; g.sB = sB; g.sC = sC;
  g.tiles_m = (M + p.bm - 1) / p.bm; g.tiles_n = (N + p.bn - 1) / p.bn; g.stages = (K + p.bk - 1) / p.bk;
  FPSG_REQUIRE((long)g.stages * 3 * (p.bk / 8) * p.bm * 16 < (1L << 31), FPSG_E_LIMIT, "fpsg_gemm_split_nn_packed: K too long");
  const long blocks = (long)batch * g.tiles_m * g.tiles_n;
  FPSG_REQUIRE(blocks < (1L << 30), FPSG_E_LIMIT, "fpsg_gemm_split_nn_packed: too many tiles");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), block(p.threads);
  switch (t) {
    case 0: hipLaunchKernelGGL((gemm_split_pa_kernel<256, 128, 16, 2, 4, 4>), grid, block, 0, s, g); break;
    case 1: hipLaunchKernelGGL((gemm_split_pa_kernel<256, 256, 16, 2, 4, 2>), grid, block, 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_split_pa_kernel<128, 128, 16, 2, 2, 3>), grid, block, 0, s, g); break;
  }
  return launch_status("fpsg_gemm_split_nn_packed");
}

extern "C" size_t fpsg_gemm_split_workspace_floats(int batch, int M, int N, int K, int transB, int variant) {
  fpsg::Plan p;
  if (batch <= 0 || M <= 0 || N <= 0 || K <= 0 || !fpsg::plan_for(batch, M, N, K, transB, variant, &p)) return 0;
  return p.splits > 1 ? (size_t)p.splits * batch * M * N : 0;
}

extern "C" int fpsg_gemm_split(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb,
                               int ldc, long sA, long sB, long sC, int transB, int variant, float* ws, size_t ws_floats,
                               fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0, FPSG_E_SHAPE, "fpsg_gemm_split: batch, M, N, K must be positive");
  FPSG_REQUIRE(lda >= K && ldc >= N && ldb >= (transB ? K : N), FPSG_E_SHAPE, "fpsg_gemm_split: leading dimension below the row length");
  FPSG_REQUIRE_PTR(A);
  FPSG_REQUIRE_PTR(B);
  FPSG_REQUIRE_PTR(C);
  Plan p;
  FPSG_REQUIRE(plan_for(batch, M, N, K, transB, variant, &p), FPSG_E_SHAPE, "fpsg_gemm_split: unknown variant %d", variant);
  const long blocks = (long)batch * p.tiles_m * p.tiles_n * p.splits;
  FPSG_REQUIRE(blocks < (1L << 30), FPSG_E_LIMIT, "fpsg_gemm_split: too many tiles");
  GemmArgs g;
  g.A = A; g.B = B;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb;
  g.sA = sA; g.sB = sB;
  g.tiles_m = p.tiles_m; g.tiles_n = p.tiles_n; g.splits = p.splits; g.k_per_split = p.k_per_split;
  if (p.splits > 1) {
    FPSG_REQUIRE(ws != nullptr && ws_floats >= (size_t)p.splits * batch * M * N, FPSG_E_SHAPE,
                 "fpsg_gemm_split: workspace of %zu floats needed", (size_t)p.splits * batch * M * N);
    FPSG_REQUIRE(ldc == N && sC == (long)M * N, FPSG_E_SHAPE, "fpsg_gemm_split: a split reduction needs a dense output");
    g.C = ws; g.ldc = N; g.sC = (long)M * N; g.s_split = (long)batch * M * N;
  } else {
    g.C = C; g.ldc = ldc; g.sC = sC; g.s_split = 0;
  }
  FPSG_REQUIRE((long)M * lda < (1L << 29) && (long)(transB ? N : K) * ldb < (1L << 29) && (long)M * ldc < (1L << 29),
               FPSG_E_LIMIT, "fpsg_gemm_split: a matrix of one batch entry must stay below 2 GiB (32-bit buffer offsets)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), block(kTiles[p.tile].threads);
  const bool prio = p.prio;
#define FPSG_GEMM_LAUNCH_PIPE(BM, BN, BK, WR, WC, MINW)                                                           \
  do {                                                                                                            \
    if (transB) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, true, true, true>), grid, block, 0, s, g);  \
    else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, false, true, true>), grid, block, 0, s, g);        \
  } while (0)
#define FPSG_GEMM_LAUNCH(BM, BN, BK, WR, WC, MINW)                                                                \
  do {                                                                                                            \
    if (prio) {                                                                                                    \
      if (transB) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, true, true>), grid, block, 0, s, g);  \
      else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, false, true>), grid, block, 0, s, g);        \
    } else {                                                                                                       \
      if (transB) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, true, false>), grid, block, 0, s, g); \
      else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, BK, WR, WC, MINW, false, false>), grid, block, 0, s, g);       \
    }                                                                                                              \
  } while (0)
  if (p.abl > 0) {        // measurement builds (tools/bench_gemm_split.py --variants 2000 ...): wrong results by design
    FPSG_REQUIRE(!transB && (p.tile == 0 || p.tile == 2) && p.abl <= 3, FPSG_E_SHAPE, "fpsg_gemm_split: no such ablation build");
#define FPSG_ABL(A)                                                                                                     \
    do {                                                                                                                \
      if (p.tile == 0) hipLaunchKernelGGL((gemm_split_kernel<256, 256, 16, 2, 4, 2, false, true, false, A>), grid, block, 0, s, g); \
      else hipLaunchKernelGGL((gemm_split_kernel<128, 128, 16, 2, 2, 3, false, true, false, A>), grid, block, 0, s, g);             \
    } while (0)
    if (p.abl == 1) FPSG_ABL(1);
    else if (p.abl == 2) FPSG_ABL(2);
    else FPSG_ABL(3);
#undef FPSG_ABL
    return launch_status("fpsg_gemm_split (ablation)");
  }
  switch (p.tile) {
    case 0: FPSG_GEMM_LAUNCH(256, 256, 16, 2, 4, 2); break;
    case 1: FPSG_GEMM_LAUNCH(256, 128, 32, 2, 4, 2); break;
    case 2: FPSG_GEMM_LAUNCH(128, 128, 16, 2, 2, 3); break;
    case 3: FPSG_GEMM_LAUNCH(128, 256, 16, 1, 4, 2); break;
    case 4: FPSG_GEMM_LAUNCH_PIPE(256, 256, 16, 2, 4, 2); break;
    case 5: FPSG_GEMM_LAUNCH_PIPE(128, 128, 16, 2, 2, 3); break;
    default: FPSG_GEMM_LAUNCH_PIPE(128, 256, 16, 1, 4, 2); break;
  }
#undef FPSG_GEMM_LAUNCH
#undef FPSG_GEMM_LAUNCH_PIPE
  int rc = launch_status("fpsg_gemm_split");
  if (rc != 0 || p.splits == 1) return rc;
  const long n = (long)batch * M * N;
  FPSG_REQUIRE(ldc == N, FPSG_E_SHAPE, "fpsg_gemm_split: dense output expected");
  const int rblocks = (int)std::min<long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ws, C, n, g.s_split, p.splits);
  return launch_status("fpsg_gemm_split (reduce)");
}
