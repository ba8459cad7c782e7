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
  int nt_c;                  // non-temporal stores of C
  long packed_floats;        // PACKA builds: A is the packed image (fpsg_gemm_split_pack_a, 128-row tiles), this many floats
  int stages_a;              //               and ceil(K / 16) stages per row tile
  int stagger_mode;          // measurement (variant >= 100000): the first round's workgroups of one CU start stagger_ticks
  int stagger_per_cu;        //   (10 ns units) apart, so that their tile stores do not coincide; mode 1: co-resident
  int stagger_ticks;         //   workgroups = blockIdx 256 apart, mode 2: consecutive blockIdx
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

// MFMA fragments are carried as four dwords (one ds_read_b128) and cast to 8 x bf16 only AT the MFMA call.  Cast where
// they are loaded -- bf16x8 values flowing through loops and branches -- hipcc (ROCm 7.2) re-packs every register with a
// v_lshrrev + v_perm pair: 8 vector instructions per fragment, 6 per MFMA in these kernels, on the critical path of
// every product (found with SQ_INSTS_VALU: 37 M in a build that should have had none; the first versions of this
// file ran at half their MFMA rate because of it).
__device__ __forceinline__ void keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }   // measurement builds

typedef u32x4 frag_t;      // carried as four dwords through the control flow, cast to 8 x bf16 only AT the MFMA
__device__ __forceinline__ frag_t lds_frag(const unsigned char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ f32x16 mfma_bf16(frag_t a, frag_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr unsigned kOut = 0x7fffffffu;     // a lane offset beyond every buffer: the load returns 0, the store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// The accumulators of a wave's TI x TJ blocks -> C.  The MFMAs take the COLUMN-side fragment as their first operand
// (D = B^T A^T), so block (i, j) holds, in lane (r = lane & 31, h = lane >> 5), row 32 i + r of the wave's share and in
// registers 4 gq .. 4 gq + 3 the four consecutive columns 32 j + 8 gq + 4 h + (0..3): one 16-byte store per group instead
// of four 4-byte ones (128 dword stores per lane were a third of a 16-step tile's time).  Rows beyond M and columns
// beyond col_end are dropped; ragged ends fall back to dword stores.  nt: non-temporal (an output beyond the Infinity
// Cache that is read back only by the next kernel).
template <int TI, int TJ>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[TI][TJ], float* Cb, int M, int N, int ldc, int row0, int col0, int col_end,
                                           int lane, int nt) {
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cb, (long)(M - 1) * ldc + N);
  const int r = lane & 31, h = lane >> 5;
  auto body = [&](auto aux_c) {
    constexpr int AUX = decltype(aux_c)::value;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int row = row0 + 32 * i + r;
      const bool row_ok = row < M;
      const unsigned rbase = row_ok ? (unsigned)row * (unsigned)ldc : 0u;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int cl = col0 + 32 * j + 8 * gq + 4 * h;
          if (row_ok && cl + 3 < col_end) {
            const u32x4 v = {__float_as_uint(acc[i][j][4 * gq]), __float_as_uint(acc[i][j][4 * gq + 1]),
                             __float_as_uint(acc[i][j][4 * gq + 2]), __float_as_uint(acc[i][j][4 * gq + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsC, (rbase + (unsigned)cl) * 4u, 0, AUX);
          } else if (row_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (cl + e < col_end)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][4 * gq + e]), rsC, (rbase + (unsigned)(cl + e)) * 4u, 0, AUX);
          }
        }
    }
  };
  if (nt) body(std::integral_constant<int, 2>{});
  else body(std::integral_constant<int, 0>{});
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

// The A operand already split (fpsg_gemm_split_pack_a: [batch * row tile][stage][piece][k-group][R rows] x 16 B, zero-padded
// in rows and k): item = (row, k-group) as above, three 16-byte loads (one per piece, 1 KB contiguous per wave instruction)
// straight into the three LDS stores -- no vector instruction.  Same interface as ContigStage; init takes the stage count
// and the packed row-tile index where ContigStage takes the leading dimension and the first row.
template <int R, int BK, int kThreads>
struct PackedStage {
  static constexpr int KG = BK / 8;
  static constexpr int PS = R * 16 + 128 / KG;
  static constexpr int BYTES = 3 * KG * PS;
  static constexpr int ITEMS = (R * KG + kThreads - 1) / kThreads;
  static constexpr bool PARTIAL = R * KG % kThreads != 0;
  static constexpr int CHUNKS = 3 * KG * (R / 64);       // 1 KB pieces per stage
  static_assert(!PARTIAL || ITEMS == 1, "tile / thread mismatch");
  static_assert(BK == 16, "the packed image has 16-deep stages");
  u32x4 p[ITEMS][3];
  unsigned off[ITEMS];
  unsigned qbase;                                        // (row tile index) * stages
  __device__ __forceinline__ void init(int stages, int q, int /*rows*/, int tid) {
    qbase = (unsigned)q * (unsigned)stages;
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      const int row = it % R, g = it / R;
      off[n] = it < R * KG ? (unsigned)((g * (R / 64) + row / 64) * 1024 + (row % 64) * 16) : kOut;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int k0) {
    const unsigned sbase = (qbase + (unsigned)(k0 / BK)) * (unsigned)(CHUNKS * 1024);
#pragma unroll
    for (int n = 0; n < ITEMS; ++n)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        p[n][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[n] + t * (KG * (R / 64) * 1024), sbase, 0));
  }
  __device__ __forceinline__ void mask_tail(int, int, int) {}
  __device__ __forceinline__ void split_only() {}
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      if (PARTIAL && it >= R * KG) break;
      const int row = it % R, g = it / R;
      unsigned char* d = lds + g * PS + row * 16;
#pragma unroll
      for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4*>(d + t * KG * PS) = p[n][t];
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
template <int BM, int BN, int BK, int WR, int WC, int MINW, bool TRANSB, bool PRIO, bool PIPE = false, int ABL = 0, bool PACKA = false>
__global__ __launch_bounds__(64 * WR * WC, MINW) void gemm_split_kernel(const GemmArgs g) {
  constexpr int kThreads = 64 * WR * WC;
  constexpr int WM = BM / WR, WN = BN / WC, TI = WM / 32, TJ = WN / 32;
  constexpr int KG = BK / 8;
  using StageA = typename std::conditional<PACKA, PackedStage<BM, BK, kThreads>, ContigStage<BM, BK, kThreads>>::type;
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
  if (g.stagger_mode && bid < 256 * g.stagger_per_cu) {
    const int phase = g.stagger_mode == 1 ? (bid / 256) % g.stagger_per_cu : bid % g.stagger_per_cu;
    if (phase) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), wait = (unsigned long long)phase * g.stagger_ticks;
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
  }
  // Co-resident waves of one SIMD run the same program; under the default (round-robin) arbitration they fall into
  // lock-step -- all in their MFMA block, then all in their conversion block with the matrix pipe idle (measured:
  // SQ_WAIT_INST_ANY = 3x the MFMA busy time, pipe 46 % busy).  Distinct static priorities make the arbitration strict:
  // the higher wave runs its MFMA block at full rate and the others fill the pipe while it converts.
  if (PRIO) set_wave_priority(WR * WC >= 8 ? (2 * (bid & 1) + (wave >= WR * WC / 2 ? 1 : 0)) : bid % 3);
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  const __amdgpu_buffer_rsrc_t rsA =
      PACKA ? make_rsrc(g.A, g.packed_floats) : make_rsrc(g.A + batch * g.sA, (long)(g.M - 1) * g.lda + g.K);
  const __amdgpu_buffer_rsrc_t rsB =
      make_rsrc(g.B + batch * g.sB, TRANSB ? (long)(g.N - 1) * g.ldb + g.K : (long)(g.K - 1) * g.ldb + g.N);

  StageA ra;
  StageB rb;
  if constexpr (PACKA) ra.init(g.stages_a, batch * g.tiles_m + tm, g.M, tid);
  else ra.init(g.lda, m0, g.M, tid);
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
      frag_t fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, frag_t (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = lds_frag((cur + a_off + (t * KG + 2 * s) * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = lds_frag((cur + b_off + (t * KG + 2 * s) * PSB + j * 512));
      read_a(0, fa[0]);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (i + 1 < TI) read_a(i + 1, fa[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const frag_t (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = mfma_bf16(fb[0][j], a[2], c);
          c = mfma_bf16(fb[1][j], a[1], c);
          c = mfma_bf16(fb[2][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[1], c);
          c = mfma_bf16(fb[1][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[0], c);
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
      frag_t fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, frag_t (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = lds_frag((cur + a_off + t * KG * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = lds_frag((cur + b_off + t * KG * PSB + j * 512));
      read_a(0, fa[0]);
      if (TI > 1) read_a(1, fa[1]);
      if constexpr (STORE && MASK) {
        const int k0 = k_begin + (kt + 1) * BK;
        ra.mask_tail(k0, k_end, tid);
        rb.mask_tail(k0, k_end, tid);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const frag_t (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = mfma_bf16(fb[0][j], a[2], c);
          c = mfma_bf16(fb[1][j], a[1], c);
          c = mfma_bf16(fb[2][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[1], c);
          c = mfma_bf16(fb[1][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[0], c);
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

  store_tile<TI, TJ>(acc, g.C + split * g.s_split + batch * g.sC, g.M, g.N, g.ldc, m0 + wm0, n0 + wn0, g.N, lane, g.nt_c);
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
      frag_t fb[3][TJ], fa[2][3];
      auto read_a = [&](int i, frag_t (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = lds_frag((cur + a_off + (t * KG + 2 * s) * PSA + i * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = lds_frag((cur + b_off + (t * KG + 2 * s) * PSB + j * 512));
      read_a(0, fa[0]);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (i + 1 < TI) read_a(i + 1, fa[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const frag_t (&a)[3] = fa[i & 1];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = mfma_bf16(fb[0][j], a[2], c);
          c = mfma_bf16(fb[1][j], a[1], c);
          c = mfma_bf16(fb[2][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[1], c);
          c = mfma_bf16(fb[1][j], a[0], c);
          c = mfma_bf16(fb[0][j], a[0], c);
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

  store_tile<TI, TJ>(acc, g.C + batch * g.sC, g.M, g.N, g.ldc, m0 + wm0, n0 + wn0, g.N, lane, 0);
}

// ---- the forward / data-gradient form as ONE persistent launch (round 5, third version).  Measured on the tiled
// kernels above (profiles/r05/gemm_split_ablation.txt): with every staging step removed a 256 x 256 tile kernel still
// took 205 us at 256 -> 256 @56 against 114 us of MFMA time -- a workgroup's launch, its two global round trips before
// the first product and its 128 dword stores per lane are a third of a 16-step tile, and 1044 tiles on 256 CUs are 5
// rounds where 4.08 would do.  Here a workgroup per CU walks a contiguous range of the flattened (batch, row tile,
// column) space, cut at multiples of 32 columns -- equal shares whatever the tile count -- and the software pipeline
// (A by LDS-DMA one stage ahead, B in registers two stages ahead, split + LDS stores of the next stage among the MFMAs
// of this one) runs across tile boundaries: the next tile's first stages are in flight while the last products of a
// tile issue, its stores go out behind the barrier, and the matrix pipe never waits for a launch.
struct PersistArgs {
  const u32x4* Ap;
  const float* B;
  float* C;
  int M, N, K;
  int ldb, ldc;
  long sB, sC;
  int tiles_m, stages, batchq;     // stages = ceil(K / 16), batchq = batch * tiles_m
  long total;                      // batchq * N: the flattened column space
  int nt_c;                        // non-temporal stores of C
};

constexpr int kMaxPieces = 40;

// ABL (measurements only, wrong results): 1 = no DMA in the loop, 2 = no B loads / split / LDS stores, 3 = neither,
// 4 = 3 without the tiles' stores
template <int BN, bool KTAIL, int ABL = 0>       // KTAIL: K is no multiple of 16 (the last k-step's rows beyond K are zeroed)
__global__ __launch_bounds__(512, 2) void gemm_split_pnn_kernel(const PersistArgs g) {
  // 8 waves, each 32 rows x BN columns of the tile: the COLUMNS are the serial dimension of a wave (TJ blocks of 32), so a
  // piece narrower than BN costs every wave proportionally less -- equal column shares are equal times, whatever the
  // tile count (with the columns spread over waves a 100-column piece cost a full tile: measured 25 % imbalance).
  // The MFMA takes the B fragment as its first operand: D[column block][row block]^T, so a lane's four consecutive
  // accumulator registers are four consecutive COLUMNS of one row of C -- one 16-byte store instead of four 4-byte ones
  // (dword stores took 128 instructions per lane and tile, a third of a 16-step tile's time).
  constexpr int BM = 256, BK = 16, KG = 2, NW = 8;
  constexpr int TJ = BN / 32;
  constexpr int PSA = BM * 16 + 64, PSB = BN * 16 + 64;
  constexpr int A_BYTES = 3 * KG * PSA, B_BYTES = 3 * KG * PSB, STAGE = A_BYTES + B_BYTES;
  constexpr int CHUNKS = 3 * KG * BM / 64, DMA_PER_WAVE = CHUNKS / NW;      // 24 one-KB pieces per packed stage, 3 per wave
  constexpr int B_ITEMS = BN * KG;                                          // (column, k-group) items of a stage
  static_assert(B_ITEMS == 512 || B_ITEMS == 256, "one item per thread (or per thread of waves 0-3)");
  static_assert(CHUNKS % NW == 0, "DMA pieces per wave");
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE + kMaxPieces * 16];   // a ring of three stages
  int* ptab = reinterpret_cast<int*>(lds + 3 * STAGE);                      // [piece] {q, c0, w, -}: same array as the stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsteps = g.stages;

  // this workgroup's range of the flattened space, cut at multiples of 32 columns inside a (batch, row tile) segment
  if (tid == 0) {
    auto cut = [&](long b) {
      if (b >= gridDim.x) return g.total;
      const long raw = g.total / gridDim.x * b + g.total % gridDim.x * b / gridDim.x;
      const long q = raw / g.N, c = (raw % g.N) & ~31L;
      return q * g.N + c;
    };
    // consecutive ranges to the workgroups of one XCD (blockIdx mod 8 names the XCD's workgroups): neighbours share a
    // batch entry's packed A through that XCD's L2
    const long nb = gridDim.x, bid = blockIdx.x, q8 = nb / 8, r8 = nb % 8, x8 = bid % 8;
    const long id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + bid / 8;
    long pos = cut(id);
    const long end = cut(id + 1);
    int np = 0;
    while (pos < end && np < kMaxPieces - 1) {
      const long q = pos / g.N, c = pos % g.N;
      long w = g.N - c;
      if (w > BN) w = BN;
      if (w > end - pos) w = end - pos;
      ptab[4 * np] = (int)q; ptab[4 * np + 1] = (int)c; ptab[4 * np + 2] = (int)w;
      pos += w;
      ++np;
    }
    ptab[4 * (kMaxPieces - 1)] = np;
  }
  __syncthreads();
  const int npieces = __builtin_amdgcn_readfirstlane(ptab[4 * (kMaxPieces - 1)]);
  if (npieces == 0) return;
  const int S = npieces * nsteps;                                           // stages of the whole stream
  set_wave_priority(wave >= NW / 2 ? 1 : 0);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(g.Ap), 0, (int)((long)g.batchq * nsteps * (CHUNKS * 1024)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.B), 0, (int)(((long)(g.batchq / g.tiles_m - 1) * g.sB + (long)(g.K - 1) * g.ldb + g.N) * 4), 0x00020000);
  const unsigned ld4 = (unsigned)g.ldb * 4u;

  // ---- the three streams: A by DMA and B into registers (both two stages ahead), products.  They advance one stage
  // per iteration: piece and k-step are carried, the piece table is read only when a stream enters a new piece
  auto piece_q = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p]); };
  auto piece_c = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 1]); };
  auto piece_w = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 2]); };
  struct Cursor { int p, kt, q, c0; };
  auto enter = [&](Cursor& cu, int p) {
    cu.p = p; cu.kt = 0; cu.q = piece_q(p); cu.c0 = piece_c(p);
  };
  auto advance = [&](Cursor& cu) {                                         // to the next stage; stays on the last one at the end
    if (cu.kt + 1 < nsteps) { ++cu.kt; }
    else if (cu.p + 1 < npieces) enter(cu, cu.p + 1);
  };
  auto dma_a = [&](const Cursor& cu, int slot) {                           // packed stage at the cursor -> ring slot
    const unsigned base = ((unsigned)cu.q * (unsigned)nsteps + (unsigned)cu.kt) * (unsigned)(CHUNKS * 1024);
    unsigned char* d = lds + slot * STAGE;
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) {
      const int c = wave + q * NW;
      const int plane = c / (BM / 64), r64 = c % (BM / 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(d + plane * PSA + r64 * 1024), 16,
                                               lane * 16, base + c * 1024, 0, 0);
    }
  };
  const bool converts = B_ITEMS == 512 || wave < 4;
  const int b_col = tid % BN, b_g = tid / BN;                               // this thread's item of a B stage
  float bv[8];
  auto load_b = [&](const Cursor& cu) {                                    // stage at the cursor -> registers
    const int batch = cu.q / g.tiles_m;
    const int col = cu.c0 + b_col;
    const unsigned voff = (col < g.N && converts) ? (unsigned)col * 4u + (unsigned)b_g * 8u * ld4 : kOut;
    const unsigned sbase = (unsigned)((long)batch * g.sB * 4) + (unsigned)(cu.kt * BK) * ld4;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, voff, sbase + j * ld4, 0));
  };
  u32x4 bp[3];                                                              // the split stage, between split_b and store_b
  auto split_pair = [&](int q) {                                           // elements 2q, 2q+1 of the item (11 vector instructions)
    const float a = bv[2 * q], b = bv[2 * q + 1];
    const unsigned hh = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
    const unsigned mm = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
    bp[0][q] = hh; bp[1][q] = mm; bp[2][q] = cvt_pk_bf16(sa, sb);
  };
  auto mask_b = [&](int kt_of_stage) {
    if constexpr (KTAIL) {      // (a branch here would make the compiler wait for every outstanding load, the DMA included)
      const int k0 = kt_of_stage * BK + b_g * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = k0 + j < g.K ? bv[j] : 0.f;
    }
  };
  auto store_b = [&](int slot) {                                           // the split stage -> ring slot
    if (!converts) return;
    unsigned char* d = lds + slot * STAGE + A_BYTES + b_g * PSB + b_col * 16;
    *reinterpret_cast<u32x4*>(d) = bp[0];
    *reinterpret_cast<u32x4*>(d + KG * PSB) = bp[1];
    *reinterpret_cast<u32x4*>(d + 2 * KG * PSB) = bp[2];
  };

  f32x16 acc[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int a_off = h * PSA + (wave * 32 + r) * 16;                         // this wave's 32 rows
  const int b_off = A_BYTES + h * PSB + r * 16;                             // + 512 per column block

  auto epilogue = [&](int p) {                                             // the finished tile of piece p
    const int q = piece_q(p), c0 = piece_c(p), w = piece_w(p);
    const int batch = q / g.tiles_m, tm = q - batch * g.tiles_m;
    f32x16 (&tile)[1][TJ] = reinterpret_cast<f32x16 (&)[1][TJ]>(acc);
    store_tile<1, TJ>(tile, g.C + batch * g.sC, g.M, g.N, g.ldc, tm * BM + wave * 32, c0, c0 + w, lane, g.nt_c);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  };

  // Ring of three LDS stages.  Iteration s multiplies stage s (slot s % 3), splits B(s+1) -- in registers since the
  // previous iteration -- into slot (s+1) % 3, then issues the register loads of B(s+2) and, LAST, the DMA of A(s+2) into
  // slot (s+2) % 3 (read last in iteration s-1: the barrier behind it has passed).  Everything a wait can see was
  // issued at least one MFMA block earlier: hipcc does not count LDS-DMA loads in its own vmcnt bookkeeping (its
  // "vmcnt(0)" for the register loads also waits for every DMA in flight), so a DMA must never be the youngest
  // operation when register loads are consumed.  Before the barrier: A(s+1) -- issued at the end of iteration s-1 --
  // must have landed; younger than it are this iteration's 8 register loads and DMA_PER_WAVE DMAs.  (A finished tile's
  // stores go out right behind the barrier, i.e. OLDER than everything this count leaves in flight: the same wait makes
  // them complete one MFMA block later, when they long are.)
  Cursor ca, cb;
  enter(ca, 0);
  enter(cb, 0);
  dma_a(ca, 0);
  load_b(cb);
  mask_b(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(q);
  store_b(0);
  advance(ca);
  advance(cb);
  int kt_b1 = cb.kt;                   // k-step of the stage the registers hold (for the K tail)
  load_b(cb);
  if (S > 1) dma_a(ca, 1);
  advance(ca);
  advance(cb);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int kt = 0, piece = 0, slot = 0;
  int nj = (piece_w(0) + 31) / 32;                                          // column blocks of the current piece
  for (int s = 0; s < S; ++s) {
    const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
    {
      const unsigned char* cur = lds + slot * STAGE;
      frag_t fa[3], fb[2][3];
      auto read_b = [&](int j, frag_t (&dst)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          dst[t] = lds_frag((cur + b_off + t * KG * PSB + j * 512));
      };
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fa[t] = lds_frag((cur + a_off + t * KG * PSA));
      read_b(0, fb[0]);
      if (ABL != 2 && ABL < 3) mask_b(kt_b1);
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        if (j < nj) {                                                       // wave-uniform
          if (j + 1 < TJ) read_b(j + 1, fb[(j + 1) & 1]);
          const frag_t (&b)[3] = fb[j & 1];
          f32x16 c = acc[j];
          c = mfma_bf16(b[0], fa[2], c);
          c = mfma_bf16(b[1], fa[1], c);
          c = mfma_bf16(b[2], fa[0], c);
          c = mfma_bf16(b[0], fa[1], c);
          c = mfma_bf16(b[1], fa[0], c);
          c = mfma_bf16(b[0], fa[0], c);
          acc[j] = c;
        }
        // the split of B(s+1) in four parts behind the first column blocks' MFMAs (11 vector instructions each)
        if (ABL != 2 && ABL < 3 && j < 4 && converts) split_pair(j);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ABL != 2 && ABL < 3) {
        if (TJ < 4) {
#pragma unroll
          for (int q = TJ; q < 4; ++q) split_pair(q);
        }
        store_b(slot1);
        kt_b1 = cb.kt;
        load_b(cb);                      // (beyond the end of the stream: a harmless repeat of the last stage)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < S && ABL != 1 && ABL < 3) dma_a(ca, slot2);                // the youngest operations of the iteration
    if (ABL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (s + 2 < S) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    advance(ca);
    advance(cb);
    slot = slot1;
    if (++kt == nsteps) {                                                  // tile finished: its stores go out behind the barrier
      if (ABL == 4) {                                                      // (measurement: the products kept alive, nothing stored)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          keep_alive(acc[j]);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        }
      } else
      epilogue(piece);
      kt = 0;
      ++piece;
      if (piece < npieces) nj = (piece_w(piece) + 31) / 32;
    }
  }
}

// ---- the persistent forward form with the waves SPECIALISED (round 5, fourth version): waves 0-3 only multiply, waves 4-7
// only stage.  What the measurements of the earlier forms say (profiles/r05/gemm_split_ablation.txt): the product loop
// alone is worth 1.5-1.8 PFLOP/s, but (1) a wave that also loads must wait for its own tile stores before it can consume a
// later load (the vmcnt counter is in order); (2) an LDS-DMA instruction stalls the issuing wave 100-180 cycles; (3) the
// split's vector instructions and LDS stores sit in the MFMA waves' streams.  With the roles split, a SIMD holds one
// consumer (fragment reads + 48 MFMAs per step; its stores are never waited for: it issues no load) and one producer.
// Producers move BOTH operands by LDS-DMA (A: the packed pieces straight into the ring slot two stages ahead; B: the raw
// fp32 rows into a three-deep raw ring, three stages ahead -- a register-staged B could not be prefetched deeper than one
// step without hipcc's own vmcnt waits, which do not count DMAs, draining the younger loads: measured 250 us), then split
// the raw stage that landed a step ago from LDS into the bf16 planes.  Every vmcnt wait is written by hand; one
// workgroup-wide s_barrier per step hands a ring slot over.  Same ranges / pieces as gemm_split_pnn_kernel; tile 256 x BN.
// Needs ldb, sB, the range starts (multiples of 32) and B itself aligned to 4 floats (16-byte DMA of B rows).
// ABL (measurement builds, results meaningless): 1 = producers only meet the barriers, 3 = no B path, 4 = consumers only
// meet the barriers, 5 = consumers without the tile stores, 6 = 1 + 5
// Two builds: BM 256 / rings of 3 (one workgroup per CU, 138 KB of LDS), and BM 128 / A ring of 2 (two workgroups per CU,
// 76 KB each, 128 registers: one workgroup's store burst and barrier waits overlap the other's products).
template <int BM, int BN, int RA, int RR, int MINW, bool KTAIL, int ABL = 0>
__global__ __launch_bounds__(512, MINW) void gemm_split_ws_kernel(const PersistArgs g) {
  constexpr int BK = 16, KG = 2, NC = 4, NP = 4;                           // consumer / producer waves
  constexpr int TI = BM / NC / 32, TJ = BN / 32;                           // a consumer: BM / 4 rows x BN columns
  constexpr int PSA = BM * 16 + 64, PSB = BN * 16 + 64;
  constexpr int A_BYTES = 3 * KG * PSA, B_BYTES = 3 * KG * PSB;
  constexpr int CHUNKS = 3 * KG * BM / 64, DMA_A = CHUNKS / NP;             // one-KB pieces per packed stage
  constexpr int RAW = BK * BN * 4, RAW_CHUNKS = RAW / 1024, DMA_B = RAW_CHUNKS / NP;   // raw B stage: [16 k][BN] fp32
  constexpr int ROWS_PER_DMA = 1024 / (BN * 4), LANES_PER_ROW = BN / 4;    // 16 B per lane
  constexpr int B_ITEMS = BN * KG, B_PER_THREAD = B_ITEMS / (64 * NP);      // (column, k-group) items per producer thread
  static_assert(B_ITEMS % (64 * NP) == 0 && CHUNKS % NP == 0 && RAW_CHUNKS % NP == 0, "stage / producer mismatch");
  static_assert((RA == 2 || RA == 3) && (RR == 2 || RR == 3), "ring depths");
  // LDS: RA slots of packed A | 2 slots of split B | RR slots of raw B | the piece table
  __shared__ __attribute__((aligned(16))) unsigned char lds[RA * A_BYTES + 2 * B_BYTES + RR * RAW + kMaxPieces * 16];
  unsigned char* const ldsB = lds + RA * A_BYTES;
  unsigned char* const raw_ring = ldsB + 2 * B_BYTES;
  int* ptab = reinterpret_cast<int*>(raw_ring + RR * RAW);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsteps = g.stages;

  if (tid == 0) {
    auto cut = [&](long b) {
      if (b >= gridDim.x) return g.total;
      const long raw = g.total / gridDim.x * b + g.total % gridDim.x * b / gridDim.x;
      const long q = raw / g.N, c = (raw % g.N) & ~31L;
      return q * g.N + c;
    };
    const long nb = gridDim.x, bid = blockIdx.x, q8 = nb / 8, r8 = nb % 8, x8 = bid % 8;
    const long id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + bid / 8;
    long pos = cut(id);
    const long end = cut(id + 1);
    int np = 0;
    // g.nt_c bit 1: a first piece of 32 / 64 / 96 / 128 columns by workgroup, so that the CUs do not finish their tiles (and
    // burst their stores) all at the same moment
    long first = (g.nt_c & 2) ? 32 * (1 + (id & 3)) : BN;
    while (pos < end && np < kMaxPieces - 1) {
      const long q = pos / g.N, c = pos % g.N;
      long w = g.N - c;
      if (w > BN) w = BN;
      if (w > first) w = first;
      first = BN;
      if (w > end - pos) w = end - pos;
      ptab[4 * np] = (int)q; ptab[4 * np + 1] = (int)c; ptab[4 * np + 2] = (int)w;
      pos += w;
      ++np;
    }
    ptab[4 * (kMaxPieces - 1)] = np;
  }
  __syncthreads();
  const int npieces = __builtin_amdgcn_readfirstlane(ptab[4 * (kMaxPieces - 1)]);
  if (npieces == 0) return;
  const int S = npieces * nsteps;
  auto piece_q = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p]); };
  auto piece_c = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 1]); };
  auto piece_w = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 2]); };

  if (wave >= NC) {
    // ------------------------------------------------------------------ producers
    const int pw = wave - NC, pt = tid - 64 * NC;
    if constexpr (ABL == 1 || ABL == 6) {
      __builtin_amdgcn_s_barrier();
      for (int s = 0; s < S; ++s) __builtin_amdgcn_s_barrier();
      return;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(g.Ap), 0, (int)((long)g.batchq * nsteps * (CHUNKS * 1024)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.B), 0, (int)(((long)(g.batchq / g.tiles_m - 1) * g.sB + (long)(g.K - 1) * g.ldb + g.N) * 4), 0x00020000);
    const unsigned ld4 = (unsigned)g.ldb * 4u;
    struct Cursor { int p, kt, q, c0; };
    auto enter = [&](Cursor& cu, int p) { cu.p = p; cu.kt = 0; cu.q = piece_q(p); cu.c0 = piece_c(p); };
    auto advance = [&](Cursor& cu) {
      if (cu.kt + 1 < nsteps) { ++cu.kt; }
      else if (cu.p + 1 < npieces) enter(cu, cu.p + 1);
    };
    auto dma_a = [&](const Cursor& cu, int slot) {
      const unsigned base = ((unsigned)cu.q * (unsigned)nsteps + (unsigned)cu.kt) * (unsigned)(CHUNKS * 1024);
      unsigned char* d = lds + slot * A_BYTES;
#pragma unroll
      for (int q = 0; q < DMA_A; ++q) {
        const int c = pw + q * NP;
        const int plane = c / (BM / 64), r64 = c % (BM / 64);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(d + plane * PSA + r64 * 1024), 16,
                                                 lane * 16, base + c * 1024, 0, 0);
      }
    };
    // raw B stage [16 k][BN] fp32: one DMA = ROWS_PER_DMA k-rows; columns at or beyond N and k-rows at or beyond K read zeros
    const int b_lrow = lane / LANES_PER_ROW, b_lcol = (lane % LANES_PER_ROW) * 4;
    auto dma_b = [&](const Cursor& cu, int rslot) {
      const int batch = cu.q / g.tiles_m;
      const unsigned sbase = (unsigned)((long)batch * g.sB * 4) + (unsigned)(cu.kt * BK) * ld4;
      const int col = cu.c0 + b_lcol;
      unsigned char* d = raw_ring + rslot * RAW;
#pragma unroll
      for (int q = 0; q < DMA_B; ++q) {
        const int c = pw + q * NP;
        const int krow = c * ROWS_PER_DMA + b_lrow;
        const bool ok = col < g.N && (!KTAIL || cu.kt * BK + krow < g.K);
        const unsigned voff = ok ? (unsigned)col * 4u + (unsigned)krow * ld4 : kOut;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(d + c * 1024), 16, voff, sbase, 0, 0);
      }
    };
    auto convert_b = [&](int rslot, int slot) {                             // raw fp32 (LDS) -> three bf16 planes (LDS)
#pragma unroll
      for (int n = 0; n < B_PER_THREAD; ++n) {
        const int it = pt + n * 64 * NP;
        const int col = it % BN, gq = it / BN;
        const float* src = reinterpret_cast<const float*>(raw_ring + rslot * RAW) + gq * 8 * BN + col;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * BN];
        u32x4 p1, p2, p3;
        split8(v, p1, p2, p3);
        unsigned char* d = ldsB + slot * B_BYTES + gq * PSB + col * 16;
        *reinterpret_cast<u32x4*>(d) = p1;
        *reinterpret_cast<u32x4*>(d + KG * PSB) = p2;
        *reinterpret_cast<u32x4*>(d + 2 * KG * PSB) = p3;
      }
    };
    // stage t: packed A in A slot t % RA (brought in during step t - RA + 1), raw B in raw slot t % RR (during step t - RR),
    // split B in B slot t % 2 (converted during step t - 1)
    Cursor ca, cb;
    enter(ca, 0);
    enter(cb, 0);
#pragma unroll
    for (int t = 0; t < RR; ++t) { dma_b(cb, t); advance(cb); }
#pragma unroll
    for (int t = 0; t < RA - 1; ++t) { dma_a(ca, t); advance(ca); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // (P) every producer's DMAs have landed
    convert_b(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // (0) stage 0 complete
    int sa = 0, sr = 0, sb = 0;                      // s % RA, s % RR, s % 2
    for (int s = 0; s < S; ++s) {
      const int sa_new = sa == 0 ? RA - 1 : sa - 1;  // (s + RA - 1) % RA: the slot stage s - 1 was read from
      const int sr1 = sr == RR - 1 ? 0 : sr + 1;
      // raw B(s+1) landed before the previous barrier
      if constexpr (ABL != 3) convert_b(sr1, sb ^ 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // before stage s+1 is handed over, A(s+1) and raw B(s+2) must have landed: with RA = 3 both were issued a step ago
      // (only this step's operations may be outstanding); with RA = 2 A(s+1) is issued now, FIRST, and only this step's raw
      // B(s+RR) may be outstanding
      if constexpr (RA == 3) {
        if constexpr (ABL != 3) dma_b(cb, sr);       // raw(s+RR) -> the slot of raw(s), converted in step s-1
        dma_a(ca, sa_new);
        if constexpr (ABL != 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_A + DMA_B) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_A) : "memory");
      } else {
        dma_a(ca, sa_new);
        if constexpr (ABL != 3) dma_b(cb, sr);
        if constexpr (ABL != 3 && RR == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_B) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      advance(ca);
      advance(cb);
      sa = sa == RA - 1 ? 0 : sa + 1;
      sr = sr1;
      sb ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (DMAs past the end repeat the last stage into free slots)
    return;
  }

  // -------------------------------------------------------------------- consumers
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  const int a_off = h * PSA + (wave * (BM / NC) + r) * 16;                 // + 512 per row block
  const int b_off = h * PSB + r * 16;                                      // + 512 per column block
  if constexpr (ABL != 1 && ABL != 6) __builtin_amdgcn_s_barrier();        // (P)
  __builtin_amdgcn_s_barrier();                                            // (0)
  if constexpr (ABL == 4) {
    for (int s = 0; s < S; ++s) __builtin_amdgcn_s_barrier();
    return;
  }
  int kt = 0, piece = 0, sa = 0, sb = 0;
  int nj = (piece_w(0) + 31) / 32;
  for (int s = 0; s < S; ++s) {
    const unsigned char* curA = lds + sa * A_BYTES + a_off;
    const unsigned char* curB = ldsB + sb * B_BYTES + b_off;
    frag_t fa[TI][3], fb[2][3];
    auto read_b = [&](int j, frag_t (&dst)[3]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) dst[t] = lds_frag((curB + t * KG * PSB + j * 512));
    };
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) fa[i][t] = lds_frag((curA + t * KG * PSA + i * 512));
    read_b(0, fb[0]);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      if (j < nj) {                                                         // wave-uniform: a narrower piece costs less
        if (j + 1 < TJ) read_b(j + 1, fb[(j + 1) & 1]);
        const frag_t (&b)[3] = fb[j & 1];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          f32x16 c = acc[i][j];
          c = mfma_bf16(b[0], fa[i][2], c);
          c = mfma_bf16(b[1], fa[i][1], c);
          c = mfma_bf16(b[2], fa[i][0], c);
          c = mfma_bf16(b[0], fa[i][1], c);
          c = mfma_bf16(b[1], fa[i][0], c);
          c = mfma_bf16(b[0], fa[i][0], c);
          acc[i][j] = c;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++kt == nsteps) {                                                  // tile finished: stores nobody waits for
      const int q = piece_q(piece), c0 = piece_c(piece), w = piece_w(piece);
      const int batch = q / g.tiles_m, tm = q - batch * g.tiles_m;
      if ((ABL != 5 && ABL != 6) || g.M < 0)
        store_tile<TI, TJ>(acc, g.C + batch * g.sC, g.M, g.N, g.ldc, tm * BM + wave * (BM / NC), c0, c0 + w, lane, 0);
      else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) keep_alive(acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      kt = 0;
      ++piece;
      if (piece < npieces) nj = (piece_w(piece) + 31) / 32;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    sa = sa == RA - 1 ? 0 : sa + 1;
    sb ^= 1;
  }
}

// ---- K11 (round 5): the SAME batched product on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), hand-written ---------------
// The headline step keeps fp32 MFMA arithmetic; its library GEMMs run at 0.73-0.90 of the pipe's sustained rate (137 TFLOP/s
// at the ~2.09 GHz the card holds under fp32 MFMAs).  This is the specialised-wave skeleton of gemm_split_ws_kernel without
// the split: four producer waves bring fp32 A (k-contiguous rows -> four planes [256 rows][4 k], 16 bytes per lane with a
// per-lane source address: no packing pass) and fp32 B (rows of BN columns as they lie) into a ring of three LDS stages by
// LDS-DMA, two stages ahead; four consumer waves (one per SIMD) read their fragments -- A as two ds_read_b128 per 32 rows
// (8 k values per lane), B as one ds_read_b32 per MFMA and column block -- and issue 64 MFMAs of 64 cycles per k-step of
// 16: staging, barrier and stores are a small fraction of that.  Persistent ranges / pieces as in the split kernels
// (1044 tiles on 256 CUs cost 4.08 rounds, not 5).  lane (x, h) of an MFMA holds k = 8 h + jj for the stage's jj-th
// MFMA (any pairing of the 16 k values is a valid order of the fp32 sum).  Needs lda, ldb, sA, sB multiples of 4 floats,
// 16-byte aligned bases and K a multiple of 4 (16-byte DMA pieces must not straddle the end of a row).
struct F32Args {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long sA, sB, sC;
  int tiles_m, stages, batchq;
  long total;
  int stagger;
};

// Two builds: 256 x 128 tiles, consumers 4 x 1, one workgroup per CU; 128 x 128 tiles, consumers 2 x 2, TWO workgroups per
// CU (128 registers): one workgroup's tile stores and barrier waits overlap the other's MFMAs.
template <int BM, int BN, int WR, int MINW, bool KTAIL>
__global__ __launch_bounds__(512, MINW) void gemm_f32_ws_kernel(const F32Args g) {
  constexpr int BK = 16, NC = 4, NP = 4, WC = NC / WR;
  constexpr int TI = BM / WR / 32, TJ = BN / WC / 32;
  constexpr int PLA = BM * 16 + 64;                                        // one k-quad plane of A: [256 rows][4 floats]
  constexpr int A_BYTES = 4 * PLA, B_BYTES = BK * BN * 4, STAGE = A_BYTES + B_BYTES;
  constexpr int A_CHUNKS = 4 * (BM / 64), DMA_A = A_CHUNKS / NP;            // one-KB pieces of A per stage
  constexpr int B_CHUNKS = B_BYTES / 1024, DMA_B = B_CHUNKS / NP;
  constexpr int ROWS_PER_DMA = 1024 / (BN * 4), LANES_PER_ROW = BN / 4;
  static_assert(B_CHUNKS % NP == 0 && A_CHUNKS % NP == 0 && WR * WC == NC, "stage / producer mismatch");
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE + kMaxPieces * 16];
  int* ptab = reinterpret_cast<int*>(lds + 3 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsteps = g.stages;

  if (tid == 0) {
    auto cut = [&](long b) {
      if (b >= gridDim.x) return g.total;
      const long raw = g.total / gridDim.x * b + g.total % gridDim.x * b / gridDim.x;
      const long q = raw / g.N, c = (raw % g.N) & ~31L;
      return q * g.N + c;
    };
    const long nb = gridDim.x, bid = blockIdx.x, q8 = nb / 8, r8 = nb % 8, x8 = bid % 8;
    const long id = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + bid / 8;
    long pos = cut(id);
    const long end = cut(id + 1);
    int np = 0;
    long first = g.stagger ? 32 * (1 + (id & 3)) : BN;
    while (pos < end && np < kMaxPieces - 1) {
      const long q = pos / g.N, c = pos % g.N;
      long w = g.N - c;
      if (w > BN) w = BN;
      if (w > first) w = first;
      first = BN;
      if (w > end - pos) w = end - pos;
      ptab[4 * np] = (int)q; ptab[4 * np + 1] = (int)c; ptab[4 * np + 2] = (int)w;
      pos += w;
      ++np;
    }
    ptab[4 * (kMaxPieces - 1)] = np;
  }
  __syncthreads();
  const int npieces = __builtin_amdgcn_readfirstlane(ptab[4 * (kMaxPieces - 1)]);
  if (npieces == 0) return;
  const int S = npieces * nsteps;
  auto piece_q = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p]); };
  auto piece_c = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 1]); };
  auto piece_w = [&](int p) { return __builtin_amdgcn_readfirstlane(ptab[4 * p + 2]); };

  if (wave >= NC) {
    // ------------------------------------------------------------------ producers
    const int pw = wave - NC;
    const int nbatch = g.batchq / g.tiles_m;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A, (long)(nbatch - 1) * g.sA + (long)(g.M - 1) * g.lda + g.K);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(g.B, (long)(nbatch - 1) * g.sB + (long)(g.K - 1) * g.ldb + g.N);
    const unsigned lda4 = (unsigned)g.lda * 4u, ldb4 = (unsigned)g.ldb * 4u;
    struct Cursor { int p, kt, q, c0; };
    auto enter = [&](Cursor& cu, int p) { cu.p = p; cu.kt = 0; cu.q = piece_q(p); cu.c0 = piece_c(p); };
    auto advance = [&](Cursor& cu) {
      if (cu.kt + 1 < nsteps) { ++cu.kt; }
      else if (cu.p + 1 < npieces) enter(cu, cu.p + 1);
    };
    const int b_lrow = lane / LANES_PER_ROW, b_lcol = (lane % LANES_PER_ROW) * 4;
    auto dma = [&](const Cursor& cu, int slot) {
      const int batch = cu.q / g.tiles_m, tm = cu.q - batch * g.tiles_m;
      unsigned char* d = lds + slot * STAGE;
      {   // A: pieces (k-quad plane p, group of 64 rows)
        const unsigned sbase = (unsigned)((long)batch * g.sA * 4) + (unsigned)(cu.kt * BK) * 4u;
#pragma unroll
        for (int q = 0; q < DMA_A; ++q) {
          const int c = pw + q * NP;
          const int p = c / (BM / 64), rg = c % (BM / 64);
          const int row = tm * BM + rg * 64 + lane;
          const bool ok = row < g.M && (!KTAIL || cu.kt * BK + 4 * p < g.K);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(d + p * PLA + rg * 1024), 16,
                                                   ok ? (unsigned)row * lda4 + p * 16u : kOut, sbase, 0, 0);
        }
      }
      {   // B: rows of BN columns
        const unsigned sbase = (unsigned)((long)batch * g.sB * 4) + (unsigned)(cu.kt * BK) * ldb4;
        const int col = cu.c0 + b_lcol;
#pragma unroll
        for (int q = 0; q < DMA_B; ++q) {
          const int c = pw + q * NP;
          const int krow = c * ROWS_PER_DMA + b_lrow;
          const bool ok = col < g.N && (!KTAIL || cu.kt * BK + krow < g.K);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(d + A_BYTES + c * 1024), 16,
                                                   ok ? (unsigned)col * 4u + (unsigned)krow * ldb4 : kOut, sbase, 0, 0);
        }
      }
    };
    Cursor cu;
    enter(cu, 0);
    dma(cu, 0); advance(cu);
    dma(cu, 1); advance(cu);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_A + DMA_B) : "memory");      // stage 0 has landed (stage 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int s = 0; s < S; ++s) {
      const int slot2 = slot == 0 ? 2 : slot - 1;                            // (s + 2) % 3: the slot stage s - 1 was read from
      dma(cu, slot2);                                                        // stage s + 2 (beyond the end: a harmless repeat)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_A + DMA_B) : "memory");    // stage s + 1 has landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      advance(cu);
      slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- consumers
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;                                 // this consumer's (row, column) share of the tile
  const int a_off = 2 * h * PLA + (wr * (BM / WR) + r) * 16;                // planes 2h, 2h + 1; + 512 per row block
  const int b_off = A_BYTES + (8 * h * BN + wc * (BN / WC) + r) * 4;        // + jj * BN * 4 per MFMA, + 128 per column block
  __builtin_amdgcn_s_barrier();
  int kt = 0, piece = 0, slot = 0;
  auto blocks_of = [&](int w) {                                             // 32-column blocks of this consumer inside a piece
    const int mine = w - wc * (BN / WC);
    return mine <= 0 ? 0 : (mine + 31) / 32;
  };
  int nj = blocks_of(piece_w(0));
  // The step's barrier sits in front of its LAST column block's MFMAs: by then every fragment of the stage is in registers,
  // so the slot can be handed back, and the next stage's first fragments are requested right behind the barrier -- their
  // LDS latency and the barrier's skew hide under that block's 8 TI MFMAs instead of opening every step.
  float fa[TI][8], fb[2][8];
  auto read_a = [&](const unsigned char* cur, float (&dst)[TI][8]) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(cur + a_off + q * PLA + i * 512);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[i][4 * q + t] = __uint_as_float(v[t]);
      }
  };
  auto read_b = [&](const unsigned char* cur, int j, float (&dst)[8]) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) dst[jj] = *reinterpret_cast<const float*>(cur + b_off + jj * BN * 4 + j * 128);
  };
  read_a(lds, fa);
  read_b(lds, 0, fb[0]);
  for (int s = 0; s < S; ++s) {
    const unsigned char* cur = lds + slot * STAGE;
    const int slot1 = slot == 2 ? 0 : slot + 1;
    float fa_next[TI][8];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      if (j + 1 < TJ) {
        read_b(cur, j + 1, fb[(j + 1) & 1]);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_a(lds + slot1 * STAGE, fa_next);                               // (past the last stage: a stale slot, unused)
        read_b(lds + slot1 * STAGE, 0, fb[TJ & 1]);
      }
      if (j < nj) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
          for (int i = 0; i < TI; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j & 1][jj], fa[i][jj], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++kt == nsteps) {
      const int q = piece_q(piece), c0 = piece_c(piece), w = piece_w(piece);
      const int batch = q / g.tiles_m, tm = q - batch * g.tiles_m;
      store_tile<TI, TJ>(acc, g.C + batch * g.sC, g.M, g.N, g.ldc, tm * BM + wr * (BM / WR), c0 + wc * (BN / WC), c0 + w, lane, 0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      kt = 0;
      ++piece;
      if (piece < npieces) nj = blocks_of(piece_w(piece));
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[i][t] = fa_next[i][t];
    slot = slot1;
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
    {256, 128, 16, 512, 2},   // 7: 2 x 4 waves of 128 x 32 within 128 registers, two workgroups per CU (half the B
                              //    conversions of the 128-row tiles when M >= 256)
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

struct Plan {
  int tile, bm, bn, bk, tiles_m, tiles_n, splits, k_per_split;
  bool prio;
  int abl;
  int stagger_mode, stagger_us;
};

// variant: -1 automatic; else tile + 10 * splits (splits 0 = automatic)
bool plan_for(int batch, int M, int N, int K, int transB, int variant, Plan* p) {
  p->stagger_mode = p->stagger_us = 0;
  if (variant >= 100000) {      // measurement: 100000 * mode + 10 * microseconds + tile (GemmArgs::stagger_*)
    p->stagger_mode = variant / 100000;
    p->stagger_us = (variant % 100000) / 10;
    if (p->stagger_mode > 2) return false;
    variant %= 10;
  }
  int tile = variant < 0 ? -1 : variant % 10;
  int splits = variant < 0 ? 0 : (variant / 10) % 100;
  p->prio = variant < 0 || variant < 1000 || variant >= 2000;   // + 1000: without the static wave priorities (A/B)
  p->abl = variant >= 2000 ? variant / 1000 - 1 : 0;             // + 2000 / 3000 / 4000: ablation builds (tiles 0, 2; NN)
  if (tile >= kNumTiles) return false;
  const long cus = 256;
  // measured (tools/bench_gemm_split.py, profiles/r05/gemm_split_variants.txt): the 128 x 128 tiles with three
  // workgroups per CU and the split placed among the MFMAs everywhere, except the long reductions of the weight
  // gradient with at least 256 x 256 outputs: one 256 x 256 tile per CU, the reduction split over the chip
  if (tile < 0) tile = (transB && M >= 256 && N >= 256 && K >= 1024) ? 0 : 5;
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
// 3 = the generic tiled kernel (128 x 128 x 16, three per CU, the split of B among the MFMAs) with A staged through
// registers from the image packed for variant 2
constexpr PaTile kPaTiles[] = {{256, 128, 16, 512}, {256, 256, 16, 512}, {128, 128, 16, 256}, {128, 128, 16, 256}};
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
  if (t == 3) {
    GemmArgs a{};
    a.A = static_cast<const float*>(Ap); a.B = B; a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = ldb; a.ldc = ldc; a.sA = 0; a.sB = sB; a.sC = sC;
    a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128; a.splits = 1; a.k_per_split = ((K + 15) / 16) * 16;
    a.s_split = 0; a.nt_c = 0; a.stages_a = (K + 15) / 16;
    a.packed_floats = (long)batch * a.tiles_m * a.stages_a * (3 * 2 * 128 * 16) / 4;
    FPSG_REQUIRE(a.packed_floats < (1L << 30), FPSG_E_LIMIT, "fpsg_gemm_split_nn_packed: the packed A must stay below 4 GiB");
    const long blocks = (long)batch * a.tiles_m * a.tiles_n;
    FPSG_REQUIRE(blocks < (1L << 30), FPSG_E_LIMIT, "fpsg_gemm_split_nn_packed: too many tiles");
    hipLaunchKernelGGL((gemm_split_kernel<128, 128, 16, 2, 2, 3, false, true, true, 0, true>), dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return launch_status("fpsg_gemm_split_nn_packed (register-staged)");
  }
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

extern "C" int fpsg_gemm_split_nn_persistent(const void* Ap, const float* B, float* C, int batch, int M, int N, int K, int ldb,
                                             int ldc, long sB, long sC, int variant, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && ldb >= N && ldc >= N, FPSG_E_SHAPE, "fpsg_gemm_split_nn_persistent: bad shape");
  FPSG_REQUIRE(variant >= -1 && variant <= 14, FPSG_E_SHAPE, "fpsg_gemm_split_nn_persistent: unknown variant %d", variant);
  FPSG_REQUIRE_PTR(Ap);
  FPSG_REQUIRE_PTR(B);
  FPSG_REQUIRE_PTR(C);
  const int bn = (variant == 1 || variant >= 6) ? 128 : 256;
  PersistArgs g;
  g.Ap = static_cast<const u32x4*>(Ap); g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldb = ldb; g.ldc = ldc; g.sB = sB; g.sC = sC;
  const int bm = variant >= 13 ? 128 : 256;
  g.tiles_m = (M + bm - 1) / bm; g.stages = (K + 15) / 16; g.batchq = batch * g.tiles_m;
  g.total = (long)g.batchq * N;
  g.nt_c = 0;     // (non-temporal stores measured and rejected: see fpsg_gemm_split)
  FPSG_REQUIRE(((long)(batch - 1) * sB + (long)K * ldb) < (1L << 30) && (long)M * ldc < (1L << 29) &&
                   (long)g.batchq * g.stages * 24576 < (1L << 31),
               FPSG_E_LIMIT, "fpsg_gemm_split_nn_persistent: B must stay below 4 GiB, one C matrix and packed A below 2 GiB");
  // one workgroup per CU; more (a multiple of 256) only when a range would hold more tiles than the piece table
  long grid = variant >= 13 ? 512 : 256;
  while ((g.total + grid - 1) / grid > (long)(kMaxPieces - 4) * bn) grid += 256;
  if (g.total / 32 < grid) grid = g.total / 32 > 0 ? g.total / 32 : 1;     // tiny problems: at least 32 columns each
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool ktail = K % 16 != 0;
  if (variant >= 6) {
    FPSG_REQUIRE(ldb % 4 == 0 && sB % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0, FPSG_E_SHAPE,
                 "fpsg_gemm_split_nn_persistent: the specialised-wave form moves B rows by 16-byte DMA (ldb, sB, B aligned to 4 floats)");
    // 6 / 12: 256 x 128 tiles, one workgroup per CU (12: staggered first pieces); 13 / 14: 128 x 128 tiles, two per CU
    // (A packed for 128-row tiles: fpsg_gemm_split_pack_a variant 2); 7..11: ablation builds of 6 (ABL 1, 6, 3, 4, 5)
    const dim3 gr((unsigned)grid), bl(512);
#define FPSG_WS(BM_, RA_, RR_, MINW_, ABL_)                                                                               \
  do {                                                                                                                    \
    if (ktail) hipLaunchKernelGGL((gemm_split_ws_kernel<BM_, 128, RA_, RR_, MINW_, true, ABL_>), gr, bl, 0, s, g);        \
    else hipLaunchKernelGGL((gemm_split_ws_kernel<BM_, 128, RA_, RR_, MINW_, false, ABL_>), gr, bl, 0, s, g);             \
  } while (0)
    switch (variant) {
      case 12: g.nt_c |= 2; [[fallthrough]];
      case 6: FPSG_WS(256, 3, 3, 2, 0); break;
      case 14: g.nt_c |= 2; [[fallthrough]];
      case 13: FPSG_WS(128, 2, 3, 4, 0); break;
      case 7: FPSG_WS(256, 3, 3, 2, 1); break;
      case 8: FPSG_WS(256, 3, 3, 2, 6); break;
      case 9: FPSG_WS(256, 3, 3, 2, 3); break;
      case 10: FPSG_WS(256, 3, 3, 2, 4); break;
      default: FPSG_WS(256, 3, 3, 2, 5); break;
    }
#undef FPSG_WS
    return launch_status("fpsg_gemm_split_nn_persistent (specialised waves)");
  }
  if (variant >= 2) {       // ablation builds (measurements: tools/bench_gemm_split.py --persistent 2,3,4,5)
    if (variant == 2) hipLaunchKernelGGL((gemm_split_pnn_kernel<256, false, 1>), dim3((unsigned)grid), dim3(512), 0, s, g);
    else if (variant == 3) hipLaunchKernelGGL((gemm_split_pnn_kernel<256, false, 2>), dim3((unsigned)grid), dim3(512), 0, s, g);
    else if (variant == 4) hipLaunchKernelGGL((gemm_split_pnn_kernel<256, false, 3>), dim3((unsigned)grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm_split_pnn_kernel<256, false, 4>), dim3((unsigned)grid), dim3(512), 0, s, g);
    return launch_status("fpsg_gemm_split_nn_persistent (ablation)");
  }
  if (bn == 256) {
    if (ktail) hipLaunchKernelGGL((gemm_split_pnn_kernel<256, true>), dim3((unsigned)grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm_split_pnn_kernel<256, false>), dim3((unsigned)grid), dim3(512), 0, s, g);
  } else {
    if (ktail) hipLaunchKernelGGL((gemm_split_pnn_kernel<128, true>), dim3((unsigned)grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm_split_pnn_kernel<128, false>), dim3((unsigned)grid), dim3(512), 0, s, g);
  }
  return launch_status("fpsg_gemm_split_nn_persistent");
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
  GemmArgs g{};
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
  g.stagger_mode = p.stagger_mode; g.stagger_per_cu = kTiles[p.tile].per_cu; g.stagger_ticks = p.stagger_us * 100;
  g.nt_c = 0;     // measured: non-temporal stores of an output beyond the Infinity Cache (128 -> 128 @112: 535 MB) 260 -> 480 us
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
    case 6: FPSG_GEMM_LAUNCH_PIPE(128, 256, 16, 1, 4, 2); break;
    default: FPSG_GEMM_LAUNCH_PIPE(256, 128, 16, 2, 4, 4); break;
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

extern "C" int fpsg_gemm_f32_nn(const float* A, const float* B, float* C, int batch, int M, int N, int K, int lda, int ldb,
                                int ldc, long sA, long sB, long sC, int variant, fpsg_stream_t stream) {
  using namespace fpsg;
  FPSG_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && lda >= K && ldb >= N && ldc >= N, FPSG_E_SHAPE, "fpsg_gemm_f32_nn: bad shape");
  FPSG_REQUIRE(variant >= -1 && variant <= 3, FPSG_E_SHAPE, "fpsg_gemm_f32_nn: unknown variant %d", variant);
  FPSG_REQUIRE_PTR(A);
  FPSG_REQUIRE_PTR(B);
  FPSG_REQUIRE_PTR(C);
  FPSG_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0,
               FPSG_E_ALIGN, "fpsg_gemm_f32_nn: K, lda, ldb, sA, sB must be multiples of 4 floats and A, B 16-byte aligned (16-byte DMA pieces)");
  F32Args g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  // variants: 0 / 1 = 256 x 128 tiles, one workgroup per CU (1: staggered first pieces); 2 / 3 = 128 x 128, two per CU
  const bool small = variant < 0 || variant >= 2;
  const int bm = small ? 128 : 256;
  g.tiles_m = (M + bm - 1) / bm; g.stages = (K + 15) / 16; g.batchq = batch * g.tiles_m;
  g.total = (long)g.batchq * N;
  g.stagger = variant == 1 || variant == 3;
  FPSG_REQUIRE(((long)(batch - 1) * sA + (long)M * lda) < (1L << 30) && ((long)(batch - 1) * sB + (long)K * ldb) < (1L << 30) &&
                   (long)M * ldc < (1L << 29),
               FPSG_E_LIMIT, "fpsg_gemm_f32_nn: A and B must stay below 4 GiB in total, one C matrix below 2 GiB");
  long grid = small ? 512 : 256;
  while ((g.total + grid - 1) / grid > (long)(kMaxPieces - 4) * 128) grid += 256;
  if (g.total / 32 < grid) grid = g.total / 32 > 0 ? g.total / 32 : 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 gr((unsigned)grid), bl(512);
  if (small) {
    if (K % 16 != 0) hipLaunchKernelGGL((gemm_f32_ws_kernel<128, 128, 2, 4, true>), gr, bl, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_ws_kernel<128, 128, 2, 4, false>), gr, bl, 0, s, g);
  } else {
    if (K % 16 != 0) hipLaunchKernelGGL((gemm_f32_ws_kernel<256, 128, 4, 2, true>), gr, bl, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_ws_kernel<256, 128, 4, 2, false>), gr, bl, 0, s, g);
  }
  return launch_status("fpsg_gemm_f32_nn");
}
