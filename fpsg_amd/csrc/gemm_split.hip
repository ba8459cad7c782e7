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

constexpr int kThreads = 512;

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

constexpr unsigned kOut = 0x7fffffffu;     // a lane offset beyond every buffer: the load returns 0, the store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// One operand tile of R rows x BK k, K-contiguous in memory ([rows][K]): item = (row, k-group), 8 consecutive k per
// item as two 16-byte buffer loads: descriptor + the lane's fixed byte offset + the step's scalar offset -- no address
// arithmetic in the loop; a row beyond the matrix has an offset beyond the buffer and reads zeros.
template <int R, int BK>
struct ContigStage {
  static constexpr int KG = BK / 8;
  static constexpr int PS = R * 16 + 128 / KG;          // plane stride (bytes)
  static constexpr int BYTES = 3 * KG * PS;
  static constexpr int ITEMS = R * KG / kThreads;
  static_assert(R * KG % kThreads == 0, "tile / thread mismatch");
  float v[ITEMS][8];
  unsigned off[ITEMS];                                   // byte offset of the item's first element at k = 0

  __device__ __forceinline__ void init(int ld, int row0, int rows, int tid) {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      const int row = row0 + it / KG;
      off[n] = row < rows ? ((unsigned)row * (unsigned)ld + (it % KG) * 8) * 4u : kOut;
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
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
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
template <int R, int BK>
struct StridedStage {
  static constexpr int KG = BK / 8;
  static constexpr int PS = R * 16 + 128 / KG;
  static constexpr int BYTES = 3 * KG * PS;
  static constexpr int ITEMS = R * KG / kThreads;
  static_assert(R * KG % kThreads == 0, "tile / thread mismatch");
  float v[ITEMS][8];
  unsigned off[ITEMS];
  unsigned ld4;

  __device__ __forceinline__ void init(int ld, int col0, int cols, int tid) {
    ld4 = (unsigned)ld * 4u;
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
      const int col = col0 + it % R;
      off[n] = col < cols ? (unsigned)col * 4u + (unsigned)(it / R) * 8u * ld4 : kOut;
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
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
#pragma unroll
    for (int n = 0; n < ITEMS; ++n) {
      const int it = tid + n * kThreads;
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

template <int BM, int BN, int BK, bool TRANSB>
__global__ __launch_bounds__(kThreads) void gemm_split_kernel(const GemmArgs g) {
  constexpr int WR = 2, WC = 4;                        // waves: rows x columns
  constexpr int WM = BM / WR, WN = BN / WC, TI = WM / 32, TJ = WN / 32;
  constexpr int KG = BK / 8;
  using StageA = ContigStage<BM, BK>;
  using StageB = typename std::conditional<TRANSB, ContigStage<BN, BK>, StridedStage<BN, BK>>::type;
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
    ra.load(rsA, k0);
    rb.load(rsB, k0);
  };
  auto store = [&](int kt) {                            // registers -> split -> LDS stage kt & 1
    const int k0 = k_begin + kt * BK;
    if (k0 + BK > k_end) {                              // workgroup-uniform
      ra.mask_tail(k0, k_end, tid);
      rb.mask_tail(k0, k_end, tid);
    }
    unsigned char* d = lds + (kt & 1) * STAGE;
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
      bf16x8 fa[3][TI], fb[3][TJ];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
          fa[t][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + a_off + (t * KG + 2 * s) * PSA + i * 512));
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          fb[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + b_off + (t * KG + 2 * s) * PSB + j * 512));
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  };

  // Stage kt+1 is split and written (into the buffer step kt-1 read: the barrier behind it has passed) while stage kt
  // is multiplied; its global loads were issued one step earlier, those of stage kt+2 follow into the freed registers.
  // The two waves of a SIMD (w and w+4) take the two halves in opposite order, so that one's vector work runs beside
  // the other's MFMAs instead of both idling the matrix pipe together.
  load(0);
  store(0);
  if (nsteps > 1) load(1);
  __syncthreads();
  const bool convert_first = wave < 4;
  for (int kt = 0; kt < nsteps; ++kt) {
    if (convert_first) {
      if (kt + 1 < nsteps) store(kt + 1);
      if (kt + 2 < nsteps) load(kt + 2);
      compute(kt);
    } else {
      compute(kt);
      if (kt + 1 < nsteps) store(kt + 1);
      if (kt + 2 < nsteps) load(kt + 2);
    }
    __syncthreads();
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

struct Plan {
  int bm, bn, bk, tiles_m, tiles_n, splits, k_per_split;
};

// variant: -1 automatic; else tile + 10 * splits with tile 0 = 256x256x16, 1 = 256x128x32 (splits 0 = automatic)
bool plan_for(int batch, int M, int N, int K, int transB, int variant, Plan* p) {
  int tile = variant < 0 ? -1 : variant % 10;
  int splits = variant < 0 ? 0 : variant / 10;
  if (tile > 1) return false;
  const long cus = 256;
  if (tile < 0) {
    // the smaller tile when the large one would leave the chip's last round less than 60 % full (or fill no round)
    const long t0 = (long)batch * ((M + 255) / 256) * ((N + 255) / 256);
    const long t1 = (long)batch * ((M + 255) / 256) * ((N + 127) / 128);
    const double e0 = (double)t0 / (double)(((t0 + cus - 1) / cus) * cus);
    const double e1 = (double)t1 / (double)(((t1 + cus - 1) / cus) * cus);
    tile = (transB || e0 >= e1 - 0.02) ? 0 : 1;
  }
  p->bm = 256;
  p->bn = tile == 0 ? 256 : 128;
  p->bk = tile == 0 ? 16 : 32;
  p->tiles_m = (M + p->bm - 1) / p->bm;
  p->tiles_n = (N + p->bn - 1) / p->bn;
  const long tiles = (long)batch * p->tiles_m * p->tiles_n;
  if (splits <= 0) {
    splits = 1;
    if (transB && tiles < cus) {                          // long reduction, few tiles: fill the chip with K ranges
      splits = (int)(cus / tiles);
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
  const dim3 grid((unsigned)blocks), block(kThreads);
  if (p.bn == 256) {
    if (transB) hipLaunchKernelGGL((gemm_split_kernel<256, 256, 16, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_split_kernel<256, 256, 16, false>), grid, block, 0, s, g);
  } else {
    if (transB) hipLaunchKernelGGL((gemm_split_kernel<256, 128, 32, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_split_kernel<256, 128, 32, false>), grid, block, 0, s, g);
  }
  int rc = launch_status("fpsg_gemm_split");
  if (rc != 0 || p.splits == 1) return rc;
  const long n = (long)batch * M * N;
  FPSG_REQUIRE(ldc == N, FPSG_E_SHAPE, "fpsg_gemm_split: dense output expected");
  const int rblocks = (int)std::min<long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ws, C, n, g.s_split, p.splits);
  return launch_status("fpsg_gemm_split (reduce)");
}
