// chamfer_tiled.hip -- K1, one-pass form: every pair distance d(i,j) of a cloud pair is evaluated
// ONCE and serves both directions (row minimum for xyz1[i], column minimum for xyz2[j]); and the
// sorted backward.  gfx950 (MI355X).  Replaces Kaolin 0.9.0's sided_distance forward / backward behind
// kaolin.metrics.pointcloud.chamfer_distance (reference call sites src/models/few_shot.py:110,117,167).
//
// d(i,j) = fma(dz,dz, fma(dy,dy, dx*dx)) with dx = xyz2[j].x - xyz1[i].x is bit-symmetric in the two
// directions (the other direction negates dx, dy, dz; squares and the fma chain are unchanged), so the
// two-pass kernel of chamfer.hip evaluates every pair twice.  Here:
//
//   chamfer_tile_kernel<R,W>: a workgroup owns a 2-D tile: 64*R rows (points of xyz1; R consecutive
//     rows per lane, in VGPRs) x W*cpw*16 candidates (points of xyz2, staged once in LDS as SoA and
//     read as broadcast ds_read_b128 = 4 candidates per instruction; wave w scans its own cpw chunks of
//     16).  Only the tile's 3 KB of candidates are staged (the two-pass kernel stages the whole 24 KB
//     cloud in every workgroup).  Per chunk and lane: 16*R distances (packed FP32, 3 instructions per
//     pair), the row side's running chunk minimum (v_min3: 0.5 per pair) and the column side's minimum
//     over the lane's R rows per candidate (v_min3: 0.5 per pair).  The 16 per-lane column minima are
//     then reduced across the 64 lanes through a wave-private LDS transpose: lane (s,c) reads the 16
//     partials of candidate c from the lanes of segment s, keeps the minimum and the first lane that
//     holds it; two butterfly steps merge the four segments (lower lane on ties).  Partial results are
//     32-bit keys (round 4; 64-bit before: the partial keys were 6.6x the op's algorithmic HBM bytes)
//       rows:    dist_bits & ~63 | candidate chunk inside the tile (< 64) -> row_keys[b][cs][i]
//       columns: dist_bits & ~63 | lane = group of R rows inside the tile -> col_keys[b][rt][j]
//     i.e. the tile's exact minimum with its six lowest mantissa bits replaced by where it was found.
//   chamfer_finalize_kernel<LOSS>: the truncation is monotone, so the tile that holds a point's true minimum is
//     among those whose truncated distance equals the smallest truncated distance (nearly always exactly one);
//     the kernel re-evaluates the named range of every such tile with the identical arithmetic, strict '<',
//     ascending tile and index (16 candidates for a row; R rows for a column) = Kaolin's first-minimum rule,
//     bit for bit.  LOSS: it also leaves the sum of each block's 256 distances (fixed tree) for the episode's
//     loss sums (chamfer_loss_reduce_kernel; fpsg_chamfer_fwd_tiled_losses).
//
// Instruction count: ~4.7 per d(i,j) = 2.35 per directed pair evaluation against 3.7 for the two-pass
// kernel.
//
//   chamfer_bwd_sorted_kernel: one workgroup per (cloud pair, side).  The other side's argmin list is
//     inverted by SORTING (target << 12 | source) keys with a bitonic network in LDS (no atomics, time
//     independent of how many sources share a target); every output point then walks its contiguous run
//     in ascending source index: ga_i = 2 g_a[i] (a_i - b[idx_a[i]]), then fma(2 g_b[j], a_i - b_j, .)
//     for j ascending -- the order of the oracle's sequential loop.
#include "fpsg_common.h"

namespace fpsg {
namespace {

constexpr int kChunk = 16;          // candidates per chunk (row-side argmin granule)
constexpr int kTStride = 20;        // floats per lane in the transpose buffer (16 + pad, 16-B aligned)
constexpr float kRowPad = 3.0e38f;  // coordinates of padded rows: every distance overflows to +inf

__device__ __forceinline__ float sq_dist(float qx, float qy, float qz, float cx, float cy, float cz) {
  float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return fma_rn(dz, dz, fma_rn(dy, dy, dx * dx));
}

// Squared distances of TWO rows (their coordinates share register pairs: q?.x = row a, q?.y = row b) to
// FOUR candidates (X, Y, Z = x, y, z of candidates 0..3): d = fma(dz,dz, fma(dy,dy, dx*dx)), dx = c - q,
// two candidates per packed instruction -- bit-identical to the scalar form.  Written out as one block:
//   * the row coordinate is splatted by op_sel from the shared pair (the compiler materialises {q,q} pairs:
//     48 VGPRs at R = 8) -- low half for row a, high half for row b;
//   * the four dependent chains (row a / b x candidates 01 / 23) are interleaved, so no packed operation
//     follows its producer (hipcc schedules the chains depth-first under register pressure and pads every
//     dependent pair with s_nop).  Outputs are early-clobber: they are written before the last input is read.
__device__ __forceinline__ void dist_2rows_4cands(v4f X, v4f Y, v4f Z, v2f qx, v2f qy, v2f qz,
                                                  v2f& a01, v2f& a23, v2f& b01, v2f& b23) {
  v2f t0, t1, t2, t3;
  const v2f x01 = X.xy, x23 = X.zw, y01 = Y.xy, y23 = Y.zw, z01 = Z.xy, z23 = Z.zw;
  asm volatile(
      "v_pk_add_f32 %0, %8, %14 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // a01 = x01 - qx.a
      "v_pk_add_f32 %1, %9, %14 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // a23 = x23 - qx.a
      "v_pk_add_f32 %2, %8, %14 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // b01 = x01 - qx.b
      "v_pk_add_f32 %3, %9, %14 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // b23 = x23 - qx.b
      "v_pk_add_f32 %4, %10, %15 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"    // t0 = y01 - qy.a
      "v_pk_add_f32 %5, %11, %15 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"    // t1 = y23 - qy.a
      "v_pk_add_f32 %6, %10, %15 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    // t2 = y01 - qy.b
      "v_pk_add_f32 %7, %11, %15 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    // t3 = y23 - qy.b
      "v_pk_mul_f32 %0, %0, %0\n\t"                                                              // dx*dx
      "v_pk_mul_f32 %1, %1, %1\n\t"
      "v_pk_mul_f32 %2, %2, %2\n\t"
      "v_pk_mul_f32 %3, %3, %3\n\t"
      "v_pk_fma_f32 %0, %4, %4, %0\n\t"                                                          // fma(dy,dy,.)
      "v_pk_fma_f32 %1, %5, %5, %1\n\t"
      "v_pk_fma_f32 %2, %6, %6, %2\n\t"
      "v_pk_fma_f32 %3, %7, %7, %3\n\t"
      "v_pk_add_f32 %4, %12, %16 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"    // t0 = z01 - qz.a
      "v_pk_add_f32 %5, %13, %16 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %6, %12, %16 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %7, %13, %16 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %4, %4, %0\n\t"                                                          // fma(dz,dz,.)
      "v_pk_fma_f32 %1, %5, %5, %1\n\t"
      "v_pk_fma_f32 %2, %6, %6, %2\n\t"
      "v_pk_fma_f32 %3, %7, %7, %3"
      : "=&v"(a01), "=&v"(a23), "=&v"(b01), "=&v"(b23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(x01), "v"(x23), "v"(y01), "v"(y23), "v"(z01), "v"(z23), "v"(qx), "v"(qy), "v"(qz));
}

// XCD-aware work order: workgroup ids are dealt round-robin over the 8 XCDs; XCD x takes the x-th
// contiguous eighth of the (cloud-pair-major) work list, so one pair's tiles share an L2 (speed only).
__device__ __forceinline__ int xcd_work_index() {
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
}

// LDS: [3][CW] candidate SoA | [W][64][kTStride] transpose buffers | [W][64*R] u64 row keys (W > 1)
template <int R, int W>
__global__ __launch_bounds__(64 * W) void chamfer_tile_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, int N, int M, int RT, int CS,
    int cpw, unsigned* __restrict__ row_keys, unsigned* __restrict__ col_keys) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int CW = W * cpw * kChunk;
  const int work = xcd_work_index();
  const int cs = work % CS;
  const int rt = (work / CS) % RT;
  const int b = work / (CS * RT);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  float* lx = lds;
  float* ly = lds + CW;
  float* lz = lds + 2 * CW;
  float* tbuf = lds + 3 * CW + wave * (64 * kTStride);

  const float* __restrict__ P1 = xyz1 + (size_t)b * N * 3;
  const float* __restrict__ P2 = xyz2 + (size_t)b * M * 3;

  // ---- this lane's R consecutive rows
  const int row0 = rt * (64 * R) + lane * R;
  static_assert(R % 2 == 0, "rows are kept as register pairs");
  v2f qx[R / 2], qy[R / 2], qz[R / 2];              // .x = row 2p, .y = row 2p + 1
  {
    const float* src = P1 + (size_t)row0 * 3;
    const bool vec = ((reinterpret_cast<uintptr_t>(P1 + (size_t)rt * (64 * R) * 3) & 15) == 0) && ((3 * R) % 4 == 0);
    float f[3 * R];
    if (vec && row0 + R <= N) {
      const v4f* s4 = reinterpret_cast<const v4f*>(src);
#pragma unroll
      for (int u = 0; u < (3 * R) / 4; ++u) {
        const v4f v = s4[u];
        f[4 * u + 0] = v.x; f[4 * u + 1] = v.y; f[4 * u + 2] = v.z; f[4 * u + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 3 * R; ++e) f[e] = (row0 + e / 3 < N) ? src[e] : kRowPad;
    }
#pragma unroll
    for (int p2 = 0; p2 < R / 2; ++p2) {
      qx[p2].x = f[6 * p2 + 0]; qy[p2].x = f[6 * p2 + 1]; qz[p2].x = f[6 * p2 + 2];
      qx[p2].y = f[6 * p2 + 3]; qy[p2].y = f[6 * p2 + 4]; qz[p2].y = f[6 * p2 + 5];
    }
  }

  // ---- stage this tile's candidates [c0, c0+CW) as SoA, +inf padded (never wins a strict '<')
  const int c0 = cs * CW;
  {
    const int remain = M - c0;
    const int valid = (remain < CW ? (remain > 0 ? remain : 0) : CW) * 3;
    const float* __restrict__ src = P2 + (size_t)c0 * 3;
    if (((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((valid & 3) == 0)) {
      const v4f* __restrict__ src4 = reinterpret_cast<const v4f*>(src);
      for (int e4 = tid; e4 < (3 * CW) / 4; e4 += 64 * W) {
        const int e = 4 * e4;
        v4f v = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
        if (e < valid) v = src4[e4];
        int j = e / 3;
        int c = e - 3 * j;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          lds[c * CW + j] = v[u];
          if (++c == 3) { c = 0; ++j; }
        }
      }
    } else {
      for (int e = tid; e < 3 * CW; e += 64 * W) {
        const float v = e < valid ? src[e] : __builtin_inff();
        const int j = e / 3;
        const int c = e - 3 * j;
        lds[c * CW + j] = v;
      }
    }
  }
  __syncthreads();

  float best[R];
  int bestc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { best[r] = __builtin_inff(); bestc[r] = c0 / kChunk; }

  const int seg = lane >> 4;          // 16-lane segment = 16*R consecutive rows
  const int cl = lane & 15;           // candidate of the chunk this lane reduces

  for (int ch = 0; ch < cpw; ++ch) {
    const int cbase = (wave * cpw + ch) * kChunk;      // first candidate of the chunk inside the tile
    float cm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cm[r] = __builtin_inff();
    float cp[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) cp[u] = __builtin_inff();
    const v4f* px = reinterpret_cast<const v4f*>(lx + cbase);
    const v4f* py = reinterpret_cast<const v4f*>(ly + cbase);
    const v4f* pz = reinterpret_cast<const v4f*>(lz + cbase);
#pragma unroll
    for (int g = 0; g < kChunk / 4; ++g) {
      const v4f X = px[g], Y = py[g], Z = pz[g];
#pragma unroll
      for (int p2 = 0; p2 < R / 2; ++p2) {
        v2f a01, a23, b01, b23;
        dist_2rows_4cands(X, Y, Z, qx[p2], qy[p2], qz[p2], a01, a23, b01, b23);
        cm[2 * p2] = __builtin_fminf(__builtin_fminf(cm[2 * p2], a01.x), a01.y);
        cm[2 * p2] = __builtin_fminf(__builtin_fminf(cm[2 * p2], a23.x), a23.y);
        cm[2 * p2 + 1] = __builtin_fminf(__builtin_fminf(cm[2 * p2 + 1], b01.x), b01.y);
        cm[2 * p2 + 1] = __builtin_fminf(__builtin_fminf(cm[2 * p2 + 1], b23.x), b23.y);
        // column side: minimum over this lane's rows, two rows per v_min3
        cp[4 * g + 0] = __builtin_fminf(__builtin_fminf(cp[4 * g + 0], a01.x), b01.x);
        cp[4 * g + 1] = __builtin_fminf(__builtin_fminf(cp[4 * g + 1], a01.y), b01.y);
        cp[4 * g + 2] = __builtin_fminf(__builtin_fminf(cp[4 * g + 2], a23.x), b23.x);
        cp[4 * g + 3] = __builtin_fminf(__builtin_fminf(cp[4 * g + 3], a23.y), b23.y);
      }
    }
    // ---- rows: one compare + two selects per chunk
    const int chunk_global = (c0 + cbase) / kChunk;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool lt = cm[r] < best[r];
      bestc[r] = lt ? chunk_global : bestc[r];
      best[r] = lt ? cm[r] : best[r];
    }
    // ---- columns: transpose the 64 x 16 partial minima through the wave's LDS buffer.  One wave's LDS
    // operations execute in order, so the reads below see the stores above without a barrier.
    {
      v4f* tw = reinterpret_cast<v4f*>(tbuf + lane * kTStride);
#pragma unroll
      for (int u = 0; u < kChunk / 4; ++u) {
        const v4f v = {cp[4 * u + 0], cp[4 * u + 1], cp[4 * u + 2], cp[4 * u + 3]};
        tw[u] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float* tr = tbuf + (seg * 16) * kTStride + cl;
      float vt[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) vt[t] = tr[t * kTStride];
      __builtin_amdgcn_wave_barrier();          // the next chunk's stores come after these reads
      float v = __builtin_inff();
#pragma unroll
      for (int t = 0; t < 16; t += 2) v = __builtin_fminf(__builtin_fminf(v, vt[t]), vt[t + 1]);
      int ln = 15;                              // first lane of the segment that holds the minimum
#pragma unroll
      for (int t = 14; t >= 0; --t) ln = (vt[t] == v) ? t : ln;
      ln += seg * 16;
#pragma unroll
      for (int sft = 16; sft <= 32; sft <<= 1) {
        const float ov = __shfl_xor(v, sft, 64);
        const int ol = __shfl_xor(ln, sft, 64);
        const bool take = (ov < v) || (ov == v && ol < ln);
        v = take ? ov : v;
        ln = take ? ol : ln;
      }
      const int cand = c0 + cbase + cl;
      if (lane < 16 && cand < M) {
        // low six bits: the lane whose R consecutive rows hold the column minimum
        col_keys[((size_t)b * RT + rt) * M + cand] = (__float_as_uint(v) & ~63u) | (unsigned)ln;
      }
    }
  }

  // ---- rows: merge the W waves' partial minima (exact 64-bit keys in LDS), store the tile's 32-bit row keys
  const unsigned chunk0 = (unsigned)(c0 / kChunk);       // bestc - chunk0 < W * cpw <= 36
  if (W > 1) {
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds + 3 * CW + W * 64 * kTStride);
#pragma unroll
    for (int r = 0; r < R; ++r)
      keys[wave * (64 * R) + lane * R + r] =
          ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)bestc[r];
    __syncthreads();
    for (int ql = tid; ql < 64 * R; ql += 64 * W) {
      const int i = rt * (64 * R) + ql;
      if (i >= N) continue;
      unsigned long long key = keys[ql];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const unsigned long long k2 = keys[w * (64 * R) + ql];
        key = k2 < key ? k2 : key;
      }
      row_keys[((size_t)b * CS + cs) * N + i] =
          ((unsigned)(key >> 32) & ~63u) | (((unsigned)key - chunk0) & 63u);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = row0 + r;
      if (i < N)
        row_keys[((size_t)b * CS + cs) * N + i] =
            (__float_as_uint(best[r]) & ~63u) | (((unsigned)bestc[r] - chunk0) & 63u);
    }
  }
}

constexpr int kFinThreads = 256;

// One thread per output point.  Blocks [0, nb1) of a cloud pair serve xyz1's points (rows), the
// others xyz2's points (columns).  A point's partial keys are scanned once (smallest truncated distance, the
// first tile that has it, how many have it); the first such tile's range is re-evaluated by every lane together,
// further ones (two tiles whose minima agree in all but the six lowest mantissa bits: rare) in a second loop.
// LOSS: block_sums[b * (nb1 + nb2) + blk] = the sum of the block's distances, points past the end as +0:
//   wave w = tree sum over its 64 lanes (wave_sum), block = ((w0 + w1) + w2) + w3.
template <bool LOSS>
__global__ __launch_bounds__(kFinThreads) void chamfer_finalize_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, int N, int M, int RT, int CS,
    int grp_rows, int tile_chunks, const unsigned* __restrict__ row_keys,
    const unsigned* __restrict__ col_keys, float* __restrict__ dist1,
    int32_t* __restrict__ idx1, float* __restrict__ dist2, int32_t* __restrict__ idx2,
    float* __restrict__ block_sums) {
  const int nb1 = (N + kFinThreads - 1) / kFinThreads;
  const int nb2 = (M + kFinThreads - 1) / kFinThreads;
  const int work = xcd_work_index();
  const int blk = work % (nb1 + nb2);
  const int b = work / (nb1 + nb2);
  const int tid = threadIdx.x;
  const float* __restrict__ P1 = xyz1 + (size_t)b * N * 3;
  const float* __restrict__ P2 = xyz2 + (size_t)b * M * 3;
  float mine = 0.f;                                      // this thread's distance (LOSS)

  if (blk < nb1) {
    // ---- a point of xyz1: best candidate chunk(s), 16 candidates each in ascending order
    const int i = blk * kFinThreads + tid;
    if (i < N) {
      const unsigned* kp = row_keys + ((size_t)b * CS) * N + i;
      unsigned tmin = 0xffffffffu, kfirst = 0;
      int sfirst = 0, same = 0;
      auto take = [&](unsigned k, int s) {
        const unsigned t = k & ~63u;
        same = t < tmin ? 1 : same + (t == tmin ? 1 : 0);
        sfirst = t < tmin ? s : sfirst;
        kfirst = t < tmin ? k : kfirst;
        tmin = t < tmin ? t : tmin;
      };
      int s = 0;
      for (; s + 4 <= CS; s += 4) {                      // four independent loads in flight
        const unsigned k0 = kp[(size_t)s * N], k1 = kp[(size_t)(s + 1) * N];
        const unsigned k2 = kp[(size_t)(s + 2) * N], k3 = kp[(size_t)(s + 3) * N];
        take(k0, s); take(k1, s + 1); take(k2, s + 2); take(k3, s + 3);
      }
      for (; s < CS; ++s) take(kp[(size_t)s * N], s);
      const float x = P1[3 * i + 0], y = P1[3 * i + 1], z = P1[3 * i + 2];
      float bd = __builtin_inff();
      int bi = 0;
      auto chunk = [&](int j0) {
        float cx[kChunk], cy[kChunk], cz[kChunk];
        if (((reinterpret_cast<uintptr_t>(P2) & 15) == 0) && j0 + kChunk <= M) {
          // a chunk is 192 contiguous, 16-byte aligned bytes: twelve 16-byte loads instead of 48 scalar ones
          float f[3 * kChunk];
          const v4f* c4 = reinterpret_cast<const v4f*>(P2 + 3 * j0);
#pragma unroll
          for (int u = 0; u < (3 * kChunk) / 4; ++u) {
            const v4f v = c4[u];
            f[4 * u] = v.x; f[4 * u + 1] = v.y; f[4 * u + 2] = v.z; f[4 * u + 3] = v.w;
          }
#pragma unroll
          for (int u = 0; u < kChunk; ++u) { cx[u] = f[3 * u]; cy[u] = f[3 * u + 1]; cz[u] = f[3 * u + 2]; }
        } else {
#pragma unroll
          for (int u = 0; u < kChunk; ++u) {             // all loads first; candidates past M read the last one
            const int j = (j0 + u) < M ? (j0 + u) : (M - 1);
            cx[u] = P2[3 * j + 0]; cy[u] = P2[3 * j + 1]; cz[u] = P2[3 * j + 2];
          }
        }
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
          const float d = (j0 + u) < M ? sq_dist(x, y, z, cx[u], cy[u], cz[u]) : __builtin_inff();
          const bool lt = d < bd;
          bi = lt ? j0 + u : bi;
          bd = lt ? d : bd;
        }
      };
      chunk((sfirst * tile_chunks + (int)(kfirst & 63u)) * kChunk);
      if (same > 1) {
        for (s = sfirst + 1; s < CS; ++s) {
          const unsigned k = kp[(size_t)s * N];
          if ((k & ~63u) == tmin) chunk((s * tile_chunks + (int)(k & 63u)) * kChunk);
        }
      }
      dist1[(size_t)b * N + i] = bd;
      idx1[(size_t)b * N + i] = bi;
      mine = bd;
    }
  } else {
    // ---- a point of xyz2: best group(s) of grp_rows consecutive rows (one lane's rows of a tile)
    const int j = (blk - nb1) * kFinThreads + tid;
    if (j < M) {
      const unsigned* kp = col_keys + ((size_t)b * RT) * M + j;
      unsigned tmin = 0xffffffffu, kfirst = 0;
      int tfirst = 0, same = 0;
      auto take = [&](unsigned k, int t_) {
        const unsigned t = k & ~63u;
        same = t < tmin ? 1 : same + (t == tmin ? 1 : 0);
        tfirst = t < tmin ? t_ : tfirst;
        kfirst = t < tmin ? k : kfirst;
        tmin = t < tmin ? t : tmin;
      };
      int t = 0;
      for (; t + 4 <= RT; t += 4) {
        const unsigned k0 = kp[(size_t)t * M], k1 = kp[(size_t)(t + 1) * M];
        const unsigned k2 = kp[(size_t)(t + 2) * M], k3 = kp[(size_t)(t + 3) * M];
        take(k0, t); take(k1, t + 1); take(k2, t + 2); take(k3, t + 3);
      }
      for (; t < RT; ++t) take(kp[(size_t)t * M], t);
      const float x = P2[3 * j + 0], y = P2[3 * j + 1], z = P2[3 * j + 2];
      float bd = __builtin_inff();
      int bi = 0;
      // the reference's second direction evaluates d(q = xyz2[j], c = xyz1[i]); same bits either way
      auto group = [&](int i0) {
        for (int u0 = 0; u0 < grp_rows; u0 += 4) {       // grp_rows is 4 or 8
          float rx[4], ry[4], rz[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = (i0 + u0 + u) < N ? (i0 + u0 + u) : (N - 1);
            rx[u] = P1[3 * i + 0]; ry[u] = P1[3 * i + 1]; rz[u] = P1[3 * i + 2];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float d = (i0 + u0 + u) < N ? sq_dist(x, y, z, rx[u], ry[u], rz[u]) : __builtin_inff();
            const bool lt = d < bd;
            bi = lt ? i0 + u0 + u : bi;
            bd = lt ? d : bd;
          }
        }
      };
      group((tfirst * 64 + (int)(kfirst & 63u)) * grp_rows);
      if (same > 1) {
        for (t = tfirst + 1; t < RT; ++t) {
          const unsigned k = kp[(size_t)t * M];
          if ((k & ~63u) == tmin) group((t * 64 + (int)(k & 63u)) * grp_rows);
        }
      }
      dist2[(size_t)b * M + j] = bd;
      idx2[(size_t)b * M + j] = bi;
      mine = bd;
    }
  }

  if (LOSS) {
    __shared__ float wsum[kFinThreads / 64];
    const float w = wave_sum(mine);
    if ((tid & 63) == 0) wsum[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) block_sums[(size_t)b * (nb1 + nb2) + blk] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
}

// The last stage of the episode's loss sums (few_shot.py:110-124), one workgroup.  Thread (b, side) adds that row's
// block sums in ascending block order from +0; cd_b = s1 / N + s2 / M; the two group sums (pairs below n_first / the
// rest; a pair outside the group counts as +0): lane l of wave 0 adds cd_b for b = l, l + 64, ... in ascending order,
// then the balanced tree over the lanes (wave_sum).  out3 = { q, r, w_first * q + w_rest * r }.
constexpr int kLossPairsMax = 4096;
constexpr int kRedThreads = 256;
__global__ __launch_bounds__(kRedThreads) void chamfer_loss_reduce_kernel(
    const float* __restrict__ block_sums, int B, int N, int M, int n_first, float w_first, float w_rest,
    float* __restrict__ out) {
  __shared__ float cd[kLossPairsMax];
  __shared__ float stage[2048];                          // block sums of up to kRedThreads / 2 pairs at a time
  const int nb1 = (N + kFinThreads - 1) / kFinThreads;   // <= 16 each (N, M <= 4096)
  const int nb2 = (M + kFinThreads - 1) / kFinThreads;
  const int nb = nb1 + nb2;
  const int tid = threadIdx.x;
  const int per = 2048 / nb < kRedThreads / 2 ? 2048 / nb : kRedThreads / 2;   // pairs per round
  for (int b0 = 0; b0 < B; b0 += per) {
    const int cnt = (B - b0 < per ? B - b0 : per) * nb;
    for (int e = tid; e < cnt; e += kRedThreads) stage[e] = block_sums[(size_t)b0 * nb + e];   // independent, coalesced
    __syncthreads();
    const int bl = tid >> 1, side = tid & 1;
    float sum = 0.f;
    if (bl < per && b0 + bl < B) {
      const float* p = stage + bl * nb + (side ? nb1 : 0);
      const int n = side ? nb2 : nb1;
      for (int k = 0; k < n; ++k) sum += p[k];
    }
    const float other = __uint_as_float(lane_xor<1>(__float_as_uint(sum)));
    if (side == 0 && bl < per && b0 + bl < B) cd[b0 + bl] = sum * (1.0f / (float)N) + other * (1.0f / (float)M);
    __syncthreads();
  }
  if (tid < 64) {
    float q = 0.f, r = 0.f;
    for (int b = tid; b < B; b += 64) {
      const float v = cd[b];
      if (b < n_first) q += v; else r += v;
    }
    q = wave_sum(q);
    r = wave_sum(r);
    if (tid == 0) {
      out[0] = q;
      out[1] = r;
      out[2] = w_first * q + w_rest * r;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Tile configuration.  variant = R_code * 100 + W * 10 + log2(cpw-ish); chosen from B, N, M.
struct TileCfg {
  int R, W, cpw, RT, CS;
};

inline bool tile_cfg(int B, int N, int M, int variant, TileCfg* c) {
  if (N > 4096 || M > 4096) return false;            // partial keys per point grow with N*M: larger clouds take the two-pass kernel
  int R, W, cpw;
  if (variant >= 0) {
    R = (variant / 100) ? 8 : 4;
    W = (variant / 10) % 10;
    cpw = variant % 10;
    if (!(W == 1 || W == 2 || W == 4) || cpw < 1) return false;
  } else {
    // From a sweep on MI355X at N = M = 2048 (profiles/r02/k1_variant_sweep.txt): below ~7 cloud pairs the
    // two-pass kernel's many small workgroups fill the chip better; then small tiles (R = 4, 2 chunks
    // per wave), then R = 8 with 4 and 8 chunks per wave.  One launch's time is a whole number of wave
    // rounds per SIMD, so the tile size follows the number of wave-sized work units.
    const double n_eq = (double)B * ((double)N + (double)M) / 4096.0;         // 2048-point cloud pairs
    if (n_eq < 7.0) return false;
    const double units = (double)B * ((N + 511) / 512) * ((M + 15) / 16) / 4096.0;   // chunk-units per 4 SIMD slots at R = 8
    W = 4;
    if (units < 1.7) { R = 4; cpw = 2; }
    else { R = 8; cpw = units >= 8.0 ? 8 : 4; }
  }
  c->R = R; c->W = W; c->cpw = cpw;
  c->RT = (N + 64 * R - 1) / (64 * R);
  c->CS = (M + W * cpw * kChunk - 1) / (W * cpw * kChunk);
  return true;
}

template <int R, int W>
int launch_tiles(const float* xyz1, const float* xyz2, int B, int N, int M, const TileCfg& c,
                 unsigned* row_keys, unsigned* col_keys, hipStream_t s) {
  const int CW = W * c.cpw * kChunk;
  const size_t lds_bytes = (size_t)3 * CW * 4 + (size_t)W * 64 * kTStride * 4 + (W > 1 ? (size_t)W * 64 * R * 8 : 0);
  dim3 grid((unsigned)((size_t)B * c.RT * c.CS));
  hipLaunchKernelGGL((chamfer_tile_kernel<R, W>), grid, dim3(64 * W), lds_bytes, s, xyz1, xyz2, N, M,
                     c.RT, c.CS, c.cpw, row_keys, col_keys);
  return launch_status("fpsg_chamfer_fwd_tiled (tiles)");
}

// ------------------------------------------------------------------------------------------------
// Backward.  Summation order of an output point i (Kaolin scatters with float atomics: no defined order):
//   the sources j with idx_b[j] == i in ascending j are cut into blocks of 32; block k is summed from +0 by
//   S_k = fma(2 g_b[j], a_i - b_j, S_k) in ascending j;  ga_i = own term, then ga_i += S_0, += S_1, ...
// Blocks are independent, so a point chosen by hundreds of sources (routine early in training, when the
// generated cloud is a small blob) is summed by many threads at once; the oracle follows the same order.
constexpr int kSortThreads = 1024;
constexpr int kSortMax = 4096;       // sources sorted in LDS; j needs 12 bits of the key
constexpr int kBlk = 32;             // sources per summation block

template <int M>
__device__ __forceinline__ void sort_step_lanes(unsigned& a, unsigned& b, bool asc, int tid) {
  const unsigned pa = lane_xor<M>(a), pb = lane_xor<M>(b);
  const bool keep_min = (((tid & M) == 0) == asc);
  a = keep_min ? (a < pa ? a : pa) : (a > pa ? a : pa);
  b = keep_min ? (b < pb ? b : pb) : (b > pb ? b : pb);
}

// Bitonic network over 2048 keys, two per thread (positions 2t, 2t+1), in registers: partners closer than
// 128 positions live in the same wave (lane exchange), the ten farthest steps go through LDS.
__device__ __forceinline__ void sort2048_regs(unsigned& a, unsigned& b, unsigned* xch, int tid) {
  int flip = 0;                                       // which half of xch the next LDS step uses
#pragma unroll
  for (int k = 2; k <= 2048; k <<= 1) {
    const bool asc = (((2 * tid) & k) == 0);          // same for both of the thread's positions (k >= 2)
#pragma unroll
    for (int j = k >> 1; j >= 128; j >>= 1) {          // partner thread tid ^ (j/2) in another wave
      const int m = j >> 1;
      unsigned* buf = xch + flip * 2048;
      flip ^= 1;
      buf[2 * tid] = a;
      buf[2 * tid + 1] = b;
      __syncthreads();
      const unsigned pa = buf[2 * (tid ^ m)], pb = buf[2 * (tid ^ m) + 1];
      const bool keep_min = (((tid & m) == 0) == asc);
      a = keep_min ? (a < pa ? a : pa) : (a > pa ? a : pa);
      b = keep_min ? (b < pb ? b : pb) : (b > pb ? b : pb);
    }
    if (k >= 128) sort_step_lanes<32>(a, b, asc, tid);
    if (k >= 64) sort_step_lanes<16>(a, b, asc, tid);
    if (k >= 32) sort_step_lanes<8>(a, b, asc, tid);
    if (k >= 16) sort_step_lanes<4>(a, b, asc, tid);
    if (k >= 8) sort_step_lanes<2>(a, b, asc, tid);
    if (k >= 4) sort_step_lanes<1>(a, b, asc, tid);
    // j == 1: the thread's own pair
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    a = asc ? lo : hi;
    b = asc ? hi : lo;
  }
}

// PAIR: the upstream gradient of a distance is one value per cloud pair and side -- the episode's loss sums
// (fpsg_chamfer_bwd_losses): g = g_total * (b < n_first ? w_first : w_rest) + (b < n_first ? g_first : g_rest),
// dist1's gradient g * (1/N), dist2's g * (1/M), formed here exactly as chamfer_loss_grads_kernel forms them (g1 = g_first,
// g2 = g_rest then point at device scalars, null = none) instead of being read from two constant [B,N] arrays.
struct PairGrad {
  const float* g_total;
  int n_first;
  float w_first, w_rest;
};

template <bool REG_SORT, bool PAIR>
__global__ __launch_bounds__(kSortThreads) void chamfer_bwd_sorted_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2,
    const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
    const float* __restrict__ g1, const float* __restrict__ g2, PairGrad pg, int N, int M, int NBP_max,
    float* __restrict__ gxyz1, float* __restrict__ gxyz2) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  __shared__ int n_blocks;
  const int work = xcd_work_index();
  const int side = work & 1;
  const int b = work >> 1;
  const int na = side ? M : N;
  const int nb = side ? N : M;
  const float* __restrict__ A = (side ? xyz2 : xyz1) + (size_t)b * na * 3;
  const float* __restrict__ Bc = (side ? xyz1 : xyz2) + (size_t)b * nb * 3;
  const int32_t* __restrict__ ia = (side ? idx2 : idx1) + (size_t)b * na;
  const int32_t* __restrict__ ib = (side ? idx1 : idx2) + (size_t)b * nb;
  const float* __restrict__ ga_up = PAIR ? nullptr : (side ? g2 : g1) + (size_t)b * na;
  const float* __restrict__ gb_up = PAIR ? nullptr : (side ? g1 : g2) + (size_t)b * nb;
  float ga_pair = 0.f, gb_pair = 0.f;                    // PAIR: the constants of this pair's two sides
  if (PAIR) {
    const bool first = b < pg.n_first;
    float g = 0.f;
    if (pg.g_total) g = *pg.g_total * (first ? pg.w_first : pg.w_rest);
    const float* own = first ? g1 : g2;
    if (own) g += *own;
    const float v1 = g * (1.0f / (float)N), v2 = g * (1.0f / (float)M);
    ga_pair = side ? v2 : v1;
    gb_pair = side ? v1 : v2;
  }
  float* __restrict__ out = (side ? gxyz2 : gxyz1) + (size_t)b * na * 3;

  // LDS (NBP_max = power of two >= max(N, M), >= 2048):
  unsigned* keys = smem;                                            // [NBP_max] sorted (target << 12 | source)
  int* start = reinterpret_cast<int*>(smem + NBP_max);              // [NBP_max] first sorted position of a target
  int* count = start + NBP_max;                                     // [NBP_max] its number of sources
  float* sb = reinterpret_cast<float*>(smem + 3 * NBP_max);         // [3 * NBP_max] the other cloud (AoS)
  float* sg = sb + 3 * NBP_max;                                     // [NBP_max] its upstream gradient
  unsigned* xch = smem + NBP_max;                                   // [2 * NBP_max] sort exchange: the start / count area, not yet in use
  unsigned* tbl = smem + 7 * NBP_max;                               // [5 * NBP_max / 16] block table + partial sums
  const int tid = threadIdx.x;

  for (int e = tid; e < 3 * nb; e += kSortThreads) sb[e] = Bc[e];
  if (!PAIR)
    for (int j = tid; j < nb; j += kSortThreads) sg[j] = gb_up[j];
  if (tid == 0) n_blocks = 0;

  auto make_key = [&](int j) -> unsigned {
    if (j >= nb) return 0xffffffffu;
    int t = ib[j];
    t = t < 0 ? 0 : (t >= na ? na - 1 : t);       // never index outside the cloud
    return ((unsigned)t << 12) | (unsigned)j;
  };

  // this thread's output points: their global data is fetched now, under the sort
  constexpr int kPts = kSortMax / kSortThreads;
  float apx[kPts], apy[kPts], apz[kPts], aup[kPts];
  int aidx[kPts];
#pragma unroll
  for (int r = 0; r < kPts; ++r) {
    const int i = tid + r * kSortThreads;
    const int ic = i < na ? i : na - 1;
    apx[r] = A[3 * ic + 0]; apy[r] = A[3 * ic + 1]; apz[r] = A[3 * ic + 2];
    aup[r] = PAIR ? ga_pair : ga_up[ic];
    aidx[r] = ia[ic];
  }

  int NBP;
  if (REG_SORT) {
    NBP = 2048;
    unsigned a = make_key(2 * tid), c = make_key(2 * tid + 1);
    sort2048_regs(a, c, xch, tid);
    keys[2 * tid] = a;
    keys[2 * tid + 1] = c;
    __syncthreads();
  } else {
    NBP = 64;
    while (NBP < nb) NBP <<= 1;
    for (int j = tid; j < NBP; j += kSortThreads) keys[j] = make_key(j);
    __syncthreads();
    for (int k = 2; k <= NBP; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int p = tid; p < (NBP >> 1); p += kSortThreads) {
          const int lo = ((p / j) * (j << 1)) + (p % j);
          const int hi = lo + j;
          const unsigned a = keys[lo], c = keys[hi];
          const bool up = (lo & k) == 0;
          if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
    }
  }

  // ---- runs: first position and end of every target's sources
  for (int i = tid; i < na; i += kSortThreads) { start[i] = 0; count[i] = 0; }
  __syncthreads();
  for (int p = tid; p < nb; p += kSortThreads) {
    const unsigned t = keys[p] >> 12;
    if (p == 0 || (keys[p - 1] >> 12) != t) start[t] = p;
    if (p == nb - 1 || (keys[p + 1] >> 12) != t) count[t] = p + 1;     // end for now
  }
  __syncthreads();

  int* blk_pos = reinterpret_cast<int*>(tbl);                       // [<= NBP/16] first sorted position of a block
  int* blk_tgt = blk_pos + (NBP_max >> 4);                          // its target
  float* part = reinterpret_cast<float*>(blk_tgt + (NBP_max >> 4)); // [<= NBP/16][3] its sum

  auto block_sum = [&](int p0, int cnt, float px, float py, float pz, float& sx, float& sy, float& sz) {
    sx = 0.0f; sy = 0.0f; sz = 0.0f;
    for (int u0 = 0; u0 < cnt; u0 += 4) {
      int jj[4];
      float tt[4], bx[4], by[4], bz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                 // loads of four sources in flight
        const int p = (u0 + u) < cnt ? (p0 + u0 + u) : p0;
        jj[u] = (int)(keys[p] & 0xfffu);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tt[u] = 2.0f * (PAIR ? gb_pair : sg[jj[u]]);
        bx[u] = sb[3 * jj[u] + 0]; by[u] = sb[3 * jj[u] + 1]; bz[u] = sb[3 * jj[u] + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u0 + u < cnt) {
          sx = fma_rn(tt[u], px - bx[u], sx);
          sy = fma_rn(tt[u], py - by[u], sy);
          sz = fma_rn(tt[u], pz - bz[u], sz);
        }
      }
    }
  };

  // ---- phase A: every output point: own term; a single block is summed right away, more blocks are queued
  float ax[kPts], ay[kPts], az[kPts];
  int slot[kPts], nblk[kPts];
#pragma unroll
  for (int r = 0; r < kPts; ++r) {
    const int i = tid + r * kSortThreads;
    nblk[r] = 0; slot[r] = 0;
    if (i >= na) continue;
    const float px = apx[r], py = apy[r], pz = apz[r];
    {
      int j = aidx[r];
      j = j < 0 ? 0 : (j >= nb ? nb - 1 : j);
      const float t = 2.0f * aup[r];
      ax[r] = t * (px - sb[3 * j + 0]);
      ay[r] = t * (py - sb[3 * j + 1]);
      az[r] = t * (pz - sb[3 * j + 2]);
    }
    const int p0 = start[i];
    const int n = count[i] ? count[i] - p0 : 0;
    nblk[r] = (n + kBlk - 1) / kBlk;
    if (nblk[r] == 1) {
      float sx, sy, sz;
      block_sum(p0, n, px, py, pz, sx, sy, sz);
      ax[r] += sx; ay[r] += sy; az[r] += sz;
    } else if (nblk[r] > 1) {
      slot[r] = atomicAdd(&n_blocks, nblk[r]);      // any order: the table records which block is whose
      for (int k = 0; k < nblk[r]; ++k) {
        blk_pos[slot[r] + k] = p0 + k * kBlk;
        blk_tgt[slot[r] + k] = i;
      }
    }
    if (nblk[r] <= 1) {
      out[3 * i + 0] = ax[r];
      out[3 * i + 1] = ay[r];
      out[3 * i + 2] = az[r];
    }
  }
  __syncthreads();

  // ---- phase B: one thread per queued block
  const int nq = n_blocks;
  for (int q = tid; q < nq; q += kSortThreads) {
    const int i = blk_tgt[q];
    const int p0 = blk_pos[q];
    const int end = count[i];
    const int cnt = (end - p0) < kBlk ? (end - p0) : kBlk;
    float sx, sy, sz;
    block_sum(p0, cnt, A[3 * i + 0], A[3 * i + 1], A[3 * i + 2], sx, sy, sz);
    part[3 * q + 0] = sx; part[3 * q + 1] = sy; part[3 * q + 2] = sz;
  }
  __syncthreads();

  // ---- phase C: points with several blocks add their partial sums in block order
#pragma unroll
  for (int r = 0; r < kPts; ++r) {
    if (nblk[r] <= 1) continue;
    const int i = tid + r * kSortThreads;
    for (int k = 0; k < nblk[r]; ++k) {
      ax[r] += part[3 * (slot[r] + k) + 0];
      ay[r] += part[3 * (slot[r] + k) + 1];
      az[r] += part[3 * (slot[r] + k) + 2];
    }
    out[3 * i + 0] = ax[r];
    out[3 * i + 1] = ay[r];
    out[3 * i + 2] = az[r];
  }
}

}  // namespace
}  // namespace fpsg

namespace fpsg {
namespace {

inline size_t tile_ws_bytes(int B, int N, int M, const TileCfg& c) {
  const size_t nblk = (size_t)((N + kFinThreads - 1) / kFinThreads + (M + kFinThreads - 1) / kFinThreads);
  // 32-bit partial keys of both sides, then the finalize blocks' distance sums (fpsg_chamfer_fwd_tiled_losses)
  return ((size_t)B * c.CS * N + (size_t)B * c.RT * M) * sizeof(unsigned) + (size_t)B * nblk * sizeof(float);
}

// tiles + finalize (+ the loss sums' last stage when out3 is given)
int tiled_forward(const char* who, const float* xyz1, const float* xyz2, int B, int N, int M, float* dist1,
                  int32_t* idx1, float* dist2, int32_t* idx2, void* ws, size_t ws_bytes, int variant,
                  bool loss, int n_first, float w_first, float w_rest, float* out3, fpsg_stream_t stream) {
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE, "%s: B,N,M must be positive (got %d,%d,%d)", who, B, N, M);
  FPSG_REQUIRE_PTR(xyz1); FPSG_REQUIRE_PTR(xyz2);
  FPSG_REQUIRE_PTR(dist1); FPSG_REQUIRE_PTR(idx1);
  FPSG_REQUIRE_PTR(dist2); FPSG_REQUIRE_PTR(idx2);
  TileCfg c;
  FPSG_REQUIRE(tile_cfg(B, N, M, variant, &c), FPSG_E_LIMIT,
               "%s: N=%d, M=%d (limit 4096 each) or variant %d unsupported; use fpsg_chamfer_fwd", who, N, M, variant);
  const size_t need = tile_ws_bytes(B, N, M, c);
  FPSG_REQUIRE(ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, FPSG_E_NULL,
               "%s: workspace missing or not 8-byte aligned", who);
  FPSG_REQUIRE(ws_bytes >= need, FPSG_E_LIMIT, "%s: workspace of %zu bytes, %zu needed", who, ws_bytes, need);
  FPSG_REQUIRE((size_t)B * c.RT * c.CS < (1u << 31), FPSG_E_LIMIT, "%s: grid too large", who);
  if (loss) {
    FPSG_REQUIRE_PTR(out3);
    FPSG_REQUIRE(B <= kLossPairsMax, FPSG_E_LIMIT, "%s: B=%d exceeds %d", who, B, kLossPairsMax);
    FPSG_REQUIRE(n_first >= 0 && n_first <= B, FPSG_E_SHAPE, "%s: n_first=%d outside [0,%d]", who, n_first, B);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned* row_keys = static_cast<unsigned*>(ws);
  unsigned* col_keys = row_keys + (size_t)B * c.CS * N;
  float* block_sums = reinterpret_cast<float*>(col_keys + (size_t)B * c.RT * M);
  int rc;
  if (c.R == 8) {
    rc = c.W == 4 ? launch_tiles<8, 4>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s)
       : c.W == 2 ? launch_tiles<8, 2>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s)
                  : launch_tiles<8, 1>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s);
  } else {
    rc = c.W == 4 ? launch_tiles<4, 4>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s)
       : c.W == 2 ? launch_tiles<4, 2>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s)
                  : launch_tiles<4, 1>(xyz1, xyz2, B, N, M, c, row_keys, col_keys, s);
  }
  if (rc != 0) return rc;
  const int nblk = (N + kFinThreads - 1) / kFinThreads + (M + kFinThreads - 1) / kFinThreads;
  const dim3 grid((unsigned)((size_t)B * nblk));
  if (loss)
    hipLaunchKernelGGL(chamfer_finalize_kernel<true>, grid, dim3(kFinThreads), 0, s, xyz1, xyz2, N, M, c.RT, c.CS, c.R,
                       c.W * c.cpw, row_keys, col_keys, dist1, idx1, dist2, idx2, block_sums);
  else
    hipLaunchKernelGGL(chamfer_finalize_kernel<false>, grid, dim3(kFinThreads), 0, s, xyz1, xyz2, N, M, c.RT, c.CS, c.R,
                       c.W * c.cpw, row_keys, col_keys, dist1, idx1, dist2, idx2, block_sums);
  rc = launch_status("fpsg_chamfer_fwd_tiled (finalize)");
  if (rc != 0 || !loss) return rc;
  hipLaunchKernelGGL(chamfer_loss_reduce_kernel, dim3(1), dim3(kRedThreads), 0, s, block_sums, B, N, M, n_first, w_first,
                     w_rest, out3);
  return launch_status("fpsg_chamfer_fwd_tiled_losses (sums)");
}

int sorted_backward(const char* who, bool pair, const float* xyz1, const float* xyz2, const int32_t* idx1,
                    const int32_t* idx2, const float* g1, const float* g2, PairGrad pg, int B, int N, int M,
                    float* gxyz1, float* gxyz2, fpsg_stream_t stream) {
  FPSG_REQUIRE(B > 0 && N > 0 && M > 0, FPSG_E_SHAPE, "%s: B,N,M must be positive (got %d,%d,%d)", who, B, N, M);
  FPSG_REQUIRE(N <= kSortMax && M <= kSortMax, FPSG_E_LIMIT, "%s: N=%d, M=%d beyond %d; use fpsg_chamfer_bwd", who, N,
               M, kSortMax);
  FPSG_REQUIRE(B < (1 << 30), FPSG_E_LIMIT, "%s: B=%d too large", who, B);
  FPSG_REQUIRE_PTR(xyz1); FPSG_REQUIRE_PTR(xyz2); FPSG_REQUIRE_PTR(idx1); FPSG_REQUIRE_PTR(idx2);
  FPSG_REQUIRE_PTR(gxyz1); FPSG_REQUIRE_PTR(gxyz2);
  const int nmax = N > M ? N : M;
  const int NBP = nmax <= 2048 ? 2048 : 4096;
  const size_t lds_bytes = (size_t)NBP * 4 * 7 + (size_t)(NBP / 16) * 4 * 5;   // keys, start, count, 3 coordinates, gradient; block table
  const void* fn = NBP == 2048
      ? (pair ? reinterpret_cast<const void*>(chamfer_bwd_sorted_kernel<true, true>)
              : reinterpret_cast<const void*>(chamfer_bwd_sorted_kernel<true, false>))
      : (pair ? reinterpret_cast<const void*>(chamfer_bwd_sorted_kernel<false, true>)
              : reinterpret_cast<const void*>(chamfer_bwd_sorted_kernel<false, false>));
  if (lds_bytes > 65536) {                              // dynamic LDS beyond 64 KiB has to be requested (no state kept)
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); return (int)e; }
  }
  const dim3 grid((unsigned)(2 * B)), block(kSortThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define FPSG_BWD_LAUNCH(RS, PR)                                                                                  \
  hipLaunchKernelGGL((chamfer_bwd_sorted_kernel<RS, PR>), grid, block, lds_bytes, s, xyz1, xyz2, idx1, idx2, g1, \
                     g2, pg, N, M, NBP, gxyz1, gxyz2)
  if (NBP == 2048) { if (pair) FPSG_BWD_LAUNCH(true, true); else FPSG_BWD_LAUNCH(true, false); }
  else             { if (pair) FPSG_BWD_LAUNCH(false, true); else FPSG_BWD_LAUNCH(false, false); }
#undef FPSG_BWD_LAUNCH
  return launch_status(who);
}

}  // namespace
}  // namespace fpsg

extern "C" size_t fpsg_chamfer_workspace_bytes(int B, int N, int M, int variant) {
  fpsg::TileCfg c;
  if (B <= 0 || N <= 0 || M <= 0 || !fpsg::tile_cfg(B, N, M, variant, &c)) return 0;
  return fpsg::tile_ws_bytes(B, N, M, c);
}

extern "C" int fpsg_chamfer_fwd_tiled(const float* xyz1, const float* xyz2, int B, int N, int M,
                                      float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                                      void* ws, size_t ws_bytes, int variant, fpsg_stream_t stream) {
  return fpsg::tiled_forward("fpsg_chamfer_fwd_tiled", xyz1, xyz2, B, N, M, dist1, idx1, dist2, idx2, ws, ws_bytes,
                             variant, false, 0, 0.f, 0.f, nullptr, stream);
}

extern "C" int fpsg_chamfer_fwd_tiled_losses(const float* xyz1, const float* xyz2, int B, int N, int M,
                                             float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                                             void* ws, size_t ws_bytes, int variant, int n_first, float w_first,
                                             float w_rest, float* out3, fpsg_stream_t stream) {
  return fpsg::tiled_forward("fpsg_chamfer_fwd_tiled_losses", xyz1, xyz2, B, N, M, dist1, idx1, dist2, idx2, ws,
                             ws_bytes, variant, true, n_first, w_first, w_rest, out3, stream);
}

extern "C" int fpsg_chamfer_bwd_sorted(const float* xyz1, const float* xyz2, const int32_t* idx1,
                                       const int32_t* idx2, const float* g1, const float* g2, int B,
                                       int N, int M, float* gxyz1, float* gxyz2, fpsg_stream_t stream) {
  FPSG_REQUIRE_PTR(g1); FPSG_REQUIRE_PTR(g2);
  return fpsg::sorted_backward("fpsg_chamfer_bwd_sorted", false, xyz1, xyz2, idx1, idx2, g1, g2,
                               fpsg::PairGrad{nullptr, 0, 0.f, 0.f}, B, N, M, gxyz1, gxyz2, stream);
}

extern "C" int fpsg_chamfer_bwd_losses(const float* xyz1, const float* xyz2, const int32_t* idx1,
                                       const int32_t* idx2, const float* g_first, const float* g_rest,
                                       const float* g_total, int B, int N, int M, int n_first, float w_first,
                                       float w_rest, float* gxyz1, float* gxyz2, fpsg_stream_t stream) {
  FPSG_REQUIRE(n_first >= 0 && n_first <= B, FPSG_E_SHAPE, "fpsg_chamfer_bwd_losses: n_first=%d outside [0,%d]", n_first, B);
  return fpsg::sorted_backward("fpsg_chamfer_bwd_losses", true, xyz1, xyz2, idx1, idx2, g_first, g_rest,
                               fpsg::PairGrad{g_total, n_first, w_first, w_rest}, B, N, M, gxyz1, gxyz2, stream);
}
