// ARCHIVED EXPERIMENT (round 4) -- not part of libfpsg_hip.so: measured slower than the streaming kernel (342 vs 256 us,
// profiles/r04/k3_row_per_lane_kernel_rejected.txt).  To rebuild it: copy next to knn_internal.h, add it to the Makefile,
// declare knn_rows_launch in knn_internal.h and call it from fpsg_knn_ex for C <= 4.
//
// knn_rows.hip -- K3 for C <= 4 (DGCNN's first EdgeConv layer: raw xyz): one query row per LANE, distances on the
// vector ALU, selection in per-lane buffers.  gfx950.  Replaces `knn` of reference src/dgcnn/model.py:13-20 for that
// layer; the streaming MFMA kernel (knn_stream.hip) keeps C = 64 / 128.
//
// Why a second kernel (round 4): at C = 3 the streaming kernel's 257 us are selection, not distance -- the MFMA part is
// 1 M of its 128 M instructions -- and its selection is shaped by the MFMA's output layout: a row's scores sit in 4
// lanes, so its buffer, slot counter and threshold are shared (one LDS atomic per tile pair, events that split a row over
// 16 lanes): 26 instructions per score-lane.  With three channels the distance is five packed instructions per TWO
// scores on the vector ALU, which leaves the row <-> lane mapping free:
//   * a workgroup = 256 query rows, one per lane; the cloud's candidates pass through LDS in stages of 256 (SoA: x, y,
//     z, (w,) |x|^2; two buffers, one barrier per stage); every lane reads the same address: ds_read_b128 = four
//     candidates per instruction and coordinate (broadcast, conflict-free);
//   * score  pd_ij = fma(2, dot_ij, -|x_j|^2) - |x_i|^2,  dot = fma chain over the channels in ascending order from +0
//     -- the oracle's expressions (oracle_knn), two candidates per packed instruction;
//   * a lane keeps its row's threshold T (the k-th best score so far, -inf until k candidates were seen) and slot
//     counter in REGISTERS; a score above T is appended as a 64-bit key (orderable score << 32 | ~index: descending
//     score, then ascending index -- the oracle's order) to the lane's 32-entry column of an LDS buffer: no atomics;
//   * when any lane of a wave has fewer than 4 free slots, every lane of the wave loads its column into registers,
//     sorts it with a 32-input bitonic network (240 compare-exchanges on 64-bit keys, static indices), writes the best
//     k back and takes the k-th key's score as T.  A later candidate with score == T has a larger index than every kept
//     entry of that score, so `score > T` loses nothing.  Expected ~10 such events per wave and cloud;
//   * after the sweep one more sort, and the first k indices are written.
// Results are bit-identical to oracle_knn (tests/test_dgcnn_gpu.py: the same shapes, ties and adversarial orders as the
// other two kernels).  Deterministic.
#include "../../fpsg_amd/csrc/knn_internal.h"

namespace fpsg {
namespace {

constexpr int kRThreads = 256;        // query rows per workgroup
constexpr int kRCap = 32;             // entries per lane (k <= 24 kept + appends of up to two quads)
constexpr int kRStage = 256;          // candidates per stage

template <bool C4>
constexpr size_t rows_lds_bytes() {
  return (size_t)2 * (C4 ? 5 : 4) * kRStage * sizeof(float) + (size_t)kRCap * kRThreads * sizeof(unsigned long long);
}

__device__ __forceinline__ float rows_unorderable(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

// descending bitonic sort of 32 keys held in registers (every index is a compile-time constant)
__device__ __forceinline__ void sort32_desc(unsigned long long (&K)[kRCap]) {
#pragma unroll
  for (int k2 = 2; k2 <= kRCap; k2 <<= 1) {
#pragma unroll
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
#pragma unroll
      for (int i = 0; i < kRCap; ++i) {
        const int l = i ^ j2;
        if (l > i) {
          const bool desc = (i & k2) == 0;                 // blocks alternate; the last merge (k2 = 32) is all descending
          const unsigned long long a = K[i], c = K[l];
          const bool sw = desc ? (a < c) : (a > c);
          K[i] = sw ? c : a;
          K[l] = sw ? a : c;
        }
      }
    }
  }
}

template <bool C4>
__global__ __launch_bounds__(kRThreads) void knn_rows_kernel(const float* __restrict__ xk /*[B][N][4]*/,
                                                             const float* __restrict__ xx /*[B][N]*/, int B, int N, int k,
                                                             int32_t* __restrict__ idx) {
  constexpr int NA = C4 ? 5 : 4;                           // staged arrays: x, y, z, (w,) |x|^2
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* stage = lds;                                                            // [2][NA][kRStage]
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(lds + 2 * NA * kRStage);   // [kRCap][kRThreads]

  // clouds -> XCDs as in the streaming kernel: a cloud's row blocks take workgroup ids of one residue class mod 8
  const int nblk = (N + kRThreads - 1) / kRThreads;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / nblk) * 8 + xcd;
  const int blk = slot - (slot / nblk) * nblk;
  if (b >= B) return;                                      // whole workgroup, before any barrier

  const int tid = threadIdx.x;
  const float* __restrict__ xkb = xk + (size_t)b * N * 4;
  const float* __restrict__ xxb = xx + (size_t)b * N;
  const int i = blk * kRThreads + tid;
  const int ic = i < N ? i : N - 1;
  const v4f q = *reinterpret_cast<const v4f*>(xkb + (size_t)ic * 4);
  const float xxq = xxb[ic];
  const v2f qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z}, qw = {q.w, q.w};
  const v2f nxxq = {-xxq, -xxq}, two = {2.0f, 2.0f}, zero = {0.0f, 0.0f};

  unsigned long long* const col = buf + tid;               // this lane's entries: col[e * kRThreads]
  float T = -__builtin_inff();
  int cnt = 0;

  auto event = [&]() {
    unsigned long long K[kRCap];
#pragma unroll
    for (int e = 0; e < kRCap; ++e) {
      const unsigned long long v = col[e * kRThreads];
      K[e] = e < cnt ? v : 0ull;                           // 0 sorts below every key
    }
    sort32_desc(K);
    cnt = cnt < k ? cnt : k;
#pragma unroll
    for (int e = 0; e < 24; ++e) col[e * kRThreads] = K[e];     // k <= 24 (slots past cnt are rewritten before they are read)
    if (cnt == k) T = rows_unorderable((unsigned)(col[(k - 1) * kRThreads] >> 32));
  };

  v4f pre;
  float prexx;
  auto load_stage = [&](int s) {
    const int j = s * kRStage + tid;
    pre = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    prexx = __builtin_inff();                              // padding scores -inf: never above a threshold
    if (j < N) { pre = *reinterpret_cast<const v4f*>(xkb + (size_t)j * 4); prexx = xxb[j]; }
  };
  auto store_stage = [&](int sel) {
    float* st = stage + sel * NA * kRStage;
    st[tid] = pre.x;
    st[kRStage + tid] = pre.y;
    st[2 * kRStage + tid] = pre.z;
    if (C4) st[3 * kRStage + tid] = pre.w;
    st[(NA - 1) * kRStage + tid] = prexx;
  };

  const int n_stages = (N + kRStage - 1) / kRStage;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < n_stages; ++s) {
    const int sel = s & 1;
    if (s + 1 < n_stages) load_stage(s + 1);
    const float* st = stage + sel * NA * kRStage;
    const v4f* px = reinterpret_cast<const v4f*>(st);
    const v4f* py = reinterpret_cast<const v4f*>(st + kRStage);
    const v4f* pz = reinterpret_cast<const v4f*>(st + 2 * kRStage);
    const v4f* pw = reinterpret_cast<const v4f*>(st + 3 * kRStage);
    const v4f* pxx = reinterpret_cast<const v4f*>(st + (NA - 1) * kRStage);
    const int jbase = s * kRStage;
    const int quads = (N - jbase < kRStage ? N - jbase + 3 : kRStage) >> 2;      // quads holding a real candidate
    for (int g = 0; g < quads; ++g) {
      const v4f X = px[g], Y = py[g], Z = pz[g], XX = pxx[g];
      v4f Wc = {0.0f, 0.0f, 0.0f, 0.0f};
      if (C4) Wc = pw[g];
      float sc[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const v2f cx = h ? v2f{X[2], X[3]} : v2f{X[0], X[1]};
        const v2f cy = h ? v2f{Y[2], Y[3]} : v2f{Y[0], Y[1]};
        const v2f cz = h ? v2f{Z[2], Z[3]} : v2f{Z[0], Z[1]};
        const v2f cxx = h ? v2f{XX[2], XX[3]} : v2f{XX[0], XX[1]};
        v2f acc = fma_rn(qx, cx, zero);
        acc = fma_rn(qy, cy, acc);
        acc = fma_rn(qz, cz, acc);
        if (C4) { const v2f cw = h ? v2f{Wc[2], Wc[3]} : v2f{Wc[0], Wc[1]}; acc = fma_rn(qw, cw, acc); }
        const v2f pd = fma_rn(two, acc, -cxx) + nxxq;
        sc[2 * h] = pd.x;
        sc[2 * h + 1] = pd.y;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sc[u] > T) {
          const unsigned j = (unsigned)(jbase + 4 * g + u);
          col[cnt * kRThreads] = ((unsigned long long)knn_orderable(sc[u] + 0.0f) << 32) | (unsigned)~j;
          ++cnt;
        }
      }
      if (__builtin_amdgcn_ballot_w64(cnt > kRCap - 4) != 0ull) event();
    }
    if (s + 1 < n_stages) store_stage(sel ^ 1);
    __syncthreads();
  }
  event();
  if (i < N) {
    int32_t* out = idx + ((size_t)b * N + i) * k;
    for (int r = 0; r < k; ++r) out[r] = (int32_t)~(unsigned)col[r * kRThreads];
  }
}

template <bool C4>
int launch_rows(const float* xk, const float* xx, int B, int N, int k, int32_t* idx, hipStream_t s) {
  constexpr size_t lds_bytes = rows_lds_bytes<C4>();
  static_assert(2 * lds_bytes <= 160 * 1024, "two workgroups per CU");
  auto kern = knn_rows_kernel<C4>;
  const hipError_t optin = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (optin != hipSuccess) {
    set_error("fpsg_knn: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(optin));
    return (int)optin;
  }
  const int nblk = (N + kRThreads - 1) / kRThreads;
  const int grid = 8 * nblk * ((B + 7) / 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kRThreads), lds_bytes, s, xk, xx, B, N, k, idx);
  return launch_status("fpsg_knn(rows)");
}

}  // namespace

int knn_rows_launch(const float* xk, const float* xx, int B, int C, int N, int k, int32_t* idx, hipStream_t s) {
  if (C > 4 || k > 24) { set_error("fpsg_knn(rows): C=%d, k=%d not served (C <= 4, k <= 24)", C, k); return FPSG_E_LIMIT; }
  return C == 4 ? launch_rows<true>(xk, xx, B, N, k, idx, s) : launch_rows<false>(xk, xx, B, N, k, idx, s);
}

}  // namespace fpsg
