// knn_internal.h -- what knn.hip (the C-ABI entry points and the score-tile kernel) and knn_stream.hip (the
// streaming kernel) share.
#pragma once
#include "fpsg_common.h"

namespace fpsg {

// fp32 -> unsigned with the same order (larger float = larger unsigned); callers canonicalise -0 with `+ 0.0f` first
__device__ __forceinline__ unsigned knn_orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// channels after padding for the streaming kernel (0: not served by it)
inline int knn_stream_cpad(int C) { return C <= 4 ? 4 : (C <= 64 ? 64 : (C <= 128 ? 128 : 0)); }
// the streaming kernel keeps ~1.4 k candidates per row after a compaction of its 124-entry row buffers
inline bool knn_stream_serves(int C, int k) { return knn_stream_cpad(C) != 0 && k <= 24; }

// xk [B][N][cpad] k-interleaved point-major features, xx [B][N] squared norms (knn_stream_prepare makes both)
int knn_stream_prepare(const float* x, bool point_major, int B, int C, int N, float* xx, float* xk, hipStream_t s);
int knn_stream_launch(const float* xk, const float* xx, int B, int C, int N, int k, int flags, int32_t* idx,
                      hipStream_t s);

}  // namespace fpsg
