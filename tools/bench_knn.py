"""K3 micro-benchmark: the streaming kernel against the score-tile kernel at DGCNN's layer shapes (64 clouds of 2048
points, k = 20), channel-major and point-major input.  Usage (GPU box): python tools/bench_knn.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.dgcnn import KNN_FORCE_TILE, knn_int32  # noqa: E402

PEAK = 157.3e12


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    B, N, k = 64, 2048, 20
    for C in (3, 64, 128):
        x = torch.randn(B, C, N, device=dev)
        xp = x.transpose(1, 2).contiguous()
        ref = knn_int32(x, k, flags=KNN_FORCE_TILE)
        same = bool((knn_int32(x, k) == ref).all()) and bool((knn_int32(xp, k, point_major=True) == ref).all())
        flop = 2.0 * C * N * N * B
        row = [f"C={C:3d}", f"identical={same}"]
        for name, fn in (("tile", lambda: knn_int32(x, k, flags=KNN_FORCE_TILE)), ("stream", lambda: knn_int32(x, k)),
                         ("stream_pm", lambda: knn_int32(xp, k, point_major=True))):
            t = timeit(fn, reps)
            row.append(f"{name} {t*1e6:8.1f} us {flop/t/1e12:6.2f} TFLOP/s ({flop/t/PEAK:5.3f} of the fp32 MFMA peak)")
        print("  ".join(row), flush=True)
    x = torch.randn(B, 64, N, device=dev)
    print("stream, C=64, time against k:", {kk: round(timeit(lambda: knn_int32(x, kk), reps) * 1e6, 1) for kk in (1, 5, 10, 20, 24)})


if __name__ == "__main__":
    main()
