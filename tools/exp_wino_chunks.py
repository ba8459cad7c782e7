"""Experiment: the three-kernel Winograd form (input transform -> 36 batched products -> output transform) over
all 37 images at once against the same work in image chunks whose V and M share two workspaces small enough to
stay in the 256 MB Infinity Cache.  Per repetition the caches are swept by a 1 GB fill outside the timed region.

    python tools/exp_wino_chunks.py            # on the GPU box
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import _hip, gemm_tuning  # noqa: E402
from fpsg_amd import winograd as wg  # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.load()


def chain(m, x, U, y, Vws, Mws, chunk, transforms=True, products=True):
    N, C, H, W = x.shape
    K = U.shape[1]
    tiles = (H // m) * (W // m)
    a2 = (m + 2) ** 2
    st = torch.cuda.current_stream().cuda_stream
    for n0 in range(0, N, chunk):
        n = min(chunk, N - n0)
        V = Vws[: a2 * C * n * tiles].view(a2, C, n * tiles)
        M = Mws[: a2 * K * n * tiles].view(a2, K, n * tiles)
        if transforms:
            wg._call("fpsg_wino_input_transform", m, x[n0].data_ptr(), n, C, H, W, V.data_ptr(), 0, st)
        if products:
            torch.bmm(U, V, out=M)
        if transforms:
            wg._call("fpsg_wino_output_transform", m, M.data_ptr(), n, K, H, W, y[n0].data_ptr(), 0, st)


def timed(fn, reps=6):
    flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    tot = 0.0
    for i in range(reps + 1):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i:
            tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    # unknown product shapes are timed by TunableOp on first use (a chunk's GEMM has its own best library kernel)
    gemm_tuning.enable(path="gpurun_out/exp_chunk_gemm_records.csv", tune=True)
    n_img = 37
    for C, K, H, m, chunks in ((128, 128, 112, 4, (37, 8, 4)), (256, 256, 56, 4, (37, 16, 8)),
                               (512, 512, 28, 4, (37, 19, 13))):
        x = torch.randn(n_img, C, H, H, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        U = wg._filter(m, w, False)
        y = torch.empty(n_img, K, H, H, device=dev)
        tiles = (H // m) ** 2
        a2 = (m + 2) ** 2
        Vws = torch.empty(a2 * C * n_img * tiles, device=dev)
        Mws = torch.empty(a2 * K * n_img * tiles, device=dev)
        ref = None
        for chunk in chunks:
            chain(m, x, U, y, Vws, Mws, chunk)                 # tunes the chunk's product
            t = timed(lambda: chain(m, x, U, y, Vws, Mws, chunk))
            tt = timed(lambda: chain(m, x, U, y, Vws, Mws, chunk, True, False))
            tp = timed(lambda: chain(m, x, U, y, Vws, Mws, chunk, False, True))
            if ref is None:
                ref = y.clone()
                same = True
            else:
                same = bool(torch.equal(ref, y))
            vm = a2 * (C + K) * chunk * tiles * 4 / 2**20
            print(f"{C:3d}->{K:3d} @{H:3d} m={m} chunk {chunk:2d} images (V+M {vm:6.0f} MB): {t:8.1f} us  (transforms alone {tt:7.1f}, products alone {tp:7.1f})  identical={same}", flush=True)
        del x, y, Vws, Mws


if __name__ == "__main__":
    main()
