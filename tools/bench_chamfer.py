"""K1 micro-benchmark: N=M=2048, several batch sizes; the two-pass forward, the
one-pass tiled forward (tiles + finalize, timed together as one op), both backward kernels.
Usage (GPU box): python tools/bench_chamfer.py [--reps 200] [--sweep] [--blob]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import _hip  # noqa: E402

FWD_BYTES = 81920      # per cloud pair, N=M=2048
BWD_BYTES = 131072
PAIRS = 2 * 2048 * 2048
PEAK_HBM = 8.0e12
PEAK_F32 = 157.3e12
TILED = [148, 144, 142, 141, 124, 122, 114, 112, 48, 44, 42, 24, 18]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--sweep", action="store_true", help="time every variant of the forward kernels")
    ap.add_argument("--blob", action="store_true", help="second cloud collapsed to a small blob (early training)")
    ap.add_argument("--batches", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.load()
    N = M = args.n
    scale = (N / 2048.0)
    batches = [int(b) for b in args.batches.split(",")] if args.batches else \
        ((1, 2, 5, 8, 16, 32, 37, 64, 128, 256) if args.sweep else (1, 5, 32, 37, 64, 256, 1024))
    # bring the chip to its working clock before anything is timed
    warm = torch.randn(4096, 4096, device=dev)
    t_end = torch.cuda.Event(enable_timing=True)
    for _ in range(200):
        warm = torch.tanh(warm)
    torch.cuda.synchronize()
    for B in batches:
        p1 = torch.rand(B, N, 3, device=dev) * 2 - 1
        p2 = torch.tanh(torch.randn(B, M, 3, device=dev) * (0.02 if args.blob else 1.0))
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, M, device=dev)
        i1 = torch.empty(B, N, device=dev, dtype=torch.int32); i2 = torch.empty(B, M, device=dev, dtype=torch.int32)
        g1 = torch.randn(B, N, device=dev); g2 = torch.randn(B, M, device=dev)
        gx1 = torch.empty_like(p1); gx2 = torch.empty_like(p2)
        s = torch.cuda.current_stream().cuda_stream
        ws_max = max(lib.fpsg_chamfer_workspace_bytes(B, N, M, v) for v in [-1] + TILED)
        ws = torch.empty(max(ws_max, 8), dtype=torch.uint8, device=dev)

        def two_pass(cfg):
            return lambda: lib.fpsg_chamfer_fwd_variant(p1.data_ptr(), p2.data_ptr(), B, N, M, d1.data_ptr(), i1.data_ptr(),
                                                        d2.data_ptr(), i2.data_ptr(), cfg, s)

        def tiled(variant):
            nb = lib.fpsg_chamfer_workspace_bytes(B, N, M, variant)
            return lambda: lib.fpsg_chamfer_fwd_tiled(p1.data_ptr(), p2.data_ptr(), B, N, M, d1.data_ptr(), i1.data_ptr(),
                                                      d2.data_ptr(), i2.data_ptr(), ws.data_ptr(), nb, variant, s)

        def bwd(name):
            return lambda: getattr(lib, name)(p1.data_ptr(), p2.data_ptr(), i1.data_ptr(), i2.data_ptr(), g1.data_ptr(),
                                              g2.data_ptr(), B, N, M, gx1.data_ptr(), gx2.data_ptr(), s)

        runs = [("fwd two-pass", two_pass(-1), FWD_BYTES)]
        if lib.fpsg_chamfer_workspace_bytes(B, N, M, -1) > 0:       # the one-pass form does not serve fewer than ~7 pairs
            runs.append(("fwd tiled", tiled(-1), FWD_BYTES))
        if args.sweep:
            runs += [(f"fwd two-pass[cfg{c}]", two_pass(c), FWD_BYTES) for c in range(7)]
            runs += [(f"fwd tiled[{v}]", tiled(v), FWD_BYTES) for v in TILED]
        runs += [("bwd sorted", bwd("fpsg_chamfer_bwd_sorted"), BWD_BYTES), ("bwd scan", bwd("fpsg_chamfer_bwd_scan"), BWD_BYTES)]
        for name, fn, nbytes in runs:
            for _ in range(50):
                assert fn() == 0, lib.fpsg_last_error()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / args.reps
            gbps = B * nbytes * scale / t / 1e9
            line = f"{name:22s} B={B:5d} N={N}: {t*1e6:9.2f} us  {gbps:8.1f} GB/s ({gbps*1e9/PEAK_HBM*100:5.2f}% HBM)"
            if name.startswith("fwd"):
                # executed work: the one-pass tiled form evaluates each of the N*M distances once (it serves both
                # directions), the two-pass form 2*N*M; the second column is SURVEY.md 8(d)'s 2*N*M convention
                tf2 = B * PAIRS * scale * scale * 8 / t
                tfx = tf2 * (0.5 if "tiled" in name else 1.0)
                line += (f"  {tfx/1e12:7.2f} TFLOP/s executed ({tfx/PEAK_F32*100:5.1f}% fp32 peak)"
                         f"  [2NM convention: {tf2/1e12:7.2f} TFLOP/s, {tf2/PEAK_F32*100:5.1f}%]")
            print(line, flush=True)


if __name__ == "__main__":
    main()
