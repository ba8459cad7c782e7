"""K1 micro-benchmark: N=M=2048, several batch sizes, fwd and bwd.
Usage (GPU box): python tools/bench_chamfer.py [--reps 200]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--sweep", action="store_true", help="time every (R,W) variant of the forward kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.load()
    N = M = args.n
    scale = (N / 2048.0)
    for B in ((1, 2, 3, 5, 8, 12, 16, 24, 32, 37, 48, 64, 96, 128, 256) if args.sweep else (1, 5, 32, 37, 64, 256, 1024)):
        p1 = torch.rand(B, N, 3, device=dev) * 2 - 1
        p2 = torch.tanh(torch.randn(B, M, 3, device=dev))
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, M, device=dev)
        i1 = torch.empty(B, N, device=dev, dtype=torch.int32); i2 = torch.empty(B, M, device=dev, dtype=torch.int32)
        g1 = torch.randn(B, N, device=dev); g2 = torch.randn(B, M, device=dev)
        gx1 = torch.empty_like(p1); gx2 = torch.empty_like(p2)
        s = torch.cuda.current_stream().cuda_stream

        def fwd():
            return lib.fpsg_chamfer_fwd(p1.data_ptr(), p2.data_ptr(), B, N, M, d1.data_ptr(), i1.data_ptr(),
                                        d2.data_ptr(), i2.data_ptr(), s)

        def bwd():
            return lib.fpsg_chamfer_bwd(p1.data_ptr(), p2.data_ptr(), i1.data_ptr(), i2.data_ptr(), g1.data_ptr(),
                                        g2.data_ptr(), B, N, M, gx1.data_ptr(), gx2.data_ptr(), s)

        runs = [("fwd", fwd, FWD_BYTES, -1), ("bwd", bwd, BWD_BYTES, -1)]
        if args.sweep:
            runs = [(f"fwd[cfg{c}]", fwd, FWD_BYTES, c) for c in range(7)] + runs
        for name, fn, nbytes, cfg in runs:
            lib.fpsg_chamfer_set_config(cfg)
            for _ in range(20):
                assert fn() == 0
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / args.reps
            gbps = B * nbytes * scale / t / 1e9
            line = f"{name} B={B:5d} N={N}: {t*1e6:9.2f} us  {gbps:8.1f} GB/s ({gbps*1e9/PEAK_HBM*100:5.2f}% HBM)"
            if name.startswith("fwd"):
                tf = B * PAIRS * scale * scale * 8 / t
                line += f"  {tf/1e12:7.2f} TFLOP/s ({tf/PEAK_F32*100:5.1f}% fp32 peak)  {B*PAIRS*scale*scale/t/1e12:6.3f} Tpair/s"
            print(line, flush=True)


if __name__ == "__main__":
    main()
