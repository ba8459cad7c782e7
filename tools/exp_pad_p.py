"""Experiment (round 5): does the row stride of the transform-domain tensors matter?  V / M / dM are ``[36][C][P]`` with
P = images x tiles = 29008 / 7252 / 1813 / 592 floats per row -- 16-byte aligned at best (1813 is odd), never a whole
number of 128-byte lines, so every row piece a GEMM tile reads or writes starts and ends inside a line shared with the
neighbouring tile.  Times the library fp32 GEMMs (tuned online for the padded shapes, records for the exact ones) and
K10 with P padded up to a multiple of 32 / 64 floats (the pad columns are part of the problem: N = Pp).

    python tools/exp_pad_p.py [--pads 1,32,64] [--rounds 3] [--reps 10] [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd import gemm_tuning  # noqa: E402
from fpsg_amd.gemm_split import bmm_split  # noqa: E402


def _time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pads", default="1,32,64")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-split", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fd, records = tempfile.mkstemp(prefix="fpsg_pad_records_", suffix=".csv")
    os.close(fd)
    shutil.copyfile(gemm_tuning.DEFAULT_FILE, records)
    print(json.dumps(gemm_tuning.enable(records, tune=True)), file=sys.stderr)
    shapes = ((256, 256, 56), (512, 512, 28)) if args.quick else ((128, 128, 112), (128, 256, 56), (256, 256, 56),
                                                                 (256, 512, 28), (512, 512, 28), (512, 512, 14))
    pads = [int(p) for p in args.pads.split(",")]
    for C, K, H in shapes:
        P = 37 * ((H + 3) // 4) ** 2
        legs = {}
        for pad in pads:
            Pp = (P + pad - 1) // pad * pad
            U = torch.randn(36, K, C, device=dev)
            Ut = torch.randn(36, C, K, device=dev)
            V = torch.randn(36, C, Pp, device=dev)
            dM = torch.randn(36, K, Pp, device=dev)
            M = torch.empty(36, K, Pp, device=dev)
            dU = torch.empty(36, K, C, device=dev)
            legs[(pad, "lib_fwd")] = (Pp, lambda U=U, V=V, M=M: torch.bmm(U, V, out=M))
            legs[(pad, "lib_dw")] = (Pp, lambda dM=dM, V=V, dU=dU: torch.bmm(dM, V.transpose(1, 2), out=dU))
            if not args.no_split:
                legs[(pad, "k10_fwd")] = (Pp, lambda U=U, V=V, M=M: bmm_split(U, V, False, -1, out=M))
                legs[(pad, "k10_dw")] = (Pp, lambda dM=dM, V=V, dU=dU: bmm_split(dM, V, True, -1, out=dU))
        for fn in legs.values():        # tuning + warm-up
            fn[1]()
            fn[1]()
        torch.cuda.synchronize()
        best = {k: float("inf") for k in legs}
        for _ in range(args.rounds):
            for k, (Pp, fn) in legs.items():
                best[k] = min(best[k], _time(fn, args.reps))
        row = {"C": C, "K": K, "H": H, "P": P}
        for (pad, name), us in best.items():
            row[f"{name}_pad{pad}_us"] = round(us, 1)
            row[f"Pp_pad{pad}"] = legs[(pad, name)][0]
        print(json.dumps(row), flush=True)
    os.remove(records) if os.path.exists(records) else None


if __name__ == "__main__":
    main()
