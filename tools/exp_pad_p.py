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


def decoder(args, dev):
    """The patch MLPs' wide layers (16 patches, 37 x 128 points): w [16, out, in] . h [16, in, 4736] and the two products
    of its backward, with the odd feature dimensions 1539 / 769 (a) as they are, (b) only the leading dimension of the
    weights padded to 8 floats (views of padded buffers), (c) the feature dimensions padded to 8 / 32 with zero rows and
    columns (what a padded activation layout would give)."""
    G, BP = 16, args.n_img * 128
    for cin, cout in ((1539, 769), (769, 384)):
        legs = {}
        for tag, pad in (("exact", 1), ("pad8", 8), ("pad32", 32)):
            ci, co = (cin + pad - 1) // pad * pad, (cout + pad - 1) // pad * pad
            w = torch.randn(G, co, ci, device=dev)
            h = torch.randn(G, ci, BP, device=dev)
            g = torch.randn(G, co, BP, device=dev)
            wt = w.transpose(1, 2).contiguous()
            legs[(tag, "fwd")] = lambda w=w, h=h: torch.bmm(w, h)
            legs[(tag, "dgrad")] = lambda wt=wt, g=g: torch.bmm(wt, g)
            legs[(tag, "wgrad")] = lambda g=g, h=h: torch.bmm(g, h.transpose(1, 2))
        # (b): the true sizes, only the weights' rows 32-byte aligned
        wp = torch.randn(G, cout, (cin + 7) // 8 * 8, device=dev)[:, :, :cin]
        wtp = torch.randn(G, cin, (cout + 7) // 8 * 8, device=dev)[:, :, :cout]
        h = torch.randn(G, cin, BP, device=dev)
        g = torch.randn(G, cout, BP, device=dev)
        legs[("ld8", "fwd")] = lambda wp=wp, h=h: torch.bmm(wp, h)
        legs[("ld8", "dgrad")] = lambda wtp=wtp, g=g: torch.bmm(wtp, g)
        for fn in legs.values():
            fn()
            fn()
        torch.cuda.synchronize()
        best = {k: float("inf") for k in legs}
        for _ in range(args.rounds):
            for k, fn in legs.items():
                best[k] = min(best[k], _time(fn, args.reps))
        print(json.dumps({"layer": f"{cin}->{cout}", **{f"{t}_{n}_us": round(v, 1) for (t, n), v in best.items()}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pads", default="1,32,64")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-split", action="store_true")
    ap.add_argument("--n-img", type=int, default=37, help="images per product (two episodes batched: 74)")
    ap.add_argument("--decoder", action="store_true", help="the decoder's wide layers with padded feature dimensions instead")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fd, records = tempfile.mkstemp(prefix="fpsg_pad_records_", suffix=".csv")
    os.close(fd)
    shutil.copyfile(gemm_tuning.DEFAULT_FILE, records)
    print(json.dumps(gemm_tuning.enable(records, tune=True)), file=sys.stderr)
    if args.decoder:
        decoder(args, dev)
        os.remove(records) if os.path.exists(records) else None
        return
    shapes = ((256, 256, 56), (512, 512, 28)) if args.quick else ((128, 128, 112), (128, 256, 56), (256, 256, 56),
                                                                 (256, 512, 28), (512, 512, 28), (512, 512, 14))
    pads = [int(p) for p in args.pads.split(",")]
    for C, K, H in shapes:
        P = args.n_img * ((H + 3) // 4) ** 2
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
