"""K10 (fpsg_gemm_split: batched fp32 GEMM on the bf16 matrix pipe, three-way split operands) against the library fp32
GEMM (torch.bmm -> rocBLAS / hipBLASLt on the fp32 MFMA) at the Winograd-domain shapes of the c5 step (bench.py's K6g legs):
time, interleaved in one process, and the error of both against a float64 product of the same operands.

    python tools/bench_gemm_split.py [--variants -1,0,1] [--rounds 5] [--quick]

VERDICT r4 item 1's gate: >= 1.6x at 256 -> 256 @56 and max / median error no worse than 1.5x the library's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpsg_amd.winograd import row_stride  # noqa: E402
from fpsg_amd.gemm_split import bmm_f32, bmm_packed, bmm_persistent, bmm_split as gemm_split, pack_a  # noqa: E402


def _time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def _errors(C, ref64):
    """max and median |C - ref| over the reference's RMS (one scale per problem)."""
    d = (C.double() - ref64).abs()
    scale = ref64.pow(2).mean().sqrt()
    return float(d.max() / scale), float(d.flatten()[:: max(1, d.numel() // 4_000_000)].median() / scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="-1")
    ap.add_argument("--packed", default="", help="variants of the packed-A forward form to time (v100.. columns)")
    ap.add_argument("--persistent", default="", help="variants of the persistent forward form to time (v200.. columns)")
    ap.add_argument("--f32", default="", help="variants of K11 (fpsg_gemm_f32_nn, fp32 MFMA) to time on the forward leg (v300.. columns)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n-img", type=int, default=37)
    ap.add_argument("--quick", action="store_true", help="256 -> 256 @56 only")
    ap.add_argument("--err-batches", type=int, default=3)
    ap.add_argument("--dist", default="randn", choices=["randn", "wino"])
    ap.add_argument("--decoder", action="store_true", help="the decoder's wide layers instead of the trunk's products")
    ap.add_argument("--no-tuning", action="store_true", help="library GEMMs by the libraries' default heuristic")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if not args.no_tuning:      # the library side as the step runs it: the recorded kernel per shape (fpsg_amd/tuning)
        from fpsg_amd import gemm_tuning
        print(json.dumps(gemm_tuning.enable()), file=sys.stderr)
    base_variants = [int(v) for v in args.variants.split(",")]
    shapes = ((256, 256, 56),) if args.quick else ((128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28),
                                                   (512, 512, 28), (512, 512, 14))
    if args.decoder:     # the patch MLPs' wide layers: 16 patches, (in -> out) on 37 x 128 points (H = 0 marks them)
        shapes = ((1539, 769, 0), (769, 384, 0), (769, 1539, 0))
    for C, K, H in shapes:
        P = row_stride(args.n_img * ((H + 3) // 4) ** 2) if H else args.n_img * 128      # the step's padded rows
        nb_ = 36 if H else 16
        U = torch.randn(nb_, K, C, device=dev)
        V = torch.randn(nb_, C, P, device=dev)
        gM = torch.randn(nb_, K, P, device=dev)
        if args.dist == "wino":       # the spread of magnitudes of transform-domain data: per-xi scales over 3 decades
            s = torch.logspace(-1.5, 1.5, 36, device=dev).view(36, 1, 1)
            U, V, gM = U / s, V * s, gM * s
        flop = 2.0 * nb_ * K * C * P
        packed = {int(v): pack_a(U, int(v)) for v in args.packed.split(",") if v != ""}

        if args.persistent and 0 not in packed:
            packed[0] = pack_a(U, 0)
        if args.persistent and 2 not in packed:      # 128-row tiles (the persistent form's variants 13, 14)
            packed[2] = pack_a(U, 2)

        def fwd_fn(v):      # variant ids 100 ..: the packed-A form's variant v - 100; 200 ..: the persistent form's
            if 300 <= v < 310:
                return bmm_f32(U, V, v - 300)
            if 200 <= v < 220:
                return bmm_persistent(packed[2 if v - 200 >= 13 else 0], U.shape, V, v - 200)
            return bmm_packed(packed[v - 100], U.shape, V, v - 100) if 100 <= v < 110 else gemm_split(U, V, False, v)

        legs = {
            "fwd": (lambda: torch.bmm(U, V), fwd_fn),
            "dw": (lambda: torch.bmm(gM, V.transpose(1, 2)), lambda v: gemm_split(gM, V, True, v)),
        }
        for leg, (lib_fn, split_fn) in legs.items():
            variants = base_variants + ([100 + v for v in packed if str(v) in args.packed.split(",")] +
                                        [200 + int(v) for v in args.persistent.split(",") if v != ""] +
                                        [300 + int(v) for v in args.f32.split(",") if v != ""] if leg == "fwd" else [])
            # errors on the first batches against float64
            nb = args.err_batches
            if leg == "fwd":
                ref = torch.bmm(U[:nb].double(), V[:nb].double())
            else:
                ref = torch.bmm(gM[:nb].double(), V[:nb].double().transpose(1, 2))
            e_lib = _errors(lib_fn()[:nb], ref)
            row = {"shape": f"{C}->{K} @{H}", "leg": leg, "dims": f"{nb_} x [{K}x{C}].[{C}x{P}]" if leg == "fwd" else f"{nb_} x [{K}x{P}].[{P}x{C}]",
                   "lib_err_max": e_lib[0], "lib_err_med": e_lib[1]}
            outs = {}
            for v in variants:
                try:
                    Cs = split_fn(v)
                    torch.cuda.synchronize()
                except Exception as ex:  # a variant that does not apply to the shape
                    row[f"v{v}"] = f"n/a ({ex})"
                    continue
                outs[v] = _errors(Cs[:nb], ref)
                del Cs
            del ref
            # interleaved timing rounds
            lib_fn(); torch.cuda.synchronize()
            t_lib, t_v = [], {v: [] for v in outs}
            for _ in range(args.rounds):
                t_lib.append(_time(lib_fn, args.reps))
                for v in outs:
                    t_v[v].append(_time(lambda: split_fn(v), args.reps))
            tl = sorted(t_lib)[len(t_lib) // 2]
            row["lib_us"] = round(tl * 1e6, 1)
            row["lib_TFLOPs"] = round(flop / tl / 1e12, 1)
            for v in outs:
                tv = sorted(t_v[v])[len(t_v[v]) // 2]
                row[f"v{v}"] = {"us": round(tv * 1e6, 1), "min_us": round(min(t_v[v]) * 1e6, 1), "speedup": round(tl / tv, 3),
                                "fp32_equiv_TFLOPs": round(flop / tv / 1e12, 1), "bf16_TFLOPs": round(6 * flop / tv / 1e12, 1),
                                "err_max": outs[v][0], "err_med": outs[v][1],
                                "err_max_ratio": round(outs[v][0] / e_lib[0], 3), "err_med_ratio": round(outs[v][1] / e_lib[1], 3)}
            print(json.dumps(row), flush=True)
        if packed:
            t = _time(lambda: pack_a(U, next(iter(packed))), args.reps)
            print(json.dumps({"shape": f"{C}->{K} @{H}", "leg": "pack_a", "us": round(t * 1e6, 1)}), flush=True)
        del U, V, gM, packed


if __name__ == "__main__":
    main()
